#!/usr/bin/env python3
"""h-Edit with Plug-and-Play on the HIP path: same flags, dataset format and output-path scheme as the reference's
``text-guided/main_plugnplay.py`` (:56-87 flags, :117-118 strings, :124-248 loop) for the h-Edit modes
(``h_edit_R_pnp``, ``h_edit_D_pnp``).  Additions: ``--model_path`` / ``--random_init`` / ``--tiny`` / ``--seed`` as in
main_p2p.py (``--tiny`` = the four-level toy UNet: the injection indexes up_blocks[1..3]); entries are sharded
over ranks under torch.distributed.run.  The comparison baselines (ef_pnp, pnp_inv_w_pnp, nt_pnp, np_pnp, nmg_pnp)
are not part of this build and are refused."""
import argparse
import calendar
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from hedit import dist as D  # noqa: E402
from hedit.text import prescan_prompts  # noqa: E402
from hedit.inversion.ddim_inversion import ddim_inversion  # noqa: E402
from hedit.inversion.ddpm_inversion import inversion_forward_process_ddpm  # noqa: E402
from hedit.inversion.pnp_h_edit import h_Edit_PnP_implicit  # noqa: E402
from hedit.plug_n_play import register_attention_control_efficient, register_conv_control_efficient  # noqa: E402
from hedit.scheduler import DDIMScheduler  # noqa: E402
from hedit.utils import image_grid  # noqa: E402
from main_p2p import load_model  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument('--data_path', type=str, default="./PIE_Bench_Data")
    p.add_argument('--output_path', type=str, default="./results/pnp")
    p.add_argument('--edit_category_list', nargs='+', type=str, default=[str(i) for i in range(10)])
    p.add_argument("--mode", default="h_edit_R_pnp", help="modes: h_edit_R_pnp, h_edit_D_pnp")
    p.add_argument("--num_diffusion_steps", type=int, default=50)
    p.add_argument("--skip", type=int, default=0)
    p.add_argument("--eta", type=float, default=1.0)
    p.add_argument("--cfg_src", type=float, default=1.0)
    p.add_argument("--cfg_src_edit", type=float, default=5.0)
    p.add_argument("--cfg_tar", type=float, default=7.5)
    p.add_argument("--implicit", action='store_true', help="Use implicit form of h-Edit")
    p.add_argument("--optimization_steps", type=int, default=1)
    p.add_argument("--weight_reconstruction", type=float, default=0.1)
    p.add_argument("--pnp_f_t", type=float, default=0.45)
    p.add_argument("--pnp_attn_t", type=float, default=0.35)
    p.add_argument("--model_path", type=str, default=None, help="local SD-1.x checkpoint directory (diffusers layout)")
    p.add_argument("--random_init", action="store_true", help="synthetic SD-1.x-shaped weights (no checkpoint)")
    p.add_argument("--tiny", action="store_true", help="with --random_init: the small test configuration")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--batch", type=int, default=1, help="dataset entries edited in lock-step per pass (batched engine)")
    return p


def load_image(image_path, device, size=512):
    """main_plugnplay.py (its load_image twin of main_masactrl.py:39-44): RGB uint8 -> [-1, 1], nearest resize to size x size (F.interpolate's default)."""
    from PIL import Image
    a = torch.from_numpy(np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8).copy()).permute(2, 0, 1)
    image = a[:3].unsqueeze(0).float() / 127.5 - 1.
    return torch.nn.functional.interpolate(image, (size, size)).to(device)


def load_pnp_model(args, device):
    if args.random_init and args.tiny:
        # the injection needs the four-level SD layout (up_blocks[1..3] with attention)
        from hedit.pipeline import HEditPipeline
        from hedit.vae import TINY_VAE_CONFIG
        ucfg = dict(in_channels=4, out_channels=4, sample_size=32, block_out_channels=(64, 64, 128, 128),
                    down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                    up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, layers_per_block=2,
                    cross_attention_dim=64, attention_head_dim=2, norm_num_groups=32)
        ucfg["sample_size"] = 64
        vcfg = dict(TINY_VAE_CONFIG)
        vcfg.update(block_out_channels=(64, 64, 128))                 # f = 4: 256 x 256 images
        return HEditPipeline.from_random(ucfg, seed=args.seed, device=device, text_layers=2, vae_config=vcfg)
    return load_model(args, device)


def edit_group(args, model, entries, scale, size, device):
    """--batch N: the n entries of one group in lock-step (hedit.engine.HEditEngine.run_pnp: the injected pass has rows
    [x_orig|src]*n, [x_k|tar]*n and row n + i takes row i's q, k / features).  entries: [(item, image_path, save_path)]."""
    from hedit.engine import HEditEngine
    eng = HEditEngine(model)
    eta = args.eta
    is_ddim_inversion = eta == 0
    if is_ddim_inversion:
        model.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                        clip_sample=False, set_alpha_to_one=False)
    model.scheduler.config.timestep_spacing = "leading"
    model.scheduler.set_timesteps(args.num_diffusion_steps)
    xs = torch.cat([load_image(ip, device, size) for _, ip, _ in entries])
    w0 = (model.vae.encode(xs).latent_dist.mean * scale).float()
    src_p = [item["original_prompt"].replace("[", "").replace("]", "") for item, _, _ in entries]
    tar_p = [item["editing_prompt"].replace("[", "").replace("]", "") for item, _, _ in entries]
    if is_ddim_inversion:
        _, zs, wts = eng.ddim_inversion(w0, src_p, args.cfg_src)
        eta = 1.0
    elif 0 < eta <= 1:
        zs, wts = eng.ddpm_inversion(w0, src_p, eta=eta, cfg_src=args.cfg_src)
    else:
        raise SystemExit("Warning: out of range for eta")
    after = args.num_diffusion_steps - args.skip
    pnp_f_t, pnp_attn_t = int(after * args.pnp_f_t), int(after * args.pnp_attn_t)
    register_attention_control_efficient(model, model.scheduler.timesteps[:pnp_attn_t] if pnp_attn_t >= 0 else [])
    register_conv_control_efficient(model, model.scheduler.timesteps[:pnp_f_t] if pnp_f_t >= 0 else [])
    edited, _ = eng.run_pnp(wts[after].contiguous(), zs[:after].contiguous(), [[a, b] for a, b in zip(src_p, tar_p)],
                            [args.cfg_src, args.cfg_src_edit, args.cfg_tar], eta=eta, K=args.optimization_steps,
                            after_skip_steps=after, ddim_inv=is_ddim_inversion)
    x0_dec = model.vae.decode(1 / scale * edited).sample
    out = []
    for i, (_, _, save_path) in enumerate(entries):
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        image_grid(x0_dec[i:i + 1]).save(save_path)
        out.append(save_path)
    return out


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == "h_edit_D_pnp":
        assert args.eta == 0.0, "eta should be 0.0 for h-Edit-D"
    elif args.mode == "h_edit_R_pnp":
        assert args.eta == 1.0, "eta should be 1.0 for h-Edit-R"
    else:
        raise NotImplementedError(f"mode {args.mode}: only h_edit_R_pnp / h_edit_D_pnp are built")
    print(f'Arguments: {args}')
    rank, world, local_rank = D.env_rank_world()
    device = f"cuda:{local_rank if world > 1 else args.device_num}"
    torch.cuda.set_device(device)
    D.init_from_env(device)      # several ranks: RCCL group; rank 0 reads the checkpoints and broadcasts them
    data_path, output_path = args.data_path, args.output_path
    with open(os.path.join(data_path, 'mapping_file.json')) as f:
        full_data = json.load(f)
    time_stamp = calendar.timegm(time.gmtime())
    step_layer_string = f'_f_t_{args.pnp_f_t}_attn_t_{args.pnp_attn_t}_'
    weight_string = (f'implicit_{args.implicit}_eta_{args.eta}_src_orig_{args.cfg_src}_src_edit_{args.cfg_src_edit}'
                     f'_tar_scale_{args.cfg_tar}_w_rec_{args.weight_reconstruction}_n_opts_{args.optimization_steps}'
                     f'_time_{time_stamp}')
    model = load_pnp_model(args, device)
    prescan_prompts(model.tokenizer, full_data.values())      # (stand-in tokenizer only: word ids independent of order / shard)
    if model.vae is None:
        raise SystemExit("the checkpoint has no vae/ sub-folder: images cannot be encoded / decoded")
    scale = model.vae.config["scaling_factor"]
    size = model.unet.sample_size * model.vae.factor
    keys = [k for k, item in full_data.items() if item["editing_type_id"] in args.edit_category_list]
    written = []
    if args.batch > 1:
        mine = list(D.shard(len(keys), rank, world))
        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + step_layer_string)
        for lo in range(0, len(mine), args.batch):
            entries = []
            for idx in mine[lo:lo + args.batch]:
                item = full_data[keys[idx]]
                image_path = os.path.join(f"{data_path}/annotation_images", item["image_path"])
                entries.append((item, image_path, image_path.replace(data_path, os.path.join(output_path, sub))))
            written += edit_group(args, model, entries, scale, size, device)
        print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
        return written
    for idx in D.shard(len(keys), rank, world):
        item = full_data[keys[idx]]
        eta = args.eta
        is_ddim_inversion = eta == 0
        original_prompt = item["original_prompt"].replace("[", "").replace("]", "")
        editing_prompt = item["editing_prompt"].replace("[", "").replace("]", "")
        image_path = os.path.join(f"{data_path}/annotation_images", item["image_path"])
        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + step_layer_string)
        save_path = image_path.replace(data_path, os.path.join(output_path, sub))
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        if is_ddim_inversion:
            model.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                            clip_sample=False, set_alpha_to_one=False)
        model.scheduler.config.timestep_spacing = "leading"
        model.scheduler.set_timesteps(args.num_diffusion_steps)
        x0 = load_image(image_path, device, size)
        w0 = model.vae.encode(x0).latent_dist.mean * scale
        if is_ddim_inversion:
            wt, zs, wts = ddim_inversion(model, w0, original_prompt, args.cfg_src)
            eta = 1.0
        elif 0 < eta <= 1:
            wt, zs, wts, _ = inversion_forward_process_ddpm(model, w0, etas=eta, prompt=original_prompt,
                                                            cfg_scale_src=args.cfg_src,
                                                            num_inference_steps=args.num_diffusion_steps)
        else:
            raise SystemExit("Warning: out of range for eta")
        after_skip_steps = args.num_diffusion_steps - args.skip
        pnp_f_t, pnp_attn_t = int(after_skip_steps * args.pnp_f_t), int(after_skip_steps * args.pnp_attn_t)
        qk_injection_timesteps = model.scheduler.timesteps[:pnp_attn_t] if pnp_attn_t >= 0 else []
        conv_injection_timesteps = model.scheduler.timesteps[:pnp_f_t] if pnp_f_t >= 0 else []
        register_attention_control_efficient(model, qk_injection_timesteps)
        register_conv_control_efficient(model, conv_injection_timesteps)
        edited_w0, _ = h_Edit_PnP_implicit(model, xT=wts[after_skip_steps], eta=eta, prompts=[original_prompt, editing_prompt],
                                           cfg_scales=[args.cfg_src, args.cfg_src_edit, args.cfg_tar], prog_bar=True,
                                           zs=zs[:after_skip_steps], optimization_steps=args.optimization_steps,
                                           after_skip_steps=after_skip_steps, is_ddim_inversion=is_ddim_inversion)
        x0_dec = model.vae.decode(1 / scale * edited_w0).sample
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        image_grid(x0_dec).save(save_path)
        written.append(save_path)
    print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
    return written


if __name__ == "__main__":
    main()
