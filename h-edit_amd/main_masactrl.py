#!/usr/bin/env python3
"""h-Edit with MasaCtrl on the HIP path: same flags, dataset format and output-path scheme as the reference's
``text-guided/main_masactrl.py`` (:55-88 flags, :118-124 strings, :128-241 loop) for the h-Edit modes
(``h_edit_D_masactrl``, ``h_edit_R_masactrl``).  Like the reference the source prompt is the empty string
(:178).  (The reference reads ``args.LAYER`` for the ``--layer`` flag, :198, which argparse never sets; this
driver uses ``args.layer``.)  Additions: ``--model_path`` / ``--random_init`` / ``--tiny`` / ``--seed`` as in
main_p2p.py; entries are sharded over ranks under torch.distributed.run.  The comparison baselines
(pnp_inv_masactrl, ef_masactrl) are not part of this build and are refused."""
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
from hedit.inversion.masactrl_h_edit import h_Edit_masactrl_implicit  # noqa: E402
from hedit.masactrl import MutualSelfAttentionControl, regiter_attention_editor_diffusers  # noqa: E402
from hedit.scheduler import DDIMScheduler  # noqa: E402
from hedit.utils import image_grid  # noqa: E402
from main_p2p import load_model  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument('--data_path', type=str, default="./PIE_Bench_Data")
    p.add_argument('--output_path', type=str, default="./results/masactrl")
    p.add_argument('--edit_category_list', nargs='+', type=str, default=[str(i) for i in range(10)])
    p.add_argument("--mode", default="h_edit_D_masactrl", help="modes: h_edit_D_masactrl, h_edit_R_masactrl")
    p.add_argument("--num_diffusion_steps", type=int, default=50)
    p.add_argument("--skip", type=int, default=0)
    p.add_argument("--eta", type=float, default=0.0)
    p.add_argument("--cfg_src", type=float, default=1.0)
    p.add_argument("--cfg_src_edit", type=float, default=5.0)
    p.add_argument("--cfg_tar", type=float, default=7.5)
    p.add_argument("--implicit", action='store_true', help="Use implicit form of h-Edit")
    p.add_argument("--optimization_steps", type=int, default=1)
    p.add_argument("--weight_reconstruction", type=float, default=0.1)
    p.add_argument("--layer", type=int, default=10)
    p.add_argument("--step", type=int, default=4)
    p.add_argument("--model_path", type=str, default=None, help="local SD-1.x checkpoint directory (diffusers layout)")
    p.add_argument("--random_init", action="store_true", help="synthetic SD-1.x-shaped weights (no checkpoint)")
    p.add_argument("--tiny", action="store_true", help="with --random_init: the small test configuration")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--batch", type=int, default=1, help="dataset entries edited in lock-step per pass (batched engine)")
    return p


def load_image(image_path, device, size=512):
    """main_masactrl.py:39-44: RGB uint8 -> [-1, 1], nearest resize to size x size (F.interpolate's default)."""
    from PIL import Image
    a = torch.from_numpy(np.asarray(Image.open(image_path).convert("RGB"), dtype=np.uint8).copy()).permute(2, 0, 1)
    image = a[:3].unsqueeze(0).float() / 127.5 - 1.
    return torch.nn.functional.interpolate(image, (size, size)).to(device)


def edit_group(args, model, entries, scale, size, device):
    """--batch N: the n entries of one group in lock-step on hedit.engine.HEditEngine (rows [x_orig|null]*n,
    [x_k|null]*n, [x_orig|src]*n, [x_k|tar]*n: the mutual self-attention plan already maps row (kind, i) to the source
    row of image i).  entries: [(item, image_path, save_path)].  Same per-image arithmetic as the one-image path."""
    from hedit.engine import HEditEngine
    eng = HEditEngine(model)
    eta = args.eta
    is_ddim_inversion = eta == 0
    if is_ddim_inversion:
        model.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                        clip_sample=False, set_alpha_to_one=False)
    model.scheduler.config.timestep_spacing = "leading"
    model.scheduler.set_timesteps(args.num_diffusion_steps)
    n = len(entries)
    xs = torch.cat([load_image(ip, device, size) for _, ip, _ in entries])
    w0 = (model.vae.encode(xs).latent_dist.mean * scale).float()
    src_p = [""] * n                      # MasaCtrl runs without the source prompt (main_masactrl.py:178)
    tar_p = [item["editing_prompt"].replace("[", "").replace("]", "") for item, _, _ in entries]
    if is_ddim_inversion:
        _, zs, wts = eng.ddim_inversion(w0, src_p, args.cfg_src)
        eta = 1.0
    elif 0 < eta <= 1:
        zs, wts = eng.ddpm_inversion(w0, src_p, eta=eta, cfg_src=args.cfg_src)
    else:
        raise SystemExit("Warning: out of range for eta")
    editor = MutualSelfAttentionControl(args.step, args.layer)
    regiter_attention_editor_diffusers(model, editor)
    after = args.num_diffusion_steps - args.skip
    edited, _ = eng.run(wts[after].contiguous(), zs[:after].contiguous(), [[a, b] for a, b in zip(src_p, tar_p)],
                        [args.cfg_src, args.cfg_src_edit, args.cfg_tar], editor, eta=eta, p2p=True, implicit=True,
                        K=args.optimization_steps, after_skip_steps=after, ddim_inv=is_ddim_inversion, rec_pull=False)
    x0_dec = model.vae.decode(1 / scale * edited).sample
    out = []
    for i, (_, _, save_path) in enumerate(entries):
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        image_grid(x0_dec[i:i + 1]).save(save_path)
        out.append(save_path)
    return out


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == "h_edit_D_masactrl":
        assert args.eta == 0.0, "eta should be 0.0 for h-Edit-D"
    elif args.mode == "h_edit_R_masactrl":
        assert args.eta == 1.0, "eta should be 1.0 for h-Edit-R"
    else:
        raise NotImplementedError(f"mode {args.mode}: only h_edit_D_masactrl / h_edit_R_masactrl are built")
    print(f'Arguments: {args}')
    rank, world, local_rank = D.env_rank_world()
    device = f"cuda:{local_rank if world > 1 else args.device_num}"
    torch.cuda.set_device(device)
    D.init_from_env(device)      # several ranks: RCCL group; rank 0 reads the checkpoints and broadcasts them
    data_path, output_path = args.data_path, args.output_path
    with open(os.path.join(data_path, 'mapping_file.json')) as f:
        full_data = json.load(f)
    time_stamp = calendar.timegm(time.gmtime())
    step_layer_string = f'_step_{args.step}_layer_{args.layer}_'
    weight_string = (f'implicit_{args.implicit}_eta_{args.eta}_src_orig_{args.cfg_src}_src_edit_{args.cfg_src_edit}'
                     f'_tar_scale_{args.cfg_tar}_w_rec_{args.weight_reconstruction}_n_opts_{args.optimization_steps}'
                     f'_time_{time_stamp}')
    model = load_model(args, device)
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
        original_prompt = ""              # MasaCtrl runs without the source prompt (main_masactrl.py:178)
        if is_ddim_inversion:
            wt, zs, wts = ddim_inversion(model, w0, original_prompt, args.cfg_src)
            eta = 1.0
        elif 0 < eta <= 1:
            wt, zs, wts, _ = inversion_forward_process_ddpm(model, w0, etas=eta, prompt=original_prompt,
                                                            cfg_scale_src=args.cfg_src,
                                                            num_inference_steps=args.num_diffusion_steps)
        else:
            raise SystemExit("Warning: out of range for eta")
        editor = MutualSelfAttentionControl(args.step, args.layer)
        regiter_attention_editor_diffusers(model, editor)
        after_skip_steps = args.num_diffusion_steps - args.skip
        edited_w0, _ = h_Edit_masactrl_implicit(model, xT=wts[after_skip_steps], eta=eta, prompts=[original_prompt, editing_prompt],
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
