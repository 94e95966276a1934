#!/usr/bin/env python3
"""Text-guided editing driver on the HIP path: same flags, dataset format and output-path scheme as
the reference's ``text-guided/main_p2p.py`` (:38-70 flags, :76-103 strings, :110-275 loop) for the
h-Edit modes (h_edit_R, h_edit_R_p2p, h_edit_D_p2p; explicit or ``--implicit``).

Differences, all additive:
  * ``--model_path DIR``: a LOCAL checkpoint directory in the diffusers layout (no network here);
    ``--random_init`` builds SD-1.x-shaped synthetic weights instead (``--tiny`` = the small test
    configuration at 256x256).
  * launched under ``torch.distributed.run`` the dataset entries are sharded across the ranks
    (one process per GPU, no data-path collective).
  * ``--batch N``: N dataset entries are edited in lock-step by the batched engine (the reference edits one image
    at a time, main_p2p.py:110; images are independent, and the kernels are batch-invariant bit for bit, so every
    image comes out exactly as it does alone -- at several times the throughput).
The comparison baselines of the reference driver (ef, ef_p2p, nmg_p2p, pnp_inv_p2p) are not part of
this build and are refused.
"""
import argparse
import calendar
import json
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from hedit import dist as D  # noqa: E402
from hedit.text import prescan_prompts  # noqa: E402
from hedit.inversion.ddim_inversion import ddim_inversion  # noqa: E402
from hedit.inversion.ddpm_inversion import inversion_forward_process_ddpm  # noqa: E402
from hedit.inversion.p2p_h_edit import (h_Edit_p2p_explicit, h_Edit_p2p_implicit, h_Edit_R_explicit,  # noqa: E402
                                        h_Edit_R_implicit)
from hedit.p2p.ptp_classes import AttentionStore, load_512  # noqa: E402
from hedit.p2p.ptp_controller_utils import make_controller  # noqa: E402
from hedit.p2p.ptp_utils import register_attention_control  # noqa: E402
from hedit.pipeline import HEditPipeline  # noqa: E402
from hedit.scheduler import DDIMScheduler  # noqa: E402
from hedit.utils import image_grid  # noqa: E402

# entries for which the reference switches to the Replace controller when source and target have
# the same number of words (main_p2p.py:183-186)
_REPLACE_KEYS_DDIM = {'111000000001', '111000000004', '111000000009', '121000000007', '122000000006',
                      '121000000000', '121000000001'}
_REPLACE_KEYS_DDPM = {'122000000005', '122000000006', '000000000099', '214000000009'}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument('--data_path', type=str, default="./PIE_Bench_Data")
    p.add_argument('--output_path', type=str, default="./results/p2p")
    p.add_argument('--edit_category_list', nargs='+', type=str, default=[str(i) for i in range(10)])
    p.add_argument("--mode", default="h_edit_R_p2p", help="modes: h_edit_R, h_edit_D_p2p, h_edit_R_p2p")
    p.add_argument("--num_diffusion_steps", type=int, default=50)
    p.add_argument("--skip", type=int, default=0)
    p.add_argument("--eta", type=float, default=1.0)
    p.add_argument("--cfg_src", type=float, default=1.0)
    p.add_argument("--cfg_src_edit", type=float, default=5.0)
    p.add_argument("--cfg_tar", type=float, default=7.5)
    p.add_argument("--implicit", action='store_true', help="Use implicit form of h-Edit")
    p.add_argument("--optimization_steps", type=int, default=1)
    p.add_argument("--weight_reconstruction", type=float, default=0.1)
    p.add_argument("--xa", type=float, default=0.4)
    p.add_argument("--sa", type=float, default=0.35)
    # additions of this build
    p.add_argument("--model_path", type=str, default=None, help="local SD-1.x checkpoint directory (diffusers layout)")
    p.add_argument("--random_init", action="store_true", help="synthetic SD-1.x-shaped weights (no checkpoint)")
    p.add_argument("--tiny", action="store_true", help="with --random_init: the small test configuration")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--batch", type=int, default=1, help="dataset entries edited in lock-step per pass")
    return p


def load_model(args, device):
    if args.random_init:
        if args.tiny:
            from hedit.unet import TINY_CONFIG
            from hedit.vae import TINY_VAE_CONFIG
            vcfg = dict(TINY_VAE_CONFIG)
            vcfg.update(block_out_channels=(64, 64, 128, 128))          # f = 8 like SD
            return HEditPipeline.from_random(TINY_CONFIG, seed=args.seed, device=device, text_layers=2, vae_config=vcfg)
        return HEditPipeline.from_random(seed=args.seed, device=device, with_vae=True)
    if not args.model_path:
        raise SystemExit("give --model_path DIR (local diffusers-layout checkpoint) or --random_init")
    return HEditPipeline.from_pretrained(args.model_path, device=device)


def edit_group(args, model, entries, scale, size, device):
    """--batch N: the n entries of one group in lock-step on hedit.engine.HEditEngine -- VAE encode, inversion
    (2n-row UNet calls), the loop (4n / 5n-row calls, one controller per image in a ControllerBatch) and VAE decode.
    entries: [(key, item, image_path, save_path)].  Same per-image arithmetic as the single-image path."""
    from hedit.engine import HEditEngine
    from hedit.p2p.ptp_classes import ControllerBatch
    eng = HEditEngine(model)
    n = len(entries)
    eta = args.eta
    is_ddim_inversion = eta == 0
    if is_ddim_inversion:
        model.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                        clip_sample=False, set_alpha_to_one=False)
    model.scheduler.config.timestep_spacing = "leading"
    model.scheduler.set_timesteps(args.num_diffusion_steps)
    T = args.num_diffusion_steps
    xs, src_p, tar_p, ctrls = [], [], [], []
    after_skip_steps = T - args.skip
    for key, item, image_path, _ in entries:
        x0 = load_512(image_path, 0, 0, 0, 0, device)
        if x0.shape[-1] != size:
            x0 = torch.nn.functional.interpolate(x0, size=(size, size), mode="bilinear", align_corners=False)
        xs.append(x0)
        original_prompt = item["original_prompt"].replace("[", "").replace("]", "")
        editing_prompt = item["editing_prompt"].replace("[", "").replace("]", "")
        src_p.append(original_prompt)
        tar_p.append(editing_prompt)
        blended_word = item["blended_word"].split(" ") if item["blended_word"] != "" else []
        same_len = len(original_prompt.split(" ")) == len(editing_prompt.split(" "))
        replace = same_len and key in (_REPLACE_KEYS_DDIM if is_ddim_inversion else _REPLACE_KEYS_DDPM)
        if "_replace" in item:            # (the demo driver's rule: main_demo.py)
            replace = item["_replace"]
        if args.mode.endswith('p2p'):
            blend_word = ((blended_word[0],), (blended_word[1],)) if len(blended_word) else None
            eq_val = 1.25 if args.optimization_steps > 1 else 2.0
            eq_params = {"words": (blended_word[1],), "values": (eq_val,)} if len(blended_word) else None
            extra = item.get("_eq_extra")
            if extra is not None:
                eq_params = extra if eq_params is None else {"words": eq_params["words"] + extra["words"],
                                                             "values": eq_params["values"] + extra["values"]}
            ctrls.append(make_controller(prompts=[original_prompt, editing_prompt], is_replace_controller=replace,
                                         cross_replace_steps=args.xa, self_replace_steps=args.sa, blend_word=blend_word,
                                         equilizer_params=eq_params, num_steps=after_skip_steps, tokenizer=model.tokenizer,
                                         device=model.device))
    w0 = (model.vae.encode(torch.cat(xs)).latent_dist.mode() * scale).float()
    if is_ddim_inversion:
        _, zs, wts = eng.ddim_inversion(w0, src_p, args.cfg_src)
        eta = 1.0
    elif 0 < eta <= 1:
        zs, wts = eng.ddpm_inversion(w0, src_p, eta=eta, cfg_src=args.cfg_src)
    else:
        raise SystemExit("Warning: out of range for eta")
    p2p = args.mode.endswith('p2p')
    controller = ControllerBatch(ctrls) if p2p else AttentionStore()
    register_attention_control(model, controller)
    edited, _ = eng.run(wts[after_skip_steps].contiguous(), zs[:after_skip_steps].contiguous(), [[a, b] for a, b in zip(src_p, tar_p)],
                        [args.cfg_src, args.cfg_src_edit, args.cfg_tar], controller, eta=eta, p2p=p2p, implicit=args.implicit,
                        K=args.optimization_steps, w_rec=args.weight_reconstruction, after_skip_steps=after_skip_steps,
                        ddim_inv=is_ddim_inversion, fuse_src_pass=p2p and args.implicit)
    x0_dec = model.vae.decode(1 / scale * edited).sample
    out = []
    for i, (_, _, _, save_path) in enumerate(entries):
        os.makedirs(os.path.dirname(save_path), exist_ok=True)
        image_grid(x0_dec[i:i + 1]).save(save_path)
        out.append(save_path)
    model.unet.zero_grad()
    return out


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.mode == "h_edit_D_p2p":
        assert args.eta == 0.0, "eta should be 0.0 for h-Edit-D"
    elif args.mode in ("h_edit_R", "h_edit_R_p2p"):
        assert args.eta == 1.0, "eta should be 1.0 for h-Edit-R"
    else:
        raise NotImplementedError(f"mode {args.mode}: only the h-Edit modes are built (h_edit_R, h_edit_R_p2p, h_edit_D_p2p)")
    print(f'Arguments: {args}')

    rank, world, local_rank = D.env_rank_world()
    device = f"cuda:{local_rank if world > 1 else args.device_num}"
    torch.cuda.set_device(device)
    D.init_from_env(device)      # several ranks: RCCL group; rank 0 reads the checkpoints and broadcasts them
    data_path, output_path = args.data_path, args.output_path
    with open(os.path.join(data_path, 'mapping_file.json')) as f:
        full_data = json.load(f)
    time_stamp = calendar.timegm(time.gmtime())

    xa_sa_string = f'_xa_{args.xa}_sa{args.sa}_' if args.mode in ('h_edit_D_p2p', 'h_edit_R_p2p') else '_'
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
    mine = D.shard(len(keys), rank, world)
    if args.batch > 1:
        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + xa_sa_string)
        for lo in range(0, len(mine), args.batch):
            entries = []
            for idx in mine[lo:lo + args.batch]:
                item = full_data[keys[idx]]
                image_path = os.path.join(f"{data_path}/annotation_images", item["image_path"])
                entries.append((keys[idx], item, image_path, image_path.replace(data_path, os.path.join(output_path, sub))))
            written += edit_group(args, model, entries, scale, size, device)
        print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
        return written
    for idx in mine:
        key = keys[idx]
        item = full_data[key]
        eta = args.eta
        is_ddim_inversion = eta == 0
        original_prompt = item["original_prompt"].replace("[", "").replace("]", "")
        editing_prompt = item["editing_prompt"].replace("[", "").replace("]", "")
        image_path = os.path.join(f"{data_path}/annotation_images", item["image_path"])
        blended_word = item["blended_word"].split(" ") if item["blended_word"] != "" else []
        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + xa_sa_string)
        save_path = image_path.replace(data_path, os.path.join(output_path, sub))
        os.makedirs(os.path.dirname(save_path), exist_ok=True)

        # scheduler (main_p2p.py:139-146): explicit SD betas for DDIM inversion, the checkpoint's otherwise
        if is_ddim_inversion:
            model.scheduler = DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                            clip_sample=False, set_alpha_to_one=False)
        model.scheduler.config.timestep_spacing = "leading"
        model.scheduler.set_timesteps(args.num_diffusion_steps)

        x0 = load_512(image_path, 0, 0, 0, 0, device)
        if x0.shape[-1] != size:          # the tiny test configuration works at a smaller resolution
            x0 = torch.nn.functional.interpolate(x0, size=(size, size), mode="bilinear", align_corners=False)
        w0 = (model.vae.encode(x0).latent_dist.mode() * scale).float()

        if is_ddim_inversion:
            wt, zs, wts = ddim_inversion(model, w0, original_prompt, args.cfg_src)
            eta = 1.0                     # accounts for u_t^orig (main_p2p.py:165)
        elif 0 < eta <= 1:
            wt, zs, wts, _ = inversion_forward_process_ddpm(model, w0, etas=eta, prompt=original_prompt,
                                                            cfg_scale_src=args.cfg_src,
                                                            num_inference_steps=args.num_diffusion_steps)
        else:
            raise SystemExit("Warning: out of range for eta")

        after_skip_steps = args.num_diffusion_steps - args.skip
        same_len = len(original_prompt.split(" ")) == len(editing_prompt.split(" "))
        replace = same_len and key in (_REPLACE_KEYS_DDIM if is_ddim_inversion else _REPLACE_KEYS_DDPM)
        if "_replace" in item:            # (the demo driver's rule: main_demo.py)
            replace = item["_replace"]
        prompts = [original_prompt, editing_prompt]
        if args.mode.endswith('p2p'):
            blend_word = ((blended_word[0],), (blended_word[1],)) if len(blended_word) else None
            eq_val = 1.25 if args.optimization_steps > 1 else 2.0
            eq_params = {"words": (blended_word[1],), "values": (eq_val,)} if len(blended_word) else None
            controller = make_controller(prompts=prompts, is_replace_controller=replace,
                                         cross_replace_steps=args.xa, self_replace_steps=args.sa,
                                         blend_word=blend_word, equilizer_params=eq_params, num_steps=after_skip_steps,
                                         tokenizer=model.tokenizer, device=model.device)
        else:
            controller = AttentionStore()
        register_attention_control(model, controller)

        kw = dict(xT=wts[after_skip_steps], eta=eta, prompts=prompts, cfg_scales=[args.cfg_src, args.cfg_src_edit, args.cfg_tar],
                  prog_bar=True, zs=zs[:after_skip_steps], controller=controller, after_skip_steps=after_skip_steps,
                  is_ddim_inversion=is_ddim_inversion)
        if args.implicit:
            fn = h_Edit_R_implicit if args.mode == 'h_edit_R' else h_Edit_p2p_implicit
            edited_w0, _ = fn(model, weight_reconstruction=args.weight_reconstruction,
                              optimization_steps=args.optimization_steps, **kw)
        else:
            fn = h_Edit_R_explicit if args.mode == 'h_edit_R' else h_Edit_p2p_explicit
            edited_w0, _ = fn(model, **kw)

        x0_dec = model.vae.decode(1 / scale * edited_w0).sample
        if x0_dec.dim() < 4:
            x0_dec = x0_dec[None]
        image_grid(x0_dec).save(save_path)
        model.unet.zero_grad()
        written.append(save_path)
    print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
    return written


if __name__ == "__main__":
    main()
