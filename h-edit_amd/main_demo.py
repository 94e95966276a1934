#!/usr/bin/env python3
"""Demo driver on the HIP path: same flags, dataset format (``<data_path>/demo.yaml``: a list of
``image / source_prompt / target_prompt / blended_word / editing_instruction`` entries) and output-path
scheme as the reference's ``text-guided/main_demo.py`` (:49-84 flags, :113-118 strings, :124-262 loop)
for the h-Edit modes (h_edit_R, h_edit_R_p2p, h_edit_D_p2p; explicit or ``--implicit``).  Unlike
main_p2p.py it picks the Replace controller whenever source and target have the same number of words
(:181) and merges the heuristic equaliser of ``preprocessing`` with the dataset's blended word (:196-213).

Differences, all additive:
  * ``--model_path DIR``: a LOCAL checkpoint directory in the diffusers layout (no network here);
    ``--random_init`` builds SD-1.x-shaped synthetic weights instead (``--tiny`` = the small test
    configuration at 256x256).
  * launched under ``torch.distributed.run`` the dataset entries are sharded across the ranks
    (one process per GPU, no data-path collective).
The comparison baselines of the reference driver (ef, ef_p2p, nmg_p2p, pnp_inv_p2p) are not part of
this build and are refused.
"""
import argparse
import calendar
import json  # noqa: F401
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
from hedit.p2p.ptp_controller_utils import make_controller, preprocessing  # noqa: E402
from hedit.p2p.ptp_utils import register_attention_control  # noqa: E402
from hedit.pipeline import HEditPipeline  # noqa: E402
from hedit.scheduler import DDIMScheduler  # noqa: E402
from hedit.utils import image_grid  # noqa: E402

def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument('--data_path', type=str, default="./assets/demo")
    p.add_argument('--output_path', type=str, default="./results/demo")
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
    p.add_argument("--batch", type=int, default=1, help="demo entries edited in lock-step per pass (batched engine)")
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
    import yaml
    with open(data_path + "/demo.yaml") as f:
        full_data = yaml.safe_load(f)
    time_stamp = calendar.timegm(time.gmtime())

    xa_sa_string = f'_xa_{args.xa}_sa{args.sa}_' if args.mode in ('h_edit_D_p2p', 'h_edit_R_p2p') else '_'
    weight_string = (f'implicit_{args.implicit}_eta_{args.eta}_src_orig_{args.cfg_src}_src_edit_{args.cfg_src_edit}'
                     f'_tar_scale_{args.cfg_tar}_w_rec_{args.weight_reconstruction}_n_opts_{args.optimization_steps}'
                     f'_time_{time_stamp}')
    model = load_model(args, device)
    prescan_prompts(model.tokenizer, (full_data.values() if isinstance(full_data, dict) else full_data))      # (stand-in tokenizer only: word ids independent of order / shard)
    if model.vae is None:
        raise SystemExit("the checkpoint has no vae/ sub-folder: images cannot be encoded / decoded")
    scale = model.vae.config["scaling_factor"]
    size = model.unet.sample_size * model.vae.factor

    written = []
    if args.batch > 1:
        # lock-step groups on the batched engine (main_p2p.edit_group), with this driver's replace / equalizer rules
        from main_p2p import edit_group
        mine = list(D.shard(len(full_data), rank, world))
        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + xa_sa_string)
        for lo in range(0, len(mine), args.batch):
            entries = []
            for idx in mine[lo:lo + args.batch]:
                it = full_data[idx]
                src = it.get("source_prompt", "").replace("[", "").replace("]", "")
                tar = it.get("target_prompt", "").replace("[", "").replace("]", "")
                _, eq_heuristic = preprocessing(src, tar, is_global_edit=True)
                item = {"original_prompt": src, "editing_prompt": tar, "blended_word": it["blended_word"],
                        "_replace": len(src.split(" ")) == len(tar.split(" ")), "_eq_extra": eq_heuristic}
                image_path = data_path + it["image"]
                entries.append((str(idx), item, image_path, image_path.replace(data_path, os.path.join(output_path, sub))))
            written += edit_group(args, model, entries, scale, size, device)
        print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
        return written
    for idx in D.shard(len(full_data), rank, world):
        item = full_data[idx]
        eta = args.eta
        is_ddim_inversion = eta == 0
        original_prompt = item.get("source_prompt", "").replace("[", "").replace("]", "")
        editing_prompt = item.get("target_prompt", "").replace("[", "").replace("]", "")
        image_path = data_path + item["image"]
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
        replace = len(original_prompt.split(" ")) == len(editing_prompt.split(" "))
        prompts = [original_prompt, editing_prompt]
        if args.mode.endswith('p2p'):
            _, eq_heuristic = preprocessing(original_prompt, editing_prompt, is_global_edit=True)
            blend_word = ((blended_word[0],), (blended_word[1],)) if len(blended_word) else None
            eq_val = 1.25 if args.optimization_steps > 1 else 2.0
            eq_params = {"words": (blended_word[1],), "values": (eq_val,)} if len(blended_word) else None
            if eq_heuristic is not None:
                eq_params = eq_heuristic if eq_params is None else {
                    "words": eq_params["words"] + eq_heuristic["words"], "values": eq_params["values"] + eq_heuristic["values"]}
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
