#!/usr/bin/env python3
"""Combined text-guided + style editing driver on the HIP path: same flags, dataset format
(``<dataset>/demo.json`` with ``image_path / original_prompt / editing_prompt / blended_word / style``)
and output-path scheme as the reference's ``text-guided-n-style/main_edit.py`` (:33-73 flags,
:91-99 strings, :103-247 loop) for its h-Edit mode (``h_edit_R_p2p``).

Differences, all additive:
  * ``--model_path DIR``: LOCAL Stable-Diffusion checkpoint directory (diffusers layout);
    ``--clip_path FILE``: LOCAL OpenAI CLIP ViT-B/16 checkpoint (the reference downloads both);
    ``--random_init`` builds SD-1.x- / ViT-B/16-shaped synthetic weights (``--tiny``: small test sizes).
  * under ``torch.distributed.run`` the dataset entries are sharded across the ranks, one process per
    GPU, no data-path collective (BASELINE configs[4]).
The ``ef_p2p`` comparison baseline of the reference driver is not part of this build and is refused.
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
from hedit.clip_guidance import CLIPEncoder  # noqa: E402
from hedit.inversion.ddpm_inversion import inversion_forward_process_ddpm  # noqa: E402
from hedit.inversion.h_edit import h_Edit_p2p_implicit  # noqa: E402
from hedit.p2p.ptp_classes import AttentionStore, load_512  # noqa: E402
from hedit.p2p.ptp_controller_utils import make_controller  # noqa: E402
from hedit.p2p.ptp_utils import register_attention_control  # noqa: E402
from hedit.utils import dataset_from_json, image_grid  # noqa: E402
from main_p2p import load_model  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument('--dataset', type=str, default="./assets/demo/")
    p.add_argument('--output_path', type=str, default="./results/demo/")
    p.add_argument("--mode", default="h_edit_R_p2p", help="modes: h_edit_R_p2p")
    p.add_argument("--num_diffusion_steps", type=int, default=50)
    p.add_argument("--skip", type=int, default=0)
    p.add_argument("--eta", type=float, default=1.0)
    p.add_argument("--cfg_src", type=float, default=1.0)
    p.add_argument("--cfg_src_edit", type=float, default=5.0)
    p.add_argument("--cfg_tar", type=float, default=7.5)
    p.add_argument("--implicit", action='store_false', help="Use implicit form of h-Edit")
    p.add_argument("--optimization_steps", type=int, default=1)
    p.add_argument("--xa", type=float, default=0.4)
    p.add_argument("--sa", type=float, default=0.35)
    p.add_argument("--weight_edit_clip", type=float, default=0.5)
    p.add_argument("--weight_edit_clip_for_ef", type=float, default=1.5)
    # additions of this build
    p.add_argument("--model_path", type=str, default=None, help="local SD-1.x checkpoint directory (diffusers layout)")
    p.add_argument("--clip_path", type=str, default=None, help="local OpenAI CLIP ViT-B/16 checkpoint (.pt)")
    p.add_argument("--random_init", action="store_true", help="synthetic SD-1.x / ViT-B/16-shaped weights")
    p.add_argument("--tiny", action="store_true", help="with --random_init: the small test configuration")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--batch", type=int, default=1, help="dataset entries edited in lock-step per pass (one style encoder each)")
    return p


def load_style_encoder(args, ref_path, device):
    if args.clip_path:
        return CLIPEncoder(need_ref=True, ref_path=ref_path, clip_path=args.clip_path, device=device)
    if not args.random_init:
        raise SystemExit("give --clip_path FILE (local OpenAI CLIP ViT-B/16 checkpoint) or --random_init")
    if args.tiny:
        from hedit.clip_guidance.base_clip import ClipVisualPrefix
        m = ClipVisualPrefix(width=64, layers=3, heads=1, patch_size=32, input_resolution=224).init_random(args.seed)
        return CLIPEncoder(need_ref=True, ref_path=ref_path, clip_model=m.half(), device=device)
    return CLIPEncoder(need_ref=True, ref_path=ref_path, device=device, seed=args.seed)


def edit_group(args, model, entries, scale, size, device):
    """--batch N: n entries in lock-step on the batched engine; every image keeps its own style reference
    (CLIPEncoder.get_gram_matrix_residual reads batch item 0 only, clip_guidance/base_clip.py:60-65) on ONE copy of the
    encoder (CLIPEncoder.sibling).  entries: [(item, image_path, save_path)]."""
    from hedit.engine import HEditEngine
    from hedit.p2p.ptp_classes import ControllerBatch
    eng = HEditEngine(model)
    model.scheduler.config.timestep_spacing = "leading"
    model.scheduler.set_timesteps(args.num_diffusion_steps)
    after_skip_steps = args.num_diffusion_steps - args.skip
    xs, pairs, ctrls, encs = [], [], [], []
    for item, image_path, _ in entries:
        original_prompt = item["original_prompt"].replace("[", "").replace("]", "")
        editing_prompt = item["editing_prompt"].replace("[", "").replace("]", "")
        blended_word = item["blended_word"].split(" ") if item["blended_word"] != "" else []
        encs.append(encs[0].sibling(args.dataset + item['style']) if encs else load_style_encoder(args, args.dataset + item['style'], device))
        x0 = load_512(image_path, 0, 0, 0, 0, device)
        if x0.shape[-1] != size:
            x0 = torch.nn.functional.interpolate(x0, size=(size, size), mode="bilinear", align_corners=False)
        xs.append(x0)
        pairs.append([original_prompt, editing_prompt])
        same_len = len(original_prompt.split(" ")) == len(editing_prompt.split(" "))
        eq_params = {"words": (blended_word[1],), "values": (2.0,)} if len(blended_word) else None
        ctrls.append(make_controller(prompts=pairs[-1], is_replace_controller=same_len, cross_replace_steps=args.xa,
                                     self_replace_steps=args.sa, blend_word=None, equilizer_params=eq_params,
                                     num_steps=after_skip_steps, tokenizer=model.tokenizer, device=model.device))
    w0 = (model.vae.encode(torch.cat(xs)).latent_dist.mode() * scale).float()
    zs, wts = eng.ddpm_inversion(w0, [p[0] for p in pairs], eta=args.eta, cfg_src=args.cfg_src)
    controller = ControllerBatch(ctrls)
    register_attention_control(model, controller)
    edited, _ = eng.run(wts[after_skip_steps].contiguous(), zs[:after_skip_steps].contiguous(), pairs,
                        [args.cfg_src, args.cfg_src_edit, args.cfg_tar], controller, eta=args.eta, p2p=True, implicit=True,
                        K=args.optimization_steps, after_skip_steps=after_skip_steps, ddim_inv=False, fuse_src_pass=True,
                        style=(encs, args.weight_edit_clip))
    out = []
    with torch.no_grad():
        x0_dec = model.vae.decode(1 / scale * edited).sample
        for i, (_, _, save_path) in enumerate(entries):
            print(f'loss from CLIP: {torch.linalg.norm(encs[i].get_gram_matrix_residual(x0_dec[i:i + 1])).item()}')
            os.makedirs(os.path.dirname(save_path), exist_ok=True)
            image_grid(x0_dec[i:i + 1]).save(save_path)
            out.append(save_path)
    return out


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.eta == 1.0, "eta should be set to 1.0 for this experiment"
    assert args.optimization_steps == 1, "we have not tested multiple optimization steps for this experiment"
    assert args.implicit, "we only demo the implicit form for this experiment"
    if args.mode != 'h_edit_R_p2p':
        raise NotImplementedError(f"mode {args.mode}: only h_edit_R_p2p is built")
    print(f'Arguments: {args}')

    rank, world, local_rank = D.env_rank_world()
    device = f"cuda:{local_rank if world > 1 else args.device_num}"
    torch.cuda.set_device(device)
    D.init_from_env(device)      # several ranks: RCCL group; rank 0 reads the checkpoints and broadcasts them
    full_data = dataset_from_json(args.dataset + "demo.json")
    time_stamp = calendar.timegm(time.gmtime())
    xa_sa_string = f'_xa_{args.xa}_sa{args.sa}_'
    weight_string = (f'implicit_{args.implicit}_eta_{args.eta}_src_orig_{args.cfg_src}_src_edit_{args.cfg_src_edit}'
                     f'_tar_scale_{args.cfg_tar}_w_style_{args.weight_edit_clip}_n_opts_{args.optimization_steps}'
                     f'_time_{time_stamp}')
    model = load_model(args, device)
    prescan_prompts(model.tokenizer, (full_data.values() if isinstance(full_data, dict) else full_data))      # (stand-in tokenizer only: word ids independent of order / shard)
    if model.vae is None:
        raise SystemExit("the checkpoint has no vae/ sub-folder: the style closure decodes through it")
    scale = model.vae.config["scaling_factor"]
    size = model.unet.sample_size * model.vae.factor

    keys = list(full_data.keys())
    written = []
    mine = D.shard(len(keys), rank, world)
    if args.batch > 1:
        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + xa_sa_string)
        for lo in range(0, len(mine), args.batch):
            entries = []
            for idx in mine[lo:lo + args.batch]:
                item = full_data[keys[idx]]
                image_path = args.dataset + item['image_path']
                entries.append((item, image_path, image_path.replace(args.dataset, os.path.join(args.output_path, sub))))
            written += edit_group(args, model, entries, scale, size, device)
        print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
        return written
    for idx in mine:
        item = full_data[keys[idx]]
        eta = args.eta
        image_path = args.dataset + item['image_path']
        original_prompt = item["original_prompt"].replace("[", "").replace("]", "")
        editing_prompt = item["editing_prompt"].replace("[", "").replace("]", "")
        blended_word = item["blended_word"].split(" ") if item["blended_word"] != "" else []
        image_encoder = load_style_encoder(args, args.dataset + item['style'], device)

        model.scheduler.config.timestep_spacing = "leading"
        model.scheduler.set_timesteps(args.num_diffusion_steps)
        x0 = load_512(image_path, 0, 0, 0, 0, device)
        if x0.shape[-1] != size:
            x0 = torch.nn.functional.interpolate(x0, size=(size, size), mode="bilinear", align_corners=False)
        w0 = (model.vae.encode(x0).latent_dist.mode() * scale).float()
        wt, zs, wts, _ = inversion_forward_process_ddpm(model, w0, etas=eta, prompt=original_prompt,
                                                        cfg_scale_src=args.cfg_src,
                                                        num_inference_steps=args.num_diffusion_steps)

        sub = (args.mode + '_total_steps_' + str(args.num_diffusion_steps) + '_skip_' + str(args.skip) + '_' +
               weight_string + xa_sa_string)
        save_path = image_path.replace(args.dataset, os.path.join(args.output_path, sub))
        os.makedirs(os.path.dirname(save_path), exist_ok=True)

        after_skip_steps = args.num_diffusion_steps - args.skip
        same_len = len(original_prompt.split(" ")) == len(editing_prompt.split(" "))
        prompts = [original_prompt, editing_prompt]
        # main_edit.py:185-214: no local blend, no heuristic equaliser for the combined task; the dataset's
        # blended word is re-weighted by 2.0
        eq_params = {"words": (blended_word[1],), "values": (2.0,)} if len(blended_word) else None
        controller = make_controller(prompts=prompts, is_replace_controller=same_len, cross_replace_steps=args.xa,
                                     self_replace_steps=args.sa, blend_word=None, equilizer_params=eq_params,
                                     num_steps=after_skip_steps, tokenizer=model.tokenizer, device=model.device)
        register_attention_control(model, controller)

        edited_w0, _ = h_Edit_p2p_implicit(model, image_encoder=image_encoder, xT=wts[after_skip_steps], eta=eta,
                                           prompts=prompts, cfg_scales=[args.cfg_src, args.cfg_src_edit, args.cfg_tar],
                                           prog_bar=True, zs=zs[:after_skip_steps], controller=controller,
                                           weight_edit_clip=args.weight_edit_clip,
                                           optimization_steps=args.optimization_steps,
                                           after_skip_steps=after_skip_steps, is_ddim_inversion=False)
        with torch.no_grad():
            x0_dec = model.vae.decode(1 / scale * edited_w0).sample
            if x0_dec.dim() < 4:
                x0_dec = x0_dec[None]
            loss_from_clip = torch.linalg.norm(image_encoder.get_gram_matrix_residual(x0_dec))
        print(f'loss from CLIP: {loss_from_clip.item()}')
        image_grid(x0_dec).save(save_path)
        written.append(save_path)
    print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
    return written


if __name__ == "__main__":
    main()
