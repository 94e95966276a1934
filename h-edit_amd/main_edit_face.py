#!/usr/bin/env python3
"""Face-swapping driver on the HIP path: same flags, dataset format (``--json_file``: a list of ``{idx, source, ref}``)
and output naming as the reference's ``face-swapping/main_edit.py`` (:35-60 flags, :134-224 loop) for its h-Edit mode
(``h_edit_R``): pixel DDPM UNet on the HIP executor, SDE inversion, h_Edit_R with the ArcFace identity reward.

Differences, all additive:
  * ``--ddpm_ckpt`` / ``--arcface_ckpt``: LOCAL checkpoint files (the reference hard-codes ./diffusion/weights/celeba_hq.ckpt
    and ./arcface/weights/model_ir_se50.pth); ``--random_init`` (``--tiny``) = synthetic weights.
  * the LPIPS term needs the third-party ``lpips`` package: used when importable, otherwise skipped (the loop guards
    ``lpipsloss=None`` exactly like the reference, h_edit_R.py:124).
  * the post-processing face mask comes from the reference's face-parsing network (BiSeNet checkpoint), which is not
    part of this build: pass ``--mask_dir DIR`` with precomputed label images ``<source stem>.png`` (face-parsing class
    ids) to enable it; without it the result is saved unblended.
  * under torch.distributed.run the pairs are sharded over the ranks (one process per GPU).
The ``ef`` baseline mode is refused."""
import argparse
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

from hedit import dist as D  # noqa: E402
from hedit.arcface import IDLoss  # noqa: E402
from hedit.arcface.arcface_model import load_face_image  # noqa: E402
from hedit.arcface.face_utils import SoftErosion, encode_segmentation  # noqa: E402
from hedit.diffusion import Model, TINY_DDPM_CONFIG  # noqa: E402
from hedit.inversion.h_edit_R import h_Edit_R  # noqa: E402
from hedit.inversion.sde_inversion import inversion_forward_process_sde  # noqa: E402
from hedit.utils import image_grid  # noqa: E402


def get_source_ref_paths(json_file):
    with open(json_file, 'r') as f:
        for pair in json.load(f):
            yield pair['idx'], pair['source'], pair['ref']


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--device_num", type=int, default=0)
    p.add_argument("--json_file", type=str, default="./assets/demo/demo.json")
    p.add_argument("--image_path", type=str, default="./assets/demo/")
    p.add_argument('--output_path', type=str, default="./results/demo/")
    p.add_argument("--mode", default="h_edit_R", help="modes: h_edit_R")
    p.add_argument("--num_diffusion_steps", type=int, default=100)
    p.add_argument("--skip", type=int, default=0)
    p.add_argument("--eta", type=float, default=1.0)
    p.add_argument("--optimization_steps", type=int, default=3)
    p.add_argument("--post_processing", action='store_false', help="Apply mask as post-processing")
    p.add_argument("--weight_edit_face", type=float, default=50.0)
    # additions of this build
    p.add_argument("--ddpm_ckpt", type=str, default=None, help="local CelebA-HQ DDPM checkpoint (state_dict)")
    p.add_argument("--arcface_ckpt", type=str, default=None, help="local model_ir_se50.pth")
    p.add_argument("--mask_dir", type=str, default=None, help="precomputed face-parsing label images (<source stem>.png)")
    p.add_argument("--random_init", action="store_true")
    p.add_argument("--tiny", action="store_true", help="with --random_init: 32 x 32 toy UNet")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--lpips_ckpt", type=str, default=None, help="local file with a saved lpips.LPIPS(net='vgg').state_dict()")
    p.add_argument("--no_lpips", action="store_true", help="drop the LPIPS term (lpipsloss=None, which the loop guards)")
    p.add_argument("--batch", type=int, default=1, help="(source, reference) pairs swapped in lock-step per pass")
    return p


def linear_betas(device):
    return torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float64)).float().to(device)


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.eta == 1.0, "eta should be set to 1.0 for this experiment"
    if args.mode != "h_edit_R":
        raise NotImplementedError(f"mode {args.mode}: only h_edit_R is built")
    rank, world, local_rank = D.env_rank_world()
    device = f"cuda:{local_rank if world > 1 else args.device_num}"
    torch.cuda.set_device(device)
    D.init_from_env(device)      # several ranks: RCCL group; rank 0 reads the checkpoints and broadcasts them
    if args.random_init:
        model = Model(TINY_DDPM_CONFIG if args.tiny else None, device=device)
        model.init_random(args.seed)
    elif args.ddpm_ckpt:
        model = Model(device=device)

        def read_ddpm():
            states = torch.load(args.ddpm_ckpt, map_location="cpu")
            if isinstance(states, list):                     # [model, ..., ema]: DataParallel-prefixed first entry
                states = {k[7:]: v for k, v in states[0].items()}
            return states
        model.load_state_dict(D.state_dict_from_rank0(read_ddpm, model.param_shapes, device=device))
    else:
        raise SystemExit("give --ddpm_ckpt FILE (local CelebA-HQ DDPM checkpoint) or --random_init")
    S = model.resolution
    betas = linear_betas(device)
    skip_per_step = betas.shape[0] // args.num_diffusion_steps
    seq = (np.arange(0, betas.shape[0], skip_per_step) + 1)[::-1]
    from hedit.arcface.lpips_loss import LPIPS_Loss
    have_lpips = not args.no_lpips
    if have_lpips and not args.random_init and not args.lpips_ckpt:
        raise SystemExit("give --lpips_ckpt FILE (saved lpips.LPIPS(net='vgg').state_dict()), or --no_lpips")
    pairs = list(get_source_ref_paths(args.json_file))
    written = []
    mine = D.shard(len(pairs), rank, world)
    if args.batch > 1:
        # --batch N: the inversions run pair by pair (each reseeds with 42 like the reference), the h-Edit loop runs the
        # N pairs in lock-step with one reference face / one source image per batch item (per_image: every pair is
        # swapped exactly as it would be alone)
        after_skip_steps = args.num_diffusion_steps - args.skip
        save_path = args.output_path + (f"{args.mode}/steps_{args.num_diffusion_steps}_skip_{args.skip}_weight_{args.weight_edit_face}"
                                        f"_opts_{args.optimization_steps}")
        os.makedirs(save_path, exist_ok=True)
        for lo in range(0, len(mine), args.batch):
            grp = [pairs[i] for i in mine[lo:lo + args.batch]]
            srcs = [load_face_image(os.path.join(args.image_path, sp), S).to(device) for _, sp, _ in grp]
            refs = [load_face_image(os.path.join(args.image_path, rp), S).to(device) for _, _, rp in grp]
            to256 = lambda t: t if S == 256 else torch.nn.functional.interpolate(t, size=(256, 256), mode="bilinear", align_corners=False)
            idloss = IDLoss(ref=torch.cat([to256(r) for r in refs]), weights=None if args.random_init else args.arcface_ckpt,
                            device=device, seed=args.seed)
            lpipsloss = LPIPS_Loss(src=torch.cat(srcs), weights=None if args.random_init else args.lpips_ckpt, device=device,
                                   seed=args.seed) if have_lpips else None
            zs_l, xs_l = [], []
            for src in srcs:
                _, zs, xts, _ = inversion_forward_process_sde(model, src, betas, seq, etas=args.eta,
                                                              num_inference_steps=args.num_diffusion_steps, device=device)
                zs_l.append(zs[:after_skip_steps])
                xs_l.append(xts[after_skip_steps])
            edited = h_Edit_R(model, lpipsloss, idloss, torch.stack(xs_l), betas, seq, eta=args.eta, zs=torch.stack(zs_l, 1),
                              weight_edit_face=args.weight_edit_face, optimization_steps=args.optimization_steps,
                              after_skip_steps=after_skip_steps, num_inference_steps=args.num_diffusion_steps, soft_face_mask=None,
                              per_image=True).detach()
            with torch.no_grad():
                print(f'Cosine Similarity: {idloss.get_cosine_sim(to256(edited)).mean().item()}')
            for k, (_, sp, rp) in enumerate(grp):
                img = image_grid([refs[k].cpu(), srcs[k].cpu(), edited[k:k + 1].cpu()])
                key = f"{rp.split('/')[-1].split('.')[0]}_{sp.split('/')[-1].split('.')[0]}"
                full = os.path.join(save_path, f'item_{key}.png')
                img.save(full)
                written.append(full)
        print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
        return written
    for i in mine:
        idx, source_path, ref_path = pairs[i]
        source = load_face_image(os.path.join(args.image_path, source_path), S).to(device)
        ref = load_face_image(os.path.join(args.image_path, ref_path), S).to(device)
        ref256 = ref if S == 256 else torch.nn.functional.interpolate(ref, size=(256, 256), mode="bilinear", align_corners=False)
        idloss = IDLoss(ref=ref256, weights=None if args.random_init else args.arcface_ckpt, device=device, seed=args.seed)
        lpipsloss = LPIPS_Loss(src=source, weights=None if args.random_init else args.lpips_ckpt, device=device,
                               seed=args.seed) if have_lpips else None
        save_path = args.output_path + (f"{args.mode}/steps_{args.num_diffusion_steps}_skip_{args.skip}_weight_{args.weight_edit_face}"
                                        f"_opts_{args.optimization_steps}")
        os.makedirs(save_path, exist_ok=True)
        xt, zs, xts, _ = inversion_forward_process_sde(model, source, betas, seq, etas=args.eta,
                                                       num_inference_steps=args.num_diffusion_steps, device=device)
        soft_face_mask = None
        if args.mask_dir:
            from PIL import Image
            lab = np.asarray(Image.open(os.path.join(args.mask_dir, os.path.splitext(os.path.basename(source_path))[0] + ".png")))
            seg = torch.from_numpy(lab.astype(np.int64))[None, None].to(device)
            if seg.shape[-1] != S:
                seg = torch.nn.functional.interpolate(seg.float(), size=(S, S), mode="nearest").long()
            enc = encode_segmentation(seg)
            soft_face_mask, _ = SoftErosion(kernel_size=13, threshold=0.9, iterations=7).to(device)(enc[:, 0, None] + enc[:, 1, None])
        after_skip_steps = args.num_diffusion_steps - args.skip
        edited = h_Edit_R(model, lpipsloss, idloss, xts[after_skip_steps], betas, seq, eta=args.eta, zs=zs[:after_skip_steps],
                          weight_edit_face=args.weight_edit_face, optimization_steps=args.optimization_steps,
                          after_skip_steps=after_skip_steps, num_inference_steps=args.num_diffusion_steps, soft_face_mask=None)
        x0_dec = edited.detach()
        if args.post_processing and soft_face_mask is not None:
            x0_dec = x0_dec * soft_face_mask + source * (1 - soft_face_mask)
        with torch.no_grad():
            x256 = x0_dec if S == 256 else torch.nn.functional.interpolate(x0_dec, size=(256, 256), mode="bilinear", align_corners=False)
            print(f'Cosine Similarity: {idloss.get_cosine_sim(x256).mean().item()}')
        img = image_grid([ref.cpu(), source.cpu(), x0_dec.cpu()])
        key = f"{ref_path.split('/')[-1].split('.')[0]}_{source_path.split('/')[-1].split('.')[0]}"
        full = os.path.join(save_path, f'item_{key}.png')
        img.save(full)
        written.append(full)
    print(f"rank {rank}/{world}: wrote {len(written)} image(s)")
    return written


if __name__ == "__main__":
    main()
