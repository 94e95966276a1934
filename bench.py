#!/usr/bin/env python3
"""bench.py -- h-Edit sampling loop on MI355X (BASELINE.json metric: edited images/s).

    python bench.py --gpus 1 --steps K --warmup W            # one process
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one batch of synthetic input: the complete reverse-time
bridge sampling loop of BASELINE.json configs[1] (text-guided h_Edit_p2p_implicit, SD-1.5-shaped
random-init UNet, 64x64 latents, 50 DDIM steps, K_opt = 1 implicit "Langevin" step, P2P
Replace/Refine + Reweight + LocalBlend) for `--images` independent images per GPU run in
lock-step: 50 x (1 base pass of 4n rows + 1 source pass of n rows + 1 P2P pass of 4n rows) =
450 UNet sample-forwards per image, the reference's evaluations one for one (by default the n source
rows ride along in the P2P pass as un-edited rows; --no-fuse-src issues them as their own call).  Inputs (weights,
inverted latents x_T, noise maps z_t, text embeddings) are resident in HBM before the timed
region; DDPM inversion is outside it (it is the step BEFORE the path).

Multi-GPU: one process per GPU, images sharded across ranks (weak scaling, no data-path
collective); rank 0 creates the weights and broadcasts them over RCCL/xGMI once at start-up.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))

import torch  # noqa: E402

FLOP_PER_SAMPLE_FWD = 0.8032e12       # 401.6 GMAC per UNet sample-forward
MFMA_PEAK_TFLOPS = 2500.0             # dense bf16 MFMA peak, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0

DEMO_PAIRS = [
    # (source, target, (blend_src, blend_tar), replace-controller?)  -- the reference's demo prompts
    ("a green lizard is sitting on a branch", "a brown lizard is sitting on a branch", ("lizard", "lizard"), True),
    ("an orange van with surfboards on top", "an orange van with flowers on top", ("surfboards", "flowers"), True),
    ("a round cake with orange frosting on a wooden plate", "a square cake with orange frosting on a wooden plate",
     ("cake", "cake"), True),
    ("a cat sitting next to a mirror", "a silver cat sculpture sitting next to a mirror", ("cat", "cat"), False),
]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--images", type=int, default=24, help="images edited in lock-step per GPU (one 'step'); 24 = 120-row UNet calls: +5 %% over 8 and +2.5 %% over 16, flat beyond")
    ap.add_argument("--diffusion-steps", type=int, default=50)
    ap.add_argument("--opt-steps", type=int, default=1, help="implicit optimisation ('Langevin') steps K")
    ap.add_argument("--workload", choices=("p2p", "style", "face"), default="p2p",
                    help="p2p = BASELINE configs[1] (the quoted metric, default); style = configs[4], combined "
                         "text + CLIP-style editing: every step adds VAE decode forward + backward and the style encoder")
    ap.add_argument("--storage", choices=("bf16", "f16"), default=None,
                    help="16-bit storage format of activations / weights: bf16 = BASELINE configs[1] and the default line; f16 = the "
                         "half-storage build of the same kernels (libhedit_hip_f16.so, eps error 1.5e-3 instead of 1.2e-2 vs fp32). "
                         "Default: the HEDIT_STORAGE environment variable, else bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prof-every", type=int, default=25, help="bracket every n-th UNet call with HIP events")
    ap.add_argument("--tiny", action="store_true", help="debug: tiny network instead of SD-1.5 shape")
    ap.add_argument("--no-fuse-src", action="store_true",
                    help="issue the source-prompt pass as its own UNet call like the reference instead of "
                         "as n extra rows of the P2P pass (same arithmetic either way)")
    ap.add_argument("--force-dist", action="store_true", help="init the RCCL process group even for one rank")
    ap.add_argument("--no-single", action="store_true", help="skip the auxiliary one-image latency measurement")
    ap.add_argument("--no-half-storage", action="store_true", help="skip the auxiliary half-storage re-run of the loop (a child process)")
    ap.add_argument("--no-config2", action="store_true",
                    help="skip the auxiliary blocks after the timed region: BASELINE configs[2] (32 images, K = 3), configs[3] (face "
                         "swapping, 32 and 8 faces), configs[4] (text + style, 16 images), one pass each")
    ap.add_argument("--reuse-orig-eps", action="store_true",
                    help="opt-in: reuse eps(x_orig, t-1, {null,src}) of the P2P pass in the next base pass "
                         "(7 instead of 9 sample-forwards per step; NOT the reference's evaluation count)")
    return ap.parse_args()


def face_pass(args, rank, dev, dist, n, T, K, steps, warmup):
    """BASELINE configs[3] per GPU: face-swapping h-Edit-R (face-swapping/inversion/h_edit_R.py) -- pixel DDPM UNet
    (HIP, CelebA-HQ 256 shape, random init) guided by the ArcFace identity reward (IR-SE50) and the LPIPS-VGG
    perceptual reward, both native executors (loss + image gradient in one call each), T steps, K implicit steps:
    T + 2 K (T - 1) eps evaluations per image (694 at T = 100, K = 3) and as many reward evaluations; n faces in
    lock-step per GPU, one reference face and one source image per face.  -> the block of the JSON line."""
    import numpy as np
    from hedit.arcface import IDLoss
    from hedit.arcface.lpips_loss import LPIPS_Loss
    from hedit.diffusion import Model, TINY_DDPM_CONFIG
    from hedit.inversion.h_edit_R import h_Edit_R
    model = Model(TINY_DDPM_CONFIG if args.tiny else None, device=dev)
    model.init_random(0)
    S = model.resolution
    g = torch.Generator().manual_seed(5 + rank)
    # one reference face (identity reward) and one source image (LPIPS) per face, both native (csrc/irse.hip, lpips.hip)
    idloss = IDLoss(ref=torch.randn(n, 3, 256, 256, generator=g) * 0.4, device=dev, seed=1)
    lpipsloss = LPIPS_Loss(src=torch.randn(n, 3, S, S, generator=g) * 0.4, device=dev, seed=2) if S % 16 == 0 else None
    betas = torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float64)).float().to(dev)
    seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
    xT = torch.randn(n, 3, S, S, generator=g).to(dev)
    zs = torch.randn(T, n, 3, S, S, generator=g).to(dev)

    def one_step(steps_run=T):
        return h_Edit_R(model, lpipsloss, idloss, xT, betas, seq, eta=1.0, zs=zs[:steps_run], weight_edit_face=50.0, optimization_steps=K,
                        after_skip_steps=steps_run, num_inference_steps=T, per_image=True)

    if warmup == 0:
        one_step(2)          # two sampler steps: handles created, weights packed, kernels loaded
    for _ in range(warmup):
        one_step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    world = 1
    if dist is not None:
        from hedit import dist as HD
        elapsed = HD.max_over_ranks(elapsed, device=dev)
        world = dist.get_world_size()
    # the eps-network alone, same batch, HIP events on the launch stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        model(xT, 501.0)
    ev1.record()
    torch.cuda.synchronize()
    unet_ms = ev0.elapsed_time(ev1) / 10
    evals = T + 2 * K * (T - 1)
    flop_fwd = 0.497e12 if not args.tiny else 0.0
    imgs = steps * n * world
    ach = flop_fwd * n / (unet_ms * 1e-3) / 1e12 if unet_ms > 0 else None
    return {
        "metric": "face-swapped images/sec (256^2, 100 steps, K=3)", "value": round(imgs / elapsed, 4), "unit": "images/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.storage, "data": "synthetic",
        "config": {"workload": "BASELINE configs[3] per GPU: face-swapping h_Edit_R, CelebA-HQ-256-shaped random-init pixel DDPM "
                               f"UNet (113.7M), ArcFace IR-SE50 identity reward + LPIPS-VGG16 reward (native, random init), {T} steps, "
                               f"K={K}; {n} faces per GPU in lock-step",
                   "images_per_gpu": n, "eps_evaluations_per_image": evals, "parallelism": f"replica-dp{world}"},
        "achieved_tflops_per_s_per_gpu": round(imgs * evals * flop_fwd / elapsed / 1e12 / world, 1),
        "ms_per_eps_evaluation_batch": round(unet_ms, 3),
        "unet_share_of_step": round(evals * unet_ms * 1e-3 / (elapsed / steps), 4),
        "roofline": {"bound": "mfma", "kernel": "hedit_ddpm_forward (all kernels of one eps evaluation)", "achieved": None if ach is None else round(ach, 1),
                     "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None if ach is None else round(ach / MFMA_PEAK_TFLOPS, 4),
                     "traffic": None},
        "cpu_baseline": None, "finite": bool(torch.isfinite(out).all()),
    }


def run_face(args, world, rank, local, dev, dist):
    """--workload face: configs[3] as the timed workload (default 32 faces: 0.87 / 1.04 / 1.14 faces/s at 8 / 16 / 32)"""
    n = args.images if args.images != 24 else 32
    T = args.diffusion_steps if args.diffusion_steps != 50 else 100
    K = args.opt_steps if args.opt_steps != 1 else 3
    out_json = face_pass(args, rank, dev, dist, n, T, K, args.steps, args.warmup)
    if rank == 0:
        print(json.dumps(out_json))
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.storage:                        # decided before hedit is first imported: one storage format per process
        os.environ["HEDIT_STORAGE"] = args.storage
    args.storage = os.environ.get("HEDIT_STORAGE", "bf16").lower()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    if args.workload == "face":
        return run_face(args, world, rank, local, dev, dist)

    from hedit.engine import HEditEngine
    from hedit.p2p import ptp_controller_utils as PCU
    from hedit.p2p.ptp_classes import ControllerBatch
    from hedit.p2p.ptp_utils import register_attention_control
    from hedit.pipeline import HEditPipeline
    from hedit.scheduler import DDIMScheduler
    from hedit.text import ClipTextEncoder, WordTokenizer
    from hedit.unet import SD15_CONFIG, TINY_CONFIG, UNet2DConditionModel, random_state_dict

    cfg = dict(TINY_CONFIG if args.tiny else SD15_CONFIG)
    unet = UNet2DConditionModel(cfg, device=dev)

    # ---- weights: rank 0 creates them, RCCL broadcast to the other GPUs (no steady-state traffic)
    t_w0 = time.time()
    sd_cpu = None
    if dist is None:
        sd_cpu = random_state_dict(unet.param_shapes, seed=0)
        unet.load_state_dict(sd_cpu)
    else:
        from hedit import dist as HD
        if rank == 0:
            sd_cpu = random_state_dict(unet.param_shapes, seed=0)
        sd_dev = HD.broadcast_state_dict(unet.param_shapes, sd_cpu, src=0, device=dev, bf16_names=unet.bf16_exact)   # RCCL over xGMI: 1.7 GB bf16 + the fp32 rest
        unet.load_state_dict(sd_dev)
        del sd_dev
        assert unet._lib.hedit_unet_missing(unet._h) == 0
    t_weights = time.time() - t_w0

    tok = WordTokenizer()
    enc = ClipTextEncoder(dim=cfg["cross_attention_dim"], layers=2 if args.tiny else 12,
                          heads=4 if args.tiny else 12, seed=7).to(dev)
    model = HEditPipeline(unet, DDIMScheduler(), tok, enc, None, dev)
    T = args.diffusion_steps
    model.scheduler.set_timesteps(T)
    eng = HEditEngine(model)

    S = cfg["sample_size"]
    cfg_scales = [1.0, 5.0, 7.5]
    with torch.no_grad():
        null = eng.encode([""])
    STYLE_FLOP_PER_INNER_STEP = 2 * (2 * 1.2575e12)      # decoder forward (1257.5 GMAC) + its input-gradient pass

    def make_style():
        from hedit.clip_guidance import CLIPEncoder
        from hedit.clip_guidance.base_clip import ClipVisualPrefix
        from hedit.vae import AutoencoderKL, TINY_VAE_CONFIG
        if model.vae is None:
            model.vae = AutoencoderKL(TINY_VAE_CONFIG if args.tiny else None, device=dev)
            model.vae.init_random(11)
        clip = (ClipVisualPrefix(width=64, layers=3, heads=1, patch_size=32, input_resolution=224) if args.tiny
                else ClipVisualPrefix()).init_random(13).half()
        senc = CLIPEncoder(clip_model=clip, device=dev)
        senc.set_reference(torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(17)).to(dev))
        return (senc, 0.5)

    def build_workload(n, K, seed_off=0, style=None):
        """n images in lock-step with K implicit steps: inverted latents / noise maps / embeddings resident in HBM,
        returns (one_step, w0, t_inversion).  DDPM inversion = the step BEFORE the path, not timed."""
        pairs = [DEMO_PAIRS[(rank * n + i) % len(DEMO_PAIRS)] for i in range(n)]
        prompt_pairs = [[p[0], p[1]] for p in pairs]
        w0 = torch.stack([torch.randn(4, S, S, generator=torch.Generator().manual_seed(1 + seed_off + rank * n + i)) * 0.8
                          for i in range(n)]).to(dev)
        with torch.no_grad():
            src = eng.encode([p[0] for p in prompt_pairs])
            tar = eng.encode([p[1] for p in prompt_pairs])
        g = torch.Generator(device=dev).manual_seed(3 + seed_off + rank)
        t_i0 = time.time()
        zs, xts = eng.ddpm_inversion(w0, [p[0] for p in prompt_pairs], eta=1.0, cfg_src=1.0, generator=g)
        torch.cuda.synchronize()
        t_inv = time.time() - t_i0
        xT = xts[T].contiguous()

        def make_batch_controller():
            ctrls = []
            for (s_, t_, bw, is_replace) in pairs:
                ctrls.append(PCU.make_controller(
                    prompts=[s_, t_], is_replace_controller=is_replace, cross_replace_steps=0.4,
                    self_replace_steps=0.35, blend_word=((bw[0],), (bw[1],)),
                    equilizer_params={"words": (bw[1],), "values": (2.0 if K == 1 else 1.25,)},
                    num_steps=T, tokenizer=tok, device=dev))
            return ControllerBatch(ctrls)

        def one_step(reuse=None):
            cb = make_batch_controller()
            register_attention_control(model, cb)
            return eng.run(xT, zs, prompt_pairs, cfg_scales, cb, eta=1.0, p2p=True, implicit=True, K=K, w_rec=0.1,
                           after_skip_steps=T, ddim_inv=False, ctx=(null, src, tar), fuse_src_pass=not args.no_fuse_src,
                           reuse_orig_eps=args.reuse_orig_eps if reuse is None else reuse, style=style)

        def one_image():
            c1 = PCU.make_controller(prompts=list(prompt_pairs[0]), is_replace_controller=pairs[0][3],
                                     cross_replace_steps=0.4, self_replace_steps=0.35,
                                     blend_word=((pairs[0][2][0],), (pairs[0][2][1],)),
                                     equilizer_params={"words": (pairs[0][2][1],), "values": (2.0 if K == 1 else 1.25,)},
                                     num_steps=T, tokenizer=tok, device=dev)
            register_attention_control(model, c1)
            return eng.run(xT[:1], zs[:, :1].contiguous(), prompt_pairs[:1], cfg_scales, c1, eta=1.0, p2p=True,
                           implicit=True, K=K, w_rec=0.1, after_skip_steps=T, ddim_inv=False,
                           ctx=(null, src[:1], tar[:1]), fuse_src_pass=not args.no_fuse_src,
                           reuse_orig_eps=args.reuse_orig_eps)
        return one_step, one_image, w0, t_inv

    n = args.images
    K = args.opt_steps
    style = make_style() if args.workload == "style" else None
    one_step_raw, one_image, w0, t_inversion = build_workload(n, K, style=style)

    # sampled launch timing: bracket every prof_every-th UNet call with HIP event pairs
    calls = {"n": 0}
    raw = unet.forward_raw

    def forward_sampled(*a, **k):
        calls["n"] += 1
        on = calls["prof"] and (calls["n"] % args.prof_every == 0)
        if on:
            unet.prof_enable(True, 16384)
        out = raw(*a, **k)
        if on:
            unet.prof_enable(False, 16384)
        return out

    calls["prof"] = False
    unet.forward_raw = forward_sampled
    unet.prof_enable(True, 16384)        # allocate the event pool outside the timed region
    unet.prof_enable(False, 16384)

    one_step = one_step_raw

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    unet.prof_reset()
    calls["prof"] = True
    calls["n"] = 0

    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        edit, recon = one_step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        from hedit import dist as HD
        elapsed = HD.max_over_ranks(elapsed, device=dev)
    calls["prof"] = False
    unet_calls = calls["n"]

    # auxiliary (outside the timed region): BASELINE configs[1] read literally = ONE image; latency
    single_s = None
    if not args.no_single and style is None:
        one_image()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        one_image()
        torch.cuda.synchronize()
        single_s = time.perf_counter() - t1

    # auxiliary (outside the timed region): the same batch with the duplicate evaluations eliminated -- the P2P pass at t-1
    # already evaluates eps(x^orig_{t-1}, t-1, null / src), which the next base pass recomputes (p2p_h_edit.py:604-616 vs
    # :644-652); the kernels are batch-invariant, so reusing them gives the SAME BITS with 7 instead of 9 sample-forwards
    # per step.  Reported beside the headline, never as the headline (it is not the reference's evaluation count).
    cse = None
    if not args.no_config2 and style is None and not args.reuse_orig_eps and world == 1:
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        e3, r3 = one_step(reuse=True)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t3
        cse = {"what": "reuse_orig_eps: common-subexpression elimination of the x^orig rows, one pass of the same batch",
               "value": round(n / dt3, 4), "unit": "images/s", "ms_per_step": round(1e3 * dt3, 1),
               "sample_forwards_evaluated_per_image": (4 + 5 * K) * T - 2 * (T - 1),
               "bit_identical_to_timed_run": bool(torch.equal(e3, edit) and torch.equal(r3, recon))}

    # auxiliary (outside the timed region): BASELINE configs[2] = 32 images in lock-step, 50 steps x K = 3
    # (text-guided/main_p2p.py:65 optimization_steps 3: (4 + 5*3) * 50 = 950 sample-forwards per image), ONE pass
    config2 = None
    if not args.no_config2 and style is None and not args.tiny and K == 1 and world == 1:
        n2, K2 = 32, 3
        step2, _, w02, t_inv2 = build_workload(n2, K2, seed_off=1000)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        e2, r2 = step2()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        fwd2 = (4 + 5 * K2) * T
        config2 = {"workload": f"BASELINE configs[2]: the same loop, {n2} images in lock-step, {T} steps x K={K2} implicit steps "
                               f"({fwd2} sample-forwards per image), equalizer 1.25; one pass, kernels warm",
                   "value": round(n2 / dt2, 4), "unit": "images/s", "ms_per_step": round(1e3 * dt2, 1),
                   "achieved_tflops_per_s": round(n2 * fwd2 * FLOP_PER_SAMPLE_FWD / dt2 / 1e12, 1),
                   "mfma_frac_whole_loop": round(n2 * fwd2 * FLOP_PER_SAMPLE_FWD / dt2 / 1e12 / MFMA_PEAK_TFLOPS, 4),
                   "finite": bool(torch.isfinite(e2).all()),
                   "recon_rel_err": round(float(((r2 - w02).norm() / w02.norm()).item()), 7),
                   "ddpm_inversion_untimed_s": round(t_inv2, 2)}
        del step2, e2, r2, w02

    # auxiliary (outside the timed region): BASELINE configs[4] per GPU = text + CLIP-style editing, 16 images in
    # lock-step, 50 steps, K = 1: every inner step adds the decoder forward + backward and the style encoder; ONE pass
    config4 = None
    if not args.no_config2 and style is None and K == 1 and world == 1:
        n4 = 16 if not args.tiny else 2
        step4, _, w04, t_inv4 = build_workload(n4, 1, seed_off=2000, style=make_style())
        if args.tiny:
            step4()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        e4, r4 = step4()
        torch.cuda.synchronize()
        dt4 = time.perf_counter() - t4
        fl4 = n4 * ((4 + 5) * T * FLOP_PER_SAMPLE_FWD + T * STYLE_FLOP_PER_INNER_STEP) if not args.tiny else 0.0
        config4 = {"workload": f"BASELINE configs[4] per GPU: text-guided-n-style h_Edit_p2p_implicit (text + CLIP-style editing), SD UNet "
                               f"+ SD VAE decoder forward / backward in every step + ViT-B/16 style encoder (native, first 3 blocks), "
                               f"{T} steps, K=1, weight_edit_clip 0.5; {n4} images per GPU in lock-step; one pass, UNet kernels warm",
                   "value": round(n4 / dt4, 4), "unit": "images/s", "ms_per_step": round(1e3 * dt4, 1),
                   "roofline": {"bound": "mfma", "kernel": "whole loop: hedit_unet_forward x 450 + hedit_vae_decode_keep / _backward x 50 per image",
                                "achieved": round(fl4 / dt4 / 1e12, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(fl4 / dt4 / 1e12 / MFMA_PEAK_TFLOPS, 4), "traffic": None},
                   "finite": bool(torch.isfinite(e4).all()),
                   "recon_rel_err": round(float(((r4 - w04).norm() / w04.norm()).item()), 7),
                   "ddpm_inversion_untimed_s": round(t_inv4, 2)}
        del step4, e4, r4, w04
        model.vae = None

    # auxiliary (outside the timed region): BASELINE configs[3] per GPU = face swapping, 32 faces in lock-step, ONE pass
    config3 = None
    if not args.no_config2 and style is None and K == 1 and world == 1:
        torch.cuda.empty_cache()
        blk = face_pass(args, rank, dev, None, 32 if not args.tiny else 2, 100 if not args.tiny else 6, 3, 1, 0)
        config3 = {k: blk[k] for k in ("value", "unit", "ms_per_step", "achieved_tflops_per_s_per_gpu", "ms_per_eps_evaluation_batch",
                                       "unet_share_of_step", "roofline", "finite")}
        config3["workload"] = blk["config"]["workload"] + "; one pass after a two-step warm-up"
        # the BASELINE batch of 8 faces as well (configs[3] literally)
        blk8 = face_pass(args, rank, dev, None, 8 if not args.tiny else 2, 100 if not args.tiny else 6, 3, 1, 0)
        config3["batch_8"] = {"value": blk8["value"], "unit": "images/s", "ms_per_step": blk8["ms_per_step"],
                              "roofline_frac_eps_network": blk8["roofline"]["frac"]}

    finite = bool(torch.isfinite(edit).all())
    recon_err = float(((recon - w0).norm() / w0.norm()).item())
    prof = unet.prof_collect()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    imgs = args.steps * n * world
    sample_fwd_per_img = (4 + 5 * K) * T          # the reference's count (algorithmic work of the metric)
    evaluated_per_img = sample_fwd_per_img - (2 * (T - 1) if args.reuse_orig_eps else 0)
    total_flops = imgs * evaluated_per_img * (FLOP_PER_SAMPLE_FWD if not args.tiny else 0.0)
    if style is not None and not args.tiny:
        # per image and inner step: decoder forward (1257.5 GMAC) + its input-gradient pass (same convs, transposed)
        total_flops += imgs * T * K * STYLE_FLOP_PER_INNER_STEP
    # dominant kernel = the class with the largest sampled time
    dom = max(prof.items(), key=lambda kv: kv[1][0])
    dk, (dms, dfl, dcnt) = dom
    kernels = {k: {"ms": round(v[0], 3), "tflops": round(v[1] / 1e12, 4), "launches": v[2],
                   "tflops_per_s": round(v[1] / 1e9 / v[0], 1) if v[0] > 0 and v[1] > 0 else None}
               for k, v in prof.items()}
    sampled_ms = sum(v[0] for v in prof.values())
    roof = {"bound": "mfma", "kernel": {"conv3x3_gemm": "3x3 conv as implicit GEMM: pconv_kernel<BN> (persistent row-sharing loop) + igemm_kernel<BN,conv>",
                                        "linear_gemm": "linear class: igemm_kernel<BN,0> / pgemm_kernel<BN> (linear / 1x1) + ffn_chain_kernel / lin_chain_kernel (the token-local chains of the C=320 level, one kernel each)",
                                        "self_attn": "self_attn_kernel<D>", "cross_attn": "cross_attn_kernel<D>",
                                        "norm": "gn_*/layernorm kernels", "other": "geglu/concat"}[dk],
            "achieved": round(dfl / 1e9 / dms, 2) if dms > 0 else None, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(dfl / 1e9 / dms / MFMA_PEAK_TFLOPS, 4) if dms > 0 else None,
            "avg_launch_us": round(1e3 * dms / max(dcnt, 1), 2), "launches_sampled": dcnt,
            "alg_flops_per_launch": round(dfl / max(dcnt, 1), 0),
            "share_of_sampled_time": round(dms / sampled_ms, 4) if sampled_ms > 0 else None,
            "traffic": None}
    # roofline.traffic = HBM bytes per launch of that class from the PMC counters (tools/pmc_traffic.sh: two separate
    # rocprofv3 --pmc passes of THIS bench configuration, FETCH_SIZE doubled as the micro-arch guide prescribes).
    # A summary taken at another configuration is refused (traffic stays null).
    alg_bytes = unet.prof_collect_bytes()
    roof["alg_bytes_per_launch"] = round(alg_bytes.get(dk, 0.0) / max(dcnt, 1), 0)
    pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
    want_cfg = {"workload": args.workload, "images_per_gpu": n, "opt_steps": K, "fuse_src_pass": not args.no_fuse_src,
                "reuse_orig_eps": bool(args.reuse_orig_eps), "tiny": bool(args.tiny)}
    if os.path.exists(pmc):
        try:
            summ = json.load(open(pmc))
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from csrc_hash import csrc_hash
            roof["csrc_sha256"] = csrc_hash(ROOT)
            if summ.get("csrc_sha256") != roof["csrc_sha256"]:
                # counters of an older build say nothing about this one's re-reads
                roof["traffic_source"] = ("profiles/pmc_summary.json was measured on other kernel sources (its csrc_sha256 "
                                          f"{str(summ.get('csrc_sha256'))[:12]} != this tree's {roof['csrc_sha256'][:12]}): refused; "
                                          "re-run tools/pmc_traffic.sh")
            elif summ.get("config") == want_cfg:
                roof["traffic"] = summ["classes"].get(dk, {}).get("hbm_bytes_per_launch")
                roof["traffic_source"] = "profiles/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this configuration and of these kernel sources)"
                if roof["traffic"] and roof["alg_bytes_per_launch"]:
                    roof["traffic_over_algorithmic"] = round(roof["traffic"] / roof["alg_bytes_per_launch"], 3)
            else:
                roof["traffic_source"] = "profiles/pmc_summary.json was measured at another configuration: refused"
        except Exception as e:      # a malformed summary must not kill the bench line
            roof["traffic_source"] = f"profiles/pmc_summary.json unreadable: {e}"

    cpu = None
    if world == 1 and not args.no_cpu_baseline and not args.tiny and style is None:
        cpu = cpu_baseline(cfg, sd_cpu, T, K, tok)
    # auxiliary (after the timed region, rank 0, one GPU): the same loop in half storage, see half_storage_block.  Never part of `value`.
    half = None
    if not args.no_config2 and not args.no_half_storage and style is None and world == 1 and not args.tiny and args.storage == "bf16":
        half = half_storage_block(args, imgs / elapsed)

    out = {
        "metric": "edited images/sec (512^2, 50 steps, K Langevin)" + (" + style guidance" if style else ""), "value": round(imgs / elapsed, 4),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.storage, "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: text-guided h_Edit_p2p_implicit (h-Edit-R + P2P), "
                                "SD-1.5-shaped random-init UNet (859.5M), 64x64 latent (512^2 image), "
                                f"{T} DDIM steps, K={K} implicit step(s), CFG (1,5,7.5), xa 0.4, sa 0.35; "
                                f"{n} independent images per GPU in lock-step") if style is None else
                               ("BASELINE configs[4] per GPU: text-guided-n-style h_Edit_p2p_implicit (text + CLIP-style "
                                "editing), SD-1.5-shaped random-init UNet + SD VAE decoder forward/backward in every "
                                f"step + ViT-B/16 style encoder (fp16, first 3 blocks), {T} steps, K={K}, "
                                f"weight_edit_clip 0.5; {n} independent images per GPU in lock-step"),
                   "images_per_gpu": n, "unet_sample_forwards_per_image": sample_fwd_per_img,
                   "unet_sample_forwards_evaluated_per_image": evaluated_per_img,
                   "reuse_orig_eps": bool(args.reuse_orig_eps),
                   "unet_calls_in_timed_region": unet_calls, "parallelism": f"replica-dp{world}",
                   "pmc_config": want_cfg},
        "achieved_tflops_per_s_per_gpu": round(total_flops / elapsed / 1e12 / world, 1),
        "mfma_frac_whole_loop": round(total_flops / elapsed / 1e12 / world / MFMA_PEAK_TFLOPS, 4),
        "ms_per_unet_sample_forward": round(1e3 * elapsed * world / (imgs * evaluated_per_img), 4),
        "roofline": roof, "kernels_sampled": kernels, "cpu_baseline": cpu,
        "single_image": None if single_s is None else {"latency_s": round(single_s, 4), "images_per_s": round(1.0 / single_s, 4),
                                                        "note": "configs[1] read literally (1 image, 450 sample-forwards), "
                                                                "measured after the timed region"},
        "configs2": config2, "configs3": config3, "configs4": config4, "half_storage": half, "cse_variant": cse,
        "finite": finite, "recon_rel_err": round(recon_err, 7),
        "setup_s": {"weights_create_broadcast_load": round(t_weights, 1), "ddpm_inversion_untimed": round(t_inversion, 2)},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def half_storage_block(args, bf16_value):
    """The WHOLE timed loop of this bench (same configuration, one step after one warm-up step) in half storage: bench.py re-invoked in
    a child process with --storage f16 (a process has one storage format).  The parent drops its torch cache first; its kernels'
    handles stay (a few GB of the 288).  Any failure becomes a note, never an exception -- an auxiliary block must not cost the line."""
    import subprocess
    torch.cuda.empty_cache()
    out = {"workload": "the timed loop of this line (configs[1], same images per GPU) re-run by a child `bench.py --storage f16 --steps 1 --warmup 1`: "
                       "libhedit_hip_f16.so, the same kernels on IEEE-half storage (csrc/common.h)",
           "accuracy": "quoted, not measured in this run: SD-1.5-shape eps error vs the fp32 oracle 1.45e-3 (half) / 1.16e-2 (bfloat16), "
                       "profiles/r05_f16_storage.txt; 50-step loop vs the oracle trajectory in both formats: profiles/r06_loop_divergence.txt; "
                       "both asserted by tests/test_gpu_unet.py / tests/test_gpu_loop_trajectory.py through tests/test_gpu_f16_suite.py"}
    try:
        cmd = [sys.executable, os.path.abspath(__file__), "--storage", "f16", "--steps", "1", "--warmup", "1", "--images", str(args.images),
               "--diffusion-steps", str(args.diffusion_steps), "--opt-steps", str(args.opt_steps), "--no-config2", "--no-cpu-baseline",
               "--no-single", "--no-half-storage"]
        env = dict(os.environ)
        env.pop("HEDIT_STORAGE", None)
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            out["error"] = (r.stderr or r.stdout)[-400:]
            return out
        d = json.loads(line[-1])
        out.update({"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "finite": d["finite"],
                    "recon_rel_err": d["recon_rel_err"], "mfma_frac_whole_loop": d["mfma_frac_whole_loop"],
                    "f16_over_bf16_images_per_s": round(d["value"] / bf16_value, 4) if bf16_value else None})
    except Exception as e:      # noqa: BLE001
        out["error"] = repr(e)[:400]
    return out


def cpu_baseline(cfg, sd_cpu, T, K, tok, text_layers=12, text_heads=12):
    """The oracle (CPU fp32 eager restatement = "port" of the reference's CPU path, oracle/loops.py + oracle/sd_unet.py)
    timed on this box's host cores on a bounded sample of BASELINE configs[0] ITSELF (C1 =
    text-guided implicit h-edit without the P2P controller, 1 image, 64x64 latent, 20 DDIM steps, K = 1): ONE sampler
    step of h_Edit_R_implicit (text-guided/inversion/p2p_h_edit.py:281-315) in the reference's loop shape = 6 of C1's 120
    sample-forwards (a 2-row base pass and a 4-row correction pass), timed TWICE after one untimed 1-row forward that
    touches the weights and spins up the thread pool; the MINIMUM of the two is the figure, and both runs are reported
    with the host's 1-minute load average before and after (the figure moves 25 -> 46 s per step with what else runs on
    the box: the quieter run is the baseline).  `value` is the bench metric's configuration (configs[1], (4 + 5K) T
    sample-forwards per image) extrapolated by sample-forward count -- an EXTRAPOLATION, x75 --; the C1 figure (x20) is
    beside it."""
    import types
    sys.path.insert(0, ROOT)
    from oracle import loops as OL
    from oracle import p2p as OP
    from oracle import sd_unet as OU
    from hedit.scheduler import DDIMScheduler
    from hedit.text import ClipTextEncoder
    T0, NS, RUNS = 20, 1, 2
    net = OU.UNet2DConditionModel(**cfg)
    net.load_state_dict(sd_cpu)
    net.eval()
    for p in net.parameters():
        p.requires_grad_(False)
    om = types.SimpleNamespace(device=torch.device("cpu"), unet=net, scheduler=DDIMScheduler(), tokenizer=tok, vae=None,
                               text_encoder=ClipTextEncoder(dim=cfg["cross_attention_dim"], layers=text_layers, heads=text_heads, seed=7))
    om.scheduler.set_timesteps(NS)       # an NS-step schedule: after_skip_steps == num_inference_steps, so the loop runs exactly NS
    src, tar, _, _ = DEMO_PAIRS[0]       # regular sampler steps (no time-ahead correction); a step's cost does not depend on t
    g = torch.Generator().manual_seed(5)
    S = cfg["sample_size"]
    x = torch.randn(1, 4, S, S, generator=g)
    z = torch.randn(NS, 4, S, S, generator=g)
    threads = torch.get_num_threads()
    runs = []
    with torch.no_grad():
        net(x, int(om.scheduler.timesteps[-1]), encoder_hidden_states=torch.randn(1, 77, cfg["cross_attention_dim"], generator=g))   # warm-up, untimed
        for _ in range(RUNS):
            oc = OP.Controller("store")
            OP.register(om, oc)
            load0 = os.getloadavg()[0]
            t0 = time.perf_counter()
            OL.h_edit_r_implicit(om, xT=x, eta=1.0, prompts=[src, tar], cfg_scales=[1.0, 5.0, 7.5], zs=z, controller=oc,
                                 weight_reconstruction=0.1, optimization_steps=1, after_skip_steps=NS, is_ddim_inversion=False)
            runs.append({"s_per_sampler_step": round((time.perf_counter() - t0) / NS, 2), "host_loadavg_1min_before": round(load0, 2),
                         "host_loadavg_1min_after": round(os.getloadavg()[0], 2)})
    dt = min(r["s_per_sampler_step"] for r in runs)      # the quieter of the two runs: the host's load moves this figure by 50 %
    per_fwd = dt / 6
    per_img = per_fwd * (4 + 5 * K) * T
    return {"value": round(1.0 / per_img, 6), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"BASELINE configs[0] (C1): one sampler step of h_Edit_R_implicit, K = 1, no P2P controller, 1 image = 6 of C1's 120 "
                      f"sample-forwards (a 2-row base pass + a 4-row correction pass, step algebra included) by the fp32 eager oracle on "
                      f"{threads} host threads, timed {RUNS} times after an untimed warm-up forward, the MINIMUM taken ({dt:.2f} s per step; every "
                      f"run with the host's load average in `runs`); value = configs[1] ({(4 + 5 * K) * T} sample-forwards per image) "
                      f"EXTRAPOLATED by sample-forward count (x{(4 + 5 * K) * T / 6:.1f}); configs0_* = C1 itself extrapolated x{T0}",
            "configs0_s_per_image": round(dt * T0, 1), "configs0_images_per_s": round(1.0 / (dt * T0), 6),
            "s_per_sampler_step": round(dt, 2), "s_per_unet_sample_forward": round(per_fwd, 3), "sampler_steps_timed": NS * RUNS,
            "runs": runs, "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
