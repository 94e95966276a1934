#!/usr/bin/env python3
"""GPU: time per loss + image gradient of the three native reward executors (csrc/vit.hip, irse.hip, lpips.hip) at
the workloads' shapes.  Accuracy is the tests' business (tests/test_gpu_{clip,arcface,lpips}.py: golden vectors of
the reference and the oracle's fp32 restatement).  python tools/reward_bench.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.arcface import IDLoss  # noqa: E402
from hedit.arcface.lpips_loss import LPIPS_Loss  # noqa: E402
from hedit.clip_guidance import CLIPEncoder  # noqa: E402
from hedit.clip_guidance.base_clip import ClipVisualPrefix  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
enc = CLIPEncoder(clip_model=ClipVisualPrefix().init_random(13).float(), device=dev)
enc.set_reference(torch.randn(1, 3, 224, 224, generator=g).to(dev))
idl = IDLoss(ref=torch.randn(1, 3, 256, 256, generator=g) * 0.4, device=dev, seed=1)
lp = LPIPS_Loss(src=torch.randn(1, 3, 256, 256, generator=g) * 0.4, device=dev, seed=1)
cases = {"style encoder (ViT-B/16 prefix + Gram), 512^2 images": (lambda x: enc.gram_residual_norms(x).sum(), (B, 3, 512, 512)),
         "identity reward (IR-SE50), 256^2": (idl.get_cosine_loss, (B, 3, 256, 256)),
         "LPIPS-VGG16, 256^2": (lp.get_lpips_loss, (B, 3, 256, 256))}
for name, (fn, shape) in cases.items():
    x = (torch.randn(*shape, generator=g) * 0.4).to(dev)
    for _ in range(3):
        xx = x.clone().requires_grad_(True)
        torch.autograd.grad(fn(xx), xx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        xx = x.clone().requires_grad_(True)
        torch.autograd.grad(fn(xx), xx)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per loss + gradient, batch {B}")
