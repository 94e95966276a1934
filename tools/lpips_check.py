#!/usr/bin/env python3
"""GPU: the native LPIPS-VGG reward (hedit_lpips_fwd_bwd) against the torch fp32 restatement on the same weights
(loss, image gradient) and its time per call.  python tools/lpips_check.py [B] [size]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.arcface.lpips_loss import LPIPS_Loss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
src = torch.randn(1, 3, S, S, generator=g) * 0.4
nat = LPIPS_Loss(src=src, device=dev, seed=1, backend="hip")
tor = LPIPS_Loss(src=src, device=dev, seed=1, backend="torch")
x = (torch.randn(B, 3, S, S, generator=g) * 0.4).to(dev)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


xs = [x.clone().requires_grad_(True) for _ in range(2)]
ln, lt = nat.get_lpips_loss(xs[0]), tor.get_lpips_loss(xs[1])
gn, gt = torch.autograd.grad(ln, xs[0])[0], torch.autograd.grad(lt, xs[1])[0]
print("loss", ln.item(), lt.item(), "grad rel err", rel(gn, gt), "grad norm", gt.norm().item())
for name, m in (("native", nat), ("torch", tor)):
    for _ in range(3):
        xx = x.clone().requires_grad_(True)
        torch.autograd.grad(m.get_lpips_loss(xx), xx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        xx = x.clone().requires_grad_(True)
        torch.autograd.grad(m.get_lpips_loss(xx), xx)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per loss + gradient, batch {B}, {S}x{S}")
