#!/usr/bin/env python3
"""GPU: the native IR-SE50 identity reward (hedit_irse50_cos_fwd_bwd) against the torch fp32 module on the same
weights (forward features, loss, image gradient) and its time per call.  python tools/irse_check.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.arcface import IDLoss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
ref = torch.randn(1, 3, 256, 256, generator=g) * 0.4
nat = IDLoss(ref=ref, device=dev, seed=1, backend="hip")
tor = IDLoss(ref=ref, device=dev, seed=1, backend="torch")
x = (torch.randn(B, 3, 256, 256, generator=g) * 0.4).to(dev)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


with torch.no_grad():
    fn, ft = nat.extract_feats(x), torch.nn.functional.normalize(tor.extract_feats(x), dim=-1)
print("features rel err", rel(fn, ft))
xs = [x.clone().requires_grad_(True) for _ in range(2)]
ln, lt = nat.get_cosine_loss(xs[0]), tor.get_cosine_loss(xs[1])
gn, gt = torch.autograd.grad(ln, xs[0])[0], torch.autograd.grad(lt, xs[1])[0]
print("loss", ln.item(), lt.item(), "grad rel err", rel(gn, gt), "grad norm", gt.norm().item())
# fp64 reference of the torch module on the CPU for the first image: how far is fp32 torch itself?
t64 = IDLoss(ref=ref.double(), seed=1, backend="torch").double()
x64 = x[:1].double().cpu().requires_grad_(True)
l64 = t64.get_cosine_loss(x64)
g64 = torch.autograd.grad(l64, x64)[0]
x1 = [x[:1].clone().requires_grad_(True) for _ in range(2)]
gn1 = torch.autograd.grad(nat.get_cosine_loss(x1[0]), x1[0])[0]
gt1 = torch.autograd.grad(tor.get_cosine_loss(x1[1]), x1[1])[0]
print("vs fp64: native grad rel err", rel(gn1.cpu(), g64), " torch-fp32 grad rel err", rel(gt1.cpu(), g64))
for name, m in (("native", nat), ("torch", tor)):
    for _ in range(3):
        xx = x.clone().requires_grad_(True)
        torch.autograd.grad(m.get_cosine_loss(xx), xx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        xx = x.clone().requires_grad_(True)
        torch.autograd.grad(m.get_cosine_loss(xx), xx)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms per loss + gradient, batch {B}")
