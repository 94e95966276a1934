"""GPU diagnostic: batch invariance of the face path's pieces (pixel UNet, identity reward, LPIPS) at the toy size."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.arcface import IDLoss  # noqa: E402
from hedit.arcface.lpips_loss import LPIPS_Loss  # noqa: E402
from hedit.diffusion import Model, TINY_DDPM_CONFIG  # noqa: E402

dev = torch.device("cuda:0")
for cfg, name in ((TINY_DDPM_CONFIG, "tiny"), (None, "full")):
    m = Model(cfg, device=dev)
    m.init_random(0)
    S = m.resolution
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 3, S, S, generator=g).to(dev)
    t = torch.ones(3) * 401
    e = m(x, t)
    for i in range(3):
        e1 = m(x[i:i + 1], t[:1])
        print(name, "ddpm", i, torch.equal(e[i:i + 1], e1), (e[i:i + 1] - e1).abs().max().item())
g = torch.Generator().manual_seed(1)
refs = torch.randn(2, 3, 256, 256, generator=g) * 0.4
x = (torch.randn(2, 3, 256, 256, generator=g) * 0.4).to(dev)
idl = IDLoss(ref=refs, device=dev, seed=0)
l2, g2 = idl._native_loss_and_grad(x)
for i in range(2):
    one = IDLoss(ref=refs[i:i + 1], device=dev, seed=0)
    l1, g1 = one._native_loss_and_grad(x[i:i + 1])
    print("idloss", i, torch.equal(l2[i:i + 1], l1), torch.equal(g2[i:i + 1] * 2, g1), ((g2[i:i + 1] * 2 - g1).abs().max() / g1.abs().max()).item())
for S in (32, 256):
    src = torch.randn(2, 3, S, S, generator=g) * 0.4
    xx = (torch.randn(2, 3, S, S, generator=g) * 0.4).to(dev)
    lp = LPIPS_Loss(src=src, device=dev, seed=0)
    l2, g2 = lp._native_loss_and_grad(xx)
    for i in range(2):
        one = LPIPS_Loss(src=src[i:i + 1], device=dev, seed=0)
        l1, g1 = one._native_loss_and_grad(xx[i:i + 1])
        print("lpips", S, i, torch.equal(l2[i:i + 1], l1), torch.equal(g2[i:i + 1] * 2, g1), ((g2[i:i + 1] * 2 - g1).abs().max() / g1.abs().max()).item())
