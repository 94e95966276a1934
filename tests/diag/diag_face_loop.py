"""GPU diagnostic: h_Edit_R on two faces in lock-step vs one at a time (toy pixel UNet), step by step."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.arcface import IDLoss  # noqa: E402
from hedit.arcface.lpips_loss import LPIPS_Loss  # noqa: E402
from hedit.diffusion import Model, TINY_DDPM_CONFIG  # noqa: E402
from hedit.inversion.h_edit_R import h_Edit_R  # noqa: E402

dev = torch.device("cuda:0")
m = Model(TINY_DDPM_CONFIG, device=dev)
m.init_random(0)
S = m.resolution
T, K = 8, 2
g = torch.Generator().manual_seed(0)
refs = torch.randn(2, 3, 256, 256, generator=g) * 0.4
srcs = torch.randn(2, 3, S, S, generator=g) * 0.4
xT = torch.randn(2, 3, S, S, generator=g).to(dev)
zs = torch.randn(T, 2, 3, S, S, generator=g).to(dev)
betas = torch.from_numpy(np.linspace(0.0001, 0.02, 1000, dtype=np.float64)).float().to(dev)
seq = (np.arange(0, 1000, 1000 // T) + 1)[::-1]
for use_id, use_lp in ((True, False), (False, True), (True, True)):
    idl = IDLoss(ref=refs, device=dev, seed=0) if use_id else None
    lp = LPIPS_Loss(src=srcs, device=dev, seed=0) if use_lp else None
    both = h_Edit_R(m, lp, idl, xT, betas, seq, eta=1.0, zs=zs, weight_edit_face=4.0, optimization_steps=K, after_skip_steps=T,
                    num_inference_steps=T, per_image=True).detach()
    for i in range(2):
        idl1 = IDLoss(ref=refs[i:i + 1], device=dev, seed=0) if use_id else None
        lp1 = LPIPS_Loss(src=srcs[i:i + 1], device=dev, seed=0) if use_lp else None
        one = h_Edit_R(m, lp1, idl1, xT[i:i + 1], betas, seq, eta=1.0, zs=zs[:, i:i + 1], weight_edit_face=4.0, optimization_steps=K,
                       after_skip_steps=T, num_inference_steps=T).detach()
        print("id" if use_id else "--", "lpips" if use_lp else "--", i, torch.equal(both[i:i + 1], one), (both[i:i + 1] - one).abs().max().item())
