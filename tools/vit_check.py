#!/usr/bin/env python3
"""GPU: the native style encoder (hedit_vit_gram_fwd_bwd) against the torch mirror (fp32 and the reference's fp16) at
ViT-B/16 shape: loss, image gradient, time per call.  python tools/vit_check.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.clip_guidance import CLIPEncoder  # noqa: E402
from hedit.clip_guidance.base_clip import ClipVisualPrefix  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
m = ClipVisualPrefix().init_random(13)
ref = torch.randn(1, 3, 224, 224, generator=torch.Generator().manual_seed(17)).to(dev)
encs = {"native": CLIPEncoder(clip_model=m.float(), device=dev, backend="hip"),
        "torch fp32": CLIPEncoder(clip_model=ClipVisualPrefix().init_random(13).float(), device=dev, backend="torch"),
        "torch fp16": CLIPEncoder(clip_model=ClipVisualPrefix().init_random(13).half(), device=dev, backend="torch")}
for e in encs.values():
    e.set_reference(ref)
ims = (torch.randn(B, 3, 512, 512, generator=torch.Generator().manual_seed(3)) * 0.5).to(dev)


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm()).item()


res = {}
for name, e in encs.items():
    x = ims.clone().requires_grad_(True)
    l = e.gram_residual_norms(x)
    res[name] = (l.detach(), torch.autograd.grad(l.sum(), x)[0])
for name in ("native", "torch fp16"):
    print(f"{name} vs torch fp32: loss rel {rel(res[name][0], res['torch fp32'][0]):.2e}  grad rel {rel(res[name][1], res['torch fp32'][1]):.2e}")
for name, e in encs.items():
    for _ in range(3):
        x = ims.clone().requires_grad_(True)
        torch.autograd.grad(e.gram_residual_norms(x).sum(), x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        x = ims.clone().requires_grad_(True)
        torch.autograd.grad(e.gram_residual_norms(x).sum(), x)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per loss + gradient, batch {B}")
