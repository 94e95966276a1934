"""Repeatability and batch invariance of the UNet executor at the bench's launch sizes: REPS forwards of the same ROWS-row
batch compared bit for bit with the first, then a few rows evaluated in a smaller batch against the big batch's rows.
python tools/unet_repeat.py [rows=120] [reps=20]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit.unet import UNet2DConditionModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = "cuda:0"
unet = UNet2DConditionModel(device=dev); unet.init_random(0)
g = torch.Generator().manual_seed(1)
x = torch.randn(B, 4, 64, 64, generator=g).to(dev); ctx = torch.randn(B, 77, 768, generator=g).to(dev)
ref = unet(x, 500, encoder_hidden_states=ctx).sample.clone(); torch.cuda.synchronize()
bad = 0
for r in range(reps):
    out = unet(x, 500, encoder_hidden_states=ctx).sample; torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        d = (out - ref).abs().reshape(B, -1).max(1).values
        rows = torch.nonzero(d > 0).reshape(-1).tolist()
        print(f"  run {r}: rows {rows[:12]}{'...' if len(rows) > 12 else ''} differ, max {float(d.max()):.3e}")
print(f"rows {B}: {reps} repeats, {bad} mismatching")
for nb in (96, 48, 24, 23, 5, 2, 1):
    if nb >= B:
        continue
    sub = unet(x[:nb].contiguous(), 500, encoder_hidden_states=ctx[:nb].contiguous()).sample.clone(); torch.cuda.synchronize()
    for _ in range(5):
        again = unet(x[:nb].contiguous(), 500, encoder_hidden_states=ctx[:nb].contiguous()).sample; torch.cuda.synchronize()
        if not torch.equal(again, sub):
            print(f"   {nb}-row batch is not repeatable")
    same = torch.equal(sub, ref[:nb])
    print(f"first {nb} rows alone == inside the {B}-row batch: {same}")
    if not same:
        d = (sub - ref[:nb]).abs().reshape(nb, -1).max(1).values
        print("   rows", torch.nonzero(d > 0).reshape(-1).tolist()[:16], "max", float(d.max()))
