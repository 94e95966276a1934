"""Per-launch table of one UNet sample-forward at the bench's batch (default 120 rows = the 5n-row P2P pass of
24 images): every sampled launch with its shape, time, algorithmic TFLOP/s and GB/s, then the launches grouped by
(class, M, N, K, tag), sorted by time.  Usage: python tools/layer_prof.py [rows] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import numpy as np
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):       # tools/build_variant.sh side library (measurement builds)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
from hedit.unet import UNet2DConditionModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = "cuda:0"
if os.environ.get("HEDIT_TEST_FLAGS"):        # e.g. 8 = without the persistent linear kernel (A/B of csrc/pgemm.hip)
    _lib.check(_lib.lib().hedit_test_set_flags(int(os.environ["HEDIT_TEST_FLAGS"])))
unet = UNet2DConditionModel(device=dev); unet.init_random(0)
x = torch.randn(B, 4, 64, 64, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
for _ in range(2):
    unet(x, 500, encoder_hidden_states=ctx)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    unet(x, 500, encoder_hidden_states=ctx)
e1.record(); torch.cuda.synchronize()
print(f"# rows {B}: {e0.elapsed_time(e1) / reps:.2f} ms per sample-forward batch (no profiling)")
unet.prof_enable(True, 16384); unet.prof_reset()
for _ in range(reps):
    unet(x, 500, encoder_hidden_states=ctx)
rec = unet.prof_records()
unet.prof_enable(False, 16384)
kinds = UNet2DConditionModel.PROF_KINDS
tot = rec[:, 1].sum() / reps
print(f"# sampled: {len(rec) // reps} launches, {tot:.2f} ms per forward")
groups = {}
for r in rec:
    key = (int(r[0]), int(r[4]), int(r[5]), int(r[6]), int(r[7]), round(r[2] / 1e6), round(r[3] / 1e3))
    g = groups.setdefault(key, [0, 0.0, r[2], r[3]])
    g[0] += 1; g[1] += r[1]
print(f"{'class':>10} {'M':>7} {'N':>6} {'K':>6} {'tag':>4} {'calls':>5} {'ms/fwd':>8} {'pct':>6} {'us/call':>8} {'TF/s':>7} {'GB/s':>7}")
for key, g in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    n, ms, fl, by = g
    per = ms / n
    print(f"{kinds[key[0]]:>10} {key[1]:7d} {key[2]:6d} {key[3]:6d} {key[4]:4d} {n // reps:5d} {ms / reps:8.3f} {100 * ms / reps / tot:6.2f} "
          f"{per * 1e3:8.1f} {fl / per / 1e9:7.0f} {by / per / 1e6:7.0f}")
