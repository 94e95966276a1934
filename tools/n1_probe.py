"""Is the single-image loop launch-bound?  Times UNet calls of 2/3/5 rows: host-side issue time per call
(no sync) vs GPU time per call (events around a long back-to-back run)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit.unet import UNet2DConditionModel
dev = "cuda:0"
unet = UNet2DConditionModel(device=dev); unet.init_random(0)
for B in (2, 3, 5, 16, 40):
    x = torch.randn(B, 4, 64, 64, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
    for _ in range(3):
        unet(x, 500, encoder_hidden_states=ctx)
    torch.cuda.synchronize()
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n):
        unet(x, 500, encoder_hidden_states=ctx)
    t_issue = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"B={B:3d}: host issue {t_issue / n * 1e3:7.2f} ms/call, GPU span {e0.elapsed_time(e1) / n:7.2f} ms/call, wall {t_all / n * 1e3:7.2f} ms/call")
