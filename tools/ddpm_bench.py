"""Time the CelebA-HQ pixel DDPM UNet forward (HIP) per batch size, and one face h-Edit-R step
(1 + 2K eps evaluations) without the loss closures."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.diffusion import Model
dev = "cuda:0"
m = Model(device=dev)
m.init_random(0)
FLOP = 0.497e12      # per sample forward at 256 x 256 (248.2 GMAC conv/linear + 0.34 GMAC attention)
for B in (1, 2, 4, 8, 16):
    x = torch.randn(B, 3, 256, 256, device=dev)
    m(x, 501.0); torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        m(x, 501.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"B={B:2d} {dt*1e3:8.2f} ms  {dt*1e3/B:7.2f} ms/img  {FLOP*B/dt/1e12:6.1f} TFLOP/s  ws {m._ws.numel()/2**30:.2f} GiB")
