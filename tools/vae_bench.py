"""Time the SD-1.x image decoder forward and its vector-Jacobian product (HIP), per batch size."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit.vae import AutoencoderKL
dev = "cuda:0"
vae = AutoencoderKL(device=dev)
vae.init_random(0)
DEC_MACS = 1.2575e12   # decoder MACs per image at a 64x64 latent (counted on the CPU restatement)
for B in (1, 2, 4, 8):
    z = torch.randn(B, 4, 64, 64, device=dev)
    u = torch.randn(B, 3, 512, 512, device=dev)
    for name, fn in (("decode", lambda: vae.decode(z).sample), ("decode_vjp", lambda: vae.decode_vjp(z, u))):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        mult = 1.0 if name == "decode" else 2.0    # vjp = forward + input-gradient pass of about the same MACs
        print(f"B={B} {name:11s} {dt*1e3:8.2f} ms  {dt*1e3/B:7.2f} ms/img  {2*DEC_MACS*mult*B/dt/1e12:6.1f} TFLOP/s"
              f"  ws {vae._ws.numel()/2**30:.2f} GiB")
