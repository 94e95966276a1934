"""What the hook path for host-language controllers costs: one SD-1.5 UNet call of the reference's 4-row P2P
batch through hedit_unet_set_attn_hook (fp32 probabilities of all 32 layers materialised, a Python callback per layer, clone +
copy-back in hedit/unet.py::forward_hooked) against the fused path on the same rows."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit.unet import UNet2DConditionModel

dev = "cuda:0"
unet = UNet2DConditionModel(device=dev); unet.init_random(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(B, 4, 64, 64, device=dev); ctx = torch.randn(B, 77, 768, device=dev)
calls = [0]


def ctrl(attn, is_cross, place, save):
    calls[0] += 1
    return attn


def t(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


fused = t(lambda: unet.forward_raw(x, 500.0, ctx, None), 10)
hooked = t(lambda: unet.forward_hooked(x, 500.0, ctx, ctrl), 3)
print(f"SD-1.5 UNet call, {B} rows: fused path {fused:.1f} ms, hook path {hooked:.1f} ms ({calls[0] // 4} controller calls per pass), x{hooked / fused:.1f}")
