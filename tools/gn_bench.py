"""GroupNorm launch (statistics + apply, SiLU) through the C ABI at a few (batch, pixels, channels): time and the HBM rate of
its algorithmic traffic (x read twice, y written once).  HEDIT_LIB_VARIANT=name for A/B against a side library."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
lib = _lib.lib()
dev = torch.device("cuda:0")
shapes = [(8, 65536, 128), (8, 65536, 256), (8, 16384, 256), (32, 65536, 128), (1, 262144, 128), (120, 4096, 320), (5, 4096, 320)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, HW, Cc) in shapes:
    x = torch.randn(B, HW, Cc, device=dev).to(torch.bfloat16)
    y = torch.empty_like(x)
    g, b = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    ws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(B, HW, Cc), dtype=torch.uint8, device=dev)
    f = lambda: _lib.check(lib.hedit_k_groupnorm(_lib.ptr(x), _lib.ptr(y), _lib.ptr(g), _lib.ptr(b), B, HW, Cc, 32, 1e-5, 1, _lib.ptr(ws), None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"B={B:3d} HW={HW:6d} C={Cc:4d}: {us:8.1f} us  {3.0 * x.numel() * 2 / us / 1e6:6.2f} TB/s")
