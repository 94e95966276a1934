"""Self-attention micro-benchmark through the C ABI (shapes of one SD-1.5 UNet call with B rows)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import ctypes as C
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):       # tools/build_variant.sh side library (measurement builds)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")

lib = _lib.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
out_s = []
for (N, c) in [(4096, 320), (1024, 640), (256, 1280)]:
    heads, d = 8, c // 8
    qk = torch.randn(B * N, 2 * c, device=dev).to(torch.bfloat16) * 0.3
    vt = torch.randn(c, B * N, device=dev).to(torch.bfloat16)
    out = torch.empty(B * N, c, device=dev, dtype=torch.bfloat16)
    kv = qk[:, c:]
    f = lambda: _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * c, C.c_void_p(kv.data_ptr()), 2 * c, _lib.ptr(vt), B * N,
                                                 _lib.ptr(out), c, B, N, heads, d, None, None, None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    out_s.append(f"N={N} d={d}: {us:8.1f} us {4.0 * B * N * N * c / us / 1e6:7.1f} TF/s")
print(" | ".join(out_s))
