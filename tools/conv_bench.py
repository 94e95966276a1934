"""3x3 conv launches of the SD UNet at `rows` batch rows through the C ABI (hedit_k_gemm, mode 1): time and TFLOP/s.
HEDIT_LIB_VARIANT=name loads h-edit_amd/hedit/lib_name.so.bin (tools/build_variant.sh) instead of the product library."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
v = os.environ.get("HEDIT_LIB_VARIANT")
if v:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{v}.so.bin")
lib = _lib.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120


shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]


def conv(hw, cin, cout, res=False, iters=10):
    M = B * hw * hw
    K = 9 * cin
    A = torch.randn(M, cin, device=dev).to(torch.bfloat16)
    W = (torch.randn(cout, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(cout, device=dev)
    R = torch.randn(M, cout, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(M, cout, device=dev, dtype=torch.bfloat16)
    splits = 0
    if os.environ.get("CONV_BENCH_CANONICAL"):       # the chunking the UNet executor would use (nominal batch 4), folded in registers
        ck = lib.hedit_k_gemm_canonical_chunk(4 * hw * hw, cout, K)
        splits = -((K // 64 + ck - 1) // ck) if ck > 0 else 0
    ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, cout, K, abs(splits)), 16), dtype=torch.uint8, device=dev)
    f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(R), _lib.ptr(out), M, cout, K,
                                            cin, cout, cout, 1, hw, hw, cin, hw, hw, splits, _lib.ptr(ws), None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"conv3x3 {hw:2d}x{hw:<2d} {cin:5d}->{cout:<5d} M={M:7d} {us:9.1f} us  {2.0 * M * cout * K / us / 1e6:8.1f} TF/s  splits {splits}", flush=True)
    return out


def upconv(hin, c, iters=10):
    ho = 2 * hin
    M, K = B * ho * ho, 9 * c
    A = torch.randn(B * hin * hin, c, device=dev).to(torch.bfloat16)
    W = (torch.randn(c, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(c, device=dev)
    out = torch.empty(M, c, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, c, K,
                                            c, c, c, 3, hin, hin, c, ho, ho, 0, None, None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    print(f"upconv  {hin:2d}->{ho:<2d} {c:5d} M={M:7d} {us:9.1f} us  {2.0 * M * c * K / us / 1e6:8.1f} TF/s", flush=True)


if not shapes:
    for hin, c in [(32, 640), (16, 1280), (8, 1280)]:
        upconv(hin, c)
for hw, cin, cout in shapes or [(64, 320, 320), (64, 640, 320), (64, 960, 320), (32, 640, 640), (32, 1280, 640), (32, 1920, 640),
                      (16, 1280, 1280), (16, 2560, 1280), (8, 1280, 1280), (8, 2560, 1280)]:
    conv(hw, cin, cout, res=True)
