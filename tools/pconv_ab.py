"""Persistent row-sharing 3x3 kernel (csrc/pconv.hip) against the one-shot igemm_kernel (modes 4 / 5), shape by shape, back to back on one
box: bit equality (torch.equal), time and algorithmic TFLOP/s of both.  `hedit_test_set_flags(8)` keeps the one-shot kernels.  Rows = the
UNet batch (default 120 = the 5n-row P2P pass of 24 images).  Chunked cases use the canonical K-chunking of the UNet executor (nominal
batch 4) folded in registers.   python tools/pconv_ab.py [rows] [iters]"""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib

lib = _lib.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ST = torch.float16 if _lib.STORAGE == "f16" else torch.bfloat16


def timeit(fn, iters=ITERS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(name, rows, hin, cin, cout, res, up=False, chunked=True):
    ho = 2 * hin if up else hin
    M, K = rows * ho * ho, 9 * cin
    g = torch.Generator(device="cuda").manual_seed(rows + 3 * hin + 7 * cin + 11 * cout)
    A = torch.randn(rows * hin * hin, cin, device=dev, generator=g).to(ST)
    W = (torch.randn(cout, K, device=dev, generator=g) / math.sqrt(K)).to(ST)
    bias = torch.randn(cout, device=dev, generator=g)
    R = torch.randn(M, cout, device=dev, generator=g).to(ST) if res else None
    splits = 0
    ck = lib.hedit_k_gemm_canonical_chunk(4 * ho * ho, cout, K) if chunked else 0
    if ck > 0:
        splits = -((K // 64 + ck - 1) // ck)
    ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, cout, K, abs(splits)), 16), dtype=torch.uint8, device=dev)
    outs, times = [], []
    for flags in (8, 0, 8, 0):
        _lib.check(lib.hedit_test_set_flags(flags))
        out = torch.full((M, cout), 7.0, device=dev, dtype=ST)
        f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(R) if res else None, _lib.ptr(out), M, cout, K,
                                                cin, cout, cout, 3 if up else 1, hin, hin, cin, ho, ho, splits, _lib.ptr(ws), None))
        times.append(timeit(f))
        outs.append(out)
    _lib.check(lib.hedit_test_set_flags(0))
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)) and torch.equal(outs[0].view(torch.int16), outs[3].view(torch.int16))
    fl = 2.0 * M * cout * K
    t_old, t_new = min(times[0], times[2]), min(times[1], times[3])
    print(f"{name:30s} M={M:7d} N={cout:5d} K={K:6d} res={int(res)} chunk_kt={ck:3d}  one-shot {t_old:8.1f} us {fl / t_old / 1e6:7.0f} | persistent "
          f"{t_new:8.1f} us {fl / t_new / 1e6:7.0f}  ({t_old / t_new:5.2f}x)  bits {'equal' if same else 'DIFFER'}", flush=True)
    return same


ok = True
for res in (False, True):
    ok &= case("L0 64x64 320->320", B, 64, 320, 320, res)
    ok &= case("L0 64x64 640->320", B, 64, 640, 320, res)
    ok &= case("L0 64x64 960->320", B, 64, 960, 320, res)
    ok &= case("L1 32x32 640->640", B, 32, 640, 640, res)
    ok &= case("L1 32x32 1280->640", B, 32, 1280, 640, res)
    ok &= case("L1 32x32 1920->640", B, 32, 1920, 640, res)
    ok &= case("L2 16x16 1280->1280", B, 16, 1280, 1280, res)
    ok &= case("L2 16x16 2560->1280", B, 16, 2560, 1280, res)
ok &= case("L1 32x32 640->640 no chunk", B, 32, 640, 640, True, chunked=False)
ok &= case("up 32->64 640", B, 32, 640, 640, False, up=True)
ok &= case("up 16->32 1280 (chunk: one-shot)", B, 16, 1280, 1280, False, up=True)
ok &= case("up 16->32 1280 no chunk", B, 16, 1280, 1280, False, up=True, chunked=False)
# tiles that span images and a ragged last tile: 8x8 images (4 per tile), M = 257 * 64 = 64.25 tiles
ok &= case("8x8 1280->1280, 257 rows", 257, 8, 1280, 1280, True)
ok &= case("8x8 1280->1280, 257 rows, plain", 257, 8, 1280, 1280, False, chunked=False)
ok &= case("16x16 320->640, 131 rows", 131, 16, 320, 640, True, chunked=False)
print("ALL BITS EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
