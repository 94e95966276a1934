"""Persistent linear kernel (csrc/pgemm.hip) against the one-shot igemm_kernel, shape by shape, back to back on one box: bit equality
(torch.equal), time and algorithmic TFLOP/s of both.  `hedit_test_set_flags(8)` keeps the one-shot kernel.  Rows = the UNet batch
(default 120 = the 5n-row P2P pass of 24 images).   python tools/pgemm_ab.py [rows] [iters]"""
import math
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib

VARIANT = os.environ.get("HEDIT_LIB_VARIANT")       # tools/build_variant.sh side library (ablation builds: time only, bits not compared)
if VARIANT:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{VARIANT}.so.bin")
lib = _lib.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 120
ITERS = int(sys.argv[2]) if len(sys.argv) > 2 else 30


def timeit(fn, iters=ITERS):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def case(name, M, N, K, res, bias=True):
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + 7 * K)
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    bs = torch.randn(N, device=dev, generator=g) if bias else None
    R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16) if res else None
    outs = []
    times = []
    for flags in (8, 0, 8, 0):
        _lib.check(lib.hedit_test_set_flags(flags))
        out = torch.full((M, N), 7.0, device=dev, dtype=torch.bfloat16)
        f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bs) if bias else None, _lib.ptr(R) if res else None,
                                                _lib.ptr(out), M, N, K, K, N, N, 0, 0, 0, 0, 0, 0, 0, None, None))
        times.append(timeit(f))
        outs.append(out)
    _lib.check(lib.hedit_test_set_flags(0))
    same = VARIANT is not None or torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    fl = 2.0 * M * N * K
    t_old, t_new = min(times[0], times[2]), min(times[1], times[3])
    print(f"{name:26s} M={M:7d} N={N:5d} K={K:5d} res={int(res)}  one-shot {t_old:8.1f} us {fl / t_old / 1e6:7.0f} | persistent {t_new:8.1f} us "
          f"{fl / t_new / 1e6:7.0f}  ({t_old / t_new:5.2f}x)  bits {'n/a (' + VARIANT + ')' if VARIANT else 'equal' if same else 'DIFFER'}", flush=True)
    return same


ok = True
for lvl, (hw, c) in enumerate([(64, 320), (32, 640), (16, 1280)]):
    M = B * hw * hw
    ok &= case(f"L{lvl} linear C->C", M, c, c, False)
    ok &= case(f"L{lvl} out-proj C->C + res", M, c, c, True)
    ok &= case(f"L{lvl} qk C->2C", M, 2 * c, c, False, bias=False)
    if c * 4 // 64 < 24:
        ok &= case(f"L{lvl} ff2 4C->C + res", M, c, 4 * c, True)
# ragged shapes: M not a multiple of 256, N not a multiple of the tile
ok &= case("ragged M", 256 * 700 + 72, 640, 640, True)
ok &= case("ragged N (N = 200)", 256 * 2100, 200, 320, True)
ok &= case("N = 128 tile", 256 * 1030, 384, 512, False)
print("ALL BITS EQUAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
