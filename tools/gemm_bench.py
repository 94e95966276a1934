"""Micro-benchmark of the GEMM / attention shapes of one SD-1.5 UNet call (B rows) through the C ABI."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import ctypes as C
import torch
from hedit import _lib

if os.environ.get("HEDIT_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
lib = _lib.lib()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def gemm(M, N, K, mode=0, conv=None, name=""):
    Cin = conv[2] if conv else K
    A = torch.randn(M if mode == 0 else conv[5], K if mode == 0 else Cin, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, N, K, 0), 16), dtype=torch.uint8, device=dev)
    cv = conv[:5] if conv else (0, 0, 0, 0, 0)
    f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, N, K,
                                            K if mode == 0 else Cin, N, N, mode, *cv, 0, _lib.ptr(ws), None))
    us = timeit(f)
    fl = 2.0 * M * N * K
    print(f"{name:28s} M={M:7d} N={N:5d} K={K:6d} mode={mode} {us:9.1f} us  {fl / us / 1e6:8.1f} TF/s")
    return us


def gemm_geglu(M, inner, K, name):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(2 * inner, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(2 * inner, device=dev)
    out = torch.empty(M, inner, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.check(lib.hedit_k_gemm_geglu(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, inner, K, K, inner, None))
    us = timeit(f)
    print(f"{name:28s} M={M:7d} N={2 * inner:5d} K={K:6d} geglu  {us:9.1f} us  {4.0 * M * inner * K / us / 1e6:8.1f} TF/s")


tot = 0
for lvl, (hw, c) in enumerate([(64, 320), (32, 640), (16, 1280), (8, 1280)]):
    M = B * hw * hw
    tot += gemm(M, c, 9 * c, 1, (hw, hw, c, hw, hw, M), f"L{lvl} conv3x3 {c}->{c}")
    if lvl < 3:
        tot += gemm(M, c, c, 0, None, f"L{lvl} linear C->C")
        tot += gemm(M, 2 * c, c, 0, None, f"L{lvl} qk C->2C")
        tot += gemm(c, M, c, 0, None, f"L{lvl} v^T (swapped)")
        tot += gemm(M, 8 * c, c, 0, None, f"L{lvl} ff1 C->8C")
        tot += gemm(M, c, 4 * c, 0, None, f"L{lvl} ff2 4C->C")
for lvl, (hw, c) in enumerate([(64, 320), (32, 640), (16, 1280)]):
    gemm_geglu(B * hw * hw, 4 * c, c, f"L{lvl} ff1+geglu fused")
gemm(B * 64 * 64, 320, 9 * 960, 1, (64, 64, 960, 64, 64, B * 64 * 64), "up3 conv 960->320")
gemm(B * 16 * 16, 1280, 9 * 2560, 1, (16, 16, 2560, 16, 16, B * 256), "up1 conv 2560->1280")

# self attention
for (N, c) in [(4096, 320), (1024, 640), (256, 1280)]:
    heads, d = 8, c // 8
    qk = torch.randn(B * N, 2 * c, device=dev).to(torch.bfloat16) * 0.3
    vt = torch.randn(c, B * N, device=dev).to(torch.bfloat16)
    out = torch.empty(B * N, c, device=dev, dtype=torch.bfloat16)
    kv = qk[:, c:]
    f = lambda: _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * c, C.c_void_p(kv.data_ptr()), 2 * c, _lib.ptr(vt), B * N,
                                                 _lib.ptr(out), c, B, N, heads, d, None, None, None))
    us = timeit(f, 10)
    fl = 4.0 * B * N * N * c
    print(f"self_attn N={N} d={d} {us:9.1f} us {fl / us / 1e6:8.1f} TF/s")
