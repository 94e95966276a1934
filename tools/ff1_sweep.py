"""FF1 + GEGLU launch time against K (fixed cost per tile vs K-loop slope) and against M (one round of tiles vs many)."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
v = os.environ.get("HEDIT_LIB_VARIANT")
if v:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{v}.so.bin")
lib = _lib.lib(); dev = torch.device("cuda:0")


def geglu(M, inner, K, iters=20):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(2 * inner, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(2 * inner, device=dev)
    out = torch.empty(M, inner, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.check(lib.hedit_k_gemm_geglu(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, inner, K, K, inner, None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def plain(M, N, K, iters=20):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, N, K, K, N, N, 0, 0, 0, 0, 0, 0, 0, None, None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


M = 491520
for K in (64, 128, 192, 320, 640, 1280):
    us = geglu(M, 1280, K)
    tiles = M // 256 * 10
    print(f"geglu M={M} N=2560 K={K:5d}: {us:8.1f} us  {4.0 * M * 1280 * K / us / 1e6:7.1f} TF/s  per tile-round {us / (tiles / 256):6.2f} us", flush=True)
for rounds in (1, 2, 4, 16):
    Mr = 256 * 256 * rounds // 10 // 256 * 256
    us = geglu(Mr, 1280, 320)
    print(f"geglu K=320 M={Mr} ({Mr // 256 * 10} tiles): {us:8.1f} us", flush=True)
for K in (64, 320, 640):
    us = plain(M, 320, K)
    print(f"plain M={M} N=320 K={K}: {us:8.1f} us  {2.0 * M * 320 * K / us / 1e6:7.1f} TF/s", flush=True)
    us = plain(M, 640, K)
    print(f"plain M={M} N=640 K={K}: {us:8.1f} us  {2.0 * M * 640 * K / us / 1e6:7.1f} TF/s", flush=True)


def plain_res(M, N, K, iters=20):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(R), _lib.ptr(out), M, N, K, K, N, N, 0, 0, 0, 0, 0, 0, 0, None, None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for N, K in ((320, 320), (320, 1280), (640, 640), (640, 2560), (1280, 1280)):
    Mx = M if N == 320 else (M // 4 if N == 640 else M // 16)
    us = plain_res(Mx, N, K)
    print(f"residual M={Mx} N={N} K={K}: {us:8.1f} us  {2.0 * Mx * N * K / us / 1e6:7.1f} TF/s  {(Mx * K + 2 * Mx * N) * 2 / us / 1e6:.2f} TB/s", flush=True)
