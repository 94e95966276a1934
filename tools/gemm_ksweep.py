import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
def run(M, N, K, res=False, bias=True, iters=20):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device=dev) if bias else None
    R = torch.randn(M, N, device=dev).to(torch.bfloat16) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, N, K, 1), 16), dtype=torch.uint8, device=dev)
    f = lambda: _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(R), _lib.ptr(out), M, N, K, K, N, N, 0, 0, 0, 0, 0, 0, 1, _lib.ptr(ws), None))
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M = 163840
for N in (320, 2560):
    for K in (64, 128, 320, 640, 1280, 2560):
        us = run(M, N, K)
        print(f"M={M} N={N} K={K:5d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TF/s   out+in bytes {(M*N+M*K)*2/1e6:.0f} MB -> {(M*N+M*K)*2/us/1e6:.2f} TB/s")
print("residual:", run(M, 320, 320, res=True), "no bias:", run(M, 320, 320, bias=False))
