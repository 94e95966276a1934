"""Launch the dominant kernel shapes a few times (for rocprofv3 --pmc passes)."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import ctypes as C
import torch
from hedit import _lib
lib = _lib.lib(); dev = torch.device("cuda:0")
B = 16
def gemm(M, N, K, mode=0, conv=None):
    Cin = conv[2] if conv else K
    A = torch.randn(M, Cin if mode else K, device=dev).to(torch.bfloat16)
    W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(max(lib.hedit_k_gemm_ws_bytes(M, N, K, 0), 16), dtype=torch.uint8, device=dev)
    cv = conv[:5] if conv else (0, 0, 0, 0, 0)
    for _ in range(3):
        _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), None, _lib.ptr(out), M, N, K, Cin if mode else K, N, N, mode, *cv, 0, _lib.ptr(ws), None))
    torch.cuda.synchronize()
M = B * 4096
gemm(M, 320, 2880, 1, (64, 64, 320, 64, 64))      # L0 conv
gemm(M, 2560, 320)                                  # L0 ff1
gemm(M, 320, 1280)                                  # L0 ff2
N, c = 4096, 320
qk = torch.randn(B * N, 2 * c, device=dev).to(torch.bfloat16) * 0.3
vt = torch.randn(c, B * N, device=dev).to(torch.bfloat16)
out = torch.empty(B * N, c, device=dev, dtype=torch.bfloat16)
kv = qk[:, c:]
for _ in range(3):
    _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * c, C.c_void_p(kv.data_ptr()), 2 * c, _lib.ptr(vt), B * N, _lib.ptr(out), c, B, N, 8, 40, None, None, None))
torch.cuda.synchronize()
