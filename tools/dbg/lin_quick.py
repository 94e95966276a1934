import math, os, sys
sys.path.insert(0, "h-edit_amd")
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
rows, reps = 120, 20
lib = _lib.lib(); dev = "cuda:0"; C = 320; M = rows * 4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
a_in = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev); bo = torch.zeros(C, device=dev)
wo, wq, wk, wv = ((torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev) for _ in range(4))
ws2 = torch.empty(lib.hedit_k_lin_chain_stream_bytes(1), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_chain_pack(_lib.ptr(wo), _lib.ptr(wq), None, None, 1.0, _lib.ptr(ws2), None))
ws4 = torch.empty(lib.hedit_k_lin_chain_stream_bytes(3), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_chain_pack(_lib.ptr(wo), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), 1.0, _lib.ptr(ws4), None))
mid = torch.empty_like(x); q_c = torch.empty_like(x)
qk_c = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=dev); vt_c = torch.empty(C, M, dtype=torch.bfloat16, device=dev)
ss = torch.ones(rows, C, 2, dtype=torch.float32, device=dev)
def lin2():
    _lib.check(lib.hedit_k_lin_chain(_lib.ptr(a_in), C, _lib.ptr(x), C, None, 0, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws2), _lib.ptr(mid), C, None, 0, None, 0, _lib.ptr(q_c), C, M, C, None))
def lin4():
    _lib.check(lib.hedit_k_lin_chain(_lib.ptr(x), C, None, 0, _lib.ptr(ss), 4096, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws4), _lib.ptr(mid), C, _lib.ptr(qk_c), 2 * C, qk_c.data_ptr() + 2 * C, 2 * C, _lib.ptr(vt_c), M, M, C, None))
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print(os.environ.get("HEDIT_LIB_VARIANT", "product"), f"lin2 {timeit(lin2):.1f} us  lin4 {timeit(lin4):.1f} us")
