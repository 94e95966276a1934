"""The block tail of the 320-channel level at the bench's shape: csrc/xffn.hip (row tile in LDS, weight slices from L2 into
registers) against csrc/ffn.hip (rows in registers, weights through LDS) on the same operands: time, TFLOP/s, difference.
Usage: tools/experiments/build_xffn.sh (no GPU needed), then on the GPU box: python tools/experiments/xffn_bench.py [rows=120] [reps=10]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
import ctypes as C
from hedit import _lib
# the side library of tools/experiments/build_xffn.sh: the product library plus the experiment's units
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "lib_xffn.so.bin")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
lib = _lib.lib(); dev = "cuda:0"
lib.hedit_k_xffn_stream_bytes.restype = C.c_size_t; lib.hedit_k_xffn_stream_bytes.argtypes = []
lib.hedit_k_xffn_bias_bytes.restype = C.c_size_t; lib.hedit_k_xffn_bias_bytes.argtypes = []
lib.hedit_k_xffn_pack.restype = C.c_int; lib.hedit_k_xffn_pack.argtypes = [C.c_void_p] * 8
lib.hedit_k_xffn_chain.restype = C.c_int
lib.hedit_k_xffn_chain.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]; C = lib.hedit_k_ffn_channels(); M = rows * 4096
g = torch.Generator().manual_seed(0)
a = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
t1 = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
x = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev); beta = (0.1 * torch.randn(C, generator=g)).to(dev)
mk = lambda o, i: (torch.randn(o, i, generator=g) / math.sqrt(i)).to(dev)
wo, wpo, w1, w2 = mk(C, C), mk(C, C), mk(8 * C, C), mk(C, 4 * C)
bo, bpo, b2 = ((0.3 * torch.randn(C, generator=g)).to(dev) for _ in range(3))
b1 = (0.5 * torch.randn(8 * C, generator=g)).to(dev)
ws_n = torch.empty(lib.hedit_k_xffn_stream_bytes(), dtype=torch.uint8, device=dev)
bp_n = torch.empty(lib.hedit_k_xffn_bias_bytes(), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_xffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(wo), _lib.ptr(wpo), _lib.ptr(ws_n), _lib.ptr(bp_n), None))
ws_o = torch.empty(lib.hedit_k_ffn_stream_bytes(1), dtype=torch.uint8, device=dev)
bp_o = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(wo), _lib.ptr(wpo), _lib.ptr(ws_o), _lib.ptr(bp_o), None))
out_n, out_o = torch.empty_like(x), torch.empty_like(x)


def new():
    _lib.check(lib.hedit_k_xffn_chain(_lib.ptr(a), C, _lib.ptr(t1), C, _lib.ptr(x), C, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                      _lib.ptr(ws_n), _lib.ptr(bp_n), _lib.ptr(b2), _lib.ptr(bpo), _lib.ptr(out_n), C, M, C, None))


def old():
    _lib.check(lib.hedit_k_ffn_chain(_lib.ptr(a), C, _lib.ptr(t1), C, _lib.ptr(x), C, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws_o), _lib.ptr(bp_o), _lib.ptr(b2), _lib.ptr(bpo), _lib.ptr(out_o), C, M, C, None))


def timeit(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


flops = 2.0 * M * 14 * C * C
for name, fn in (("xffn (tile in LDS)", new), ("ffn  (rows in registers)", old), ("xffn (tile in LDS)", new), ("ffn  (rows in registers)", old)):
    ms = timeit(fn)
    print(f"{name:26s} M = {M}: {ms * 1e3:8.1f} us  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
d = (out_n.float() - out_o.float())
print(f"difference new vs old: rel L2 {float(d.norm() / out_o.float().norm()):.2e}, max abs {float(d.abs().max()):.3f}, finite {bool(torch.isfinite(out_n.float()).all())}")
