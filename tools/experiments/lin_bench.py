"""The projection chains around the attentions of the 320-channel level at the bench's shape: csrc/lintile.hip (64-row tile in
LDS, weight slices from L2 into registers, two blocks per CU) against csrc/linchain.hip (rows in registers, weights through
LDS) on the same operands: time per launch, algorithmic TFLOP/s and GB/s, difference.  LINTILE_C=320 tools/experiments/build_xffn.sh (lintile.hip builds for one channel count), then on the GPU box: python tools/experiments/lin_bench.py [rows=120] [reps=20]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
import ctypes as C
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "lib_xffn.so.bin")      # side library of tools/experiments/build_xffn.sh
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.lib(); dev = "cuda:0"
lib.hedit_k_lin_tile_stream_bytes.restype = C.c_size_t; lib.hedit_k_lin_tile_stream_bytes.argtypes = [C.c_int]
lib.hedit_k_lin_tile_pack.restype = C.c_int; lib.hedit_k_lin_tile_pack.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]
lib.hedit_k_lin_tile.restype = C.c_int; lib.hedit_k_lin_tile.argtypes = lib.hedit_k_lin_chain.argtypes; C = 320; N = 4096; M = rows * N
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
a = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
t1 = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev); beta = (0.1 * torch.randn(C, generator=g)).to(dev)
bo = (0.3 * torch.randn(C, generator=g)).to(dev)
mk = lambda: (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev)
wo, wq, wk, wv = mk(), mk(), mk(), mk()
gws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(rows, N, C), dtype=torch.uint8, device=dev)
ss = torch.empty(rows, C, 2, dtype=torch.float32, device=dev)
_lib.check(lib.hedit_k_groupnorm_affine(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), rows, N, C, 32, 1e-6, _lib.ptr(gws), _lib.ptr(ss), None))
p = _lib.ptr
res = {}
for impl in ("chain", "tile"):
    f_bytes, f_pack, f_run = (getattr(lib, f"hedit_k_lin_{impl}{s}") for s in ("_stream_bytes", "_pack", ""))
    ws2 = torch.empty(f_bytes(1), dtype=torch.uint8, device=dev); _lib.check(f_pack(p(wo), p(wq), None, None, 0.23, p(ws2), None))
    ws4 = torch.empty(f_bytes(3), dtype=torch.uint8, device=dev); _lib.check(f_pack(p(wo), p(wq), p(wk), p(wv), 0.23, p(ws4), None))
    mid = torch.empty_like(x); q = torch.empty_like(x)
    qk = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=dev); vt = torch.empty(C, M, dtype=torch.bfloat16, device=dev)

    def f_mid():
        _lib.check(f_run(p(a), C, p(t1), C, None, 0, p(bo), p(gamma), p(beta), 1e-5, p(ws2), p(mid), C, None, 0, None, 0, p(q), C, M, C, None))

    def f_front():
        _lib.check(f_run(p(x), C, None, 0, p(ss), N, p(bo), p(gamma), p(beta), 1e-5, p(ws4), p(mid), C, p(qk), 2 * C, qk.data_ptr() + 2 * C, 2 * C,
                         p(vt), M, M, C, None))
    for name, fn, layers, tensors in (("mid   (to_out + res, LN, to_q)", f_mid, 2, 4), ("front (GN, proj_in, LN, q|k|v^T)", f_front, 4, 5)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"lin_{impl:5s} {name:34s} M = {M}: {ms * 1e3:8.1f} us  {2.0 * M * layers * C * C / ms / 1e9:7.1f} TFLOP/s  {tensors * 2.0 * M * C / ms / 1e6:7.0f} GB/s", flush=True)
        res[(impl, name)] = [t.clone() for t in ((mid, q) if layers == 2 else (mid, qk, vt))]
for name in sorted({k[1] for k in res}):
    for u, v in zip(res[("chain", name)], res[("tile", name)]):
        d = (u.float() - v.float())
        print(f"  {name[:5]}: tile vs chain rel L2 {float(d.norm() / u.float().norm()):.2e}")
