"""lintile.hip at the 640-channel level (build with -DLINTILE_C=640, the default): the two projection chains of a transformer block
-- front: GroupNorm apply + proj_in + LayerNorm + q | k | v^T; mid: attn1.to_out + residual + LayerNorm + attn2.to_q -- in one
launch each, timed and compared with fp32 torch.  What they replace in the product (tools/layer_prof.py 120): front = GroupNorm
apply + proj_in (160 us) + LayerNorm (64) + q|k (279) + v^T (143); mid = to_out (160) + LayerNorm (64) + to_q (160).
python tools/experiments/lin_bench640.py [rows=120] [reps=20]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
import torch.nn.functional as F
from hedit import _lib
import ctypes as C
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "lib_xffn.so.bin")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.lib(); dev = "cuda:0"
lib.hedit_k_lin_tile_stream_bytes.restype = C.c_size_t; lib.hedit_k_lin_tile_stream_bytes.argtypes = [C.c_int]
lib.hedit_k_lin_tile_pack.restype = C.c_int; lib.hedit_k_lin_tile_pack.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_void_p]
lib.hedit_k_lin_tile.restype = C.c_int; lib.hedit_k_lin_tile.argtypes = lib.hedit_k_lin_chain.argtypes
Cc = 640; N = 1024; M = rows * N
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, Cc, generator=g) * 1.5).to(torch.bfloat16).to(dev)
a = torch.randn(M, Cc, generator=g).to(torch.bfloat16).to(dev)
t1 = (torch.randn(M, Cc, generator=g) * 1.5).to(torch.bfloat16).to(dev)
gamma = (1 + 0.1 * torch.randn(Cc, generator=g)).to(dev); beta = (0.1 * torch.randn(Cc, generator=g)).to(dev)
bo = (0.3 * torch.randn(Cc, generator=g)).to(dev)
mk = lambda: (torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc)).to(dev)
wo, wq, wk, wv = mk(), mk(), mk(), mk()
gws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(rows, N, Cc), dtype=torch.uint8, device=dev)
ss = torch.empty(rows, Cc, 2, dtype=torch.float32, device=dev)
_lib.check(lib.hedit_k_groupnorm_affine(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), rows, N, Cc, 32, 1e-6, _lib.ptr(gws), _lib.ptr(ss), None))
p = _lib.ptr
ws2 = torch.empty(lib.hedit_k_lin_tile_stream_bytes(1), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_tile_pack(p(wo), p(wq), None, None, 0.23, p(ws2), None))
ws4 = torch.empty(lib.hedit_k_lin_tile_stream_bytes(3), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_tile_pack(p(wo), p(wq), p(wk), p(wv), 0.23, p(ws4), None))
mid = torch.empty_like(x); q = torch.empty_like(x)
qk = torch.empty(M, 2 * Cc, dtype=torch.bfloat16, device=dev); vt = torch.empty(Cc, M, dtype=torch.bfloat16, device=dev)


def f_mid():
    _lib.check(lib.hedit_k_lin_tile(p(a), Cc, p(t1), Cc, None, 0, p(bo), p(gamma), p(beta), 1e-5, p(ws2), p(mid), Cc, None, 0, None, 0, p(q), Cc, M, Cc, None))


def f_front():
    _lib.check(lib.hedit_k_lin_tile(p(x), Cc, None, 0, p(ss), N, p(bo), p(gamma), p(beta), 1e-5, p(ws4), p(mid), Cc, p(qk), 2 * Cc,
                                    qk.data_ptr() + 2 * Cc, 2 * Cc, p(vt), M, M, Cc, None))


bfr = lambda t: t.to(torch.bfloat16).float()
for name, fn, layers, tensors in (("mid   (to_out + res, LN, to_q)", f_mid, 2, 4), ("front (GN, proj_in, LN, q|k|v^T)", f_front, 4, 6)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"lin_tile C=640 {name:34s} M = {M}: {ms * 1e3:8.1f} us  {2.0 * M * layers * Cc * Cc / ms / 1e9:7.1f} TFLOP/s  {tensors * 2.0 * M * Cc / ms / 1e6:7.0f} GB/s", flush=True)
# fp32 reference on the first 8 images' worth of rows (weights and inputs as the kernel sees them: bf16)
R = min(M, 8 * N)
wob, wqb, wkb, wvb = bfr(wo), bfr(wq * 0.23), bfr(wk), bfr(wv)
f_mid(); f_front(); torch.cuda.synchronize()
f_mid(); torch.cuda.synchronize()
t = a[:R].float() @ wob.t() + bo + t1[:R].float()
print("  mid  : t1  rel L2", float((mid[:R].float() - t).norm() / t.norm()))
qq = F.layer_norm(t, (Cc,), gamma, beta, 1e-5) @ wqb.t()
print("  mid  : q2  rel L2", float((q[:R].float() - qq).norm() / qq.norm()))
f_front(); torch.cuda.synchronize()
xn = x[:R].float() * ss[:R // N].repeat_interleave(N, 0)[:, :, 0] + ss[:R // N].repeat_interleave(N, 0)[:, :, 1]
t = bfr(xn) @ wob.t() + bo
print("  front: t0  rel L2", float((mid[:R].float() - t).norm() / t.norm()))
ln = F.layer_norm(t, (Cc,), gamma, beta, 1e-5)
for nm, w, got in (("q", wqb, qk[:R, :Cc]), ("k", wkb, qk[:R, Cc:]), ("v^T", wvb, vt[:, :R].t())):
    want = ln @ w.t()
    print(f"  front: {nm:3s} rel L2", float((got.float() - want).norm() / want.norm()))
