"""Repeatability of the chain kernels at the bench's level-0 shape: each kernel REPS times on the same inputs, every output
compared bit for bit with the first run's (a data race in the LDS staging / DMA windows would show up as a mismatch).
python tools/chain_repeat.py [rows=120] [reps=40]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = _lib.lib(); dev = "cuda:0"; C = lib.hedit_k_ffn_channels(); N = 4096; M = rows * N
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
a_in = torch.randn(M, C, generator=g).to(torch.bfloat16).to(dev)
t1 = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev); beta = (0.1 * torch.randn(C, generator=g)).to(dev)
bo = (0.3 * torch.randn(C, generator=g)).to(dev)
mk = lambda o, i: (torch.randn(o, i, generator=g) / math.sqrt(i)).to(dev)
wo, wq, wk, wv, wpo = mk(C, C), mk(C, C), mk(C, C), mk(C, C), mk(C, C)
w1, w2 = mk(8 * C, C), mk(C, 4 * C)
b1 = (0.5 * torch.randn(8 * C, generator=g)).to(dev); b2 = (0.5 * torch.randn(C, generator=g)).to(dev)
ws2 = torch.empty(lib.hedit_k_lin_chain_stream_bytes(1), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_chain_pack(_lib.ptr(wo), _lib.ptr(wq), None, None, 0.23, _lib.ptr(ws2), None))
ws4 = torch.empty(lib.hedit_k_lin_chain_stream_bytes(3), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_chain_pack(_lib.ptr(wo), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), 0.23, _lib.ptr(ws4), None))
wsc = torch.empty(lib.hedit_k_ffn_stream_bytes(1), dtype=torch.uint8, device=dev)
bp = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(wo), _lib.ptr(wpo), _lib.ptr(wsc), _lib.ptr(bp), None))
gws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(rows, N, C), dtype=torch.uint8, device=dev)
ss = torch.empty(rows, C, 2, dtype=torch.float32, device=dev)
_lib.check(lib.hedit_k_groupnorm_affine(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), rows, N, C, 32, 1e-6, _lib.ptr(gws), _lib.ptr(ss), None))


def lin2():
    mid = torch.empty_like(x); q = torch.empty_like(x)
    _lib.check(lib.hedit_k_lin_chain(_lib.ptr(a_in), C, _lib.ptr(t1), C, None, 0, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws2), _lib.ptr(mid), C, None, 0, None, 0, _lib.ptr(q), C, M, C, None))
    return mid, q


def lin4():
    mid = torch.empty_like(x); qk = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=dev); vt = torch.empty(C, M, dtype=torch.bfloat16, device=dev)
    _lib.check(lib.hedit_k_lin_chain(_lib.ptr(x), C, None, 0, _lib.ptr(ss), N, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws4), _lib.ptr(mid), C, _lib.ptr(qk), 2 * C, qk.data_ptr() + 2 * C, 2 * C, _lib.ptr(vt), M, M, C, None))
    return mid, qk, vt


def tail():
    out = torch.empty_like(x)
    _lib.check(lib.hedit_k_ffn_chain(_lib.ptr(a_in), C, _lib.ptr(t1), C, _lib.ptr(x), C, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(wsc), _lib.ptr(bp), _lib.ptr(b2), _lib.ptr(bo), _lib.ptr(out), C, M, C, None))
    return (out,)


for name, fn in (("lin2", lin2), ("lin4", lin4), ("tail", tail)):
    ref = fn(); torch.cuda.synchronize()
    bad = 0
    for r in range(reps):
        out = fn(); torch.cuda.synchronize()
        for u, v in zip(ref, out):
            if not torch.equal(u, v):
                bad += 1
                d = (u.float() - v.float()).abs()
                idx = torch.nonzero(d.reshape(-1) > 0)
                print(f"  {name} run {r}: {idx.numel()} elements differ, first at flat index {int(idx[0])}, max diff {float(d.max()):.3e}")
    print(f"{name}: {reps} repeats, {bad} mismatching outputs")
