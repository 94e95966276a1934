"""HEDIT_LIB_VARIANT=name loads h-edit_amd/hedit/lib_name.so.bin (tools/build_variant.sh) instead of the product library.
Fused feed-forward (csrc/ffn.hip) against the unfused chain (layernorm, FF1+GEGLU GEMM, FF2 GEMM + residual) at the
bench's level-0 shape.  Usage: python tools/ffn_bench.py [rows=120] [reps=20]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
import torch
from hedit import _lib
if os.environ.get("HEDIT_LIB_VARIANT"):       # tools/build_variant.sh side library (measurement builds)
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 120
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = _lib.lib()
dev = "cuda:0"
C = lib.hedit_k_ffn_channels()
M = rows * 4096
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
gamma = torch.ones(C, device=dev); beta = torch.zeros(C, device=dev)
w1 = (torch.randn(8 * C, C, generator=g) / math.sqrt(C)).to(dev)
b1 = torch.zeros(8 * C, device=dev)
w2 = (torch.randn(C, 4 * C, generator=g) / math.sqrt(4 * C)).to(dev)
b2 = torch.zeros(C, device=dev)
ws = torch.empty(lib.hedit_k_ffn_stream_bytes(0), dtype=torch.uint8, device=dev)
bp = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), None, None, _lib.ptr(ws), _lib.ptr(bp), None))
out = torch.empty_like(x)
xn = torch.empty_like(x)
wp = torch.empty(8 * C, C, dtype=torch.bfloat16, device=dev)
b1p = torch.empty(8 * C, dtype=torch.float32, device=dev)
_lib.check(lib.hedit_k_pack_geglu(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(wp), _lib.ptr(b1p), 4 * C, C, None))
hid = torch.empty(M, 4 * C, dtype=torch.bfloat16, device=dev)
w2b = w2.to(torch.bfloat16).contiguous()
out_k = torch.empty_like(x)


def fused():
    _lib.check(lib.hedit_k_ffn_fused(_lib.ptr(x), C, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, _lib.ptr(ws), _lib.ptr(bp),
                                     _lib.ptr(b2), _lib.ptr(out), C, M, C, None))


def unfused():
    _lib.check(lib.hedit_k_layernorm(_lib.ptr(x), _lib.ptr(xn), _lib.ptr(gamma), _lib.ptr(beta), M, C, 1e-5, None))
    _lib.check(lib.hedit_k_gemm_geglu(_lib.ptr(xn), _lib.ptr(wp), _lib.ptr(b1p), _lib.ptr(hid), M, 4 * C, C, C, 4 * C, None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(hid), _lib.ptr(w2b), _lib.ptr(b2), _lib.ptr(x), _lib.ptr(out_k), M, C, 4 * C,
                                4 * C, C, C, 0, 0, 0, 0, 0, 0, 1, None, None))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# ---- the whole block tail: attn2.to_out + residual, feed-forward, proj_out + residual
a_in = (torch.randn(M, C, generator=g)).to(torch.bfloat16).to(dev)
t1 = (torch.randn(M, C, generator=g) * 1.5).to(torch.bfloat16).to(dev)
wo = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev); wpo = (torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev)
bo = torch.zeros(C, device=dev)
wsc = torch.empty(lib.hedit_k_ffn_stream_bytes(1), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(wo), _lib.ptr(wpo), _lib.ptr(wsc), _lib.ptr(bp), None))
wob, wpb = wo.to(torch.bfloat16).contiguous(), wpo.to(torch.bfloat16).contiguous()
t2 = torch.empty_like(x); out_c = torch.empty_like(x); out_ck = torch.empty_like(x)


def chain():
    _lib.check(lib.hedit_k_ffn_chain(_lib.ptr(a_in), C, _lib.ptr(t1), C, _lib.ptr(x), C, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(wsc), _lib.ptr(bp), _lib.ptr(b2), _lib.ptr(bo), _lib.ptr(out_c), C, M, C, None))


def chain_unfused():
    _lib.check(lib.hedit_k_gemm(_lib.ptr(a_in), _lib.ptr(wob), _lib.ptr(bo), _lib.ptr(t1), _lib.ptr(t2), M, C, C, C, C, C, 0, 0, 0, 0, 0, 0, 1, None, None))
    _lib.check(lib.hedit_k_layernorm(_lib.ptr(t2), _lib.ptr(xn), _lib.ptr(gamma), _lib.ptr(beta), M, C, 1e-5, None))
    _lib.check(lib.hedit_k_gemm_geglu(_lib.ptr(xn), _lib.ptr(wp), _lib.ptr(b1p), _lib.ptr(hid), M, 4 * C, C, C, 4 * C, None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(hid), _lib.ptr(w2b), _lib.ptr(b2), _lib.ptr(t2), _lib.ptr(out_k), M, C, 4 * C,
                                4 * C, C, C, 0, 0, 0, 0, 0, 0, 1, None, None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(out_k), _lib.ptr(wpb), _lib.ptr(bo), _lib.ptr(x), _lib.ptr(out_ck), M, C, C, C, C, C, 0, 0, 0, 0, 0, 0, 1, None, None))


# ---- the projection chains around the two attentions
wq, wk, wv = ((torch.randn(C, C, generator=g) / math.sqrt(C)).to(dev) for _ in range(3))
wqb, wvb = wq.to(torch.bfloat16).contiguous(), wv.to(torch.bfloat16).contiguous()
wqkb = torch.cat([wq, wk], 0).to(torch.bfloat16).contiguous()
ws2 = torch.empty(lib.hedit_k_lin_chain_stream_bytes(1), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_chain_pack(_lib.ptr(wo), _lib.ptr(wq), None, None, 1.0, _lib.ptr(ws2), None))
ws4 = torch.empty(lib.hedit_k_lin_chain_stream_bytes(3), dtype=torch.uint8, device=dev)
_lib.check(lib.hedit_k_lin_chain_pack(_lib.ptr(wo), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), 1.0, _lib.ptr(ws4), None))
mid = torch.empty_like(x); q_c = torch.empty_like(x); q_k = torch.empty_like(x)
qk_c = torch.empty(M, 2 * C, dtype=torch.bfloat16, device=dev); qk_k = torch.empty_like(qk_c)
vt_c = torch.empty(C, M, dtype=torch.bfloat16, device=dev); vt_k = torch.empty_like(vt_c)
gws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(rows, 4096, C), dtype=torch.uint8, device=dev)
ss = torch.empty(rows, C, 2, dtype=torch.float32, device=dev)


def lin2():
    _lib.check(lib.hedit_k_lin_chain(_lib.ptr(a_in), C, _lib.ptr(t1), C, None, 0, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws2), _lib.ptr(mid), C, None, 0, None, 0, _lib.ptr(q_c), C, M, C, None))


def lin2_unfused():
    _lib.check(lib.hedit_k_gemm(_lib.ptr(a_in), _lib.ptr(wob), _lib.ptr(bo), _lib.ptr(t1), _lib.ptr(t2), M, C, C, C, C, C, 0, 0, 0, 0, 0, 0, 1, None, None))
    _lib.check(lib.hedit_k_layernorm(_lib.ptr(t2), _lib.ptr(xn), _lib.ptr(gamma), _lib.ptr(beta), M, C, 1e-5, None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(xn), _lib.ptr(wqb), None, None, _lib.ptr(q_k), M, C, C, C, C, 0, 0, 0, 0, 0, 0, 0, 1, None, None))


def lin4():
    _lib.check(lib.hedit_k_groupnorm_affine(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), rows, 4096, C, 32, 1e-6, _lib.ptr(gws), _lib.ptr(ss), None))
    _lib.check(lib.hedit_k_lin_chain(_lib.ptr(x), C, None, 0, _lib.ptr(ss), 4096, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                     _lib.ptr(ws4), _lib.ptr(mid), C, _lib.ptr(qk_c), 2 * C, qk_c.data_ptr() + 2 * C, 2 * C,
                                     _lib.ptr(vt_c), M, M, C, None))


def lin4_unfused():
    _lib.check(lib.hedit_k_groupnorm(_lib.ptr(x), _lib.ptr(out_k), _lib.ptr(gamma), _lib.ptr(beta), rows, 4096, C, 32, 1e-6, 0, _lib.ptr(gws), None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(out_k), _lib.ptr(wob), _lib.ptr(bo), None, _lib.ptr(t2), M, C, C, C, C, 0, 0, 0, 0, 0, 0, 0, 1, None, None))
    _lib.check(lib.hedit_k_layernorm(_lib.ptr(t2), _lib.ptr(xn), _lib.ptr(gamma), _lib.ptr(beta), M, C, 1e-5, None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(xn), _lib.ptr(wqkb), None, None, _lib.ptr(qk_k), M, 2 * C, C, C, 2 * C, 0, 0, 0, 0, 0, 0, 0, 1, None, None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(wvb), _lib.ptr(xn), None, None, _lib.ptr(vt_k), C, M, C, C, M, 0, 0, 0, 0, 0, 0, 0, 1, None, None))


rel = lambda u, v: ((u.float() - v.float()).norm() / v.float().norm()).item()
t2f, t2u = timeit(lin2), timeit(lin2_unfused)
print(f"rows {rows}: to_out + norm + to_q fused {t2f * 1e3:.1f} us = {2.0 * M * 2 * C * C / t2f / 1e9:.0f} TF/s | three launches {t2u * 1e3:.1f} us | "
      f"speed-up {t2u / t2f:.2f} | rel diff {rel(q_c, q_k):.2e}")
t4f, t4u = timeit(lin4), timeit(lin4_unfused)
print(f"rows {rows}: GroupNorm + proj_in + norm + q|k|v^T fused {t4f * 1e3:.1f} us = {2.0 * M * 4 * C * C / t4f / 1e9:.0f} TF/s | unfused {t4u * 1e3:.1f} us | "
      f"speed-up {t4u / t4f:.2f} | rel diff qk {rel(qk_c, qk_k):.2e} v^T {rel(vt_c, vt_k):.2e}")

flops = 2.0 * M * 12 * C * C
tc, tcu = timeit(chain), timeit(chain_unfused)
errc = ((out_c.float() - out_ck.float()).norm() / out_ck.float().norm()).item()
print(f"rows {rows}: block tail (to_out + ff + proj_out) fused {tc * 1e3:.1f} us = {2.0 * M * 14 * C * C / tc / 1e9:.0f} TF/s | five launches "
      f"{tcu * 1e3:.1f} us = {2.0 * M * 14 * C * C / tcu / 1e9:.0f} TF/s | speed-up {tcu / tc:.2f} | rel diff {errc:.2e}")
tf, tu = timeit(fused), timeit(unfused)
err = ((out.float() - out_k.float()).norm() / out_k.float().norm()).item()
print(f"rows {rows} (M = {M}): fused {tf * 1e3:.1f} us = {flops / tf / 1e9:.0f} TF/s | unfused chain {tu * 1e3:.1f} us = "
      f"{flops / tu / 1e9:.0f} TF/s | speed-up {tu / tf:.2f} | rel diff {err:.2e}")
