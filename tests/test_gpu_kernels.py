"""-m gpu: every HIP kernel, called through the C ABI, against a plain PyTorch fp32 reference of
the same op on the same (bf16-rounded) inputs.  Tolerances are stated per test: bf16 storage of
the output costs 2^-9 relative per element; accumulation is fp32."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import gpu as G  # noqa: E402
from hedit import _lib  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _lib.lib()


def run_gemm(lib, A, W, bias, res, M, N, K, lda, ldc, ldr, mode=0, conv=(0, 0, 0, 0, 0), splits=0):
    out = torch.zeros(M, ldc, dtype=_lib.storage_dtype(), device=G.dev())
    wsb = lib.hedit_k_gemm_ws_bytes(M, N, K, splits)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_gemm(_lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), M, N, K,
                                lda, ldc, ldr, mode, *conv, splits, _lib.ptr(ws), None))
    G.sync()
    return out


@pytest.mark.parametrize("M,N,K,splits", [(256, 320, 320, 0), (128, 128, 64, 1), (1000, 640, 1280, 0),
                                            (64, 1280, 1280, 0), (4096, 2560, 320, 1), (320, 1024, 320, 1),
                                            (200, 132, 192, 3), (256, 1280, 11520, 0)])
def test_gemm_linear(lib, M, N, K, splits):
    g = torch.Generator().manual_seed(M + N + K)
    A = G.bf(torch.randn(M, K, generator=g))
    W = G.bf(torch.randn(N, K, generator=g) / math.sqrt(K))
    bias = G.f32(torch.randn(N, generator=g))
    res = G.bf(torch.randn(M, N, generator=g))
    out = run_gemm(lib, A, W, bias, res, M, N, K, K, N, N, splits=splits)
    want = A.float() @ W.float().t() + bias + res.float()
    # asymmetric operands: a transposed C-write would be caught (guide rule 16)
    assert G.rel_err(out.float(), want) < 6e-3
    out2 = run_gemm(lib, A, W, None, None, M, N, K, K, N, N, splits=splits)
    assert G.rel_err(out2.float(), A.float() @ W.float().t()) < 6e-3


@pytest.mark.parametrize("mode,B,H,Wd,Cin,Cout", [(1, 2, 16, 16, 64, 128), (1, 1, 8, 8, 320, 320),
                                                    (2, 2, 16, 16, 64, 64), (3, 2, 8, 8, 128, 128),
                                                    (1, 4, 32, 32, 192, 320), (1, 2, 8, 8, 1280, 1280)])
def test_gemm_conv3x3(lib, mode, B, H, Wd, Cin, Cout):
    g = torch.Generator().manual_seed(mode * 1000 + Cin + Cout)
    x = torch.randn(B, Cin, H, Wd, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = torch.randn(Cout, generator=g)
    xb = G.bf(x.permute(0, 2, 3, 1))                       # NHWC
    wq = torch.empty(Cout * 9 * Cin, dtype=_lib.storage_dtype(), device=G.dev())
    wd = G.f32(w)
    _lib.check(lib.hedit_k_pack_conv3x3(_lib.ptr(wd), _lib.ptr(wq), Cout, Cin, None))
    xr = xb.float().permute(0, 3, 1, 2)
    wr = G.bf(w).float()
    if mode == 1:
        want = F.conv2d(xr, wr, G.f32(bias), padding=1)
    elif mode == 2:
        want = F.conv2d(xr, wr, G.f32(bias), stride=2, padding=1)
    else:
        want = F.conv2d(F.interpolate(xr, scale_factor=2.0, mode="nearest"), wr, G.f32(bias), padding=1)
    Ho, Wo = want.shape[2], want.shape[3]
    M = B * Ho * Wo
    out = run_gemm(lib, xb, wq, G.f32(bias), None, M, Cout, 9 * Cin, Cin, Cout, Cout, mode=mode,
                   conv=(H, Wd, Cin, Ho, Wo))
    got = out.float().reshape(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert G.rel_err(got, want) < 6e-3


@pytest.mark.parametrize("mode,B,H,Cin,Cout,splits", [
    (1, 100, 8, 128, 1280, 0),     # W = 8: a 16-row sub-tile spans two image rows
    (1, 100, 8, 128, 1024, -2),    # chunk fold on the 128-column tile
    (1, 101, 8, 64, 1280, 0),      # ragged last tile (rows beyond M), four images per tile
    (1, 52, 16, 64, 640, 0), (1, 13, 64, 64, 320, 0), (1, 26, 32, 128, 256, -3), (1, 3, 128, 64, 192, 0),
    (3, 56, 8, 64, 640, 0), (3, 14, 32, 64, 320, 0), (3, 52, 8, 128, 1024, -2)])
def test_conv3x3_row_sharing_loop_bits(lib, mode, B, H, Cin, Cout, splits):
    """Large launches of the stride-1 / upsampling 3x3 take the 256-row kernel whose three taps of a kernel row share one
    staged activation tile (gemm.hip, kernel modes 4 / 5).  The MFMA chain per output element is the one of every other
    loop, so: the first images equal, bit for bit, the same images convolved in a launch too small for that kernel, the
    chunk fold equals the split-K slabs, and the whole result matches torch."""
    g = torch.Generator().manual_seed(mode * 77 + B + Cin)
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = G.f32(torch.randn(Cout, generator=g))
    xb = G.bf(x.permute(0, 2, 3, 1))
    wq = torch.empty(Cout * 9 * Cin, dtype=_lib.storage_dtype(), device=G.dev())
    wd = G.f32(w)
    _lib.check(lib.hedit_k_pack_conv3x3(_lib.ptr(wd), _lib.ptr(wq), Cout, Cin, None))
    Ho = H if mode == 1 else 2 * H
    res = G.bf(torch.randn(B * Ho * Ho, Cout, generator=g))

    def conv(n, sp):
        return run_gemm(lib, xb[:n].contiguous(), wq, bias, res[: n * Ho * Ho].contiguous(), n * Ho * Ho, Cout, 9 * Cin, Cin, Cout, Cout,
                        mode=mode, conv=(H, H, Cin, Ho, Ho), splits=sp)

    big = conv(B, splits)
    bn = 160 if (Cout + 159) // 160 * 160 <= (Cout + 127) // 128 * 128 else 128
    assert (B * Ho * Ho + 255) // 256 * ((Cout + bn - 1) // bn) >= 200        # the 256-row kernel is the one that ran
    small = conv(1, splits)
    assert torch.equal(big[: Ho * Ho], small)
    if splits < 0:
        assert torch.equal(big, conv(B, -splits))
    xr = xb.float().permute(0, 3, 1, 2)
    if mode == 3:
        xr = F.interpolate(xr, scale_factor=2.0, mode="nearest")
    want = F.conv2d(xr, G.bf(w).float(), bias, padding=1).permute(0, 2, 3, 1).reshape(B * Ho * Ho, Cout)
    want = want.to(_lib.storage_dtype()).float() + res.float()
    assert G.rel_err(big.float(), want) < 6e-3


@pytest.mark.parametrize("B,HW,Cc,silu", [(2, 256, 64, 1), (4, 4096, 320, 1), (2, 64, 1280, 0), (1, 1024, 960, 1),
                                           (2, 256, 2560, 1), (3, 100, 128, 0)])
def test_groupnorm(lib, B, HW, Cc, silu):
    g = torch.Generator().manual_seed(Cc + HW)
    x = G.bf(torch.randn(B, HW, Cc, generator=g) * 2 + 0.5)
    gamma, beta = G.f32(1 + 0.1 * torch.randn(Cc, generator=g)), G.f32(0.1 * torch.randn(Cc, generator=g))
    y = torch.empty_like(x)
    ws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(B, HW, Cc), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_groupnorm(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), B, HW, Cc, 32,
                                     1e-5, silu, _lib.ptr(ws), None))
    G.sync()
    want = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps=1e-5)
    if silu:
        want = F.silu(want)
    assert G.rel_err(y.float(), want.permute(0, 2, 1)) < 5e-3


@pytest.mark.parametrize("rows,Cc", [(1000, 320), (64, 1280), (333, 64), (128, 640)])
def test_layernorm_geglu(lib, rows, Cc):
    g = torch.Generator().manual_seed(rows + Cc)
    x = G.bf(torch.randn(rows, Cc, generator=g) * 1.5 + 0.2)
    gamma, beta = G.f32(1 + 0.1 * torch.randn(Cc, generator=g)), G.f32(0.1 * torch.randn(Cc, generator=g))
    y = torch.empty_like(x)
    _lib.check(lib.hedit_k_layernorm(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), rows, Cc, 1e-5, None))
    G.sync()
    assert G.rel_err(y.float(), F.layer_norm(x.float(), (Cc,), gamma, beta, 1e-5)) < 5e-3
    inner = Cc // 2
    z = torch.empty(rows, inner, dtype=_lib.storage_dtype(), device=G.dev())
    _lib.check(lib.hedit_k_geglu(_lib.ptr(x), _lib.ptr(z), rows, inner, None))
    G.sync()
    h, gt = x.float().chunk(2, dim=-1)
    assert G.rel_err(z.float(), h * F.gelu(gt)) < 5e-3


@pytest.mark.parametrize("M,inner,K", [(256, 1280, 320), (1000, 160, 64), (128, 5120, 1280), (4096, 2560, 640)])
def test_gemm_geglu_fused(lib, M, inner, K):
    """FF1 + GEGLU in one kernel == gelu-gated product of the two halves of the plain projection
    (diffusers GEGLU: hidden, gate = proj(x).chunk(2, -1); hidden * gelu(gate), exact erf GELU)."""
    g = torch.Generator().manual_seed(M + inner)
    x = G.bf(torch.randn(M, K, generator=g))
    w = torch.randn(2 * inner, K, generator=g) / math.sqrt(K)
    b = torch.randn(2 * inner, generator=g) * 0.5
    wd, bd = G.f32(w), G.f32(b)
    wp = torch.empty(2 * inner, K, dtype=_lib.storage_dtype(), device=G.dev())
    bp = torch.empty(2 * inner, dtype=torch.float32, device=G.dev())
    _lib.check(lib.hedit_k_pack_geglu(_lib.ptr(wd), _lib.ptr(bd), _lib.ptr(wp), _lib.ptr(bp), inner, K, None))
    out = torch.zeros(M, inner, dtype=_lib.storage_dtype(), device=G.dev())
    _lib.check(lib.hedit_k_gemm_geglu(_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bp), _lib.ptr(out), M, inner, K, K, inner, None))
    G.sync()
    proj = x.float() @ wd.to(_lib.storage_dtype()).float().t() + bd
    h, gate = proj.chunk(2, dim=-1)
    want = h * F.gelu(gate)
    assert G.rel_err(out.float(), want) < 6e-3
    # pointwise: within one bf16 ulp of the exactly computed value (plus the fp32 accumulation noise)
    assert ((out.float() - want).abs() <= want.abs() * 2 ** -7 + 2e-3).all()


def _ffn_setup(lib, M, seed):
    Cc = lib.hedit_k_ffn_channels()
    g = torch.Generator().manual_seed(seed)
    x = G.bf(torch.randn(M, Cc, generator=g) * 1.5 + 0.2)
    gamma, beta = G.f32(1 + 0.1 * torch.randn(Cc, generator=g)), G.f32(0.1 * torch.randn(Cc, generator=g))
    w1 = G.f32(torch.randn(8 * Cc, Cc, generator=g) / math.sqrt(Cc))
    b1 = G.f32(torch.randn(8 * Cc, generator=g) * 0.5)
    w2 = G.f32(torch.randn(Cc, 4 * Cc, generator=g) / math.sqrt(4 * Cc))
    b2 = G.f32(torch.randn(Cc, generator=g) * 0.5)
    ws = torch.empty(lib.hedit_k_ffn_stream_bytes(0), dtype=torch.uint8, device=G.dev())
    bp = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), None, None, _lib.ptr(ws), _lib.ptr(bp), None))
    return Cc, x, gamma, beta, w1, b1, w2, b2, ws, bp


def _ffn_run(lib, x, gamma, beta, ws, bp, b2, Cc):
    out = torch.zeros_like(x)
    _lib.check(lib.hedit_k_ffn_fused(_lib.ptr(x), Cc, _lib.ptr(gamma), _lib.ptr(beta), 1e-5, _lib.ptr(ws), _lib.ptr(bp),
                                     _lib.ptr(b2), _lib.ptr(out), Cc, x.shape[0], Cc, None))
    G.sync()
    return out


@pytest.mark.parametrize("M", [128, 100, 4096, 5 * 4096 + 37])
def test_ffn_fused(lib, M):
    """LayerNorm -> FF1 -> GEGLU -> FF2 -> + residual in one kernel (csrc/ffn.hip) == the block's feed-forward
    (diffusers BasicTransformerBlock: ff(norm3(h)) + h, GEGLU with exact erf GELU; oracle/sd_unet.py) in fp32 on the
    bf16-rounded operands, and == the unfused kernel chain (layernorm, FF1+GEGLU GEMM, FF2 GEMM + residual) up to the
    summation order inside a 32-deep FF2 k-step."""
    Cc, x, gamma, beta, w1, b1, w2, b2, ws, bp = _ffn_setup(lib, M, 11 + M)
    out = _ffn_run(lib, x, gamma, beta, ws, bp, b2, Cc)
    xn = F.layer_norm(x.float(), (Cc,), gamma, beta, 1e-5).to(_lib.storage_dtype()).float()
    proj = xn @ w1.to(_lib.storage_dtype()).float().t() + b1
    h, gate = proj.chunk(2, dim=-1)
    hid = (h * F.gelu(gate)).to(_lib.storage_dtype()).float()
    y = (hid @ w2.to(_lib.storage_dtype()).float().t() + b2).to(_lib.storage_dtype()).float()
    want = y + x.float()
    assert torch.isfinite(out.float()).all()
    assert G.rel_err(out.float(), want) < 6e-3
    # the unfused chain of the same library
    xn_k = torch.empty_like(x)
    _lib.check(lib.hedit_k_layernorm(_lib.ptr(x), _lib.ptr(xn_k), _lib.ptr(gamma), _lib.ptr(beta), M, Cc, 1e-5, None))
    wp = torch.empty(8 * Cc, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    b1p = torch.empty(8 * Cc, dtype=torch.float32, device=G.dev())
    _lib.check(lib.hedit_k_pack_geglu(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(wp), _lib.ptr(b1p), 4 * Cc, Cc, None))
    hid_k = torch.empty(M, 4 * Cc, dtype=_lib.storage_dtype(), device=G.dev())
    _lib.check(lib.hedit_k_gemm_geglu(_lib.ptr(xn_k), _lib.ptr(wp), _lib.ptr(b1p), _lib.ptr(hid_k), M, 4 * Cc, Cc, Cc, 4 * Cc, None))
    w2b = w2.to(_lib.storage_dtype()).contiguous()
    out_k = torch.empty_like(x)
    _lib.check(lib.hedit_k_gemm(_lib.ptr(hid_k), _lib.ptr(w2b), _lib.ptr(b2), _lib.ptr(x), _lib.ptr(out_k), M, Cc, 4 * Cc,
                                4 * Cc, Cc, Cc, 0, 0, 0, 0, 0, 0, 1, None, None))
    G.sync()
    assert G.rel_err(out.float(), out_k.float()) < 4e-3


def test_ffn_fused_rows_are_independent_and_repeatable(lib):
    """A row's result is a function of that row alone (bitwise): alone, inside a ragged batch, twice."""
    Cc, x, gamma, beta, w1, b1, w2, b2, ws, bp = _ffn_setup(lib, 3 * 4096 + 5, 7)
    full = _ffn_run(lib, x, gamma, beta, ws, bp, b2, Cc)
    again = _ffn_run(lib, x, gamma, beta, ws, bp, b2, Cc)
    assert torch.equal(full, again)
    part = _ffn_run(lib, x[4096 + 77:4096 + 77 + 300].contiguous(), gamma, beta, ws, bp, b2, Cc)
    assert torch.equal(full[4096 + 77:4096 + 77 + 300], part)
    one = _ffn_run(lib, x[-1:].contiguous(), gamma, beta, ws, bp, b2, Cc)
    assert torch.equal(full[-1:], one)


@pytest.mark.parametrize("impl", ["chain"])
@pytest.mark.parametrize("M", [128, 4096 + 77, 3 * 4096, 11 * 4096 + 5, 70 * 4096 + 8])     # (more tiles than CUs, several per block)
def test_lin_chain_out_then_query(lib, M, impl):
    """attn1.to_out + residual -> norm2 -> attn2.to_q in one kernel (csrc/ffn.hip, hedit_k_lin_chain, one output) == the
    three layers in fp32 on the bf16-rounded operands (diffusers BasicTransformerBlock, oracle/sd_unet.py); both results
    (the residual stream and the query) are functions of their own row alone, bit for bit."""
    Cc = lib.hedit_k_ffn_channels()
    g = torch.Generator().manual_seed(31 + M)
    a = G.bf(torch.randn(M, Cc, generator=g))
    r1 = G.bf(torch.randn(M, Cc, generator=g) * 1.5 + 0.2)
    gamma, beta = G.f32(1 + 0.1 * torch.randn(Cc, generator=g)), G.f32(0.1 * torch.randn(Cc, generator=g))
    wo = G.f32(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc))
    bo = G.f32(torch.randn(Cc, generator=g) * 0.3)
    wq = G.f32(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc))
    scale = 0.2281
    f_bytes, f_pack, f_run = (getattr(lib, f"hedit_k_lin_{impl}{sfx}") for sfx in ("_stream_bytes", "_pack", ""))
    ws = torch.empty(f_bytes(1), dtype=torch.uint8, device=G.dev())
    _lib.check(f_pack(_lib.ptr(wo), _lib.ptr(wq), None, None, scale, _lib.ptr(ws), None))

    def run(a_, r1_):
        m = a_.shape[0]
        mid = torch.zeros(m, Cc, dtype=_lib.storage_dtype(), device=G.dev())
        q = torch.zeros(m, Cc, dtype=_lib.storage_dtype(), device=G.dev())
        _lib.check(f_run(_lib.ptr(a_), Cc, _lib.ptr(r1_), Cc, None, 0, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                                         _lib.ptr(ws), _lib.ptr(mid), Cc, None, 0, None, 0, _lib.ptr(q), Cc, m, Cc, None))
        G.sync()
        return mid, q
    mid, q = run(a, r1)
    bfr = lambda t: t.to(_lib.storage_dtype()).float()
    t1 = a.float() @ bfr(wo).t() + bo + r1.float()
    want_q = bfr(F.layer_norm(t1, (Cc,), gamma, beta, 1e-5)) @ bfr(wq * scale).t()
    assert torch.isfinite(q.float()).all()
    assert G.rel_err(mid.float(), t1) < 3e-3
    assert G.rel_err(q.float(), want_q) < 6e-3
    mid2, q2 = run(a, r1)
    assert torch.equal(mid, mid2) and torch.equal(q, q2)
    lo = min(M - 1, 100)
    mid3, q3 = run(a[lo:].contiguous(), r1[lo:].contiguous())
    assert torch.equal(mid3, mid[lo:]) and torch.equal(q3, q[lo:])


@pytest.mark.parametrize("impl", ["chain"])
@pytest.mark.parametrize("B,N", [(1, 128), (3, 1024), (2, 4096), (11, 4096), (5, 576), (3, 64), (70, 4096)])     # (24 x 24 / 8 x 8 tokens: images that are not whole 128-row tiles)
def test_lin_chain_groupnorm_to_qkv(lib, B, N, impl):
    """GroupNorm (applied on the fly) -> proj_in -> norm1 -> attn1.to_q | to_k | to_v^T in one kernel (hedit_k_lin_chain,
    three outputs; Transformer2DModel.norm / proj_in and the self-attention projections, oracle/sd_unet.py) == the layers
    in fp32 on the bf16-rounded operands; q and k go into one [M][2C] buffer, v comes out transposed ([C][M])."""
    Cc = lib.hedit_k_ffn_channels()
    M = B * N
    g = torch.Generator().manual_seed(41 + M)
    x = G.bf(torch.randn(B, N, Cc, generator=g) * 1.3 + 0.4 * torch.randn(1, 1, Cc, generator=g))
    gn_g, gn_b = G.f32(1 + 0.1 * torch.randn(Cc, generator=g)), G.f32(0.1 * torch.randn(Cc, generator=g))
    gamma, beta = G.f32(1 + 0.1 * torch.randn(Cc, generator=g)), G.f32(0.1 * torch.randn(Cc, generator=g))
    w_in = G.f32(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc))
    b_in = G.f32(torch.randn(Cc, generator=g) * 0.3)
    wq, wk, wv = (G.f32(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc)) for _ in range(3))
    scale = 0.2281
    f_bytes, f_pack, f_run = (getattr(lib, f"hedit_k_lin_{impl}{sfx}") for sfx in ("_stream_bytes", "_pack", ""))
    ws = torch.empty(f_bytes(3), dtype=torch.uint8, device=G.dev())
    _lib.check(f_pack(_lib.ptr(w_in), _lib.ptr(wq), _lib.ptr(wk), _lib.ptr(wv), scale, _lib.ptr(ws), None))
    gws = torch.empty(lib.hedit_k_groupnorm_ws_bytes(B, N, Cc), dtype=torch.uint8, device=G.dev())

    def run(x_):
        b = x_.shape[0]
        m = b * N
        ss = torch.empty(b, Cc, 2, dtype=torch.float32, device=G.dev())
        _lib.check(lib.hedit_k_groupnorm_affine(_lib.ptr(x_), _lib.ptr(gn_g), _lib.ptr(gn_b), b, N, Cc, 32, 1e-6, _lib.ptr(gws), _lib.ptr(ss), None))
        mid = torch.zeros(m, Cc, dtype=_lib.storage_dtype(), device=G.dev())
        qk = torch.zeros(m, 2 * Cc, dtype=_lib.storage_dtype(), device=G.dev())
        vt = torch.zeros(Cc, m, dtype=_lib.storage_dtype(), device=G.dev())
        _lib.check(f_run(_lib.ptr(x_), Cc, None, 0, _lib.ptr(ss), N, _lib.ptr(b_in), _lib.ptr(gamma), _lib.ptr(beta), 1e-5,
                         _lib.ptr(ws), _lib.ptr(mid), Cc, _lib.ptr(qk), 2 * Cc, qk.data_ptr() + 2 * Cc, 2 * Cc,
                         _lib.ptr(vt), m, m, Cc, None))
        G.sync()
        return mid, qk, vt
    mid, qk, vt = run(x)
    bfr = lambda t: t.to(_lib.storage_dtype()).float()
    xf = x.float()
    xg = F.group_norm(xf.transpose(1, 2), 32, gn_g, gn_b, 1e-6).transpose(1, 2).reshape(M, Cc)
    t0 = bfr(xg) @ bfr(w_in).t() + b_in
    tn = bfr(F.layer_norm(t0, (Cc,), gamma, beta, 1e-5))
    assert G.rel_err(mid.float(), t0) < 6e-3
    assert G.rel_err(qk[:, :Cc].float(), tn @ bfr(wq * scale).t()) < 8e-3
    assert G.rel_err(qk[:, Cc:].float(), tn @ bfr(wk).t()) < 8e-3
    assert G.rel_err(vt.float().t(), tn @ bfr(wv).t()) < 8e-3
    again = run(x)
    assert all(torch.equal(u, v) for u, v in zip((mid, qk, vt), again))
    if B > 1:                      # an image alone == the image inside the batch
        m1, qk1, vt1 = run(x[1:2].contiguous())
        assert torch.equal(m1, mid[N:2 * N]) and torch.equal(qk1, qk[N:2 * N]) and torch.equal(vt1, vt[:, N:2 * N])


@pytest.mark.parametrize("M", [128, 4096 + 77, 3 * 4096])
def test_ffn_chain(lib, M):
    """The token-local tail of a transformer block in one kernel (csrc/ffn.hip): attn2.to_out + residual -> norm3 ->
    GEGLU feed-forward + residual -> proj_out + residual == the same five layers in fp32 on the bf16-rounded operands
    (diffusers BasicTransformerBlock / Transformer2DModel order, oracle/sd_unet.py) and == the chain of single kernels
    of this library (GEMM + residual, fused feed-forward, GEMM + residual); rows are independent bit for bit."""
    Cc, _, gamma, beta, w1, b1, w2, b2, _, _ = _ffn_setup(lib, 8, 23)
    g = torch.Generator().manual_seed(5 + M)
    a = G.bf(torch.randn(M, Cc, generator=g))
    t1 = G.bf(torch.randn(M, Cc, generator=g) * 1.5 + 0.2)
    ld_x = Cc + 64                                                  # the block input may sit in a wider buffer
    xw = G.bf(torch.randn(M, ld_x, generator=g))
    x = xw[:, :Cc]
    wo = G.f32(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc))
    bo = G.f32(torch.randn(Cc, generator=g) * 0.3)
    wp = G.f32(torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc))
    bpo = G.f32(torch.randn(Cc, generator=g) * 0.3)
    ws = torch.empty(lib.hedit_k_ffn_stream_bytes(1), dtype=torch.uint8, device=G.dev())
    bp = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(wo), _lib.ptr(wp), _lib.ptr(ws), _lib.ptr(bp), None))
    ld_o = Cc + 320                                                 # ... and the result goes into a concatenation buffer
    outw = torch.zeros(M, ld_o, dtype=_lib.storage_dtype(), device=G.dev())

    def run(a_, t1_, xw_, outw_):
        _lib.check(lib.hedit_k_ffn_chain(_lib.ptr(a_), Cc, _lib.ptr(t1_), Cc, _lib.ptr(xw_), ld_x, _lib.ptr(bo), _lib.ptr(gamma), _lib.ptr(beta),
                                         1e-5, _lib.ptr(ws), _lib.ptr(bp), _lib.ptr(b2), _lib.ptr(bpo), _lib.ptr(outw_), ld_o, a_.shape[0], Cc, None))
        G.sync()
    run(a, t1, xw, outw)
    out = outw[:, :Cc]
    assert float(outw[:, Cc:].abs().max()) == 0.0
    bfr = lambda t: t.to(_lib.storage_dtype()).float()
    t2 = a.float() @ bfr(wo).t() + bo + t1.float()
    xn = bfr(F.layer_norm(t2, (Cc,), gamma, beta, 1e-5))
    proj = xn @ bfr(w1).t() + b1
    h, gate = proj.chunk(2, dim=-1)
    t3 = t2 + bfr(h * F.gelu(gate)) @ bfr(w2).t() + b2
    want = bfr(t3) @ bfr(wp).t() + bpo + x.float()
    assert torch.isfinite(out.float()).all()
    assert G.rel_err(out.float(), want) < 6e-3
    # the chain of single kernels (t2 and t3 rounded to bf16 in between, as they were in HBM)
    t2_k = torch.empty(M, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    _lib.check(lib.hedit_k_gemm(_lib.ptr(a), _lib.ptr(wo.to(_lib.storage_dtype()).contiguous()), _lib.ptr(bo), _lib.ptr(t1), _lib.ptr(t2_k), M, Cc, Cc,
                                Cc, Cc, Cc, 0, 0, 0, 0, 0, 0, 1, None, None))
    ws0 = torch.empty(lib.hedit_k_ffn_stream_bytes(0), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_ffn_pack(_lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), None, None, _lib.ptr(ws0), _lib.ptr(bp), None))
    t3_k = _ffn_run(lib, t2_k, gamma, beta, ws0, bp, b2, Cc)
    out_k = torch.empty(M, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    xc = x.contiguous()
    _lib.check(lib.hedit_k_gemm(_lib.ptr(t3_k), _lib.ptr(wp.to(_lib.storage_dtype()).contiguous()), _lib.ptr(bpo), _lib.ptr(xc), _lib.ptr(out_k), M, Cc, Cc,
                                Cc, Cc, Cc, 0, 0, 0, 0, 0, 0, 1, None, None))
    G.sync()
    assert G.rel_err(out.float(), out_k.float()) < 6e-3
    # a row is a function of itself alone, run to run and batch to batch
    again = torch.zeros_like(outw)
    run(a, t1, xw, again)
    assert torch.equal(again, outw)
    lo = min(M - 1, 100)
    part = torch.zeros(M - lo, ld_o, dtype=_lib.storage_dtype(), device=G.dev())
    run(a[lo:].contiguous(), t1[lo:].contiguous(), xw[lo:].contiguous(), part)
    assert torch.equal(part, outw[lo:])


def attn_ref(q, k, v, heads):
    """q (B,N,C) pre-scaled in log2 units, k (B,M,C), v (B,M,C) -> probs (B,h,N,M), out (B,N,C)"""
    B, N, Cc = q.shape
    d = Cc // heads
    qh = q.reshape(B, N, heads, d).transpose(1, 2)
    kh = k.reshape(B, -1, heads, d).transpose(1, 2)
    vh = v.reshape(B, -1, heads, d).transpose(1, 2)
    p = torch.softmax((qh @ kh.transpose(-1, -2)) * math.log(2.0), dim=-1)
    return p, (p @ vh).transpose(1, 2).reshape(B, N, Cc)


@pytest.mark.parametrize("d,heads,N,B", [(32, 2, 256, 2), (40, 8, 1024, 2), (64, 2, 64, 3), (80, 8, 256, 2),
                                           (160, 8, 64, 2), (40, 8, 4096, 1), (160, 8, 256, 4), (40, 2, 192, 2), (80, 2, 192, 2),
                                           (160, 2, 192, 1), (64, 2, 320, 1), (80, 4, 1024, 1)])
def test_self_attention(lib, d, heads, N, B):
    g = torch.Generator().manual_seed(d * 7 + N)
    Cc = d * heads
    scale = d ** -0.5 * math.log2(math.e)
    q = G.bf(torch.randn(B, N, Cc, generator=g) * scale * 1.5)
    k = G.bf(torch.randn(B, N, Cc, generator=g) * 1.5)
    v = G.bf(torch.randn(B, N, Cc, generator=g))
    qk = torch.cat([q, k], dim=-1).contiguous()                       # [B*N][2C]
    vt = v.reshape(B * N, Cc).t().contiguous()                        # [C][B*N]
    out = torch.zeros(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    k_view = qk.reshape(B * N, 2 * Cc)[:, Cc:]
    _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * Cc, C.c_void_p(k_view.data_ptr()), 2 * Cc, _lib.ptr(vt),
                                     B * N, _lib.ptr(out), Cc, B, N, heads, d, None, None, None))
    G.sync()
    _, want = attn_ref(q.float(), k.float(), v.float(), heads)
    assert G.rel_err(out.float(), want) < 1.2e-2
    # P2P self-replacement: row b uses q,k of row qk_src[b], v of its own row
    src = list(range(B))
    src[B - 1] = 0
    idx = torch.tensor(src, dtype=torch.int32, device=G.dev())
    out2 = torch.zeros_like(out)
    _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * Cc, C.c_void_p(k_view.data_ptr()), 2 * Cc, _lib.ptr(vt),
                                     B * N, _lib.ptr(out2), Cc, B, N, heads, d, _lib.ptr(idx), None, None))
    G.sync()
    _, want2 = attn_ref(q.float()[src], k.float()[src], v.float(), heads)
    assert G.rel_err(out2.float(), want2) < 1.2e-2


@pytest.mark.parametrize("d,heads,N", [(40, 8, 4096), (80, 8, 1024), (32, 2, 1024), (160, 4, 256)])
def test_self_attention_is_deterministic(lib, d, heads, N):
    """Identical batch rows give bit-identical outputs, launch after launch.  (Guards the software-
    managed MFMA hazards: a consumer of MFMA results the compiler cannot see -- inline asm -- reads
    stale registers now and then, which shows up here long before it moves a tolerance test.)"""
    B, Cc = 4, d * heads
    g = torch.Generator().manual_seed(d + N)
    q1 = torch.randn(1, N, Cc, generator=g) * d ** -0.5 * 3.0
    k1 = torch.randn(1, N, Cc, generator=g) * 2.0
    v1 = torch.randn(1, N, Cc, generator=g)
    qk = G.bf(torch.cat([q1, k1], -1).repeat(B, 1, 1).reshape(B * N, 2 * Cc)).contiguous()
    vt = G.bf(v1.repeat(B, 1, 1).reshape(B * N, Cc).t().contiguous())
    k_view = qk[:, Cc:]
    outs = []
    for _ in range(3):
        out = torch.zeros(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
        _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * Cc, C.c_void_p(k_view.data_ptr()), 2 * Cc, _lib.ptr(vt),
                                         B * N, _lib.ptr(out), Cc, B, N, heads, d, None, None, None))
        G.sync()
        outs.append(out)
    for b in range(1, B):
        assert torch.equal(outs[0][b], outs[0][0])
    assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0])


def test_self_attention_online_softmax_rescale(lib):
    """Force the running-max rescale branch: one key late in the sequence dominates a query."""
    d, heads, N, B = 40, 8, 512, 1
    g = torch.Generator().manual_seed(5)
    Cc = d * heads
    q = torch.randn(B, N, Cc, generator=g) * 0.3
    k = torch.randn(B, N, Cc, generator=g)
    v = torch.randn(B, N, Cc, generator=g)
    k[0, 400] = q[0, 7] * 40.0          # spike: raw q.k far above everything seen before tile 6
    q, k, v = G.bf(q), G.bf(k), G.bf(v)
    qk = torch.cat([q, k], dim=-1).contiguous()
    vt = v.reshape(B * N, Cc).t().contiguous()
    out = torch.zeros(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    k_view = qk.reshape(B * N, 2 * Cc)[:, Cc:]
    _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * Cc, C.c_void_p(k_view.data_ptr()), 2 * Cc, _lib.ptr(vt),
                                     B * N, _lib.ptr(out), Cc, B, N, heads, d, None, None, None))
    G.sync()
    _, want = attn_ref(q.float(), k.float(), v.float(), heads)
    assert G.max_err(out.float(), want) < 3e-2
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("d,heads,N", [(40, 8, 1024), (80, 8, 512), (160, 8, 256), (32, 2, 512), (64, 2, 512)])
@pytest.mark.parametrize("lift", [45.0, 90.0, 200.0])
def test_self_attention_pinned_shift_and_exact_fallback(lib, d, heads, N, lift):
    """The fast pass keeps the shift of a stream's first units and checks the denominators afterwards (attn.hip, "Pinned
    shift").  Keys late in the sequence that beat everything of the first tile by 2^45 stay on the fast pass (probabilities
    up to 2^45 are as exact in bf16 / fp32 as any other), by 2^90 trip the 2^60 denominator check, by 2^200 overflow to
    inf -- the latter two must come out of the exact pass; all three must match the fp32 softmax."""
    B = 2
    g = torch.Generator().manual_seed(int(lift) + d)
    Cc = d * heads
    q = torch.randn(B, N, Cc, generator=g) * 0.3
    k = torch.randn(B, N, Cc, generator=g)
    v = torch.randn(B, N, Cc, generator=g)
    rows = [(0, 7, 3 * N // 4 + 16), (0, 200, 70), (1, N - 1, N - 1), (1, 33, 129)]           # (batch row, query, key)
    for (b, qi, ki) in rows:
        for h in range(heads):
            qh = q[b, qi, h * d:(h + 1) * d]
            k[b, ki, h * d:(h + 1) * d] = qh * (lift / float(qh @ qh))
    q, k, v = G.bf(q), G.bf(k), G.bf(v)
    qk = torch.cat([q, k], dim=-1).contiguous()
    vt = v.reshape(B * N, Cc).t().contiguous()
    k_view = qk.reshape(B * N, 2 * Cc)[:, Cc:]
    outs = []
    for _ in range(2):
        out = torch.zeros(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
        _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk), 2 * Cc, C.c_void_p(k_view.data_ptr()), 2 * Cc, _lib.ptr(vt),
                                         B * N, _lib.ptr(out), Cc, B, N, heads, d, None, None, None))
        G.sync()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert torch.isfinite(outs[0].float()).all()
    _, want = attn_ref(q.float(), k.float(), v.float(), heads)
    assert G.rel_err(outs[0].float(), want) < 1.2e-2
    for (b, qi, ki) in rows:                                                       # the rows the spikes own
        assert G.max_err(outs[0][b, qi].float(), want[b, qi]) < 3e-2
    # a block's path depends on its own rows only: the second batch row alone gives the same bits
    out1 = torch.zeros(1, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    qk1, vt1 = qk[1:].contiguous(), v[1].reshape(N, Cc).t().contiguous()
    _lib.check(lib.hedit_k_self_attn(_lib.ptr(qk1), 2 * Cc, C.c_void_p(qk1.reshape(N, 2 * Cc)[:, Cc:].data_ptr()), 2 * Cc,
                                     _lib.ptr(vt1), N, _lib.ptr(out1), Cc, 1, N, heads, d, None, None, None))
    G.sync()
    assert torch.equal(out1[0], outs[0][1])


@pytest.mark.parametrize("d,heads,N", [(32, 2, 256), (40, 8, 1024), (64, 2, 64), (80, 8, 256), (160, 8, 64)])
def test_cross_attention_p2p(lib, d, heads, N):
    g = torch.Generator().manual_seed(d * 3 + N)
    B, Cc, CT = 4, d * heads, 80
    scale = d ** -0.5 * math.log2(math.e)
    q = G.bf(torch.randn(B, N, Cc, generator=g) * scale * 2)
    kc = torch.zeros(B, CT, Cc)
    vc = torch.zeros(B, CT, Cc)
    kc[:, :77] = torch.randn(B, 77, Cc, generator=g) * 2
    vc[:, :77] = torch.randn(B, 77, Cc, generator=g)
    kc, vc = G.bf(kc), G.bf(vc)
    vt = vc.reshape(B * CT, Cc).t().contiguous()
    # a non-trivial mixing matrix: permutation-ish gather + scaling, and a blend vector
    A = torch.zeros(77, 77)
    perm = torch.randperm(77, generator=g)
    A[perm, torch.arange(77)] = torch.rand(77, generator=g) * 1.5
    A[3, 5] = 0.5
    A[4, 5] = 0.5
    bvec = torch.rand(77, generator=g)
    mixT = torch.zeros(1, 96, 96)
    mixT[0, :77, :77] = A.t()
    bv = torch.zeros(1, 96)
    bv[0, :77] = bvec
    mixT_d, bv_d = G.bf(mixT), G.f32(bv)
    plan, keep = G.make_plan(n_pairs=1, pair_src=[2], pair_tar=[3], singles=[0, 1], mixT=mixT_d, bvec=bv_d, mode=2)
    store = torch.zeros(1, 2, heads, N, 77, dtype=torch.float32, device=G.dev())
    out = torch.zeros(B, N, Cc, dtype=_lib.storage_dtype(), device=G.dev())
    for rep in range(2):        # two passes: the store must accumulate
        _lib.check(lib.hedit_k_cross_attn(_lib.ptr(q), Cc, _lib.ptr(kc), Cc, _lib.ptr(vt), B * CT, _lib.ptr(out),
                                          Cc, B, N, heads, d, C.byref(plan), _lib.ptr(store), None))
    G.sync()
    p, o = attn_ref(q.float(), kc.float()[:, :77], vc.float()[:, :77], heads)
    A_b = mixT_d.float()[0, :77, :77].t()
    p_new = p[2] @ A_b + bv_d[0, :77] * p[3]
    vh = vc.float()[3, :77].reshape(77, heads, d).transpose(0, 1)
    o_tar = (p_new @ vh).transpose(0, 1).reshape(N, Cc)
    assert G.rel_err(out[:3].float(), o[:3]) < 1.2e-2          # plain rows and the source row
    assert G.rel_err(out[3].float(), o_tar) < 1.5e-2            # edited target row
    assert G.rel_err(store[0, 0], 2 * p[2]) < 2e-3               # source maps, two passes
    assert G.rel_err(store[0, 1], 2 * p_new) < 1e-2              # post-edit target maps
    # controller off: every row plain
    plan0, keep0 = G.make_plan(n_pairs=0, singles=[0, 1, 2, 3], mode=0)
    out0 = torch.zeros_like(out)
    _lib.check(lib.hedit_k_cross_attn(_lib.ptr(q), Cc, _lib.ptr(kc), Cc, _lib.ptr(vt), B * CT, _lib.ptr(out0),
                                      Cc, B, N, heads, d, C.byref(plan0), None, None))
    G.sync()
    assert G.rel_err(out0.float(), o) < 1.2e-2


def test_step_kernels_match_reference_vectors(lib, golden_dir):
    """hedit_step_base against the reference's reverse_step outputs (g2), then hedit_step_update
    against the oracle formulas."""
    import os
    from helpers.tiny import ddim_tables
    from hedit.engine import Schedule
    g = np.load(os.path.join(golden_dir, "g2_reverse_step.npz"))
    sch = ddim_tables(20)
    S = Schedule(sch)
    eps, x, z = (torch.from_numpy(g[k]) for k in ("eps", "x", "z"))
    elems = x[0].numel()
    for t in (951, 501, 1):
        for eta in (0.0, 1.0):
            for ddim in (False, True):
                coef = S.step_coef(t, max(t - 50, 0), eta, ddim, (1.0, 5.0, 7.5))
                # rows [x_o|0, x_e|0, x_o|src, x_e|src] with w_src = 1 -> eps = conditional rows
                e4 = G.f32(torch.cat([torch.zeros_like(eps), eps]))
                xd, zd = G.f32(x), G.f32(z)            # keep the device copies alive across the launch
                out = torch.zeros(2, *x.shape[1:], device=G.dev())
                _lib.check(lib.hedit_step_base(_lib.ptr(e4), _lib.ptr(xd), _lib.ptr(zd), _lib.ptr(out), 1,
                                               elems, 4, C.byref(coef), None))
                G.sync()
                want = torch.from_numpy(g[f"prev_t{t}_eta{int(eta)}_ddim{int(ddim)}"])
                assert G.max_err(out, want) < 2e-5 * max(1.0, want.abs().max().item())
        # hedit_step_tweedie against the reference's reverse_step_pred_x0 (inversion_utils.py:128-140; g2 tweedie_t*),
        # eps handed over as its CFG parts (e_u + w (e_c - e_u) with e_u = 0.25 eps, e_c = 0.5 eps, w = 3) and whole
        ab = float(sch.alphas_cumprod[t])
        want0 = torch.from_numpy(g[f"tweedie_t{t}"])
        for eu, ec, w in ((0.25 * eps, 0.5 * eps, 3.0), (eps, eps, 7.5)):
            eud, ecd, xd = G.f32(eu), G.f32(ec), G.f32(x)
            z0 = torch.zeros_like(xd)
            _lib.check(lib.hedit_step_tweedie(_lib.ptr(eud), _lib.ptr(ecd), elems, _lib.ptr(xd), _lib.ptr(z0), x.shape[0], elems,
                                              w, math.sqrt(ab), math.sqrt(1.0 - ab), 1.0, None))
            G.sync()
            assert G.max_err(z0, want0) < 2e-5 * max(1.0, want0.abs().max().item())
    # update kernel, k = 0 and k > 0, two images
    gen = torch.Generator().manual_seed(3)
    n, el = 2, 4 * 16 * 16
    e = torch.randn(4, n, el, generator=gen)
    xk = torch.randn(n, el, generator=gen)
    xb = xk + 0.1 * torch.randn(n, el, generator=gen)
    xb[0, :10] = xk[0, :10]                       # exact ties -> sign 0
    coef = S.step_coef(501, 451, 1.0, False, (1.0, 5.0, 7.5), w_rec=0.1)
    for k_gt0 in (0, 1):
        out = torch.zeros(n, el, device=G.dev())
        ed, xkd, xbd = G.f32(e), G.f32(xk), G.f32(xb)
        _lib.check(lib.hedit_step_update(_lib.ptr(ed[0]), _lib.ptr(ed[2]), _lib.ptr(ed[1]), _lib.ptr(ed[3]), el,
                                         _lib.ptr(xkd), _lib.ptr(xbd), _lib.ptr(out), n, el, k_gt0,
                                         C.byref(coef), None))
        G.sync()
        for i in range(n):
            ehat = e[0, i] + coef.w_hat * (e[2, i] - e[0, i])
            etar = e[1, i] + coef.w_tar * (e[3, i] - e[1, i])
            corr = etar - ehat
            rec = xk[i]
            if k_gt0:
                gr = torch.sign(xk[i] - xb[i]) / el
                rho = corr.pow(2).mean().sqrt().item() / (gr.pow(2).mean().sqrt().item() + 1e-8) * coef.w_rec
                rec = xk[i] - rho * gr
            want = rec + coef.coeff * corr
            assert G.max_err(out[i], want) < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("ci", [0, 1, 2])
def test_local_blend_substruct_words_match_reference_vectors(lib, golden_dir, ci):
    """hedit's LocalBlend(substruct_words=...) -> hedit_local_blend_sub against the reference's class (g16)."""
    import os
    from helpers.tiny import LOCAL_BLEND_SUB_CASES, PROMPT_PAIRS, WordTokenizer, hash_normal, hash_uniform
    from hedit.p2p.ptp_classes import LocalBlend
    g = np.load(os.path.join(golden_dir, "g16_local_blend_sub.npz"))
    pi, words, sub, th = LOCAL_BLEND_SUB_CASES[ci]
    src, tar = PROMPT_PAIRS[pi][:2]
    tok = WordTokenizer(split_long_words_at=6)
    lb = LocalBlend([src, tar], 10, words, substruct_words=sub, th=th, tokenizer=tok, device=G.dev())
    heads = 2
    five = [G.f32(hash_uniform((2 * heads, 256, 77), 3100 + ci * 10 + i) ** 6) for i in range(5)]
    big = torch.zeros(2 * heads, 1024, 77, device=G.dev())
    store = {"down_cross": [big, big, five[0], five[1]], "up_cross": [five[2], five[3], five[4], big]}
    x = hash_normal((2, 4, 64, 64), 3600 + ci)
    lb.counter = 5
    y = lb(G.f32(x), store)
    G.sync()
    assert torch.equal(y[0].cpu(), x[0])
    assert G.max_err(y[1], torch.from_numpy(g[f"c{ci}_y"])) < 1e-6


@pytest.mark.parametrize("pi", [0, 1, 3])
def test_local_blend_matches_reference_vectors(lib, golden_dir, pi):
    import os
    from helpers.tiny import PROMPT_PAIRS, WordTokenizer, hash_normal, hash_uniform
    from hedit.p2p import ptp_controller_utils as PCU
    g = np.load(os.path.join(golden_dir, "g5_local_blend.npz"))
    src, tar, blend, is_replace = PROMPT_PAIRS[pi]
    tok = WordTokenizer(split_long_words_at=6)
    c = PCU.make_controller([src, tar], is_replace, 0.4, 0.35, blend_word=((blend[0],), (blend[1],)),
                            equilizer_params={"words": (blend[1],), "values": (2.0,)}, num_steps=10, tokenizer=tok)
    heads = 2
    five = [G.f32(hash_uniform((2 * heads, 256, 77), 3000 + pi * 10 + i) ** 6) for i in range(5)]
    big = torch.zeros(2 * heads, 1024, 77, device=G.dev())
    store = {"down_cross": [big, big, five[0], five[1]], "up_cross": [five[2], five[3], five[4], big]}
    x = hash_normal((2, 4, 64, 64), 3500 + pi)
    for counter in (0, 2, 3):
        c.local_blend.counter = counter
        y = c.local_blend(G.f32(x), store)
        G.sync()
        assert torch.equal(y[0].cpu(), x[0])
        assert G.max_err(y[1], torch.from_numpy(g[f"p{pi}_y_counter{counter}"])) < 1e-6
