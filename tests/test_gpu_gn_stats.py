"""-m gpu: GroupNorm statistics taken by the producing convolution (h-edit_amd/csrc/gnstat.h) -- what replaces the statistics
pass of torch's GroupNorm in the pixel UNet's ResNet blocks (reference face-swapping/diffusion/diffusion.py:27-33, 115-134).

  * every producer form (128-row tile, its three-stage ring, 256-row tile, row-sharing loop, upsampling / stride-2 gathers,
    chunk fold, split-K slabs + reduce) writes, bit for bit, the pair statistics tests/helpers/gnstat_ref.py computes from
    the stored output -- the tree pinned on the CPU by tests/test_host_gn_stats.py -- and the output bits of the launch
    without statistics;
  * GroupNorm from those pairs (one producer, or the two halves of a skip concatenation with groups across the seam)
    against the kernel that reads the tensor for its statistics and against torch;
  * the CelebA-HQ pixel UNet with the path switched on: oracle parity, batch invariance bit for bit, and the distance to the
    statistics-pass path.
"""
import math
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gnstat_ref  # noqa: E402
from helpers import gpu as G  # noqa: E402
from helpers.tiny import hash_normal  # noqa: E402
from hedit import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
HALF = _lib.STORAGE == "f16"


@pytest.fixture(scope="module")
def lib():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _lib.lib()


def bits_of(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def conv_pair(lib, xb, wq, bias, res, B, H, Cin, Cout, mode, splits):
    """the convolution with and without the statistics epilogue: (out, out_plain, gn_part)"""
    Ho = H // 2 if mode == 2 else (2 * H if mode == 3 else H)
    M, K = B * Ho * Ho, 9 * Cin
    wsb = lib.hedit_k_gemm_ws_bytes(M, Cout, K, abs(splits))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=G.dev())
    out = torch.zeros(M, Cout, dtype=_lib.storage_dtype(), device=G.dev())
    plain = torch.zeros_like(out)
    part = torch.full((M // 128, Cout // 2, 2), float("nan"), dtype=torch.float32, device=G.dev())
    _lib.check(lib.hedit_k_conv_gn(_lib.ptr(xb), _lib.ptr(wq), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(out), M, Cout, K, Cout, Cout,
                                   mode, H, H, Cin, Ho, Ho, splits, _lib.ptr(ws), _lib.ptr(part), None))
    _lib.check(lib.hedit_k_gemm(_lib.ptr(xb), _lib.ptr(wq), _lib.ptr(bias), _lib.ptr(res), _lib.ptr(plain), M, Cout, K, Cin, Cout,
                                Cout, mode, H, H, Cin, Ho, Ho, splits, _lib.ptr(ws), None))
    G.sync()
    return out, plain, part


CASES = [
    # mode, B, H, Cin, Cout, splits, residual          which kernel writes the tile
    (1, 2, 32, 64, 128, 0, False),      # 128-row tile, three-stage ring (at most one block per CU)
    (1, 6, 64, 128, 256, 0, True),      # 128-row tile, two-stage loop
    (1, 8, 96, 128, 128, 0, True),      # 256-row tile, plain loop (width 96 does not divide 256)
    (1, 1, 256, 128, 128, 0, False),    # 256-row tile, row-sharing loop
    (1, 16, 64, 128, 256, -3, True),    # row-sharing loop with the chunk fold
    (1, 16, 64, 128, 256, 3, True),     # the same chunking as split-K slabs + splitk_reduce_gn_kernel
    (1, 2, 32, 256, 256, 4, False),     # split-K at a small launch
    (2, 4, 128, 128, 128, 0, False),    # stride 2
    (3, 4, 32, 128, 128, 0, True),      # on the 2x upsampled image, 128-row tile
    (3, 16, 64, 128, 128, 0, False),    # the same, row-sharing loop
    (3, 16, 64, 128, 128, -2, True),    # ... with the chunk fold
]


@pytest.mark.parametrize("mode,B,H,Cin,Cout,splits,with_res", CASES)
def test_conv_epilogue_statistics_are_the_canonical_tree(lib, mode, B, H, Cin, Cout, splits, with_res):
    g = torch.Generator().manual_seed(mode * 131 + B * 7 + H + Cin + Cout + (splits & 15))
    x = torch.randn(B, H, H, Cin, generator=g) * 1.5 + 0.2
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    bias = G.f32(torch.randn(Cout, generator=g))
    xb = G.bf(x)
    wq = torch.empty(Cout * 9 * Cin, dtype=_lib.storage_dtype(), device=G.dev())
    wd = G.f32(w)
    _lib.check(lib.hedit_k_pack_conv3x3(_lib.ptr(wd), _lib.ptr(wq), Cout, Cin, None))
    Ho = H // 2 if mode == 2 else (2 * H if mode == 3 else H)
    res = G.bf(torch.randn(B * Ho * Ho, Cout, generator=g)) if with_res else None
    out, plain, part = conv_pair(lib, xb, wq, bias, res, B, H, Cin, Cout, mode, splits)
    assert torch.equal(out, plain), "the statistics epilogue changed the output"
    want = gnstat_ref.pair_stats(bits_of(out), HALF)
    got = part.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # an image alone: the same output bits and the same statistics (plain chains and canonical chunkings alike)
    n1 = Ho * Ho
    o1, _, p1 = conv_pair(lib, xb[:1].contiguous(), wq, bias, res[:n1].contiguous() if with_res else None, 1, H, Cin, Cout, mode,
                          splits)
    assert torch.equal(o1, out[:n1])
    assert torch.equal(p1, part[: n1 // 128])


def test_chunk_fold_and_split_k_write_the_same_statistics(lib):
    g = torch.Generator().manual_seed(9)
    B, H, Cin, Cout = 16, 64, 128, 256
    xb = G.bf(torch.randn(B, H, H, Cin, generator=g))
    wq = torch.empty(Cout * 9 * Cin, dtype=_lib.storage_dtype(), device=G.dev())
    wd = G.f32(torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin))
    _lib.check(lib.hedit_k_pack_conv3x3(_lib.ptr(wd), _lib.ptr(wq), Cout, Cin, None))
    bias = G.f32(torch.randn(Cout, generator=g))
    res = G.bf(torch.randn(B * H * H, Cout, generator=g))
    o_fold, _, p_fold = conv_pair(lib, xb, wq, bias, res, B, H, Cin, Cout, 1, -3)
    o_slab, _, p_slab = conv_pair(lib, xb, wq, bias, res, B, H, Cin, Cout, 1, 3)
    assert torch.equal(o_fold, o_slab) and torch.equal(p_fold, p_slab)


@pytest.mark.parametrize("B,HW,ca,cb,silu", [(2, 4096, 128, 0, 1), (3, 1024, 256, 128, 1), (2, 16384, 128, 128, 1), (1, 1024, 512, 256, 0)])
def test_groupnorm_from_parts(lib, B, HW, ca, cb, silu):
    g = torch.Generator().manual_seed(HW + ca + cb)
    C = ca + cb
    x = G.bf(torch.randn(B, HW, C, generator=g) * 2 + 0.5)
    gamma, beta = G.f32(1 + 0.1 * torch.randn(C, generator=g)), G.f32(0.1 * torch.randn(C, generator=g))
    xa = x[:, :, :ca].contiguous()
    pa = torch.from_numpy(gnstat_ref.pair_stats(bits_of(xa).reshape(B * HW, ca), HALF)).to(G.dev())
    pb = None
    if cb:
        xbh = x[:, :, ca:].contiguous()
        pb = torch.from_numpy(gnstat_ref.pair_stats(bits_of(xbh).reshape(B * HW, cb), HALF)).to(G.dev())
    y = torch.empty_like(x)
    ws = torch.empty(B * 512, dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_groupnorm_from_parts(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), B, HW, C, 32, 1e-6, silu,
                                                _lib.ptr(pa), ca, _lib.ptr(pb), cb, _lib.ptr(ws), None))
    y2 = torch.empty_like(x)
    ws2 = torch.empty(lib.hedit_k_groupnorm_ws_bytes(B, HW, C), dtype=torch.uint8, device=G.dev())
    _lib.check(lib.hedit_k_groupnorm(_lib.ptr(x), _lib.ptr(y2), _lib.ptr(gamma), _lib.ptr(beta), B, HW, C, 32, 1e-6, silu,
                                     _lib.ptr(ws2), None))
    G.sync()
    want = F.group_norm(x.float().permute(0, 2, 1), 32, gamma, beta, eps=1e-6)
    if silu:
        want = F.silu(want)
    assert G.rel_err(y.float(), want.permute(0, 2, 1)) < 5e-3
    # same formulas on statistics that differ in the last fp32 bits: a few outputs move by one bf16 step
    assert G.rel_err(y.float(), y2.float()) < 1e-3
    # an image alone
    y1 = torch.empty_like(x[:1])
    _lib.check(lib.hedit_k_groupnorm_from_parts(_lib.ptr(x[:1].contiguous()), _lib.ptr(y1), _lib.ptr(gamma), _lib.ptr(beta), 1, HW, C, 32,
                                                1e-6, silu, _lib.ptr(pa[: HW // 128].contiguous()), ca,
                                                _lib.ptr(pb[: HW // 128].contiguous()) if cb else None, cb, _lib.ptr(ws), None))
    G.sync()
    assert torch.equal(y1, y[:1])


def test_conv_gn_rejects_shapes_without_the_128_column_tile(lib):
    x = torch.zeros(1, 32, 32, 64, dtype=_lib.storage_dtype(), device=G.dev())
    w = torch.zeros(320 * 9 * 64, dtype=_lib.storage_dtype(), device=G.dev())
    out = torch.zeros(1024, 320, dtype=_lib.storage_dtype(), device=G.dev())
    part = torch.zeros(8, 160, 2, device=G.dev())
    rc = lib.hedit_k_conv_gn(_lib.ptr(x), _lib.ptr(w), None, None, _lib.ptr(out), 1024, 320, 576, 320, 320, 1, 32, 32, 64, 32, 32, 0,
                             None, _lib.ptr(part), None)
    assert rc == -1 and b"128-column tile" in lib.hedit_last_error()


def test_celeba_unet_on_producer_statistics():
    """CelebA-HQ 256 configuration on both GroupNorm paths -- the build's default (csrc/ddpm.hip DDPM_GN_FUSE_DEFAULT = true: every
    eligible GroupNorm on the statistics its producing convolution wrote) and, with hedit_test_set_flags bit 2, the separate statistics
    pass: on EACH path a row's eps is a function of the row alone bit for bit and repeatable; oracle parity as tests/test_gpu_face.py
    asserts it; the two paths within a fraction of the bf16 error of each other."""
    from hedit.diffusion import Model
    from oracle import ddpm_unet
    lib = _lib.lib()
    hip = Model(device=G.dev())
    sd = hip.init_random(1)
    x = hash_normal((3, 3, 256, 256), 5) * 0.8
    xg = x.to(G.dev())
    res = {}
    for name, flags in (("producer statistics (default)", 0), ("statistics pass (flag 4)", 4)):
        try:
            _lib.check(lib.hedit_test_set_flags(flags))
            a = hip(xg, 501.0)
            b = hip(xg, 501.0)
            a1 = hip(xg[:1], 501.0)
            a2 = hip(xg[1:], 501.0)
            G.sync()
        finally:
            lib.hedit_test_set_flags(0)
        assert torch.isfinite(a).all() and torch.equal(a, b), name
        assert torch.equal(a1, a[:1]) and torch.equal(a2, a[1:]), name
        res[name] = a
    (n0, r0), (n1, r1) = res.items()
    assert not torch.equal(r0, r1)                        # the flag really selects another path
    # statistics that differ in their last fp32 bits flip bf16 roundings, and seventy layers amplify the flips to the
    # level of the bf16 noise itself: measured 8.3e-3 between the paths, each 1.2e-2 from the fp32 oracle
    assert G.rel_err(r0, r1) < 1.5e-2
    om = ddpm_unet.Model(**ddpm_unet.CELEBA_HQ).eval()
    om.load_state_dict(sd)
    with torch.no_grad():
        want = om(x[:1], torch.ones(1) * 501.0)
    e0, e1 = G.rel_err(r0[:1], want), G.rel_err(r1[:1], want)
    print(f"eps vs oracle: {n0} {e0:.3e}, {n1} {e1:.3e}; between the paths {G.rel_err(r0, r1):.3e}")
    assert e0 < 2.5e-2 and e1 < 2.5e-2
