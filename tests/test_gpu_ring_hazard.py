"""-m gpu: drained-schedule differential stress test of EVERY kernel whose operand ring is retired by counted
``s_waitcnt vmcnt(N)`` waits ("What's weak" 1 / "do this" 2).

Round 3's lin_chain_kernel passed every whole-loop check with a wait that waited for nothing (the compiler had merged 19
of 20 place-holder DMAs); what found it was tests/test_gpu_chain_hazard.py: the kernel against a twin whose every wait is
``vmcnt(0) lgkmcnt(0)`` in front of its barrier, bit for bit, over many launches, while a second stream sweeps HBM.  This
file runs that protocol on the other three rings:

  * igemm_kernel (csrc/gemm.hip): the three-stage ring of the 256-row tile (mode 0 / 1, both column tiles, with and
    without the in-register chunk fold), the same ring under the 128-row tile for launches of at most one block per CU and the row-sharing 3x3 loop (kernel modes 4 / 5: stride-1 and 2x-upsampled
    gather, 160-column plain, 128-column plain and chunked) -- ``igemm_kernel<..., DRAIN = true>``;
  * ffn_chain_kernel (csrc/ffn.hip): the feed-forward alone and the whole block tail -- ``ffn_chain_kernel<..., true>``;
  * self_attn_kernel (csrc/attn.hip): the head dims whose K / V^T ring runs two tiles ahead (d = 32, 40, 64) -- the
    hand-over's ``p.test_flags & 1``.  (cross_attn_kernel stages through plain __syncthreads() phases: no counted wait.)

``hedit_test_set_flags(1)`` selects the drained twins for the reference launch; the product schedule (flags 0) must
reproduce its bits.  The two-stage igemm loop (128-row tile, 256 x 256 GEGLU tile) waits vmcnt(0) as it is.
HEDIT_HAZARD_LAUNCHES: launches per case at the large shapes (default 300; tools/ring_hazard.sh runs 20 000).
Also here: the self-attention pinned-shift pass against the exact pass (flags 2), bit for bit, on inputs that trip the
denominator check (advisor, round 4)."""
import ctypes as C
import math
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hedit import _lib  # noqa: E402

if os.environ.get("HEDIT_LIB_VARIANT"):        # tools/ring_hazard.sh: a side library with a deliberately weakened wait
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), f"lib_{os.environ['HEDIT_LIB_VARIANT']}.so.bin")

pytestmark = pytest.mark.gpu
LAUNCHES = int(os.environ.get("HEDIT_HAZARD_LAUNCHES", "300"))
DEV = "cuda:0"


def _bf(t):
    return t.to(_lib.storage_dtype()).to(DEV)


def _stress(run, launches, hog):
    """run(out=None) -> tuple of tensors.  Reference = the drained twins (twice: repeatable), then `launches` product
    launches into fixed buffers with `hog` 1 GiB copies per launch on a second stream -> (bad launches, bad elements)."""
    lib = _lib.lib()
    _lib.check(lib.hedit_test_set_flags(1))
    try:
        ref = run()
        torch.cuda.synchronize()
        again = run()
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.hedit_test_set_flags(0))
    assert all(torch.equal(u, v) for u, v in zip(ref, again)), "the drained schedule itself is not repeatable"
    out = tuple(torch.empty_like(t) for t in ref)
    bad_launch = torch.zeros((), dtype=torch.int64, device=DEV)
    bad_elems = torch.zeros((), dtype=torch.int64, device=DEV)
    side = torch.cuda.Stream()
    src = dst = None
    if hog:
        src = torch.empty(1 << 29, dtype=torch.int16, device=DEV).random_(0, 1000)        # 1 GiB
        dst = torch.empty_like(src)
    for i in range(launches):
        if hog:
            with torch.cuda.stream(side):
                for _ in range(hog):
                    dst.copy_(src)
        run(out)
        n = sum(torch.count_nonzero(u.view(torch.int16) != v.view(torch.int16)) for u, v in zip(ref, out))
        bad_elems += n
        bad_launch += (n > 0).to(torch.int64)
        if i % 64 == 63:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return int(bad_launch), int(bad_elems)


# ------------------------------------------------------------------------------------------------ igemm_kernel
# (name, mode, M-or-(B, Hin, Win), Cin / K, N, chunks)   chunks: 0 = plain chain, n = canonical-style fold of n chunks in
# registers (hedit_k_gemm splits = -n), -n = n split-K slabs; the first twelve are large enough for the 256-row tile (>= 200 tiles, K >= 1024)
GEMM_CASES = [
    # >= 2 tiles per CU, no fold: the persistent kernel (csrc/pgemm.hip) -- its ring runs across tiles with the epilogue's loads
    # and stores in the vmcnt queue; both column tiles, with / without the residual rows, a ragged row tile, K = 3 tiles (the
    # shortest ring: every DMA of a tile belongs to the next one)
    ("persist 160 res    FF2 L1", 0, 122880, 2560, 640, 0),
    ("persist 160 nores  qk L1", 0, 122880, 640, 1280, 0, False),
    ("persist 160 res    out-proj L1 ragged", 0, 122880 + 72, 640, 640, 0),
    ("persist 128 res    ragged", 0, 2 * 65536 + 100, 1024, 1024, 0),
    ("persist 128 nores  K = 192", 0, 4 * 65536, 192, 384, 0, False),
    ("persist 160 res    K = 192, N = 200", 0, 4 * 65536 + 8, 192, 200, 0),
    # 200 <= tiles < 2 per CU: the one-shot three-stage ring of the 256-row tile
    ("ring3 160 plain  FF2", 0, 20480, 2560, 640, 0),
    ("ring3 160 fold   FF2 L1", 0, 122880, 2560, 640, 4),
    ("ring3 128 plain", 0, 8192 + 100, 1024, 1024, 0),               # ragged last row tile
    ("ring3 128 fold", 0, 65536, 2048, 1024, 2),
    ("ring3 160 plain  stride-2 conv", 2, (40, 64, 64), 320, 320, 0),
    # row-sharing 3x3 loop, >= 2 tiles per CU: the persistent kernel (csrc/pconv.hip) -- weight ring and activation tiles run across
    # tiles, the epilogue's loads / stores and the deferred activation DMA sit in the vmcnt queue; both column tiles, the chunk fold,
    # with / without residual rows, tiles that span images + a ragged last tile, the upsampling gather
    ("persist-conv 160 plain res   64x64 320->320", 1, (60, 64, 64), 320, 320, 0),
    ("persist-conv 160 plain nores 64x64 320->320", 1, (60, 64, 64), 320, 320, 0, False),
    ("persist-conv 128 fold  res   32x32 640->640", 1, (120, 32, 32), 640, 640, 2),
    ("persist-conv 128 fold  nores 32x32 640->640", 1, (120, 32, 32), 640, 640, 2, False),
    ("persist-conv 128 plain res   8x8 ragged", 1, (3301, 8, 8), 128, 128, 0),     # 825.25 row tiles, four images per tile
    ("persist-conv-up 160 plain res   32->64 320", 3, (40, 32, 32), 320, 320, 0),
    ("persist-conv-up 128 plain nores 16->32 128", 3, (160, 16, 16), 128, 128, 0, False),
    # 200 <= tiles < 2 per CU: the one-shot row-sharing loop of igemm_kernel (modes 4 / 5)
    ("rowshare 160 plain  64x64 320->320", 1, (12, 64, 64), 320, 320, 0),
    ("rowshare 128 fold   32x32 640->640", 1, (24, 32, 32), 640, 640, 2),
    ("rowshare 128 plain  64x64 128->128", 1, (16, 64, 64), 128, 128, 0),
    ("rowshare 128 plain  8x8 ragged", 1, (1301, 8, 8), 128, 128, 0),     # 325.25 row tiles
    ("rowshare-up 160 plain 32->64 320", 3, (12, 32, 32), 320, 320, 0),
    ("rowshare-up 128 fold  16->32 640", 3, (120, 16, 16), 640, 640, 2),
    ("rowshare-up 128 plain 32->64 128", 3, (16, 32, 32), 128, 128, 0),
    # launches with at most one 128-row block per CU: the three-stage ring of the 128-row tile (Smem DEEP); negative chunks =
    # that many split-K slabs (the form small batches run the canonical chunking in)
    ("deep128 160 linear 1280x1280", 0, 1280, 1280, 1280, 0),
    ("deep128 160 linear 640x640 ragged", 0, 5120 + 40, 640, 640, 0),
    ("deep128 128 linear", 0, 2048, 1024, 1024, 0),
    ("deep128 160 conv 16x16 split-K", 1, (2, 16, 16), 1280, 1280, -4),
    ("deep128 160 stride-2 conv", 2, (5, 32, 32), 640, 640, 0),
    ("deep128 160 upsampling conv split-K", 3, (2, 8, 8), 1280, 1280, -4),
]


class Gemm:
    def __init__(self, mode, shape, cin, N, chunks, res=True, seed=0):
        self.lib = _lib.lib()
        g = torch.Generator().manual_seed(seed)
        self.mode, self.N, self.cin = mode, N, cin
        if mode == 0:
            self.M, self.K = shape, cin
            self.A = _bf(torch.randn(self.M, cin, generator=g))
            self.conv = (0, 0, 0, 0, 0)
        else:
            B, Hin, Win = shape
            Ho, Wo = (Hin, Win) if mode == 1 else ((Hin // 2, Win // 2) if mode == 2 else (2 * Hin, 2 * Win))
            self.M, self.K = B * Ho * Wo, 9 * cin
            self.A = _bf(torch.randn(B * Hin * Win, cin, generator=g))
            self.conv = (Hin, Win, cin, Ho, Wo)
        self.W = _bf(torch.randn(N, self.K, generator=g) / math.sqrt(self.K))
        self.bias = torch.randn(N, generator=g).to(DEV)
        self.R = _bf(torch.randn(self.M, N, generator=g)) if res else None
        self.splits = -chunks               # hedit_k_gemm: < 0 folded in registers, > 0 split-K slabs + reduce
        self.ws = torch.empty(max(self.lib.hedit_k_gemm_ws_bytes(self.M, N, self.K, abs(chunks)), 16), dtype=torch.uint8, device=DEV)

    def run(self, out=None):
        o = out[0] if out else torch.empty(self.M, self.N, dtype=_lib.storage_dtype(), device=DEV)
        p = _lib.ptr
        _lib.check(self.lib.hedit_k_gemm(p(self.A), p(self.W), p(self.bias), p(self.R) if self.R is not None else None, p(o), self.M, self.N, self.K,
                                         self.cin if self.mode else self.K, self.N, self.N, self.mode, *self.conv, self.splits,
                                         p(self.ws), _lib.cur_stream()))
        return (o,)


@pytest.mark.parametrize("case", GEMM_CASES, ids=[c[0].replace(" ", "_") for c in GEMM_CASES])
def test_igemm_counted_waits_reproduce_the_drained_schedule(case):
    name, mode, shape, cin, N, chunks = case[:6]
    gm = Gemm(mode, shape, cin, N, chunks, *case[6:])
    # the case must reach a kernel WITH counted waits, i.e. differ in code from its drained twin: checked through time
    n = max(20, int(LAUNCHES * (0.5 if gm.M * N * gm.K > 2e14 else 1.0)))
    bad, elems = _stress(gm.run, n, 2)
    print(f"igemm {name}: M = {gm.M}, {n} launches under load: {bad} mismatching launches, {elems} elements")
    assert bad == 0, f"{bad} of {n} launches differ from the drained schedule ({elems} elements)"


def test_igemm_drained_twin_matches_fp32():
    """the reference of the stress test is itself checked against torch fp32 (a twin that is wrong the same way as the
    product would make the differential test vacuous)"""
    lib = _lib.lib()
    gm = Gemm(1, (8, 64, 64), 128, 128, 0, seed=5)
    _lib.check(lib.hedit_test_set_flags(1))
    try:
        (o,) = gm.run()
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.hedit_test_set_flags(0))
    x = gm.A.float().reshape(8, 64, 64, 128).permute(0, 3, 1, 2)
    w4 = gm.W.float().view(128, 3, 3, 128).permute(0, 3, 1, 2).contiguous()
    ref = torch.nn.functional.conv2d(x, w4, gm.bias, padding=1).permute(0, 2, 3, 1).reshape(gm.M, 128)
    ref = ref.to(_lib.storage_dtype()).float() + gm.R.float()
    assert float((o.float() - ref).norm() / ref.norm()) < 4e-3


# ------------------------------------------------------------------------------------------------ ffn_chain_kernel
class Ffn:
    def __init__(self, M, outer, seed=0):
        self.lib = lib = _lib.lib()
        Cc = lib.hedit_k_ffn_channels()
        g = torch.Generator().manual_seed(seed)
        self.M, self.C, self.outer = M, Cc, outer
        mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(DEV)      # noqa: E731
        w1, b1 = mk(8 * Cc, Cc, sc=Cc ** -0.5), mk(8 * Cc, sc=0.2)
        w2, self.b2 = mk(Cc, 4 * Cc, sc=(4 * Cc) ** -0.5), mk(Cc, sc=0.2)
        wpre, wpost = mk(Cc, Cc, sc=Cc ** -0.5), mk(Cc, Cc, sc=Cc ** -0.5)
        self.bpre, self.bpost = mk(Cc, sc=0.2), mk(Cc, sc=0.2)
        self.gamma, self.beta = (1 + 0.1 * torch.randn(Cc, generator=g)).to(DEV), mk(Cc, sc=0.1)
        self.stream = torch.empty(lib.hedit_k_ffn_stream_bytes(1 if outer else 0), dtype=torch.uint8, device=DEV)
        self.b1p = torch.empty(lib.hedit_k_ffn_bias_bytes(), dtype=torch.uint8, device=DEV)
        p = _lib.ptr
        _lib.check(lib.hedit_k_ffn_pack(p(w1), p(b1), p(w2), p(wpre) if outer else None, p(wpost) if outer else None,
                                        p(self.stream), p(self.b1p), None))
        self.x = _bf(torch.randn(M, Cc, generator=g) * 1.5)
        self.a = _bf(torch.randn(M, Cc, generator=g))
        self.t1 = _bf(torch.randn(M, Cc, generator=g) * 1.5)
        torch.cuda.synchronize()

    def run(self, out=None):
        o = out[0] if out else torch.empty(self.M, self.C, dtype=_lib.storage_dtype(), device=DEV)
        p, Cc = _lib.ptr, self.C
        if self.outer:
            _lib.check(self.lib.hedit_k_ffn_chain(p(self.a), Cc, p(self.t1), Cc, p(self.x), Cc, p(self.bpre), p(self.gamma), p(self.beta),
                                                  1e-5, p(self.stream), p(self.b1p), p(self.b2), p(self.bpost), p(o), Cc, self.M, Cc,
                                                  _lib.cur_stream()))
        else:
            _lib.check(self.lib.hedit_k_ffn_fused(p(self.x), Cc, p(self.gamma), p(self.beta), 1e-5, p(self.stream), p(self.b1p), p(self.b2),
                                                  p(o), Cc, self.M, Cc, _lib.cur_stream()))
        return (o,)


@pytest.mark.parametrize("outer", [True, False], ids=["block_tail", "ff_only"])
@pytest.mark.parametrize("rows,tokens,frac,hog", [(120, 4096, 1.0, 2), (96, 4096, 0.25, 2), (48, 4096, 0.25, 1), (5, 576, 0.5, 1)])
def test_ffn_chain_counted_waits_reproduce_the_drained_schedule(outer, rows, tokens, frac, hog):
    f = Ffn(rows * tokens, outer)
    n = max(20, int(LAUNCHES * frac))
    bad, elems = _stress(f.run, n, hog)
    print(f"ffn_chain outer={outer}: M = {f.M}, {n} launches, memory hog {hog}: {bad} mismatching launches, {elems} elements")
    assert bad == 0, f"{bad} of {n} launches differ from the drained schedule ({elems} elements)"


# ------------------------------------------------------------------------------------------------ self_attn_kernel
class SelfAttn:
    def __init__(self, B, N, heads, d, seed=0, spikes=()):
        self.lib = _lib.lib()
        g = torch.Generator().manual_seed(seed)
        Cc = heads * d
        self.B, self.N, self.heads, self.d, self.Cc = B, N, heads, d, Cc
        q = torch.randn(B, N, Cc, generator=g) * 0.3
        k = torch.randn(B, N, Cc, generator=g)
        v = torch.randn(B, N, Cc, generator=g)
        for (b, qi, ki, lift) in spikes:                 # key ki beats everything else of query qi by 2^lift
            for h in range(heads):
                qh = q[b, qi, h * d:(h + 1) * d]
                k[b, ki, h * d:(h + 1) * d] = qh * (lift / float(qh @ qh))
        self.qk = _bf(torch.cat([q, k], dim=-1)).contiguous()
        self.vt = _bf(v).reshape(B * N, Cc).t().contiguous()
        self.k_ptr = C.c_void_p(self.qk.reshape(B * N, 2 * Cc)[:, Cc:].data_ptr())

    def run(self, out=None):
        o = out[0] if out else torch.empty(self.B, self.N, self.Cc, dtype=_lib.storage_dtype(), device=DEV)
        _lib.check(self.lib.hedit_k_self_attn(_lib.ptr(self.qk), 2 * self.Cc, self.k_ptr, 2 * self.Cc, _lib.ptr(self.vt), self.B * self.N,
                                              _lib.ptr(o), self.Cc, self.B, self.N, self.heads, self.d, None, None, _lib.cur_stream()))
        return (o,)


@pytest.mark.parametrize("B,N,heads,d,frac,hog", [(120, 4096, 8, 40, 0.5, 2), (24, 4096, 8, 40, 1.0, 2), (20, 1024, 8, 40, 1.0, 1),
                                                   (16, 1024, 4, 64, 1.0, 1), (16, 1024, 4, 32, 1.0, 1), (5, 576, 8, 40, 1.0, 1)])
def test_self_attention_counted_waits_reproduce_the_drained_schedule(B, N, heads, d, frac, hog):
    sa = SelfAttn(B, N, heads, d)
    n = max(20, int(LAUNCHES * frac))
    bad, elems = _stress(sa.run, n, hog)
    print(f"self_attn d={d}: B = {B}, N = {N}, {n} launches, memory hog {hog}: {bad} mismatching launches, {elems} elements")
    assert bad == 0, f"{bad} of {n} launches differ from the drained schedule ({elems} elements)"


@pytest.mark.parametrize("d,heads,N", [(40, 8, 1024), (80, 8, 512), (160, 8, 256), (32, 2, 512)])
def test_self_attention_redo_branch_is_the_exact_pass_bit_for_bit(d, heads, N):
    """Advisor (round 4): a block whose pinned-shift pass fails the denominator check repeats its KV sweep with the exact
    online-softmax pass; its output must then BE the exact pass's (hedit_test_set_flags(2) = exact pass only).  Spikes of
    2^90 / 2^200 placed so that every (row, head) has one in the query block they are compared on; blocks without a spike
    stay on the fast pass and are allowed to differ in the last bit."""
    lib = _lib.lib()
    B = 2
    spikes = [(0, 7, 3 * N // 4 + 16, 90.0), (0, 200, 70 + N // 2, 200.0), (1, N - 1, N - 1, 90.0), (1, 33, N // 2 + 129 % (N // 2), 200.0)]       # all keys in the second half: behind the units the shift is taken from
    sa = SelfAttn(B, N, heads, d, seed=d, spikes=spikes)
    (fast,) = sa.run()
    torch.cuda.synchronize()
    _lib.check(lib.hedit_test_set_flags(2))
    try:
        (exact,) = sa.run()
        torch.cuda.synchronize()
    finally:
        _lib.check(lib.hedit_test_set_flags(0))
    assert torch.isfinite(fast.float()).all() and torch.isfinite(exact.float()).all()
    qb = 128 * (2 if d <= 64 else 1)                                   # queries per block (QG = 2 below d = 80)
    for (b, qi, _, _) in spikes:
        lo = (qi // qb) * qb
        assert torch.equal(fast[b, lo:lo + qb], exact[b, lo:lo + qb]), (b, qi)
    # everywhere else the two passes agree to bf16 rounding
    assert float((fast.float() - exact.float()).abs().max()) < 3e-2
