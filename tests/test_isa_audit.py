"""CPU (no GPU): audit of the gfx950 ISA of every kernel whose LDS-DMA ring is retired by counted ``s_waitcnt vmcnt(N)``.

Round 3's race was a wait whose count included 19 DMA instructions the compiler had deleted (identical place-holder LDS-DMAs
are dead stores to LLVM).  A counted wait is only as good as the number of VMEM instructions that really are in the object
code, so this test compiles the four units to assembly (device pass only, no GPU needed) and checks, per kernel,

  * the number of LDS-DMA instructions against what the source's loop structure issues (nothing merged, nothing dropped),
  * that the counted waits carry exactly the immediates the source asks for,
  * that a kernel and its drained test twin (tests/test_gpu_ring_hazard.py, test_gpu_chain_hazard.py) contain the SAME number
    of DMA instructions and that the twin has no counted ring wait left.
"""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "h-edit_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "--cuda-device-only", "-S"]
UNIT_FLAGS = {"attn": ["-fno-honor-nans"], "ffn": ["-fno-honor-nans"]}          # as h-edit_amd/build.py


def _stats(path):
    cur, out = None, {}
    pending = None           # vmcnt immediate of the last s_waitcnt, until the next instruction shows whether a barrier follows
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            out[cur] = {"dma": 0, "waits": {}, "loads": 0, "stores": 0, "ring_waits": []}
            pending = None
            continue
        if cur is None:
            continue
        ins = line.strip()
        if ins and not ins.startswith((";", ".", "s_waitcnt", "s_nop")) and pending is not None:
            if ins.startswith("s_barrier"):
                out[cur]["ring_waits"].append(pending)
            pending = None
        if re.search(r"buffer_load_dwordx4.* lds", line) or "global_load_lds" in line:
            out[cur]["dma"] += 1
        elif re.search(r"\b(buffer|global)_load_", line):
            out[cur]["loads"] += 1
        elif re.search(r"\b(buffer|global)_store_", line):
            out[cur]["stores"] += 1
        m = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", line)
        if m:
            n = int(m.group(1))
            out[cur]["waits"][n] = out[cur]["waits"].get(n, 0) + 1
            pending = n
    return out


# both builds of the sources: bfloat16 storage (the product) and half storage (-DHEDIT_STORE_F16, csrc/common.h): the rings, their
# DMA instructions and their counted waits must not depend on the operand type
@pytest.fixture(scope="module", params=["bf16", "f16"])
def isa(request):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    tmp = tempfile.mkdtemp(prefix="hedit_isa_")
    GNS_KERNELS.clear()
    extra = ["-DHEDIT_STORE_F16"] if request.param == "f16" else []

    def cc(unit):
        out = os.path.join(tmp, unit + ".s")
        r = subprocess.run([HIPCC] + FLAGS + extra + UNIT_FLAGS.get(unit, []) + ["-o", out, os.path.join(SRC, unit + ".hip")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return unit, _stats(out)
    with ThreadPoolExecutor(4) as ex:
        return dict(ex.map(cc, ["gemm", "pgemm", "pconv", "ffn", "attn", "linchain"]))


def _igemm(isa):
    out = {}
    for name, st in isa["gemm"].items():
        m = re.search(r"igemm_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])E", name)
        if m:
            key = tuple(int(v) for v in m.groups())               # (BM, BN, MODE, CHUNK, DRAIN, DEEP, GNS)
            if key[6]:
                GNS_KERNELS[key[:6]] = st
            else:
                out[key[:6]] = st
    return out


GNS_KERNELS = {}      # the instantiations with the GroupNorm-statistics epilogue (csrc/gnstat.h): same K loop, checked below


def test_igemm_rings_issue_every_dma_the_waits_count(isa):
    ks = _igemm(isa)
    seen_ring = seen_rs = seen_deep = 0
    for (bm, bn, mode, chunk, drain, deep), st in ks.items():
        if drain or bn > 160 or not (bm == 256 or deep):
            continue
        waves = bm // 32
        w_ch = (bn // 8 + waves - 1) // waves        # weight DMA instructions per wave and K-tile
        counted = {n: c for n, c in st["waits"].items() if n >= w_ch}
        if mode in (4, 5):
            # prologue: act(0) 4 | w(0..2) 3 W | act(1) first half 2;  steady 3 tiles: 3 W + 2 halves x 2;  one dummy half past the end
            assert st["dma"] == (4 + 3 * w_ch + 2) + (3 * w_ch + 4) + 2, ((bm, bn, mode, chunk), st)
            assert counted == {w_ch: 1, w_ch + 2: 3, 2 * w_ch + 2: 1}, ((bm, bn, mode, chunk), st)
            seen_rs += 1
        else:
            ndma = 4 + w_ch                            # three-stage ring: one K-tile = 4 activation + W_CH weight instructions
            assert st["dma"] % ndma == 0 and st["dma"] >= 2 * ndma, ((bm, bn, mode, chunk, deep), st)
            assert counted == {ndma: 3, 2 * ndma: 1}, ((bm, bn, mode, chunk, deep), st)
            seen_ring += 1
            seen_deep += deep
        twin = ks.get((bm, bn, mode, chunk, 1, deep))
        assert twin is not None, f"no drained twin of igemm<{bm},{bn},{mode},{chunk},deep={deep}>"
        assert twin["dma"] == st["dma"], "the drained twin must issue the same DMA instructions"
        assert not [n for n in twin["waits"] if n >= w_ch], ("drained twin still has a counted ring wait", twin)
    assert seen_ring >= 20 and seen_rs >= 6 and seen_deep >= 8
    # the two-stage loops (128-row tile, 256 x 256 GEGLU tile) wait vmcnt(0) only and have no twin
    for (bm, bn, mode, chunk, drain, deep), st in ks.items():
        if (bm == 128 and not deep) or bn == 256:
            assert not drain
            assert max(st["waits"]) <= 4, st           # (the compiler's own small waits around the residual loads of the fold)


def test_igemm_statistics_epilogue_leaves_the_k_loop_alone(isa):
    """igemm_kernel<..., GNS = true> differs from its GNS = false twin in the epilogue only: the same LDS-DMA instructions and
    the same counted waits (the ring tests of tests/test_gpu_ring_hazard.py therefore cover its K loop too)."""
    ks = _igemm(isa)
    assert len(GNS_KERNELS) >= 16
    for key, st in GNS_KERNELS.items():
        bm, bn, mode, chunk, drain, deep = key
        assert bn == 128 and mode != 0 and not drain
        twin = ks[key]
        assert st["dma"] == twin["dma"], (key, st, twin)
        waves = bm // 32
        w_ch = (bn // 8 + waves - 1) // waves
        assert {n: c for n, c in st["waits"].items() if n >= w_ch} == {n: c for n, c in twin["waits"].items() if n >= w_ch}, key


def test_ffn_chain_ring(isa):
    ks = {}
    for name, st in isa["ffn"].items():
        m = re.search(r"ffn_chain_kernelILb([01])ELb([01])ELb([01])E", name)
        if m:
            ks[tuple(int(v) for v in m.groups())] = st
    assert len(ks) == 4
    for outer in (0, 1):
        prod, twin = ks[(outer, outer, 0)], ks[(outer, outer, 1)]
        assert prod["dma"] == twin["dma"] and (prod["dma"] - 1) % 8 == 0        # 8 pieces per wave and iteration + the bias image
        iters = (prod["dma"] - 1) // 8
        # hand-over of every iteration but the drained ones: vmcnt((AHEAD - 2) * PPW) = 16
        assert prod["waits"].get(16, 0) >= iters - 4, prod
        assert twin["waits"].get(16, 0) < prod["waits"][16] - (iters - 5), twin   # what is left are the compiler's own waits


def test_self_attention_ring(isa):
    seen = 0
    for name, st in isa["attn"].items():
        m = re.search(r"self_attn_kernelILi(\d+)ELi(\d+)ELi(\d+)E", name)
        if not m:
            continue
        d = int(m.group(1))
        ti = 2 * (d // 8)                            # 1 KiB DMA instructions per KV tile (K + V^T)
        ni = (ti + 3) // 4                           # per wave
        assert st["dma"] % ni == 0 and st["dma"] >= 2 * ni, (d, st)
        if d in (32, 40, 64):                        # ring of four, two tiles ahead: the hand-over tolerates one tile = NI
            assert st["waits"].get(ni, 0) >= 2, (d, st)
        seen += 1
    assert seen == 5


def test_lin_chain_twins(isa):
    ks = {}
    for name, st in isa["linchain"].items():
        m = re.search(r"lin_chain_kernelILi(\d+)ELb([01])ELb([01])ELb([01])ELi(\d+)E", name)
        if m:
            ks[tuple(int(v) for v in m.groups())] = st
    pairs = 0
    for key, st in ks.items():
        if key[-1] != 0:
            continue
        twin = ks.get(key[:-1] + (1,))
        assert twin is not None and twin["dma"] == st["dma"], key
        assert max(twin["waits"]) <= 3, twin         # drained: only the compiler's waits for its own row loads
        assert st["waits"].get(12, 0) >= 32, st      # the ring's steady-state window
        pairs += 1
    assert pairs >= 3


def test_pgemm_ring_counts_every_vector_memory_instruction(isa):
    """csrc/pgemm.hip keeps counted waits with the epilogue's loads and stores in the queue: per tile a wave must issue EXACTLY
    RL residual loads, ST stores, one bias-row LDS-DMA and (first K-tile, middle K-tiles, deferred) NDMA LDS-DMAs each, on every path --
    masked lanes are out-of-range buffer offsets, never cleared exec bits -- and the waits in front of the ring barriers must be
    vmcnt(NDMA) (prologue, middle, last K-tile) and vmcnt(ST + 1 + NDMA) (first K-tile of a tile), all vmcnt(0) in the drained twin;
    the product kernels contain no other vmcnt(0) than the one before s_endpgm (the epilogue's scratch accesses are inline asm: a
    compiler-visible LDS write that may alias a pending LDS-DMA would be preceded by a full drain)."""
    ks = {}
    for name, st in isa["pgemm"].items():
        m = re.search(r"pgemm_kernelILi(\d+)ELb([01])ELb([01])E", name)
        if m:
            ks[tuple(int(v) for v in m.groups())] = st          # (BN, RES, DRAIN)
    assert len(ks) == 8
    for (bn, res, drain), st in ks.items():
        ni = bn // 32
        ndma = 4 + (bn // 8 + 7) // 8
        nq = (16 * (bn // 16) + 63) // 64
        stn = 4 * nq
        assert st["dma"] == 6 * ndma + 2, ((bn, res, drain), st)         # prologue 3 K-tiles | first | middle | deferred; + the bias row DMA (prologue, behind the stores)
        assert st["loads"] == (stn if res else 0), ((bn, res, drain), st)  # residual rows only: the bias comes through the wave's LDS slot
        assert st["stores"] == stn, ((bn, res, drain), st)
        assert ni > 0
        want = [0, 0, 0, 0] if drain else [ndma, stn + 1 + ndma, ndma, ndma]
        if not drain:       # no compiler-made drain inside the ring: the only vmcnt(0) is the one before s_endpgm
            assert st["waits"].get(0, 0) == 1, ((bn, res, drain), st["waits"])
        assert st["ring_waits"] == want, ((bn, res, drain), st["ring_waits"], want)
        assert ks[(bn, res, 1 - drain)]["dma"] == st["dma"]


def test_pconv_ring_counts_every_vector_memory_instruction(isa):
    """csrc/pconv.hip (persistent row-sharing 3x3 loop): the weight ring and the two activation tiles run across tiles with the epilogue's
    loads and stores in the vmcnt queue.  Per kernel the object code must contain exactly: the LDS-DMAs of the prologue (act(0) 4 |
    3 weight K-tiles | first half of act(1)), of the three K-tile bodies of the first tap row, of the middle-row loop, of the last tap row
    (whose last K-tile issues weights only) and the deferred half tile behind the epilogue = 18 + 12 W_CH, + 2 bias-row DMAs; ST residual
    loads or none; ST stores; and in front of the ring barriers the waits vmcnt(2 W + 2) (prologue), 3 x vmcnt(W) (tap column 2), 6 x vmcnt(W + 2)
    (tap columns 0 / 1) plus the two loose waits of a later tile's first tap row, vmcnt(ST + 3) and vmcnt(ST + W + 5) -- all vmcnt(0) in
    the drained twin, which must issue the same instructions."""
    from collections import Counter
    ks = {}
    for name, st in isa["pconv"].items():
        m = re.search(r"pconv_kernelILi(\d+)ELb([01])ELb([01])ELb([01])ELb([01])E", name)
        if m:
            ks[tuple(int(v) for v in m.groups())] = st          # (BN, CHUNK, UP, RES, DRAIN)
    assert len(ks) == 20, sorted(ks)                            # 160: plain x {UP} x {RES}; 128: plain x {UP} x {RES} + fold x {RES}; twins
    for (bn, chunk, up, res, drain), st in ks.items():
        ni = bn // 32
        w = (bn // 8 + 7) // 8
        nq = (16 * (bn // 16) + 63) // 64
        stn = 4 * nq
        key = (bn, chunk, up, res, drain)
        assert st["dma"] == 18 + 12 * w + 2, (key, st)                   # + the bias row DMA (prologue, behind every epilogue's stores)
        assert ni > 0 and st["loads"] == (stn if res else 0), (key, st)
        assert st["stores"] == stn, (key, st)
        if drain:       # (the loose and the plain wait of a first-row K-tile are the same instruction here: the compiler may merge the two paths)
            assert set(st["ring_waits"]) == {0} and 10 <= len(st["ring_waits"]) <= 12, (key, st["ring_waits"])
        else:
            want = Counter({2 * w + 2: 1, w: 3, w + 2: 6, stn + 3: 1, stn + w + 5: 1})
            assert Counter(st["ring_waits"]) == want, (key, st["ring_waits"], want)
        twin = ks[(bn, chunk, up, res, 1 - drain)]
        assert twin["dma"] == st["dma"] and twin["loads"] == st["loads"] and twin["stores"] == st["stores"]
