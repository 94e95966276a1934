"""CPU: the GroupNorm pair-statistics tree (h-edit_amd/csrc/gnstat.h) is ONE function of the stored tile, whatever the producer.

tests/helpers/gnstat_host.hip compiles the header's helper functions for the host and runs them in the thread layouts of the
128-row tile / split-K reduce (256 threads) and of the 256-row tile (512 threads); the numpy restatement of the tree
(tests/helpers/gnstat_ref.py, written from the header's comment) must reproduce both bit for bit, and all agree with fp64
sums.  What the -m gpu half (tests/test_gpu_gn_stats.py) then checks is that the kernels produce these bits."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import gnstat_ref  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    exe = os.path.join(tempfile.mkdtemp(prefix="hedit_gns_"), "gnstat_host")
    r = subprocess.run([HIPCC, "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-o", exe,
                        os.path.join(ROOT, "tests", "helpers", "gnstat_host.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return exe


def tile_bits(seed, scale, shift):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((256, 128)) * scale + shift).astype(np.float32)
    u = x.view(np.uint32)
    u = u + 0x7FFF + ((u >> 16) & 1)                    # round to nearest even, as the kernels store
    return (u >> 16).astype(np.uint16)


@pytest.mark.parametrize("seed,scale,shift", [(0, 1.0, 0.0), (1, 3.0, 0.5), (2, 1e-3, 40.0), (3, 300.0, -7.0)])
def test_every_producer_form_is_the_same_tree(harness, seed, scale, shift):
    bits = tile_bits(seed, scale, shift)
    d = os.path.dirname(harness)
    bits.tofile(os.path.join(d, "tile.bin"))
    subprocess.check_call([harness, os.path.join(d, "tile.bin"), os.path.join(d, "out.bin")])
    out = np.fromfile(os.path.join(d, "out.bin"), dtype=np.float32).reshape(2, 2, 64, 2)      # [form][unit][pair][s, q]
    assert np.array_equal(out[0].view(np.uint32), out[1].view(np.uint32)), "128-row and 256-row forms differ"
    want = gnstat_ref.pair_stats(bits)
    assert np.array_equal(out[0].view(np.uint32), want.view(np.uint32)), "header code and the restatement of its comment differ"
    v = gnstat_ref.bf16_bits_to_f32(bits).astype(np.float64).reshape(2, 128, 64, 2)
    s64, q64 = v.sum(axis=(1, 3)), (v * v).sum(axis=(1, 3))
    assert np.abs(out[0, :, :, 0] - s64).max() <= 2e-6 * np.abs(v).sum(axis=(1, 3)).max()
    assert np.abs(out[0, :, :, 1] - q64).max() <= 2e-6 * q64.max()


def test_group_statistics_from_pairs_match_torch_group_norm():
    """the consumer's arithmetic on top of the pairs (norm.hip gn_fold_kernel + the apply pass's mean / rstd formulas), in
    numpy: groups assembled from the pairs of a tensor alone and of a two-producer concatenation with a group that straddles
    the seam (256 + 128 channels, 12 channels per group) reproduce torch's GroupNorm statistics."""
    import torch
    rng = np.random.default_rng(5)
    HW = 1024
    mk = lambda C: tile_bits_any(rng, HW, C)
    for ca, cb in ((128, 0), (256, 128), (128, 128)):
        a = mk(ca)
        pa = gnstat_ref.pair_stats(a)
        parts = [pa]
        cols = [gnstat_ref.bf16_bits_to_f32(a)]
        if cb:
            b = mk(cb)
            parts.append(gnstat_ref.pair_stats(b))
            cols.append(gnstat_ref.bf16_bits_to_f32(b))
        pairs = np.concatenate(parts, axis=1)                       # [units][C / 2][2]
        x = np.concatenate(cols, axis=1)                            # [HW][C]
        C = ca + cb
        ppg = C // 32 // 2
        sums = pairs.astype(np.float64).sum(axis=0).reshape(32, ppg, 2).sum(axis=1)
        n = HW * (C // 32)
        mean = sums[:, 0] / n
        var = sums[:, 1] / n - mean ** 2
        xt = torch.from_numpy(x.T.copy()).reshape(1, C, HW).double()
        g = xt.reshape(1, 32, -1)
        assert np.allclose(mean, g.mean(-1)[0].numpy(), rtol=0, atol=1e-6)
        assert np.allclose(var, g.var(-1, unbiased=False)[0].numpy(), rtol=1e-5, atol=1e-7)


def tile_bits_any(rng, M, N):
    x = (rng.standard_normal((M, N)) * 2 + 0.3).astype(np.float32)
    u = x.view(np.uint32)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u >> 16).astype(np.uint16)
