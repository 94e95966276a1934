"""-m gpu: bench.py end to end at toy size -- the JSON contract of the line the driver parses, and the RCCL path
(--force-dist initialises the nccl process group with one rank: weight broadcast as ONE blob from rank 0, barrier,
max-over-ranks timing) which the 8-GPU scaling run uses unchanged."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--tiny", "--steps", "1", "--warmup", "1", "--images", "3",
           "--diffusion-steps", "6", "--no-cpu-baseline", "--prof-every", "2"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [[], ["--force-dist"], ["--force-dist", "--no-fuse-src"]])
def test_bench_line_contract(extra):
    d = _run(extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1 and d["unit"] == "images/s"
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and d["finite"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["launches_sampled"] > 0
    assert r["alg_bytes_per_launch"] > 0
    # the reconstruction branch retraces the inversion exactly, whatever the batch layout (tests/test_gpu_invariance.py)
    assert d["recon_rel_err"] < 2e-6


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script_args, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _need_two_gpus():
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 visible GPUs (the driver's 1-GPU box skips; any multi-GPU node runs it)")


def test_two_ranks_on_rccl_bench_line():
    """N > 1 on the REAL backend: two ranks on nccl (= RCCL), launched exactly as the driver launches the scaling run
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 ... bench.py --gpus 2): rank 0 builds the weights and
    broadcasts the bf16 / fp32 blobs, both ranks edit their own images, barrier + max-over-ranks timing, ONE JSON line from
    rank 0 whose value counts both ranks' images.  Skipped where fewer than two devices are visible; the gloo tests
    (tests/test_dist_gloo.py) cover the same code on the CPU."""
    _need_two_gpus()
    r = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--tiny", "--steps", "1", "--warmup", "1", "--images", "3",
                      "--diffusion-steps", "6", "--no-cpu-baseline", "--prof-every", "2"])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["finite"] is True and d["value"] > 0
    assert d["recon_rel_err"] < 2e-6
    # 3 images per rank and step: the whole job's images over the slowest rank's time
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]


def test_two_ranks_on_rccl_driver_shards_the_dataset(tmp_path):
    """the product driver under two nccl ranks: each rank edits its contiguous shard of the mapping file, rank 0 alone
    creates the weights and broadcasts them, and the union of the PNGs is the single-process result, byte for byte
    (h-Edit-D: no random numbers in the inversion)."""
    _need_two_gpus()
    import numpy as np
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_driver import _dataset, _driver
    d = _dataset(tmp_path)
    common = ["--data_path", str(d), "--random_init", "--tiny", "--num_diffusion_steps", "4", "--edit_category_list", "0", "1",
              "--mode", "h_edit_D_p2p", "--eta", "0.0", "--implicit"]
    one = _driver().main(common + ["--output_path", str(tmp_path / "r1")])
    r = _torchrun(2, [os.path.join(ROOT, "h-edit_amd", "main_p2p.py")] + common + ["--output_path", str(tmp_path / "r2")])
    assert r.returncode == 0, r.stderr[-3000:]
    two = sorted(str(p) for p in (tmp_path / "r2").rglob("*.png"))
    assert len(one) == len(two) == 2
    for a, b in zip(sorted(one), two):
        assert os.path.basename(a) == os.path.basename(b)
        assert np.array_equal(np.array(Image.open(a)), np.array(Image.open(b)))
