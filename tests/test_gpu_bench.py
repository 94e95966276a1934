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
