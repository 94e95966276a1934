"""-m gpu: the GPU suite a second time in half storage (libhedit_hip_f16.so = the same kernels compiled with -DHEDIT_STORE_F16,
csrc/common.h), so that the driver's one `pytest -m gpu` run covers both product formats (VERDICT r5 "do this" 1).

A process has one storage format, decided before the library is loaded (HEDIT_STORAGE, hedit/_lib.py), so every test file runs in a
child pytest process with HEDIT_STORAGE=f16.  The tolerances are per format: tests/helpers/gpu.py::lim / within give a HIP-vs-fp32
comparison its bfloat16 limit in the parent run and a quarter of it (or the explicit f16 limit, e.g. 3e-3 for one SD-1.5 eps
evaluation) in these children.

Two selections.  The DEFAULT ("core") is what the driver's run can afford next to the bfloat16 suite: every file, one child after the
other (three children sharing the GPU ran 15x slower each on the round-6 box: 54 minutes, a 161 s file hit its 1500 s limit --
profiles/r06_raw/pytest_gpu_full_pool3.log), and inside the five files whose time is the fp32 oracle on the host or minutes of
format-independent bitwise self-comparison only the tests that can tell the formats apart (HIP against the oracle, finite outputs, the
shortest case of each loop family).  HEDIT_F16_SUITE=full runs every test of every file (tools/f16_suite.sh does the same file by file:
profiles/r06_f16_suite_summary.txt, all green).

Not repeated in half storage in either selection: the drained-twin ring tests (test_gpu_ring_hazard.py / test_gpu_chain_hazard.py --
schedules, not formats; the ISA audit tests/test_isa_audit.py checks the rings of BOTH builds on the CPU) and
test_sd15_loops_match_oracle (minutes of fp32 oracle on the host; the SD-shape loop is covered in both formats by
tests/test_gpu_loop_trajectory.py against the committed oracle trajectory)."""
import os
import re
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
F16_LIB = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhedit_hip_f16.so")
FULL = os.environ.get("HEDIT_F16_SUITE", "core") == "full"
POOL = int(os.environ.get("HEDIT_F16_SUITE_POOL", "1"))

# (file, pytest -k selection of the core run or None = the whole file, least number of tests that must have passed: core / full)
SUITE = [
    ("test_gpu_kernels.py", None, 100, 100),
    ("test_gpu_unet.py", None, 13, 13),
    ("test_gpu_loop_trajectory.py", None, 1, 1),
    ("test_gpu_loops.py", "ddpm_inversion or null_edit or h_edit_d or (loops_match_oracle and (explicit or R_implicit))", 5, 13),
    ("test_gpu_vae.py", None, 15, 15),
    ("test_gpu_face.py", None, 10, 10),
    ("test_gpu_sd_shape_style.py", None, 3, 3),
    ("test_gpu_style.py", "style_step_matches_oracle or style_guidance_moves", 2, 8),
    ("test_gpu_masactrl.py", "reads_kv_of_another_row or changes_the_edit or needs_a_registered", 6, 10),
    ("test_gpu_foreign_controller.py", "probabilities_and_apply or failing_controller", 4, 15),
    ("test_gpu_invariance.py", "chunks or canonical or groupnorm or tiny_unet or reusing_the_source", 10, 20),
    ("test_gpu_driver.py", "driver_writes_edited_images and not face and not masactrl and not pnp and not demo and not style", 3, 18),
    ("test_gpu_bench.py", None, 3, 3),
    ("test_gpu_arcface.py", None, 4, 4),
    ("test_gpu_gn_stats.py", None, 18, 18),
    ("test_gpu_lpips.py", None, 5, 5),
    ("test_gpu_clip.py", None, 5, 5),
    ("test_gpu_pnp.py", None, 4, 4),
    ("test_gpu_errors.py", None, 4, 4),
]


def _child(entry):
    name, core_k, _, _ = entry
    env = dict(os.environ, HEDIT_STORAGE="f16")
    env.pop("PYTEST_CURRENT_TEST", None)
    extra = []
    if FULL:
        if name == "test_gpu_invariance.py":
            extra = ["-k", "not sd15_loops_match_oracle"]
    elif core_k:
        extra = ["-k", core_k]
    t0 = time.time()
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", name), "-q", "-x", "--tb=short", "-p", "no:cacheprovider",
                            "--durations=3"] + extra, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        return name, (r.returncode, r.stdout + r.stderr, time.time() - t0)
    except subprocess.TimeoutExpired as e:
        return name, (124, f"timeout after 1500 s\n{e.stdout or ''}", time.time() - t0)


@pytest.fixture(scope="module")
def f16_runs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(F16_LIB), "build it: python h-edit_amd/build.py --f16 (__graft_entry__.build() does)"
    t0 = time.time()
    with ThreadPoolExecutor(POOL) as ex:
        res = dict(ex.map(_child, SUITE))
    print(f"half-storage suite ({'full' if FULL else 'core'}): {len(SUITE)} files in {time.time() - t0:.0f} s ({POOL} at a time): "
          + ", ".join(f"{k} {v[2]:.0f}s" for k, v in res.items()))
    return res


@pytest.mark.parametrize("entry", SUITE, ids=[e[0][:-3] for e in SUITE])
def test_file_passes_in_half_storage(f16_runs, entry):
    name, _, least_core, least_full = entry
    least = least_full if FULL else least_core
    rc, out, secs = f16_runs[name]
    assert rc == 0, f"{name} under HEDIT_STORAGE=f16 (rc {rc}, {secs:.0f} s):\n{out[-3000:]}"
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) >= least, out[-500:]
    assert "failed" not in out.splitlines()[-1], out[-500:]
