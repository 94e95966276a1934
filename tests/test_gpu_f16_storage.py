"""-m gpu: the half-storage build (libhedit_hip_f16.so: the same kernels compiled with -DHEDIT_STORE_F16, csrc/common.h), driven in
child processes with HEDIT_STORAGE=f16 -- a process has one storage format, decided before the library is loaded.

  * every kernel-level test of tests/test_gpu_kernels.py (each against a plain PyTorch fp32 reference of the same op on inputs
    rounded to the storage format) passes with half storage at the tolerances written for bfloat16;
  * one SD-1.5-shaped UNet evaluation against the fp32 oracle: the reference UNet is fp32 (text-guided/main_p2p.py:106) and
    north_star words the tolerance as fp16's -- measured 1.45e-3 (bfloat16: 1.16e-2, tests/test_gpu_unet.py), limit 3e-3.
"""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "h-edit_amd"))
from hedit import _lib  # noqa: E402

pytestmark = pytest.mark.gpu
F16_LIB = os.path.join(os.path.dirname(_lib.LIB_PATH), "libhedit_hip_f16.so")


def run_f16(args, timeout):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    assert os.path.exists(F16_LIB), "build it: python h-edit_amd/build.py --f16 (__graft_entry__.build() does)"
    env = dict(os.environ, HEDIT_STORAGE="f16")
    r = subprocess.run([sys.executable] + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    return r.returncode, r.stdout + r.stderr


def test_kernel_tests_pass_in_half_storage():
    rc, out = run_f16(["-m", "pytest", os.path.join("tests", "test_gpu_kernels.py"), "-q", "-x", "--tb=short", "-p", "no:cacheprovider"], 600)
    assert rc == 0, out[-3000:]
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) >= 100, out[-500:]


def test_sd15_eps_error_in_half_storage():
    rc, out = run_f16([os.path.join("tests", "diag", "diag_storage_eps_error.py"), "1"], 600)
    assert rc == 0, out[-3000:]
    m = re.search(r"storage f16: sd15 eps error vs fp32 oracle, 1 row\(s\): ([0-9.e+-]+)", out)
    assert m, out[-1000:]
    err = float(m.group(1))
    print(f"half storage: sd15 eps error vs fp32 oracle {err:.3e}")
    assert err < 3e-3, err
