#!/usr/bin/env python3
"""Build libhedit_hip.so (gfx950) in-tree with hipcc.  `python h-edit_amd/build.py [--force] [--f16]`.

--f16: the half-storage build of the same sources (-DHEDIT_STORE_F16, csrc/common.h) -> hedit/libhedit_hip_f16.so, selected at
run time with HEDIT_STORAGE=f16 (hedit/_lib.py).  The default library, every benchmark figure and BASELINE configs[1] are bfloat16.

No cmake / setuptools indirection: one hipcc invocation per translation unit (objects cached by
source mtime under h-edit_amd/build/), one link.  The .so lands next to the Python package
(h-edit_amd/hedit/libhedit_hip.so) so it travels with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "hedit", "libhedit_hip.so")
UNITS = ["gemm.hip", "pgemm.hip", "pconv.hip", "ffn.hip", "linchain.hip", "norm.hip", "attn.hip", "step.hip", "grad.hip", "pnet.hip", "unet.hip", "vae.hip", "ddpm.hip", "irse.hip", "lpips.hip", "vit.hip", "c_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
# per-unit extras.  attn.hip: the row-max chains run on raw MFMA results; without the no-NaN promise
# every fmaxf operand is first canonicalised (v_max x,x), tripling the instruction count of the
# softmax's max pass.  (NaN inputs propagate to NaN outputs either way.)
UNIT_FLAGS = {"attn.hip": ["-fno-honor-nans"], "ffn.hip": ["-fno-honor-nans"]}   # (ffn.hip: max(x, 0) of the GELU)


def _deps_mtime():
    hdrs = [os.path.join(SRC, f) for f in os.listdir(SRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "hedit.h"))
    return max(os.path.getmtime(p) for p in hdrs)


def build(force=False, verbose=True, f16=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    OBJ = os.path.join(HERE, "build_f16" if f16 else "build")
    OUT = os.path.join(HERE, "hedit", "libhedit_hip_f16.so" if f16 else "libhedit_hip.so")
    FLAGS = globals()["FLAGS"] + (["-DHEDIT_STORE_F16"] if f16 else [])
    os.makedirs(OBJ, exist_ok=True)
    hm = _deps_mtime()
    jobs = []
    for u in UNITS:
        src = os.path.join(SRC, u)
        obj = os.path.join(OBJ, u.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hm):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + UNIT_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s in ex.map(cc, jobs):
                if verbose:
                    print("compiled", os.path.relpath(s, HERE))
    objs = [os.path.join(OBJ, u.replace(".hip", ".o")) for u in UNITS]
    if jobs or not os.path.exists(OUT):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
        if verbose:
            print("linked", os.path.relpath(OUT, HERE))
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, f16="--f16" in sys.argv)
