#!/bin/bash
set -u
out=gpurun_out/r06b
mkdir -p "$out"
timeout 600 python tools/pgemm_ab.py 120 20 > "$out/pgemm_ab_120.txt" 2>&1; echo "pgemm_ab rc=$?"; grep -v amdgpu.ids "$out/pgemm_ab_120.txt" | tail -15
timeout 900 python tools/pconv_ab.py 120 10 > "$out/pconv_ab_120.txt" 2>&1; echo "pconv_ab rc=$?"; grep -v amdgpu.ids "$out/pconv_ab_120.txt" | tail -24
HEDIT_STORAGE=f16 timeout 900 python tools/pconv_ab.py 120 3 > "$out/pconv_ab_120_f16.txt" 2>&1; echo "pconv_ab f16 rc=$?"; tail -1 "$out/pconv_ab_120_f16.txt"
HEDIT_STORAGE=f16 timeout 900 python tools/pgemm_ab.py 120 3 > "$out/pgemm_ab_120_f16.txt" 2>&1; echo "pgemm_ab f16 rc=$?"; tail -1 "$out/pgemm_ab_120_f16.txt"
timeout 1500 python -m pytest tests/test_gpu_ring_hazard.py -q -x --tb=short -p no:cacheprovider -k "igemm" -s > "$out/pytest_hazard.log" 2>&1; echo "hazard rc=$?"; tail -2 "$out/pytest_hazard.log"
for r in 120 96 40 5; do
  HEDIT_TEST_FLAGS=8 timeout 200 python tools/unet_time.py $r 5 2>&1 | grep storage | tee -a "$out/unet_ab.txt"
  timeout 200 python tools/unet_time.py $r 5 2>&1 | grep storage | tee -a "$out/unet_ab.txt"
done
timeout 900 python bench.py --steps 2 --warmup 1 --no-config2 --no-half-storage --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06b/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["tflops_per_s"] for k,v in d["kernels_sampled"].items()}, d.get("single_image"))
PY
