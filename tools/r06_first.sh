#!/bin/bash
# Round 6, first GPU call after re-entry: persistent linear kernel A/B (bits + time), its ring-hazard twins, whole-UNet A/B, bench line.
set -u
out=gpurun_out/r06
mkdir -p "$out"
timeout 600 python tools/pgemm_ab.py 120 20 > "$out/pgemm_ab_120.txt" 2>&1; echo "pgemm_ab rc=$?"
tail -16 "$out/pgemm_ab_120.txt"
for r in 120 40 5; do
  HEDIT_TEST_FLAGS=8 timeout 200 python tools/unet_time.py $r 5 2>&1 | grep storage | tee -a "$out/unet_ab.txt"
  timeout 200 python tools/unet_time.py $r 5 2>&1 | grep storage | tee -a "$out/unet_ab.txt"
done
timeout 900 python -m pytest tests/test_gpu_ring_hazard.py tests/test_gpu_kernels.py tests/test_gpu_unet.py -q -x --tb=short -p no:cacheprovider > "$out/pytest_first.log" 2>&1; echo "pytest rc=$?"; tail -3 "$out/pytest_first.log"
timeout 900 python bench.py --steps 2 --warmup 1 --no-config2 --no-half-storage --no-cpu-baseline > "$out/bench_first.json" 2> "$out/bench_first.err"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/bench_first.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["tflops_per_s"] for k,v in d["kernels_sampled"].items()}, d.get("single_image"))
PY
