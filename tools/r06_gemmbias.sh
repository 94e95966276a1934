#!/bin/bash
# A/B of the one-shot igemm bias slot (variant oldgemm = the previous gemm.hip) + kernel tests + host enqueue time
out=gpurun_out/r06g; mkdir -p $out
for r in 5 40 120 5 40 120; do
  HEDIT_LIB_VARIANT=oldgemm timeout 200 python tools/unet_time.py $r 6 2>&1 | grep storage | sed 's/^/old  /' | tee -a "$out/unet_ab.txt"
  timeout 200 python tools/unet_time.py $r 6 2>&1 | grep storage | sed 's/^/new  /' | tee -a "$out/unet_ab.txt"
done
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_ring_hazard.py tests/test_gpu_unet.py -q -x --tb=short -p no:cacheprovider > "$out/pytest.log" 2>&1; echo "pytest rc=$?"; tail -2 "$out/pytest.log"
timeout 600 python -m pytest tests/test_gpu_invariance.py -q -x --tb=short -p no:cacheprovider -k "not sd15_loops" > "$out/pytest_inv.log" 2>&1; echo "invariance rc=$?"; tail -2 "$out/pytest_inv.log"
timeout 600 python bench.py --steps 2 --warmup 1 --no-config2 --no-half-storage --no-cpu-baseline > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06g/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], {k:v["tflops_per_s"] for k,v in d["kernels_sampled"].items()}, d.get("single_image"))
PY
