#!/bin/bash
# images per GPU in lock-step vs tile quantisation: 5n / 4n-row passes on 256 CUs.  n = 24 -> 120 rows = 7.5 / 3.75 tile rounds at the 32x32 / 16x16 levels;
# n = 32 -> 160 / 128 rows = 10 / 5 and 8 / 4 rounds.
set -u
out=gpurun_out/r06
mkdir -p "$out"
for n in 24 32 24 32 48; do
  timeout 600 python bench.py --images $n --steps 1 --warmup 1 --no-config2 --no-half-storage --no-cpu-baseline --no-single > "$out/bench_n$n.json" 2> "$out/bench_n$n.err"
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
d = json.loads(open(f"gpurun_out/r06/bench_n{n}.json").read().strip().splitlines()[-1])
print("images", n, "value", d["value"], "ms/step", d["ms_per_step"], "TF/s", d["achieved_tflops_per_s_per_gpu"], {k: v["tflops_per_s"] for k, v in d["kernels_sampled"].items()}, flush=True)
PY
done 2>&1 | tee "$out/images_sweep.txt"
