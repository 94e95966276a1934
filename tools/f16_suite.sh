#!/bin/bash
# The whole -m gpu suite in half storage (HEDIT_STORAGE=f16 -> libhedit_hip_f16.so), file by file so that one failure does not
# hide the rest; then the eps diagnostic, the UNet timing in both formats and the loop-divergence curve in half storage.
# Run on the GPU box from the repository root:  gpurun --timeout 2400 -- 'bash tools/f16_suite.sh'
# (round 6: profiles/r06_f16_suite_summary.txt; the driver's own `pytest -m gpu` runs the same files through tests/test_gpu_f16_suite.py)
set -u
out=gpurun_out/f16_suite
mkdir -p "$out"
export HEDIT_STORAGE=f16
export HEDIT_LIM_REPORT="$PWD/$out/limits.tsv"      # every per-format tolerance comparison: storage, measured, limit, test (tests/helpers/gpu.py::within)
rm -f "$HEDIT_LIM_REPORT"
: > "$out/summary.txt"
for f in tests/test_gpu_*.py; do
  case "$f" in tests/test_gpu_f16_suite.py|tests/test_gpu_ring_hazard.py|tests/test_gpu_chain_hazard.py) continue ;; esac
  name=$(basename "$f" .py)
  timeout 900 python -m pytest "$f" -q --tb=short -p no:cacheprovider > "$out/$name.log" 2>&1
  echo "$name rc=$? $(tail -1 "$out/$name.log")" | tee -a "$out/summary.txt"
done
timeout 300 python tests/diag/diag_storage_eps_error.py 2 2>&1 | grep storage | tee -a "$out/summary.txt"
(timeout 120 env HEDIT_STORAGE=bf16 python tools/unet_time.py 120 5; timeout 120 python tools/unet_time.py 120 5) 2>&1 | grep storage | tee -a "$out/summary.txt"
# the 50-step loop against the committed oracle trajectory, both formats (seconds each: the oracle side is a fixture)
(env HEDIT_STORAGE=bf16 timeout 300 python tests/diag/diag_loop_divergence.py; timeout 300 python tests/diag/diag_loop_divergence.py) 2>&1 | grep -v amdgpu.ids > "$out/loop_divergence.txt"
grep "final" "$out/loop_divergence.txt" | tee -a "$out/summary.txt"
timeout 1200 python bench.py --storage f16 --steps 1 --warmup 1 --no-config2 --no-cpu-baseline > "$out/bench_f16.json" 2> "$out/bench_f16.err"
tail -c 600 "$out/bench_f16.json" | tee -a "$out/summary.txt"
