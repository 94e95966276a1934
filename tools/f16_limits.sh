#!/bin/bash
# Head-room of every per-format tolerance (tests/helpers/gpu.py::within) in BOTH storage formats: the files that compare HIP results
# with the fp32 oracle run once per format with HEDIT_LIM_REPORT set; then the 50-step divergence curve against the committed oracle
# trajectory in both formats.  gpurun --timeout 1800 -- 'bash tools/f16_limits.sh'  ->  gpurun_out/f16_limits/{limits.tsv,summary.txt,loop_divergence.txt}
set -u
out=gpurun_out/f16_limits
mkdir -p "$out"
export HEDIT_LIM_REPORT="$PWD/$out/limits.tsv"
rm -f "$HEDIT_LIM_REPORT"; : > "$out/summary.txt"
FILES="test_gpu_loop_trajectory test_gpu_unet test_gpu_loops test_gpu_vae test_gpu_sd_shape_style test_gpu_style test_gpu_face test_gpu_masactrl test_gpu_pnp"
for fmt in f16 bf16; do
  for name in $FILES; do
    HEDIT_STORAGE=$fmt timeout 900 python -m pytest tests/$name.py -q --tb=line -p no:cacheprovider > "$out/${name}_$fmt.log" 2>&1
    echo "$fmt $name rc=$? $(tail -1 "$out/${name}_$fmt.log")" | tee -a "$out/summary.txt"
  done
  HEDIT_STORAGE=$fmt timeout 900 python -m pytest tests/test_gpu_invariance.py -k "not sd15_loops_match_oracle" -q --tb=line -p no:cacheprovider > "$out/test_gpu_invariance_$fmt.log" 2>&1
  echo "$fmt test_gpu_invariance rc=$? $(tail -1 "$out/test_gpu_invariance_$fmt.log")" | tee -a "$out/summary.txt"
done
(HEDIT_STORAGE=bf16 timeout 300 python tests/diag/diag_loop_divergence.py; HEDIT_STORAGE=f16 timeout 300 python tests/diag/diag_loop_divergence.py) 2>&1 | grep -v amdgpu.ids > "$out/loop_divergence.txt"
grep "final" "$out/loop_divergence.txt" | tee -a "$out/summary.txt"
# measured / limit per comparison, worst first
sort -t$'\t' -k1,1 "$out/limits.tsv" | awk -F'\t' '{printf "%s\t%.2f\t%s\t%s\t%s\t%s\n", $1, $2/$3, $2, $3, $4, $5}' | sort -t$'\t' -k1,1 -k2,2nr > "$out/headroom.tsv"
head -25 "$out/headroom.tsv"
