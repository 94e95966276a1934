#!/bin/bash
# tools/ring_hazard.sh [N]: tests/test_gpu_ring_hazard.py at N launches per case (default 20000) for the product library, then at the
# suite's default count for side libraries with deliberately WEAKENED waits if they exist (lib_weak.so.bin: every counted wait of
# igemm / ffn_chain / self_attn tolerates one more slot than it may; lib_weakf.so.bin: ffn_chain's hand-over waits for nothing) --
# the control that the protocol can see a wait that is too weak.  Output: gpurun_out/r05/ring_hazard.txt
N=${1:-20000}
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/ring_hazard.txt
echo "# tests/test_gpu_ring_hazard.py, HEDIT_HAZARD_LAUNCHES=$N, product library" > $OUT
HEDIT_HAZARD_LAUNCHES=$N python -m pytest tests/test_gpu_ring_hazard.py -q -s -k "counted_waits" 2>&1 | grep -E "launches|passed|failed" | sed 's/^\.*//' >> $OUT
for v in weak weakf; do
  if [ -f h-edit_amd/hedit/lib_$v.so.bin ]; then
    echo "# control: side library lib_$v.so.bin (weakened waits), default launch count" >> $OUT
    HEDIT_LIB_VARIANT=$v python -m pytest tests/test_gpu_ring_hazard.py -q -s -k "counted_waits" 2>&1 | grep -E "^E +AssertionError|launches|passed|failed" | sed 's/^\.*//' >> $OUT
  fi
done
tail -5 $OUT
