#!/bin/bash
# tools/chain_hazard.sh [launches=20000] : the differential stress test of the projection chains (tests/test_gpu_chain_hazard.py)
# at full length, twice:
#   1. the product library, product schedule (sched 0) against the drained schedule -> must be 0 mismatches;
#   2. round 3's dropped IN-LAYER row-DMA variant (sched 2; exists only in a -DHEDIT_LINCHAIN_INLAYER build of linchain.hip,
#      linked here as h-edit_amd/hedit/lib_inlayer.so.bin) against the same drained schedule -> what the test was written to catch.
# Build step (hipcc, no GPU needed):  tools/chain_hazard.sh build        Run step (GPU box): tools/chain_hazard.sh [launches]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" == "build" ]; then
  mkdir -p /tmp/hedit_variants
  OBJ=/tmp/hedit_variants/linchain_inlayer.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DHEDIT_LINCHAIN_INLAYER \
      -c $ROOT/h-edit_amd/csrc/linchain.hip -o $OBJ
  OBJS=""
  for u in gemm ffn linchain norm attn step grad pnet unet vae ddpm irse lpips vit c_api; do
    if [ $u == linchain ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $ROOT/h-edit_amd/build/$u.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/h-edit_amd/hedit/lib_inlayer.so.bin $OBJS
  echo built lib_inlayer.so.bin
  exit 0
fi
N=${1:-20000}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd $ROOT
echo "== product schedule (sched 0), $N launches per form at 120 rows" | tee $OUT/chain_hazard.txt
HEDIT_HAZARD_LAUNCHES=$N timeout 3000 python -m pytest tests/test_gpu_chain_hazard.py -q -s -k counted 2>&1 | grep -E "^chain|passed|failed|Error" | tee -a $OUT/chain_hazard.txt
echo "== in-layer variant of round 3 (sched 2, lib_inlayer.so.bin), $N launches per form at 120 rows" | tee -a $OUT/chain_hazard.txt
HEDIT_LIB_VARIANT=inlayer HEDIT_CHAIN_SCHED=2 HEDIT_HAZARD_LAUNCHES=$N timeout 3000 python -m pytest tests/test_gpu_chain_hazard.py -q -s -k "counted and not 576" 2>&1 | grep -E "^chain|passed|failed|Error|differ" | tee -a $OUT/chain_hazard.txt
