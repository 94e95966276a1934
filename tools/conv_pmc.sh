#!/bin/bash
# tools/conv_pmc.sh [ROWS=120] : SQ counter passes over the two dominant 3x3 conv kernels of the SD loop:
#   igemm_kernel<256,160,4,false>  (row-sharing loop, 160-column tile: L0 64x64 320->320, unchunked)
#   igemm_kernel<256,128,4,true>   (row-sharing loop with the canonical K-chunk fold, 128-column tile: L1 32x32 640->640)
# through tools/conv_bench.py with the executor's chunking (CONV_BENCH_CANONICAL=1).  One rocprofv3 --pmc pass per counter
# group (never combined with other trace domains), per-kernel averages by tools/rocpd_pmc.py -> gpurun_out/r04/conv_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROWS=${1:-120}
OUT=$R/gpurun_out/r04
mkdir -p $OUT/conv_pmc
: > $OUT/conv_pmc.txt
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM" "SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM"; do
  i=$((i+1))
  echo "== pass $i: $grp" >> $OUT/conv_pmc.txt
  CONV_BENCH_CANONICAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $OUT/conv_pmc/p$i -o p$i -- python $R/tools/conv_bench.py $ROWS 64,320,320 32,640,640 16,1280,1280 > $OUT/conv_pmc/log$i.txt 2>&1
  db=$(find $OUT/conv_pmc/p$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_pmc.py $db igemm >> $OUT/conv_pmc.txt; else tail -3 $OUT/conv_pmc/log$i.txt >> $OUT/conv_pmc.txt; fi
  grep conv3x3 $OUT/conv_pmc/log$i.txt >> $OUT/conv_pmc.txt
  rm -rf $OUT/conv_pmc/p$i
done
tail -60 $OUT/conv_pmc.txt
