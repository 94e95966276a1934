#!/bin/bash
# PMC passes over the fused feed-forward micro-benchmark (run on the GPU box via gpurun): tools/ffn_pmc.sh [rows]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROWS=${1:-120}
mkdir -p $R/gpurun_out/ffn_pmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_TRANS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/ffn_pmc/p$i -o p$i -- python $R/tools/ffn_bench.py $ROWS 3 > $R/gpurun_out/ffn_pmc/log$i.txt 2>&1
  db=$(find $R/gpurun_out/ffn_pmc/p$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_pmc.py $db ffn_fused | grep -A8 "ffn_fused" | head -12; else tail -3 $R/gpurun_out/ffn_pmc/log$i.txt; fi
done
