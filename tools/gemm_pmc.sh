#!/bin/bash
# PMC passes over the GEMM probe shapes (tools/pmc_probe.py); run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gemm_pmc
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/gemm_pmc/p$i -o p$i -- python $R/tools/pmc_probe.py > $R/gpurun_out/gemm_pmc/log$i.txt 2>&1
  db=$(find $R/gpurun_out/gemm_pmc/p$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_pmc.py $db | grep -A4 "igemm"; else tail -3 $R/gpurun_out/gemm_pmc/log$i.txt; fi
  rm -rf $R/gpurun_out/gemm_pmc/p$i
done
