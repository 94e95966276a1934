#!/bin/bash
# memory-path PMC passes over the GEMM probe shapes (tools/pmc_probe.py); run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/gemm_pmc
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TD_LOAD_WAVEFRONT_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/gemm_pmc/m$i -o m$i -- python $R/tools/pmc_probe.py > $R/gpurun_out/gemm_pmc/mlog$i.txt 2>&1
  db=$(find $R/gpurun_out/gemm_pmc/m$i -name "*.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/rocpd_pmc.py $db | grep -A4 "igemm"; else tail -3 $R/gpurun_out/gemm_pmc/mlog$i.txt; fi
  rm -rf $R/gpurun_out/gemm_pmc/m$i
done
