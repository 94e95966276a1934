#!/bin/bash
# self-attention micro-benchmark + HBM traffic counters of the d = 40 kernel (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/sa_q
python $R/tools/sa_bench.py 120 2>&1 | tail -4
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/sa_q/$grp -o p -- python $R/tools/sa_bench.py 120 > $R/gpurun_out/sa_q/log_$grp.txt 2>&1
  db=$(find $R/gpurun_out/sa_q/$grp -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db self_attn | head -8
done
