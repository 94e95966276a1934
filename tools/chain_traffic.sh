#!/bin/bash
# HBM traffic counters of the chain kernels and of the launches they replace (tools/ffn_bench.py), two separate --pmc passes:
# per kernel the average FETCH_SIZE / WRITE_SIZE (KiB, raw) per launch.   tools/chain_traffic.sh [rows]  -> gpurun_out/r03/chain_traffic.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ROWS=${1:-120}
mkdir -p $R/gpurun_out/r03
: > $R/gpurun_out/r03/chain_traffic.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/ct_$c
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/ct_$c -o ct -- python $R/tools/ffn_bench.py $ROWS 2 > /tmp/ct_$c.log 2>&1
  db=$(find /tmp/ct_$c -name "*.db" | head -1)
  echo "== $c (KiB per launch, raw)" >> $R/gpurun_out/r03/chain_traffic.txt
  [ -n "$db" ] && python $R/tools/rocpd_pmc.py $db >> $R/gpurun_out/r03/chain_traffic.txt
done
cat $R/gpurun_out/r03/chain_traffic.txt
