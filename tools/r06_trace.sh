#!/bin/bash
# kernel trace of the bench loop (1 step) -> gpurun_out/r06/kernel_stats{,_by_grid}.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single --no-config2 --no-half-storage > $R/gpurun_out/$TAG/bench_under_rocprof.json 2> $R/gpurun_out/$TAG/rocprof.err
db=$(find $R/gpurun_out/$TAG/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python $R/tools/rocpd_stats.py $db > $R/gpurun_out/$TAG/kernel_stats.txt
  python $R/tools/rocpd_stats.py $db --by-grid > $R/gpurun_out/$TAG/kernel_stats_by_grid.txt
  head -30 $R/gpurun_out/$TAG/kernel_stats.txt | cut -c1-200
  rm -rf $R/gpurun_out/$TAG/prof
else
  tail -5 $R/gpurun_out/$TAG/rocprof.err
fi
