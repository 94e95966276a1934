#!/bin/bash
# Round-end evidence (run on the GPU box via gpurun): default bench line, the same command under
# rocprofv3 --kernel-trace --stats, summaries under gpurun_out/ (copy the ones to keep into profiles/).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
mkdir -p $R/gpurun_out/$TAG
cd $R && python bench.py > $R/gpurun_out/$TAG/bench.json 2> $R/gpurun_out/$TAG/bench.err
tail -1 $R/gpurun_out/$TAG/bench.json | cut -c1-400
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single > $R/gpurun_out/$TAG/bench_under_rocprof.json 2> $R/gpurun_out/$TAG/rocprof.err
db=$(find $R/gpurun_out/$TAG/prof -name "*.db" | head -1)
if [ -n "$db" ]; then
  python $R/tools/rocpd_stats.py $db > $R/gpurun_out/$TAG/kernel_stats.txt
  python $R/tools/rocpd_stats.py $db --by-grid > $R/gpurun_out/$TAG/kernel_stats_by_grid.txt
  head -14 $R/gpurun_out/$TAG/kernel_stats.txt
  rm -rf $R/gpurun_out/$TAG/prof
else
  tail -5 $R/gpurun_out/$TAG/rocprof.err
fi
