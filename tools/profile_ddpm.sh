#!/bin/bash
# rocprofv3 kernel statistics of the pixel DDPM UNet forward (tools/ddpm_bench.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01_face}
mkdir -p $R/gpurun_out/$TAG
cd $R && python tools/ddpm_bench.py > $R/gpurun_out/$TAG/ddpm_bench.txt 2>&1; cat $R/gpurun_out/$TAG/ddpm_bench.txt | tail -6
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o trace -- python $R/tools/ddpm_bench.py > /dev/null 2> $R/gpurun_out/$TAG/rocprof.err
db=$(find $R/gpurun_out/$TAG/prof -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $R/gpurun_out/$TAG/ddpm_kernel_stats.txt && head -24 $R/gpurun_out/$TAG/ddpm_kernel_stats.txt
rm -rf $R/gpurun_out/$TAG/prof
