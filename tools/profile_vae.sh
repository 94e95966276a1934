#!/bin/bash
# rocprofv3 kernel statistics of the decoder forward + vector-Jacobian product alone (tools/vae_bench.py)
# and of the style workload (bench.py --workload style), summaries under gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01_style}
mkdir -p $R/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof_vae -o trace -- python $R/tools/vae_bench.py > $R/gpurun_out/$TAG/vae_bench.txt 2> $R/gpurun_out/$TAG/rocprof_vae.err
db=$(find $R/gpurun_out/$TAG/prof_vae -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $R/gpurun_out/$TAG/vae_kernel_stats.txt && head -24 $R/gpurun_out/$TAG/vae_kernel_stats.txt
rm -rf $R/gpurun_out/$TAG/prof_vae
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o trace -- python $R/bench.py --workload style --images 8 --steps 1 --warmup 0 --diffusion-steps 10 > $R/gpurun_out/$TAG/bench_style_under_rocprof.json 2> $R/gpurun_out/$TAG/rocprof.err
db=$(find $R/gpurun_out/$TAG/prof -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $R/gpurun_out/$TAG/style_kernel_stats.txt && head -30 $R/gpurun_out/$TAG/style_kernel_stats.txt
rm -rf $R/gpurun_out/$TAG/prof
