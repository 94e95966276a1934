#!/bin/bash
# Round-5 evidence batch (run on the GPU box via gpurun): default bench line (with the configs2 / 3 / 4 blocks), kernel trace
# of the default workload, HBM traffic counters of the default configuration, kernel traces of the face and style
# workloads restricted to the loop (busy / span of the timed pass).  Summaries land under gpurun_out/r05/; copy the ones
# to keep into profiles/.   tools/profile_r05.sh [quick]  (quick: skip the face / style traces and the PMC passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05
mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_main -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single --no-config2 > $O/bench_under_rocprof.json 2> $O/rocprof.err
db=$(find /tmp/p_main -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats.txt && python $R/tools/rocpd_stats.py $db --by-grid > $O/kernel_stats_by_grid.txt && head -14 $O/kernel_stats.txt
rm -rf /tmp/p_main
[ "$1" == "quick" ] && exit 0
bash $R/tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1; tail -8 $O/pmc_traffic.log
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_face -o trace -- python $R/bench.py --workload face --steps 1 --warmup 0 --diffusion-steps 20 > $O/bench_face_under_rocprof.json 2> $O/rocprof_face.err
db=$(find /tmp/p_face -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db --loop > $O/face_kernel_stats.txt && head -10 $O/face_kernel_stats.txt
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db --loop --gaps > $O/face_gaps.txt && head -3 $O/face_gaps.txt
rm -rf /tmp/p_face
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_style -o trace -- python $R/bench.py --workload style --images 8 --steps 1 --warmup 0 --diffusion-steps 10 --no-cpu-baseline > $O/bench_style_under_rocprof.json 2> $O/rocprof_style.err
db=$(find /tmp/p_style -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db --loop > $O/style_kernel_stats.txt && head -10 $O/style_kernel_stats.txt
rm -rf /tmp/p_style
