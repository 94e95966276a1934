#!/bin/bash
# Round-2 evidence batch (run on the GPU box via gpurun): default bench line, kernel trace of the default workload, PMC passes
# of the self-attention kernel, kernel traces of the face and style workloads.  Summaries land under gpurun_out/r02/;
# copy the ones to keep into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02
mkdir -p $O
cd $R && python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_main -o trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single --no-config2 > $O/bench_under_rocprof.json 2> $O/rocprof.err
db=$(find /tmp/p_main -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/kernel_stats.txt && python $R/tools/rocpd_stats.py $db --by-grid > $O/kernel_stats_by_grid.txt && head -12 $O/kernel_stats.txt
rm -rf /tmp/p_main
bash $R/tools/sa_pmc.sh > $O/sa_pmc.txt 2>&1; grep -c avg $O/sa_pmc.txt
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_face -o trace -- python $R/bench.py --workload face --steps 1 --warmup 0 --diffusion-steps 20 > $O/bench_face_under_rocprof.json 2> $O/rocprof_face.err
db=$(find /tmp/p_face -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/face_kernel_stats.txt && head -8 $O/face_kernel_stats.txt
rm -rf /tmp/p_face
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_style -o trace -- python $R/bench.py --workload style --images 8 --steps 1 --warmup 0 --diffusion-steps 10 > $O/bench_style_under_rocprof.json 2> $O/rocprof_style.err
db=$(find /tmp/p_style -name "*.db" | head -1)
[ -n "$db" ] && python $R/tools/rocpd_stats.py $db > $O/style_kernel_stats.txt && head -8 $O/style_kernel_stats.txt
rm -rf /tmp/p_style
cd $R && python bench.py --workload face --steps 1 --warmup 1 > $O/bench_face.json 2>/dev/null; tail -c 300 $O/bench_face.json; echo
python bench.py --workload style --images 16 --steps 1 --warmup 1 > $O/bench_style.json 2>/dev/null; python -c "
import json;d=json.loads(open('$O/bench_style.json').read().strip().splitlines()[-1]);print('style', d['value'], d['ms_per_step'])"
