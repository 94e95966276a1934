#!/bin/bash
# Kernel trace of the face workload (configs[3]) restricted to the loop: per-kernel table, busy / span, and where the
# idle time between dispatches sits.  tools/face_trace.sh [faces]   -> gpurun_out/$ROUND_TAG/face_* (default r04)  (run on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${ROUND_TAG:-r04}
N=${1:-32}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/p_face -o trace -- python $R/bench.py --workload face --images $N --steps 1 --warmup 1 --diffusion-steps 20 > $O/bench_face_n${N}_under_rocprof.json 2> $O/rocprof_face.err
db=$(find /tmp/p_face -name "*.db" | head -1)
if [ -n "$db" ]; then
  python $R/tools/rocpd_stats.py $db --loop > $O/face_n${N}_kernel_stats.txt
  python $R/tools/rocpd_stats.py $db --loop --gaps > $O/face_n${N}_gaps.txt
  python $R/tools/rocpd_stats.py $db --loop --by-grid > $O/face_n${N}_kernel_stats_by_grid.txt
  head -12 $O/face_n${N}_kernel_stats.txt; head -45 $O/face_n${N}_gaps.txt
fi
rm -rf /tmp/p_face
