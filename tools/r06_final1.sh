#!/bin/bash
# Round-6 evidence, part 1: the default bench line (all auxiliary blocks), the kernel trace of the same loop, the face traces.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r06; mkdir -p $out
cd $R
start=$(date +%s)
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$? in $(( $(date +%s) - start )) s"
tail -1 $out/bench_default.json | cut -c1-300
bash tools/r06_trace.sh r06 | tail -3
ROUND_TAG=r06 bash tools/face_trace.sh 8 2>&1 | head -14 | cut -c1-180
ROUND_TAG=r06 bash tools/face_trace.sh 32 2>&1 | head -8 | cut -c1-180
