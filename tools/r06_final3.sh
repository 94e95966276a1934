#!/bin/bash
# Round-6 evidence on the final kernel sources: the default bench line (all auxiliary blocks) and the kernel trace of the same loop.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r06h; mkdir -p $out
cd $R
start=$(date +%s)
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$? in $(( $(date +%s) - start )) s"
tail -1 $out/bench_default.json | cut -c1-200
bash tools/r06_trace.sh r06h | tail -2
