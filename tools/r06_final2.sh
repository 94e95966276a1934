#!/bin/bash
# Round-6 evidence on the final kernel sources: PMC traffic (stamped), the default bench line, the kernel trace.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r06f; mkdir -p $out
cd $R
bash tools/pmc_traffic.sh 2>&1 | tail -12
cp $R/gpurun_out/pmc_summary.json $R/profiles/pmc_summary.json
cp $R/gpurun_out/pmc_summary.json $out/pmc_summary.json
cd $R
start=$(date +%s)
python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "default bench rc=$? in $(( $(date +%s) - start )) s"
tail -1 $out/bench_default.json | cut -c1-200
bash tools/r06_trace.sh r06f | tail -2
