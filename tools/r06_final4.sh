#!/bin/bash
# PMC traffic on the final kernel sources (stamped), then the FULL half-storage mirror of the GPU suite, sequential.
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/r06i; mkdir -p $out
cd $R
bash tools/pmc_traffic.sh 2>&1 | tail -8
cp $R/gpurun_out/pmc_summary.json $out/pmc_summary.json
cd $R
start=$(date +%s)
HEDIT_F16_SUITE=full timeout 3000 python -m pytest tests/test_gpu_f16_suite.py -x -q -s -p no:cacheprovider > $out/pytest_f16_full.log 2>&1; echo "f16 full rc=$? in $(( $(date +%s) - start )) s"
grep -a "half-storage suite" $out/pytest_f16_full.log | head -1 | cut -c1-900
tail -2 $out/pytest_f16_full.log
