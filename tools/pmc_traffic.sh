#!/bin/bash
# HBM traffic per kernel class of the bench workload (two separate --pmc passes, as the micro-arch
# guide prescribes): writes gpurun_out/pmc_summary.json (copy to profiles/pmc_summary.json).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_$c -o pm -- python $R/bench.py --steps 1 --warmup 0 --diffusion-steps 4 --prof-every 100000000 --no-cpu-baseline --no-single > /tmp/pm_$c.log 2>&1
  tail -2 /tmp/pm_$c.log | cut -c1-200
done
python $R/tools/pmc_bench_summary.py $(find /tmp/pm_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*.db" | head -1) > $R/gpurun_out/pmc_summary.json
python -c "
import json; d=json.load(open('$R/gpurun_out/pmc_summary.json'))
for k,v in d.items(): print(k, v['launches'], round(v['hbm_bytes_per_launch']/1e6,1), 'MB/launch')"
