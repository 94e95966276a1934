#!/bin/bash
# HBM traffic per kernel class of the DEFAULT bench configuration (bench.py with no workload flags: 24 images in
# lock-step, K = 1, source pass fused), measured as the micro-arch guide prescribes: two SEPARATE rocprofv3 --pmc
# passes (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only.  Per-launch traffic does not depend on the number of
# diffusion steps, so the passes run 4 of the 50 steps.  Writes gpurun_out/pmc_summary.json = {"config": the
# bench line's config.pmc_config, "classes": {...}}; copy it to profiles/pmc_summary.json -- bench.py only uses a
# summary whose config equals its own.  Extra bench flags (e.g. --images 8) may be passed through "$@".
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pm_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d /tmp/pm_$c -o pm -- python $R/bench.py --steps 1 --warmup 0 --diffusion-steps 4 --prof-every 100000000 --no-cpu-baseline --no-single --no-config2 "$@" > /tmp/pm_$c.log 2>&1
  tail -2 /tmp/pm_$c.log | cut -c1-200
done
python $R/tools/pmc_bench_summary.py $(find /tmp/pm_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pm_WRITE_SIZE -name "*.db" | head -1) /tmp/pm_FETCH_SIZE.log > $R/gpurun_out/pmc_summary.json
python -c "
import json; d=json.load(open('$R/gpurun_out/pmc_summary.json'))
print(d['config'])
for k,v in d['classes'].items(): print(k, v['launches'], round(v['hbm_bytes_per_launch']/1e6,1), 'MB/launch')"
