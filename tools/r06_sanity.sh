#!/bin/bash
# last check of the committed tree on a GPU box: smoke(), the bench line (traffic must be published from the stamped summary)
out=gpurun_out/r06z; mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 900 python bench.py --steps 2 --warmup 1 --no-config2 --no-half-storage --no-cpu-baseline > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06z/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"].get("traffic_over_algorithmic"), d["single_image"]["latency_s"])
PY
