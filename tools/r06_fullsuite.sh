#!/bin/bash
out=gpurun_out/r06; mkdir -p $out
start=$(date +%s)
timeout 3400 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/pytest_gpu_full.log 2>&1; echo "pytest -m gpu rc=$? in $(( $(date +%s) - start )) s"
tail -5 $out/pytest_gpu_full.log
grep -a "half-storage suite" $out/pytest_gpu_full.log | cut -c1-600
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -5
