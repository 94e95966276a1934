#!/bin/bash
out=gpurun_out/r06; mkdir -p $out
start=$(date +%s)
timeout 2400 python -m pytest tests/test_gpu_f16_suite.py -x -q -s -p no:cacheprovider > $out/pytest_f16_core.log 2>&1; echo "f16 core rc=$? in $(( $(date +%s) - start )) s"
grep -a "half-storage suite" $out/pytest_f16_core.log | cut -c1-900
tail -3 $out/pytest_f16_core.log
(HEDIT_STORAGE=bf16 timeout 300 python tests/diag/diag_loop_divergence.py; HEDIT_STORAGE=f16 timeout 300 python tests/diag/diag_loop_divergence.py) 2>&1 | grep -v amdgpu.ids > $out/loop_divergence.txt
grep -i "final\|step 50\|storage" $out/loop_divergence.txt | head -8
