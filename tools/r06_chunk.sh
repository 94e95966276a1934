#!/bin/bash
out=gpurun_out/r06c; mkdir -p $out
timeout 900 python tools/pconv_ab.py 120 10 > "$out/pconv_ab_120.txt" 2>&1; echo "pconv_ab rc=$?"; grep -v amdgpu.ids "$out/pconv_ab_120.txt" | grep -E "chunk_kt= *[1-9]|ALL|MISMATCH" | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_ring_hazard.py -q -x --tb=short -p no:cacheprovider -k "persist" > "$out/pytest_hazard.log" 2>&1; echo "hazard rc=$?"; tail -1 "$out/pytest_hazard.log"
for r in 120 120; do
  HEDIT_TEST_FLAGS=8 timeout 200 python tools/unet_time.py $r 5 2>&1 | grep storage | tee -a "$out/unet_ab.txt"
  timeout 200 python tools/unet_time.py $r 5 2>&1 | grep storage | tee -a "$out/unet_ab.txt"
done
