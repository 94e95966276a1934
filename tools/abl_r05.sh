#!/bin/bash
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/wdir.txt
: > $OUT
for v in "" wdir nodmaw ""; do
  echo "=== variant '${v}' conv (no chunk)" >> $OUT
  HEDIT_LIB_VARIANT=$v timeout 300 python tools/conv_bench.py 120 64,320,320 64,640,320 32,640,640 16,1280,1280 2>&1 | grep -v amdgpu.ids >> $OUT
done
cat $OUT
