#!/bin/bash
out=gpurun_out/r06; mkdir -p $out
for r in 5 5 120; do
  timeout 200 python tools/unet_time.py $r 8 2>&1 | grep storage | sed 's/^/product /' | tee -a $out/gn_cap64.txt
  HEDIT_LIB_VARIANT=cap64 timeout 200 python tools/unet_time.py $r 8 2>&1 | grep storage | sed 's/^/cap64   /' | tee -a $out/gn_cap64.txt
done
