#!/bin/bash
out=gpurun_out/r06; mkdir -p $out
for v in "" sgb3 sgb4 sgb5 "" sgb3 sgb4 sgb5; do
  echo "variant=${v:-product} $(HEDIT_LIB_VARIANT=$v timeout 300 python tools/sa_bench.py 120 2>&1 | grep 'N=')"
done | tee $out/sa_sgb.txt
