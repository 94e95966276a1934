#!/bin/bash
out=gpurun_out/r06; mkdir -p $out
for v in "" occ1 "" occ1; do
  echo "variant=${v:-product} $(HEDIT_LIB_VARIANT=$v timeout 300 python tools/sa_bench.py 120 2>&1 | grep 'N=')"
done | tee $out/sa_occ1.txt
timeout 300 python tools/layer_prof.py 5 3 > $out/layer_prof_5.txt 2>&1; head -3 $out/layer_prof_5.txt; tail -45 $out/layer_prof_5.txt | cut -c1-200
