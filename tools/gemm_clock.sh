#!/bin/bash
# shader clock of the GEMM kernel with / without its two halves (variant libraries built by tools/build_variant.sh)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/h-edit_amd/hedit/libhedit_hip.so /tmp/keep.so
for v in base nodma nomfma; do
  cp $R/h-edit_amd/hedit/lib_$v.so.bin $R/h-edit_amd/hedit/libhedit_hip.so
  rm -rf /tmp/ck_$v
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES -d /tmp/ck_$v -o ck -- python $R/tools/pmc_probe.py > /tmp/ck_$v.log 2>&1
  db=$(find /tmp/ck_$v -name "*.db" | head -1)
  echo "== $v"; python $R/tools/rocpd_clock.py $db igemm
done
cp /tmp/keep.so $R/h-edit_amd/hedit/libhedit_hip.so
