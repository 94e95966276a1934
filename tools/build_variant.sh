#!/bin/bash
# tools/build_variant.sh NAME UNIT "-DFLAG ..." : rebuild ONE unit with extra flags and link a variant library
# h-edit_amd/hedit/lib_NAME.so.bin next to the product library (for A/B timing on the GPU box).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; UNIT=$2; EXTRA=$3
mkdir -p /tmp/hedit_variants
OBJ=/tmp/hedit_variants/${UNIT}_${NAME}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-honor-nans -Wno-unused-result $EXTRA -c $ROOT/h-edit_amd/csrc/$UNIT.hip -o $OBJ
OBJS=""
for u in gemm pgemm pconv ffn linchain norm attn step grad pnet unet vae ddpm irse lpips vit c_api; do
  if [ $u == $UNIT ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $ROOT/h-edit_amd/build/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/h-edit_amd/hedit/lib_$NAME.so.bin $OBJS
echo built lib_$NAME.so.bin
