#!/bin/bash
# tools/experiments/build_xffn.sh : side library h-edit_amd/hedit/lib_xffn.so.bin = the product objects + the round-4 experiments (xffn.hip, lintile.hip)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p /tmp/hedit_variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
/opt/rocm/bin/hipcc $F -c $ROOT/tools/experiments/xffn.hip -o /tmp/hedit_variants/xffn.o
/opt/rocm/bin/hipcc $F -DLINTILE_C=${LINTILE_C:-640} -c $ROOT/tools/experiments/lintile.hip -o /tmp/hedit_variants/lintile.o
/opt/rocm/bin/hipcc $F -c $ROOT/tools/experiments/xffn_api.hip -o /tmp/hedit_variants/xffn_api.o
OBJS=""
for u in gemm ffn linchain norm attn step grad pnet unet vae ddpm irse lpips vit c_api; do OBJS="$OBJS $ROOT/h-edit_amd/build/$u.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/h-edit_amd/hedit/lib_xffn.so.bin $OBJS /tmp/hedit_variants/xffn.o /tmp/hedit_variants/lintile.o /tmp/hedit_variants/xffn_api.o
echo built lib_xffn.so.bin
