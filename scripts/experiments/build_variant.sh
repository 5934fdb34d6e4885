#!/bin/bash
# build a tuning variant of the product into build_variants/lib<NAME>.so (loaded with SF3D_PRODUCT_LIB): bash scripts/experiments/build_variant.sh NAME -DFLAG=...
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -Wno-unused-value -Wno-unused-function "$@" \
  -I$ROOT/include -I$ROOT/criteria3d_amd/csrc -x hip $ROOT/criteria3d_amd/csrc/sf3d_solver.hip $ROOT/criteria3d_amd/csrc/sf3d_api.cpp -o $ROOT/build_variants/lib$NAME.so
echo built $NAME "$@"
