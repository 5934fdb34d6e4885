#!/bin/bash
# Host side of the product under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU (no GPU needed; GPU sanitizers are not
# available on this pool): the staging model, the validation rules, the partition and strip-local model logic and the C ABI of
# criteria3d_amd/csrc/sf3d_api.cpp + the host parts of sf3d_solver.hip, driven by the CPU tests that load the product library
# (ABI, call-sequence fuzz against the oracle, regular-grid queries, the world-2 / world-3 gloo partition tests).
# usage: bash scripts/sanitize_host.sh        (about two minutes; exits non-zero on the first report)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
mkdir -p $ROOT/build_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -w \
  -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan -I$ROOT/include -I$ROOT/criteria3d_amd/csrc \
  -x hip $ROOT/criteria3d_amd/csrc/sf3d_solver.hip $ROOT/criteria3d_amd/csrc/sf3d_api.cpp -o $ROOT/build_variants/libasan_host.so
cd $ROOT
SF3D_PRODUCT_LIB=$ROOT/build_variants/libasan_host.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_abi.py tests/test_api_fuzz.py tests/test_regular_grid.py tests/test_partition_gloo.py -x -q -m "not gpu" -p no:cacheprovider
# the checker itself: the oracle built with gcc's ASan + UBSan against its golden vectors, the call-sequence fuzz and the project model
# (about six minutes); the regular build is put back afterwards
cp $ROOT/oracle/libsf3d_oracle.so /tmp/libsf3d_oracle_keep.so
trap 'cp /tmp/libsf3d_oracle_keep.so $ROOT/oracle/libsf3d_oracle.so' EXIT
g++ -std=c++17 -O1 -g -fopenmp -ffp-contract=off -fPIC -shared -fsanitize=address,undefined -I$ROOT/include -o $ROOT/oracle/libsf3d_oracle.so $ROOT/oracle/sf3d_oracle.cpp
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  python -m pytest tests/test_oracle_golden.py tests/test_api_fuzz.py tests/test_project3d.py -x -q -m "not gpu and not slow" -p no:cacheprovider
