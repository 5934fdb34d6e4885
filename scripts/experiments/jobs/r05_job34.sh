#!/bin/bash
# round 5 job 34: the -DSF3D_LIBM_GLIBC=0 build (make product-fm) with today's kernels on the oracle's fast-math twin - the diagnostic path
# of rounds 3-4 still holds
mkdir -p gpurun_out
make -C oracle oracle-fm > gpurun_out/r05_job34_build.log 2>&1
SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_fm.so SF3D_TEST_RTOL=1e-6 python -m pytest tests/test_gpu_sensitivity.py tests/test_gpu_fastmath.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -8 | tee gpurun_out/r05_job34_fastmath_build_on_the_twin.log
