#!/bin/bash
# round 5 job 6: the -DSF3D_LIBM_GLIBC=0 build on the oracle's fast-math twin (the diagnostic path of rounds 3-4 still works); the opt-in
# long tests on the final code; kernel trace of the C5 + heat hour after the unread link fluxes went
mkdir -p gpurun_out
SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_fm.so SF3D_TEST_RTOL=1e-6 python -m pytest tests/test_gpu_sensitivity.py tests/test_gpu_fastmath.py -q -s > gpurun_out/r05_job06_fastmath_build_on_the_twin.log 2>&1
grep -E "kink window|link flow|C4 F20|passed|failed|skipped" gpurun_out/r05_job06_fastmath_build_on_the_twin.log | tail -8
SF3D_LONG_TESTS=1 SF3D_FULL_MATRIX=1 python -m pytest tests -q -m gpu -k "launch_modes or c3_f60 or three_hours or heat_seven or twelve or long" --durations=8 > gpurun_out/r05_job06_long_tests.log 2>&1
tail -14 gpurun_out/r05_job06_long_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_job06_C5_heat_trace -- python3 $GRAFT_REPO_ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/r05_job06_C5_heat_under_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/r05_job06_C5_heat_under_trace.err
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/r05_job06_C5_heat_trace -name "*kernel_stats.csv" | head -1) gpurun_out/r05_job06_C5_heat_kernel_stats.csv
head -16 gpurun_out/r05_job06_C5_heat_kernel_stats.csv | cut -c1-140
rm -rf gpurun_out/r05_job06_C5_heat_trace
