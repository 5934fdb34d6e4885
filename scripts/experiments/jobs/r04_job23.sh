#!/bin/bash
# round 4, GPU job 23: the opt-in tests on the final code - full launch-mode matrix, the long runs (1 000 restore-best steps, three hours of C2 F60,
# twelve hours of heat); the kink window's two hours were job 22
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job23
mkdir -p $OUT
cd $ROOT
( time SF3D_LONG_TESTS=1 SF3D_FULL_MATRIX=1 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_flows.py tests/test_gpu_heat.py -q -m gpu --durations=12 -k "launch_modes or long or c3_f60 or three_hours or half_day or seven_hours or hours" ) > $OUT/long.log 2>&1; echo "rc=$?" >> $OUT/long.log
grep -v "^$" $OUT/long.log | tail -25
