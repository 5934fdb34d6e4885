#!/bin/bash
# round 4, GPU job 22: the kink-window test for its whole two hours on the final code (double-double norm, reworked heat kernels untouched by it)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job22
mkdir -p $OUT
cd $ROOT
( time SF3D_LONG_TESTS=1 python -m pytest tests/test_gpu_sensitivity.py -q -m gpu -s -k "kink" ) > $OUT/sens.log 2>&1; echo "rc=$?" >> $OUT/sens.log
grep -v "^$" $OUT/sens.log | tail -30
