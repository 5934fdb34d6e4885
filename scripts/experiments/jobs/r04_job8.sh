#!/bin/bash
# round 4, GPU job 8: rocprofv3 kernel trace + PMC passes of C5 + heat (one hour); the twin-based flow-sum tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job8
mkdir -p $OUT
cd $ROOT
PMC_STEPS=1 bash scripts/profile_gpu.sh r04_job8/prof --workload C5 --heat --steps 1 --warmup 0 --reps 1 > $OUT/profile_gpu.log 2>&1
python -m pytest tests/test_gpu_sensitivity.py -q -s -k "flow_sums or twin_is" > $OUT/sens.log 2>&1; echo "rc=$?" >> $OUT/sens.log
tail -n 32 $OUT/profile_gpu.log
tail -n 8 $OUT/sens.log
