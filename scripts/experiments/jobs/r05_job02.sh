#!/bin/bash
# round 5 job 2: the pin tests of tests/test_gpu_sensitivity.py with the faithful set, then the whole GPU suite
mkdir -p gpurun_out
python -m pytest tests/test_gpu_sensitivity.py -x -q -s > gpurun_out/r05_job02_sensitivity.log 2>&1
grep -E "kink window|link flow sums|C4 F20 hour 0|passed|failed|error" gpurun_out/r05_job02_sensitivity.log | tail -12
python -m pytest tests -x -q -m gpu --durations=25 > gpurun_out/r05_job02_suite.log 2>&1
tail -45 gpurun_out/r05_job02_suite.log
