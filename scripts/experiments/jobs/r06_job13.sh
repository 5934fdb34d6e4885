#!/bin/bash
# round 6 job 13: why did the multi-rank tests take minutes inside the suite (job 12)?  alone, with durations
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests/test_gpu_multirank.py -x -q --durations=15 -k "sharded_run_matches_oracle and not c4f20h0 and not ravone" ) > gpurun_out/r06_job13_multirank.txt 2>&1; tail -24 gpurun_out/r06_job13_multirank.txt
