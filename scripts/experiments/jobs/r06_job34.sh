#!/bin/bash
# round 6 job 34: after the generic (list-walk) record path left the kernels: the multi-rank tests, whole file
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu ) 2>&1 | tail -8 | tee gpurun_out/r06_job34_tests.txt
