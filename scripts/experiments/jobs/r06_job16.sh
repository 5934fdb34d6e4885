#!/bin/bash
# round 6 job 16: the rows of the resident loop software-pipelined (reads of row k + 1 before the arithmetic of row k): tests, C4E off / on, phase timers
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py -q > gpurun_out/r06_job16_tests_a.txt 2>&1; tail -3 gpurun_out/r06_job16_tests_a.txt
timeout 1200 python -m pytest tests/test_gpu_multirank.py -q -k "resident" > gpurun_out/r06_job16_tests_b.txt 2>&1; tail -3 gpurun_out/r06_job16_tests_b.txt
bash scripts/experiments/jobs/r06_job02.sh 2>&1 | grep "C4E resident"
bash scripts/experiments/jobs/r06_job03.sh 2>&1 | tail -9
