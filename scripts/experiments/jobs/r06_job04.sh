#!/bin/bash
# round 6 job 4: the resident sweep loop on strips (k_sweep_resident<DIST>: tagged records through the neighbour's window), ranks sharing the GPU
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q -k "resident" > gpurun_out/r06_job04_tests.txt 2>&1; tail -25 gpurun_out/r06_job04_tests.txt
