#!/bin/bash
# round 5 job 26: the opt-in long tests (launch-mode matrix incl. W = 14 and no-graph modes, C3 F60 / C2 F60 three hours, full-size project steps)
# with the gather-free paired sweeps
mkdir -p gpurun_out
SF3D_LONG_TESTS=1 SF3D_FULL_MATRIX=1 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_flows.py tests/test_gpu_ravone_project.py -x -q -m gpu --durations=8 2>&1 | tail -14 | tee gpurun_out/r05_job26_long_tests.log
