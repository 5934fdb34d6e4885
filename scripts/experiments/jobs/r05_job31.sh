#!/bin/bash
# round 5 job 31: after the per-W occupancy targets of the paired sweeps (W = 14: four waves per SIMD, W = 8: five): every W in the bit-identity tests
mkdir -p gpurun_out
SF3D_FULL_MATRIX=1 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -x -q -m gpu -k "launch_modes or masked or awkward or paired" 2>&1 | tail -4 | tee gpurun_out/r05_job31_tests.log
for W in 14 10; do SF3D_PAIR_W=$W python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 W=$W', d['value'], d['roofline']['avg_us'])"; done | tee gpurun_out/r05_job31_W14.txt
