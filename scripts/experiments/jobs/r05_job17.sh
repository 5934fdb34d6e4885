#!/bin/bash
# round 5 job 17: k_sweep_pair with the old iterate staged in LDS (up / down / lateral neighbours of the owned rows out of LDS, role-split loops):
# bit-identity tests first, then A/B against the library of the commit before (build_variants/libsf3d_hip_base.so), interleaved
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_golden.py -x -q -m gpu -k "launch_modes or paired or golden or sharded_run" 2>&1 | tail -5 | tee gpurun_out/r05_job17_tests.log
O=gpurun_out/r05_job17_ab.txt; : > $O
BASE=$PWD/build_variants/libsf3d_hip_base.so
for rep in 1 2 3; do
  for v in new base; do
    if [ $v = base ]; then export SF3D_PRODUCT_LIB=$BASE; else unset SF3D_PRODUCT_LIB; fi
    python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $v rep $rep', d['value'], d['roofline']['frac'], d['roofline']['avg_us'])" >> $O
  done
done
cat $O
