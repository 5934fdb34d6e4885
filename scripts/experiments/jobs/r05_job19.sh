#!/bin/bash
# round 5 job 19: k_sweep_pair without gathers (the old iterate of the patch AND one more ring of cells staged in LDS; b parked, z in
# registers) against the committed form (build_variants/libsf3d_hip_n.so): bit-identity tests, then interleaved timing
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_golden.py -x -q -m gpu -k "launch_modes or paired or golden or sharded_run" 2>&1 | tail -5 | tee gpurun_out/r05_job19_tests.log
O=gpurun_out/r05_job19_ab.txt; : > $O
for rep in 1 2 3; do
  for v in new n; do
    if [ $v = n ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_n.so; else unset SF3D_PRODUCT_LIB; fi
    python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $v rep $rep', d['value'], d['roofline']['frac'], d['roofline']['avg_us'])" >> $O
  done
done
cat $O
