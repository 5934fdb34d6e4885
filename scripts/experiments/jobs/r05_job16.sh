#!/bin/bash
# round 5 job 16: does the code size of the node kernels matter?  k_props<0,false> is 10 696 instructions with the C library's pow / exp inlined
# whole (4 647 with their edge-case helpers out of line: -DSF3D_GL_COLD='__device__ __noinline__', build_variants/libsf3d_hip_cold.so).
# A/B, interleaved: headline C4, C5 hour 0, C5 + heat; then bit-identity of the two builds on the kink pin tests.
mkdir -p gpurun_out
O=gpurun_out/r05_job16_ab.txt; : > $O
COLD=$PWD/build_variants/libsf3d_hip_cold.so
for rep in 1 2 3; do
  for v in default cold; do
    if [ $v = cold ]; then export SF3D_PRODUCT_LIB=$COLD; else unset SF3D_PRODUCT_LIB; fi
    python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $v rep $rep', d['value'], d['roofline']['frac'])" >> $O
  done
done
for v in default cold; do
  if [ $v = cold ]; then export SF3D_PRODUCT_LIB=$COLD; else unset SF3D_PRODUCT_LIB; fi
  python bench.py --workload C5 --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 $v', d['value'])" >> $O
  python bench.py --workload C5 --heat --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 heat $v', d['value'])" >> $O
done
cat $O
export SF3D_PRODUCT_LIB=$COLD
python -m pytest tests/test_gpu_fastmath.py tests/test_gpu_sensitivity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r05_job16_cold_tests.log
