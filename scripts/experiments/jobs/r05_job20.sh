#!/bin/bash
# round 5 job 20: k_sweep_pair_masked without gathers and without the index ring (old iterate of patch + outer ring staged in LDS, node
# indices two layers ahead) against the committed form (build_variants/libsf3d_hip_o.so = 00ddfb8): bit-identity, then the Ravone project
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_ravone_project.py tests/test_gpu_golden.py -x -q -m gpu -k "masked or ravone or paired or golden or launch_modes_are" 2>&1 | tail -5 | tee gpurun_out/r05_job20_tests.log
O=gpurun_out/r05_job20_ab.txt; : > $O
for rep in 1 2; do
  for v in new o; do
    if [ $v = o ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_o.so; else unset SF3D_PRODUCT_LIB; fi
    python bench.py --workload C5 --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 $v rep $rep', d['value'], d['roofline']['frac'], d['roofline']['avg_us'])" >> $O
  done
done
cat $O
