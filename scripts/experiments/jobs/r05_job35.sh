#!/bin/bash
# round 5 job 35: per-node codes of the paired sweeps as the 32 lateral bits the kernels decode (4 B per node instead of 8) against the commit
# before (build_variants/libsf3d_hip_head.so): bit-identity, then the Ravone project (the masked pass loads a code per node) and C4, interleaved
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_ravone_project.py -x -q -m gpu -k "masked or awkward or paired or launch_modes_are or window0" 2>&1 | tail -3 | tee gpurun_out/r05_job35_tests.log
O=gpurun_out/r05_job35_ab.txt; : > $O
for rep in 1 2; do
  for v in new head; do
    if [ $v = head ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_head.so; else unset SF3D_PRODUCT_LIB; fi
    python bench.py --workload C5 --no-cpu-baseline --steps 1 --warmup 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 $v rep $rep', d['value'], d['roofline']['avg_us'])" >> $O
    python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $v rep $rep', d['value'], d['roofline']['avg_us'])" >> $O
  done
done
cat $O
