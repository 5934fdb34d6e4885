#!/bin/bash
# round 5 job 18: three forms of the staged-old-iterate k_sweep_pair - n (committed 7150d6c), asm0 (the prefetched value written to LDS in
# straight-line code), asm1 (the prefetch requested and waited for in inline assembly) - bit-identity of each, then interleaved timing
mkdir -p gpurun_out
O=gpurun_out/r05_job18_ab.txt; : > $O
for v in asm0 asm1; do
  SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_$v.so python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -x -q -m gpu -k "launch_modes_are or paired" 2>&1 | tail -1 | sed "s/^/$v: /" >> $O
done
for rep in 1 2 3; do
  for v in n asm0 asm1; do
    SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_$v.so python bench.py --no-cpu-baseline --no-f60 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 $v rep $rep', d['value'], d['roofline']['frac'], d['roofline']['avg_us'])" >> $O
  done
done
cat $O
