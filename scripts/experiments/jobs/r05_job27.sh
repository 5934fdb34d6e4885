#!/bin/bash
# round 5 job 27: ablation - k_assemble's soil rows with the ten gathers of the neighbours' K replaced by a register value (WRONG results,
# a one-line local patch of assemble_soil_rows, `kj[t] = Ki * 1.0000001` instead of `v.K[j[t]]`, built as build_variants/libsf3d_hip_ablate.so and
# not kept in the tree): the ceiling of what staging K in LDS could give that kernel
mkdir -p gpurun_out
O=gpurun_out/r05_job27_assemble_gather_ablation.txt; : > $O
for v in product ablate product ablate; do
  if [ $v = ablate ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_hip_ablate.so; else unset SF3D_PRODUCT_LIB; fi
  timeout 300 python bench.py --no-cpu-baseline --no-f60 --steps 1 --warmup 0 --reps 1 --time-all-kernels 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('$v', {n:(v['launches'], round(1e3*v['total_ms']/max(1,v['launches']),1)) for n,v in k.items()})" >> $O
done
cat $O
