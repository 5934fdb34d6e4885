#!/bin/bash
# round 6 job 28: the resident sweep loop on strips with its edge rows' records stored straight to their place (ResGrid::recFast) against the library of the
# commit before (a walk through the chunk's send list per put): C4E (512 x 64 x 20) in two strips = 128 blocks per rank, both ranks resident on the one GPU
# together; the resident strip tests
mkdir -p gpurun_out
# (the strip tests of the resident loop ran in the first call of this job: 5 passed; SF3D_RESIDENT_PR=2: two rows per block = 128 blocks per rank, so that both ranks fit on the GPU together)
for lib in new prev new prev new prev; do
  if [ $lib = prev ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_prev.so; else unset SF3D_PRODUCT_LIB; fi
  SF3D_RESIDENT_PR=2 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload C4E --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job28_$lib.json 2> gpurun_out/r06_job28_$lib.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job28_$lib.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('C4E in 2 strips, library $lib', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done | tee gpurun_out/r06_job28_ab.txt
