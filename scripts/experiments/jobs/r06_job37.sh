#!/bin/bash
# round 6 job 37: the resident launch stores the halo of its result itself (from the tiles: no plain puts of the edge rows, no k_halo_copy<1> behind the
# approximation): the multi-rank tests (whole file), then C4E in two strips (SF3D_RESIDENT_PR=2) against the library of the commit before
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu ) 2>&1 | tail -6 | tee gpurun_out/r06_job37_tests.txt
for lib in new prev new prev new prev; do
  if [ $lib = prev ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_prev.so; else unset SF3D_PRODUCT_LIB; fi
  SF3D_RESIDENT_PR=2 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload C4E --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job37_$lib.json 2> gpurun_out/r06_job37_$lib.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job37_$lib.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('C4E in 2 strips, library $lib', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done | tee gpurun_out/r06_job37_ab.txt
