#!/bin/bash
# round 6 job 33: the rank's norm stored into the windows BEFORE the halo poll (block 0): resident strip tests, the phases again (libresprof.so), and the A/B
# against the library of the commit before (build_variants/libsf3d_prev.so); C4E in two strips, SF3D_RESIDENT_PR=2, both ranks on the one GPU
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "resident_sweep_loop_on_strips" 2>&1 | tail -3 | tee gpurun_out/r06_job33_tests.txt
SF3D_PRODUCT_LIB=$PWD/build_variants/libresprof.so SF3D_RESIDENT_PR=2 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload C4E --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 3 > gpurun_out/r06_job33_prof.json 2> gpurun_out/r06_job33_prof.err; grep "sf3d\]" gpurun_out/r06_job33_prof.err | tail -16 | tee gpurun_out/r06_job33_phases.txt
for lib in new prev new prev new prev; do
  if [ $lib = prev ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_prev.so; else unset SF3D_PRODUCT_LIB; fi
  SF3D_RESIDENT_PR=2 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload C4E --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job33_$lib.json 2> gpurun_out/r06_job33_$lib.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job33_$lib.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('C4E in 2 strips, library $lib', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done | tee gpurun_out/r06_job33_ab.txt
