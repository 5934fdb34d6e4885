#!/bin/bash
# round 6 job 29: the resident loop's norm between the ranks as tagged records in every rank's window (DistWindow::rrec: every block of every rank adds the
# world's partial norms itself) against the library of the commit before (block 0: mailbox all-gather, then a record to the other blocks); C4E in two strips,
# SF3D_RESIDENT_PR=2 (both ranks resident on the one GPU together); the resident strip tests (2 and 4 ranks)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "resident_sweep_loop_on_strips" 2>&1 | tail -3 | tee gpurun_out/r06_job29_tests.txt
for lib in new prev new prev new prev; do
  if [ $lib = prev ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_prev.so; else unset SF3D_PRODUCT_LIB; fi
  SF3D_RESIDENT_PR=2 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload C4E --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job29_$lib.json 2> gpurun_out/r06_job29_$lib.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job29_$lib.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('C4E in 2 strips, library $lib', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done | tee gpurun_out/r06_job29_ab.txt
