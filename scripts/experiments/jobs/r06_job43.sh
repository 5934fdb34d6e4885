#!/bin/bash
# round 6 job 43: the masked paired pass on strips - after the first pass the neighbours' previous iterate staged into LDS from the window (blocks at the cut: bit 7 of
# the patch's depth byte) instead of gathered in front of the rows' arithmetic: tests (holes, project window, the Ravone project in four strips), then config 5 hour 0
# in two / four ranks sharing the GPU against the library of the commit before
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "paired_sweep_on_strips or (sharded_run_matches_oracle and ravone) or strip_local" 2>&1 | tail -4 | tee gpurun_out/r06_job43_tests.txt
for cfg in "C5 2" "C5 4"; do set -- $cfg
for lib in new prev new prev; do
  if [ $lib = prev ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_prev.so; else unset SF3D_PRODUCT_LIB; fi
  SF3D_BENCH_SHARE_GPU=1 timeout 900 python bench.py --workload $1 --gpus $2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 1 --warmup 0 --reps 1 > gpurun_out/r06_job43_$1_$2_$lib.json 2> gpurun_out/r06_job43_$1_$2_$lib.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job43_$1_$2_$lib.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('$1 in $2 strips, library $lib', round(d['value'],4), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done; done | tee gpurun_out/r06_job43_ab.txt
