#!/bin/bash
# round 6 job 42: record hand-over, passes after the first: the neighbours' previous iterate staged into LDS from the window by the threads whose cell it is (one
# layer ahead, like every other value of the old iterate) instead of gathered per node and link in front of the edge rows' arithmetic.  Tests of the paired
# pass on strips, then the A/B against the library of the commit before: C4 in two strips, C4H in four, C4Q in two (paired pass forced, resident loop off)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "paired_sweep_on_strips or (sharded_run_matches_oracle and c4f20h0 and 2-) or host_memory" 2>&1 | tail -3 | tee gpurun_out/r06_job42_tests.txt
for cfg in "C4 2" "C4H 4" "C4Q 2"; do set -- $cfg
for lib in new prev new prev; do
  if [ $lib = prev ]; then export SF3D_PRODUCT_LIB=$PWD/build_variants/libsf3d_prev.so; else unset SF3D_PRODUCT_LIB; fi
  SF3D_RESIDENT_SWEEP=0 SF3D_PAIR_SWEEP=1 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload $1 --gpus $2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job42_$1_$2_$lib.json 2> gpurun_out/r06_job42_$1_$2_$lib.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job42_$1_$2_$lib.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('$1 in $2 strips, library $lib', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done; done | tee gpurun_out/r06_job42_ab.txt
