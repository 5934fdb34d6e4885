#!/bin/bash
# round 6 job 26: record hand-over on MASKED grids with per-node tables (DistView::recGet / recPut): bit-identity tests (holes, project window, the Ravone
# project in four strips) and the A/B on config 5 hour 0 in two and four ranks sharing the GPU (SF3D_PAIR_RECORDS=1 forces it on masked grids, =0 two launches)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "paired_sweep_on_strips or (sharded_run_matches_oracle and ravone)" 2>&1 | tail -5 | tee gpurun_out/r06_job26_tests.txt
for cfg in "C5 2" "C5 4"; do set -- $cfg
for rec in 1 0 1 0; do
  SF3D_PAIR_RECORDS=$rec SF3D_BENCH_SHARE_GPU=1 timeout 900 python bench.py --workload $1 --gpus $2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 1 --warmup 0 --reps 1 > gpurun_out/r06_job26_$1_$2_rec$rec.json 2> gpurun_out/r06_job26_$1_$2_rec$rec.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job26_$1_$2_rec$rec.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('$1 in $2 strips records=$rec', round(d['value'],4), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done; done | tee gpurun_out/r06_job26_ab.txt
