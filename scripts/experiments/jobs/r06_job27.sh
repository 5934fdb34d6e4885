#!/bin/bash
# round 6 job 27: masked strips with the per-node put / get tables in BOTH forms of the paired pass: tests, then config 5 hour 0 in two / four ranks sharing the
# GPU with the default (record hand-over) and with SF3D_PAIR_RECORDS=0 (two launches, plain puts through the table)
mkdir -p gpurun_out
SF3D_PAIR_RECORDS=0 timeout 1200 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "paired_sweep_on_strips and (holes or projwin)" 2>&1 | tail -3 | tee gpurun_out/r06_job27_tests.txt
for cfg in "C5 2" "C5 4"; do set -- $cfg
for rec in default 0 default 0; do
  if [ $rec = default ]; then unset SF3D_PAIR_RECORDS; else export SF3D_PAIR_RECORDS=$rec; fi
  SF3D_BENCH_SHARE_GPU=1 timeout 900 python bench.py --workload $1 --gpus $2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 1 --warmup 0 --reps 1 > gpurun_out/r06_job27_$1_$2_rec$rec.json 2> gpurun_out/r06_job27_$1_$2_rec$rec.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job27_$1_$2_rec$rec.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('$1 in $2 strips records=$rec', round(d['value'],4), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done; done | tee gpurun_out/r06_job27_ab.txt
