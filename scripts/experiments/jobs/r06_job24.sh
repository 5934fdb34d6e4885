#!/bin/bash
# round 6 job 24: record hand-over A/B where the ranks sharing the GPU do NOT compete for CUs: C4Q (512 x 128 x 20) in two strips and C4H in four
# (64 rows = 64 blocks of the paired pass per rank: every block has a CU to itself), paired pass forced, resident loop off
mkdir -p gpurun_out
for cfg in "C4Q 2" "C4H 4" "C4H 2"; do set -- $cfg
for rec in 1 0 1 0; do
  SF3D_RESIDENT_SWEEP=0 SF3D_PAIR_SWEEP=1 SF3D_PAIR_RECORDS=$rec SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload $1 --gpus $2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job24_$1_$2_rec$rec.json 2> gpurun_out/r06_job24_$1_$2_rec$rec.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job24_$1_$2_rec$rec.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('$1 in $2 strips records=$rec', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done; done | tee gpurun_out/r06_job24_ab.txt
