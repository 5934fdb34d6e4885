#!/bin/bash
# round 6 job 25: record hand-over with direct record addressing on regular strips (no list walk between a value and its store, the address of the polled
# record known before the step): the A/B of jobs 23 / 24 again, the bit-identity tests of the paired pass on strips, bench.py's first-contact check of it
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "paired_sweep_on_strips or (sharded_run_matches_oracle and c4f20h0 and 2-)" 2>&1 | tail -5 | tee gpurun_out/r06_job25_tests.txt
for cfg in "C4Q 2" "C4 2" "C4H 4"; do set -- $cfg
for rec in 1 0 1 0; do
  SF3D_RESIDENT_SWEEP=0 SF3D_PAIR_SWEEP=1 SF3D_PAIR_RECORDS=$rec SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --workload $1 --gpus $2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 5 > gpurun_out/r06_job25_$1_$2_rec$rec.json 2> gpurun_out/r06_job25_$1_$2_rec$rec.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job25_$1_$2_rec$rec.json').read().strip().splitlines()[-1])
e=d.get('exchange') or {}
print('$1 in $2 strips records=$rec', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), 'epochs', e.get('epochs'), 'mean wait', e.get('mean_wait_us'), list(d['parity'].values())[-1][:40])" 2>&1 | tail -2
done; done | tee gpurun_out/r06_job25_ab.txt
SF3D_BENCH_PAIR_CONTACT=1 SF3D_PAIR_RECORDS=1 SF3D_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 1 > gpurun_out/r06_job25_contact.json 2> gpurun_out/r06_job25_contact.err; grep -n "record hand-over\|parity" gpurun_out/r06_job25_contact.err | head -5
