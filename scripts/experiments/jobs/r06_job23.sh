#!/bin/bash
# round 6 job 23: the paired pass on a strip with RECORD HAND-OVER (one launch, one exchange per pass): bit-identity tests on 2-3 strips (regular and
# masked grids), C4 in two strips against the oracle and against single sweeps, and an A/B of C4 in two / four ranks sharing the GPU
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "paired_sweep_on_strips or host_memory_windows or (sharded_run_matches_oracle and (c4f20h0 or ravone)) or strip_local" 2>&1 | tail -15 | tee gpurun_out/r06_job23_tests.txt
for rec in 1 0 1 0; do
  SF3D_PAIR_RECORDS=$rec SF3D_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 2 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 3 > gpurun_out/r06_job23_2ranks_rec$rec.json 2> gpurun_out/r06_job23_2ranks_rec$rec.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job23_2ranks_rec$rec.json').read().strip().splitlines()[-1])
print('records=$rec 2 ranks', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), d.get('exchange'), d['parity'])" 2>&1 | tail -2
done | tee gpurun_out/r06_job23_ab.txt
for rec in 1 0; do
  SF3D_PAIR_RECORDS=$rec SF3D_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus 4 --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 3 > gpurun_out/r06_job23_4ranks_rec$rec.json 2> gpurun_out/r06_job23_4ranks_rec$rec.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job23_4ranks_rec$rec.json').read().strip().splitlines()[-1])
print('records=$rec 4 ranks', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), d.get('exchange'), d['parity'])" 2>&1 | tail -2
done | tee -a gpurun_out/r06_job23_ab.txt
