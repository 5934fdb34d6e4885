#!/bin/bash
# round 6 job 10: after the ranks' common decision for the resident loop (every rank or none; ranks sharing a GPU must fit on it together): the strip tests, C4 in eight
# strips on one GPU (the loop turns itself off there), the two-rank and eight-rank shared bench lines
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -k "resident or (sharded_run and c4f20h0) or (strip_local_build_runs and c4f20h0)" --durations=5 > gpurun_out/r06_job10_tests.txt 2>&1; tail -12 gpurun_out/r06_job10_tests.txt
for n in 2 8; do
  SF3D_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus $n --no-cpu-baseline --steps 6 --warmup 0 --reps 1 > gpurun_out/r06_d_bench_${n}ranks_shared.json 2> gpurun_out/r06_d_bench_${n}ranks_shared.err
  grep "exchange transport\|resident sweep loop off" gpurun_out/r06_d_bench_${n}ranks_shared.err | head -4 | cut -c1-260
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_d_bench_${n}ranks_shared.json').read().strip().splitlines()[-1]); print('$n ranks sharing', round(d['value'],2), d['roofline']['kernel'], d['exchange']['epochs'], d['parity'])"
done
