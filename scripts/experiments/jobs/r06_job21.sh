#!/bin/bash
# round 6 job 21: the whole GPU suite on the code with the double-double balance sums and the fused post-solve part; C4E; the driver's default bench line
mkdir -p gpurun_out
( time timeout 1700 python -m pytest tests/ -x -q -m gpu --durations=10 ) > gpurun_out/r06_job21_suite.txt 2>&1; tail -20 gpurun_out/r06_job21_suite.txt
for fp in 0 1; do
  SF3D_RESIDENT_POST=$fp timeout 300 python bench.py --workload C4E --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 3 --no-kernel-timing > gpurun_out/r06_job21_C4E_post$fp.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job21_C4E_post$fp.json').read().strip().splitlines()[-1]); print('C4E SF3D_RESIDENT_POST=$fp', round(d['value'],2))"
done
