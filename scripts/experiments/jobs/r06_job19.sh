#!/bin/bash
# round 6 job 19: how do the node kernels' launch times scale with the grid?  (HIP events, every kernel timed: C2 41 K nodes ... C4 5.2 M)
mkdir -p gpurun_out
for w in C2 C4E C4Q C4H C4; do
  SF3D_RESIDENT_SWEEP=0 timeout 300 python bench.py --workload $w --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 1 --time-all-kernels > gpurun_out/r06_job19_$w.json 2>/dev/null
  python3 -c "
import json
a=json.loads(open('gpurun_out/r06_job19_$w.json').read().strip().splitlines()[-1])
k=a['roofline']['kernels']
print('$w', a['config']['nodes'], {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items() if v['launches']})"
done
