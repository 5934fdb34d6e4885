#!/bin/bash
# round 6 job 18: fewer, fatter blocks for the node kernels on a strip-sized grid?  SF3D_CHUNKS_PER_WAVE = 1 (today: one chunk per wave up to 2 048 blocks) / 2 / 4 / 8
# at C4E (every kernel timed with HIP events) and at C4
mkdir -p gpurun_out
for w in C4E C4; do for c in 1 2 4 8 1; do
  SF3D_CHUNKS_PER_WAVE=$c timeout 300 python bench.py --workload $w --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 3 > gpurun_out/r06_job18_${w}_$c.json 2>/dev/null
  SF3D_CHUNKS_PER_WAVE=$c timeout 300 python bench.py --workload $w --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 1 --time-all-kernels > gpurun_out/r06_job18_${w}_${c}_all.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job18_${w}_$c.json').read().strip().splitlines()[-1])
a=json.loads(open('gpurun_out/r06_job18_${w}_${c}_all.json').read().strip().splitlines()[-1])
k=a['roofline']['kernels']
print('$w chunks/wave $c value', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items() if v['launches']})"
done; done
