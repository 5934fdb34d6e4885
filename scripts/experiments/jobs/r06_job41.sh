#!/bin/bash
# round 6 job 41: the driver's N = 8 / 4 command lines as far as one GPU allows (SF3D_BENCH_SHARE_GPU=1: gloo instead of RCCL, every rank on GPU 0): the first-contact
# checks (resident loop, then the paired pass's record hand-over on C2 in N strips), the episode, the parity keys
mkdir -p gpurun_out
for n in 8 4; do
  SF3D_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus $n --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 0 --reps 1 > gpurun_out/r06_job41_$n.json 2> gpurun_out/r06_job41_$n.err
  echo "exit $?"; grep -n "rank 0:" gpurun_out/r06_job41_$n.err | cut -c1-260 | head -8
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job41_$n.json').read().strip().splitlines()[-1])
print('$n ranks', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), (d.get('exchange') or {}).get('epochs'), d['parity'])" 2>&1 | tail -2
done
