#!/bin/bash
# round 6 job 8: inputs of the scale model on one box - C4 and its strips as grids of their own (C4H / C4Q / C4E, the last with the resident sweep loop),
# then C4 in two and in eight strips with the ranks sharing the GPU (functional: exchange epochs, transport, parity keys of the line)
mkdir -p gpurun_out
for w in C4 C4H C4Q C4E; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 > gpurun_out/r06_d_bench_$w.json 2> gpurun_out/r06_d_bench_$w.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_d_bench_$w.json').read().strip().splitlines()[-1]); print('$w', round(d['value'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1))"
done
SF3D_RESIDENT_SWEEP=0 timeout 600 python bench.py --workload C4E --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 > gpurun_out/r06_d_bench_C4E_single_sweeps.json 2>/dev/null
for n in 2 8; do
  SF3D_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus $n --no-cpu-baseline --steps 6 --warmup 0 --reps 1 > gpurun_out/r06_d_bench_${n}ranks_shared.json 2> gpurun_out/r06_d_bench_${n}ranks_shared.err
  grep "exchange transport\|parity" gpurun_out/r06_d_bench_${n}ranks_shared.err | head -12
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_d_bench_${n}ranks_shared.json').read().strip().splitlines()[-1]); print('$n ranks sharing', round(d['value'],2), d['roofline']['kernel'], d['exchange']['epochs'], d['parity'])"
done
