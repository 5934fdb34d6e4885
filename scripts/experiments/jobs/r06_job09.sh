#!/bin/bash
# round 6 job 9: C4 in eight strips with the resident sweep loop on every rank, all on one GPU (functional: eight persistent kernels take turns) - the bench line
# with its parity keys, and the strip test against the single sweeps (SF3D_LONG_TESTS=1)
mkdir -p gpurun_out
SF3D_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus 8 --no-cpu-baseline --steps 6 --warmup 0 --reps 1 > gpurun_out/r06_d_bench_8ranks_shared.json 2> gpurun_out/r06_d_bench_8ranks_shared.err
grep "exchange transport" gpurun_out/r06_d_bench_8ranks_shared.err | head -3 | cut -c1-300
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_d_bench_8ranks_shared.json').read().strip().splitlines()[-1]); print('8 ranks sharing', round(d['value'],2), d['roofline']['kernel'], d['exchange']['epochs'], d['parity'])"
( time SF3D_LONG_TESTS=1 timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -k "resident and c4f20h0" ) > gpurun_out/r06_job09_c4_8strips_resident.txt 2>&1; tail -6 gpurun_out/r06_job09_c4_8strips_resident.txt
