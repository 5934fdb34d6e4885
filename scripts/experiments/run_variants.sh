#!/bin/bash
# usage (on the GPU box): bash scripts/experiments/run_variants.sh A0 A1 ...   -> per-kernel event-timed averages of each build_variants/lib<NAME>.so
mkdir -p gpurun_out/variants
run() { SF3D_PRODUCT_LIB=$PWD/build_variants/lib$1.so timeout 300 python bench.py --no-cpu-baseline --time-all-kernels --reps 1 --steps 2 --warmup 0 2>gpurun_out/variants/$1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=(d['roofline'] or {}).get('kernels',{})
print('$1', round(d['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items() if v['launches']})"; }
for L in "$@"; do run $L; done
