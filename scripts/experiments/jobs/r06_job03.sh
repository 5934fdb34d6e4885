#!/bin/bash
# round 6 job 3: where does an iteration of the resident sweep loop spend its time?  (-DSF3D_RES_PROFILE=1: block 0 adds up the ticks of its phases)
mkdir -p gpurun_out
SF3D_PRODUCT_LIB=build_variants/libresprof.so timeout 300 python bench.py --workload C4E --no-cpu-baseline --steps 6 --warmup 1 --reps 1 --no-kernel-timing > gpurun_out/r06_job03_C4E.json 2> gpurun_out/r06_job03_C4E.err
grep "sf3d\]" gpurun_out/r06_job03_C4E.err | tail -12
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job03_C4E.json').read().strip().splitlines()[-1]); print('C4E value', d['value'])"
