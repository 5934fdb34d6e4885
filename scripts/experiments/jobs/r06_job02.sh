#!/bin/bash
# round 6 job 2: first contact of the resident-coefficient sweep loop (k_sweep_resident): bitwise against the single sweeps on nine grids,
# against the oracle through the runoff regime, then one of eight strips of C4 as a grid of its own with the loop off / on
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py -x -q > gpurun_out/r06_job02_tests.txt 2>&1; tail -15 gpurun_out/r06_job02_tests.txt
for r in 0 1; do
  SF3D_RESIDENT_SWEEP=$r timeout 300 python bench.py --workload C4E --no-cpu-baseline --steps 12 --warmup 1 > gpurun_out/r06_job02_C4E_res$r.json 2> gpurun_out/r06_job02_C4E_res$r.err
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r06_job02_C4E_res$r.json').read().strip().splitlines()[-1])
print('C4E resident=$r value', d['value'], 'dominant', d['roofline']['kernel'], d['roofline']['avg_us'], 'work', d['roofline']['step']['work'])"
  tail -3 gpurun_out/r06_job02_C4E_res$r.err
done
