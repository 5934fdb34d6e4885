#!/bin/bash
# round 6 job 7: after the register diet of the resident loop (no scratch in any shape), the 10 s exchange bound, the libm probe: resident tests (one GPU, strips), the
# launch-mode matrix (with the one-link-per-lane rows), C4E off / on + the phase timers, then the rows experiment on C4: SF3D_ASM_LINKS=0 / 1 with every kernel timed
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_resident.py tests/test_gpu_fastmath.py -q > gpurun_out/r06_job07_tests_a.txt 2>&1; tail -4 gpurun_out/r06_job07_tests_a.txt
timeout 1200 python -m pytest tests/test_gpu_multirank.py -q -k "resident" > gpurun_out/r06_job07_tests_b.txt 2>&1; tail -4 gpurun_out/r06_job07_tests_b.txt
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -q -k "launch_modes_are" > gpurun_out/r06_job07_tests_c.txt 2>&1; tail -4 gpurun_out/r06_job07_tests_c.txt
bash scripts/experiments/jobs/r06_job02.sh 2>&1 | grep "C4E resident"
bash scripts/experiments/jobs/r06_job03.sh 2>&1 | tail -9
for a in 0 1 0 1; do
  SF3D_ASM_LINKS=$a timeout 300 python bench.py --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 3 --time-all-kernels > gpurun_out/r06_job07_links$a.json 2> gpurun_out/r06_job07_links$a.err
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job07_links$a.json').read().strip().splitlines()[-1])
k=d['roofline']['kernels']
print('SF3D_ASM_LINKS=$a value', round(d['value'],2), {n:(v['launches'], round(v['total_ms']/max(v['launches'],1)*1e3,1)) for n,v in k.items() if v['launches']})"
done
