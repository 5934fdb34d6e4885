#!/bin/bash
# round 6 job 20: the balance sums as double-doubles in every path (k_post, k_restore, the separate decision kernels, the RCCL gather) and k_post's work fused into
# the resident launch: targeted tests, then C4E with the fused post off / on (phase timers), C4 for the cost of the double-double sums on the headline
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_resident.py tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_kernel_resources.py -q -x > gpurun_out/r06_job20_tests_a.txt 2>&1; tail -4 gpurun_out/r06_job20_tests_a.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q -x -k "launch_modes_are or c4_hour0 or mass_conservation" > gpurun_out/r06_job20_tests_b.txt 2>&1; tail -4 gpurun_out/r06_job20_tests_b.txt
timeout 1500 python -m pytest tests/test_gpu_multirank.py -q -x -k "resident or rccl or (sharded_run and not c4f20h0 and not ravone) or host_memory" > gpurun_out/r06_job20_tests_c.txt 2>&1; tail -4 gpurun_out/r06_job20_tests_c.txt
for fp in 0 1 0 1; do
  SF3D_RESIDENT_POST=$fp timeout 300 python bench.py --workload C4E --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 3 --no-kernel-timing > gpurun_out/r06_job20_C4E_post$fp.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job20_C4E_post$fp.json').read().strip().splitlines()[-1]); print('C4E SF3D_RESIDENT_POST=$fp', round(d['value'],2), d['config']['work'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 3 --time-all-kernels > gpurun_out/r06_job20_C4.json 2>/dev/null
python3 -c "
import json
a=json.loads(open('gpurun_out/r06_job20_C4.json').read().strip().splitlines()[-1]); k=a['roofline']['kernels']
print('C4', round(a['value'],2), {n:round(v['total_ms']/max(v['launches'],1)*1e3,1) for n,v in k.items() if v['launches']})"
