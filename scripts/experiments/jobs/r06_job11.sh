#!/bin/bash
# round 6 job 11: the end of a poll group - the device publishes the control block into the host's pinned copy and the host spins on the poll's number (k_publish)
# against the D2H copy + stream synchronisation of rounds 1-5 (SF3D_HOST_POLL=copy): parity subset, then C4E / C4 / C3-F60-hour-0 / C2 both ways, interleaved
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_resident.py tests/test_gpu_golden.py -q -x > gpurun_out/r06_job11_tests.txt 2>&1; tail -4 gpurun_out/r06_job11_tests.txt
for rep in 1 2; do for hp in copy publish; do
  for w in C4E C4 C2; do
    SF3D_HOST_POLL=$hp timeout 600 python bench.py --workload $w --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --no-kernel-timing > gpurun_out/r06_job11_${w}_$hp.json 2>/dev/null
    python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job11_${w}_$hp.json').read().strip().splitlines()[-1]); print('$w SF3D_HOST_POLL=$hp', round(d['value'],2))"
  done
  SF3D_HOST_POLL=$hp timeout 600 python bench.py --workload C3 --forcing F60 --no-cpu-baseline --steps 1 --warmup 0 --reps 3 --no-kernel-timing > gpurun_out/r06_job11_C3F60_$hp.json 2>/dev/null
  python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job11_C3F60_$hp.json').read().strip().splitlines()[-1]); print('C3 F60 hour 0 SF3D_HOST_POLL=$hp', round(d['value'],4), d['config']['work']['accepted'])"
done; done
