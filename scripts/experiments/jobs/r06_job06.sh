#!/bin/bash
# round 6 job 6: the driver's bench command with the parity self-check and the config-5 / config-3 legs on the line; how long it takes; the bench contract tests
mkdir -p gpurun_out
( time python bench.py > gpurun_out/r06_job06_bench.json 2> gpurun_out/r06_job06_bench.err ) 2> gpurun_out/r06_job06_time.txt; tail -4 gpurun_out/r06_job06_time.txt
grep "bench\]" gpurun_out/r06_job06_bench.err | tail -20
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job06_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'parity', d['parity'])
for k,v in (d.get('legs') or {}).items(): print(k, {kk: v[kk] for kk in ('value','elapsed_s','build_s','dominant_kernel') if isinstance(v, dict) and kk in v} if isinstance(v, dict) else v)
"
timeout 1500 python -m pytest tests/test_bench_contract.py -q -m gpu > gpurun_out/r06_job06_tests.txt 2>&1; tail -15 gpurun_out/r06_job06_tests.txt
