#!/bin/bash
# round 6 job 44: the driver's sequence on the last code commit of the round: the whole GPU suite (timed), smoke(), python bench.py
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r06_job44_suite.log 2>&1; tail -25 gpurun_out/r06_job44_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > gpurun_out/r06_job44_bench.json 2> gpurun_out/r06_job44_bench.err ) 2>&1 | tail -3
python3 -c "
import json
d=json.loads(open('gpurun_out/r06_job44_bench.json').read().strip().splitlines()[-1])
print(round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), round(d['roofline']['frac'],3), d['roofline'].get('traffic_is_of_this_build'), d['parity'])"
