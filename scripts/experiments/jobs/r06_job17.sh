#!/bin/bash
# round 6 job 17: the driver's bench commands on the final code (default: with legs and CPU baseline; --steps 20 --warmup 5), smoke()
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py > gpurun_out/r06_z_bench.json 2> gpurun_out/r06_z_bench.err ) 2>&1 | tail -3
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06_z_bench_driver_style_steps20_warmup5.json 2> gpurun_out/r06_z_bench_driver.err ) 2>&1 | tail -3
python3 -c "
import json
for f in ('gpurun_out/r06_z_bench.json','gpurun_out/r06_z_bench_driver_style_steps20_warmup5.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), round(d['roofline']['frac'],3), d['roofline']['traffic_is_of_this_build'], 'cpu', d['cpu_baseline']['value'] if d['cpu_baseline'] else None, {k:round(v['value'],4) for k,v in (d.get('legs') or {}).items() if isinstance(v,dict)}, 'f60', round(d['f60_hour0']['value'],2) if d.get('f60_hour0') else None)"
