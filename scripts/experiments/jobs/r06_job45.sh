#!/bin/bash
# round 6 job 45: the round's records on the FINAL code (record hand-over on strips included): rocprofv3 kernel trace + PMC of
# the driver's command, of C4E and of C5; smoke(); the driver's bench lines; the stored episode checks; C4 in two strips on the shared GPU with its parity keys
mkdir -p gpurun_out
bash scripts/profile_gpu.sh r06_z --steps 20 --warmup 5 --no-extra-legs > gpurun_out/r06_job45_profile_C4.txt 2>&1; tail -12 gpurun_out/r06_job45_profile_C4.txt
bash scripts/profile_gpu.sh r06_z_C4E --workload C4E --steps 6 --warmup 1 > gpurun_out/r06_job45_profile_C4E.txt 2>&1; tail -10 gpurun_out/r06_job45_profile_C4E.txt
PMC_STEPS=1 bash scripts/profile_gpu.sh r06_z_C5 --workload C5 --steps 1 --warmup 0 --reps 1 > gpurun_out/r06_job45_profile_C5.txt 2>&1; tail -10 gpurun_out/r06_job45_profile_C5.txt
find gpurun_out/r06_z gpurun_out/r06_z_C4E gpurun_out/r06_z_C5 -name "*.csv" -size +8M -delete
for t in r06_z r06_z_C4E r06_z_C5; do cp gpurun_out/$t/summary.json profiles/${t}_kernel_summary.json; done; cp gpurun_out/r06_z_C5/summary.json profiles/r06_z_C5_pmc_traffic.json      # (so that the bench lines below quote THIS build's traffic)
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 --reps 1 --write-episode-checks gpurun_out/c4_f20_episode_checks.json > /dev/null 2> gpurun_out/r06_job45_checks.err; tail -1 gpurun_out/r06_job45_checks.err
cp gpurun_out/c4_f20_episode_checks.json tests/golden/c4_f20_episode_checks.json
( time python bench.py > gpurun_out/r06_z_bench.json 2> gpurun_out/r06_z_bench.err ) 2>&1 | tail -3
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r06_z_bench_driver_style_steps20_warmup5.json 2> gpurun_out/r06_z_bench_driver.err ) 2>&1 | tail -3
for w in C4E; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-f60 --no-extra-legs --steps 6 --warmup 1 > gpurun_out/r06_z_bench_$w.json 2>/dev/null; done
SF3D_BENCH_SHARE_GPU=1 timeout 1200 python bench.py --gpus 2 --no-cpu-baseline --steps 6 --warmup 0 --reps 1 > gpurun_out/r06_z_bench_2ranks_shared.json 2> gpurun_out/r06_z_bench_2ranks_shared.err
python3 -c "
import json
for f in ('gpurun_out/r06_z_bench.json','gpurun_out/r06_z_bench_driver_style_steps20_warmup5.json','gpurun_out/r06_z_bench_C4E.json','gpurun_out/r06_z_bench_2ranks_shared.json'):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value'],2), 'ms_per_step', round(d['ms_per_step'],2), d['roofline']['kernel'], round(d['roofline']['avg_us'],1), round(d['roofline']['frac'],3), d['roofline'].get('traffic_is_of_this_build'), d['parity'], {k:round(v['value'],4) for k,v in (d.get('legs') or {}).items() if isinstance(v,dict)})"
