#!/bin/bash
# round 5 job 24: BASELINE config 5 for the record with the gather-free masked pass - the Ravone project, F20: hour 0 with coupled heat, the
# 6-hour episode water only and with coupled heat
mkdir -p gpurun_out
python bench.py --workload C5 --heat --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/r05_job24_C5_heat_h0.json 2> gpurun_out/r05_job24_C5_heat_h0.err
python bench.py --workload C5 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job24_C5_6h.json 2> gpurun_out/r05_job24_C5_6h.err
python bench.py --workload C5 --heat --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job24_C5_heat_6h.json 2> gpurun_out/r05_job24_C5_heat_6h.err
python - <<'PY'
import json
for n in ("C5_heat_h0","C5_6h","C5_heat_6h"):
    try:
        d=json.loads(open(f"gpurun_out/r05_job24_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["roofline"]["kernel"], d["roofline"]["avg_us"], d["roofline"]["frac"], d["config"]["work"], d["timed_region"]["per_hour_s_last_rep"])
    except Exception as e: print(n, "ERR", e, open(f"gpurun_out/r05_job24_{n}.err").read()[-500:])
PY
rm -f gpurun_out/r05_job24_*.err
