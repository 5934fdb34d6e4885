#!/bin/bash
# round 5 job 10: BASELINE config 5 for the record on the final code - the Ravone project, F20, the 6-hour episode, water only and with coupled heat
mkdir -p gpurun_out
python bench.py --workload C5 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job10_C5_6h.json 2> gpurun_out/r05_job10_C5_6h.err
python bench.py --workload C5 --heat --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job10_C5_heat_6h.json 2> gpurun_out/r05_job10_C5_heat_6h.err
python - <<'PY'
import json
for n in ("C5_6h","C5_heat_6h"):
    try:
        d=json.loads(open(f"gpurun_out/r05_job10_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["roofline"]["kernel"], d["roofline"]["avg_us"], d["roofline"]["frac"], d["config"]["work"], d["timed_region"]["per_hour_s_last_rep"])
    except Exception as e: print(n, "ERR", e, open(f"gpurun_out/r05_job10_{n}.err").read()[-500:])
PY
