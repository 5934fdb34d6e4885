#!/bin/bash
# round 5 job 25: inputs of the expected scaling curve (scripts/scale_model.py) with the gather-free kernels, one box: C4 and one of two / four /
# eight strips of it as grids of their own, two ranks sharing the GPU (exchange statistics)
mkdir -p gpurun_out
B="--no-cpu-baseline --no-f60 --steps 6 --warmup 1"
for w in C4 C4H C4Q C4E; do python bench.py --workload $w $B > gpurun_out/r05_job25_bench_$w.json 2>/dev/null; done
SF3D_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job25_bench_2ranks_shared.json 2> gpurun_out/r05_job25_bench_2ranks_shared.err
grep "exchange transport" gpurun_out/r05_job25_bench_2ranks_shared.err > gpurun_out/r05_job25_bench_2ranks_shared_exchange_lines.txt
python - <<'PY'
import json
for w in ("C4","C4H","C4Q","C4E","2ranks_shared"):
    try:
        d=json.loads(open(f"gpurun_out/r05_job25_bench_{w}.json").read().strip().splitlines()[-1]); r=d["roofline"]
        print(w, d["value"], r["kernel"], r["avg_us"], d.get("exchange"))
    except Exception as e: print(w, "ERR", e)
PY
rm -f gpurun_out/r05_job25_bench_2ranks_shared.err
