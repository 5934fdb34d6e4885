#!/bin/bash
# round 5 job 7: the strip-local build on the device (bit for bit the global build; peak memory of a rank), the two-rank bench line with it,
# then the suite the way the driver runs it (-x), smoke() and the default bench line on the final code
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multirank.py -q -s -k "strip_local" > gpurun_out/r05_job07_strip_local_build.log 2>&1; grep -E "build seconds|passed|failed|Error|error" gpurun_out/r05_job07_strip_local_build.log | tail -6
SF3D_BENCH_SHARE_GPU=1 SF3D_BENCH_STRIP_LOCAL_BUILD=1 python bench.py --gpus 2 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job07_bench_2ranks_strip_local.json 2> gpurun_out/r05_job07_bench_2ranks_strip_local.err; grep -E "graph build|exchange transport" gpurun_out/r05_job07_bench_2ranks_strip_local.err
python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r05_job07_suite.log 2>&1; tail -22 gpurun_out/r05_job07_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r05_job07_bench.json 2> gpurun_out/r05_job07_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_job07_bench_driver_style.json 2> gpurun_out/r05_job07_bench_driver_style.err
python - <<'PY'
import json
for n in ("bench","bench_driver_style","bench_2ranks_strip_local"):
    try:
        d=json.loads(open(f"gpurun_out/r05_job07_{n}.json").read().strip().splitlines()[-1])
        print(n, d["value"], d["roofline"]["frac"], d["roofline"]["avg_us"], (d.get("cpu_baseline") or {}).get("value"), ((d.get("cpu_baseline") or {}).get("tuned") or {}).get("value"), d["config"]["partition"])
    except Exception as e: print(n, "ERR", e)
PY
