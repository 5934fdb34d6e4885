#!/bin/bash
# round 5 job 8: the suite the way the driver runs it (-x) on the final code; eight ranks sharing the GPU (functional: exchange statistics of
# an 8-way run, strip-local build)
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu --durations=12 > gpurun_out/r05_job08_suite.log 2>&1; tail -20 gpurun_out/r05_job08_suite.log
SF3D_BENCH_SHARE_GPU=1 SF3D_BENCH_STRIP_LOCAL_BUILD=1 python bench.py --gpus 8 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job08_bench_8ranks_shared.json 2> gpurun_out/r05_job08_bench_8ranks_shared.err
grep -E "exchange transport|flag hop" gpurun_out/r05_job08_bench_8ranks_shared.err | cut -c1-330
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05_job08_bench_8ranks_shared.json").read().strip().splitlines()[-1])
print(d["value"], d["config"]["partition"], d["exchange"])
PY
