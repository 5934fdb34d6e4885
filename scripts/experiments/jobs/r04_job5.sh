#!/bin/bash
# round 4, GPU job 5: the whole -m gpu suite; C5 + heat; the driver's bench lines; two and eight ranks sharing the GPU
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job5
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -m gpu -q --durations=30 ) > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style_steps20_warmup5.json 2> $OUT/bench_driver.err
SF3D_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --steps 6 --warmup 1 --no-cpu-baseline > $OUT/bench_2_ranks_sharing_one_gpu.json 2> $OUT/bench_2_ranks.err
SF3D_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --steps 6 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_8_ranks_sharing_one_gpu.json 2> $OUT/bench_8_ranks.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/heat_trace -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
cd $ROOT
python - <<'PY'
import json, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job5"
for f in sorted(glob.glob(out + "/bench*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d.get("roofline") or {}
        print(os.path.basename(f), "value", round(d["value"], 4), "n_gpus", d["n_gpus"], "frac", r.get("frac"), "kernel", r.get("kernel"), "avg_us", r.get("avg_us"), "bytes", r.get("algorithmic_bytes_per_launch"),
              "step.frac", (r.get("step") or {}).get("frac"), "f60", (d.get("f60_hour0") or {}).get("value"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
f=$(ls $OUT/heat_trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -14 $f
grep -E "passed|failed|^FAILED|^ERROR|real" $OUT/suite.log | tail -12
grep -E "s call" $OUT/suite.log | head -14
tail -4 $OUT/bench_2_ranks.err $OUT/bench_8_ranks.err
