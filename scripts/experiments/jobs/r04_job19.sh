#!/bin/bash
# round 4, GPU job 19: the whole -m gpu suite the way the driver runs it (-x), smoke(), the default bench line, C5 and C5 + heat
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job19
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests/ -x -q -m gpu --durations=25 ) > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_style.json 2> $OUT/bench_driver_style.err
python bench.py --workload C5 --no-cpu-baseline > $OUT/bench_C5.json 2> $OUT/bench_C5.err
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
grep -E "passed|failed|^FAILED|^ERROR|real" $OUT/suite.log | tail -8
grep -E "s call" $OUT/suite.log | head -12
tail -n 3 $OUT/smoke.log
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job19"
for f in ("bench.json", "bench_driver_style.json", "bench_C5.json", "bench_C5_heat.json"):
    try:
        d = json.loads(open(out + "/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["frac"], d["roofline"].get("avg_us"), d["roofline"]["step"]["frac"], d.get("cpu_baseline"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
