#!/bin/bash
# round 4, GPU job 11: the double-double sweep norm - launch-mode / rank-count equalities, parity, goldens, then the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job11
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_cg.py tests/test_regular_grid.py tests/test_gpu_heat.py tests/test_gpu_fuzz.py -q -m gpu --durations=10 ) > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log
#python bench.py > $OUT/bench.json 2> $OUT/bench.err
#python bench.py --workload C5 --steps 1 --warmup 0 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
grep -E "passed|failed|^FAILED|^ERROR|real" $OUT/suite.log | tail -12
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job11"
for f in ("bench.json", "bench_c5.json"):
    try:
        d = json.loads(open(out + "/" + f).read().strip().splitlines()[-1])
        print(f, d["value"], d["roofline"]["frac"], d["roofline"].get("kernel"), d["roofline"].get("avg_us"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
