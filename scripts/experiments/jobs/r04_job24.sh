#!/bin/bash
# round 4, GPU job 24: the whole -m gpu suite the way the driver runs it (-x) on the final code, smoke(), the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job24
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests/ -x -q -m gpu --durations=25 ) > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "rc=$?" >> $OUT/smoke.log
python bench.py > $OUT/bench.json 2> $OUT/bench.err
grep -E "passed|failed|^FAILED|^ERROR|real" $OUT/suite.log | tail -8
tail -n 2 $OUT/smoke.log
cat $ROOT/gpurun_out/fullsize_worker_times.txt
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job24"
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1])
print("bench:", d["value"], d["roofline"]["frac"], d["roofline"]["traffic_source"], d["roofline"]["avg_us"], d["roofline"]["step"]["frac"], d["roofline"]["step"]["traffic_frac"])
PY
