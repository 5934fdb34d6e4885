#!/bin/bash
# round 4, GPU job 6: rocprofv3 trace + PMC passes of the driver's command on the final code; multirank tests (host staging bytes); C5 water-only
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job6
mkdir -p $OUT
cd $ROOT
bash scripts/profile_gpu.sh r04_job6/prof > $OUT/profile_gpu.log 2>&1
python -m pytest tests/test_gpu_multirank.py -q --durations=8 > $OUT/multirank.log 2>&1; echo "rc=$?" >> $OUT/multirank.log
python bench.py --workload C5 --steps 2 --warmup 0 --reps 1 > $OUT/bench_C5.json 2> $OUT/bench_C5.err
tail -n 25 $OUT/profile_gpu.log
tail -n 12 $OUT/multirank.log
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job6"
d = json.loads(open(out + "/bench_C5.json").read().strip().splitlines()[-1])
print("C5 F20 2 h:", d["value"], d["roofline"]["kernels"], d["cpu_baseline"])
PY
