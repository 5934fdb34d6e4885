#!/bin/bash
# round 4, GPU job 9: C5 + heat after the nontemporal streams in the heat link kernels (trace + PMC), heat parity tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job9
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_golden.py tests/test_gpu_heat.py -q -k "not full_size and not project_window_full_hour" --durations=5 > $OUT/heat.log 2>&1; echo "rc=$?" >> $OUT/heat.log
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
PMC_STEPS=1 bash scripts/profile_gpu.sh r04_job9/prof --workload C5 --heat --steps 1 --warmup 0 --reps 1 > $OUT/profile_gpu.log 2>&1
tail -n 6 $OUT/heat.log
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job9"
d = json.loads(open(out + "/bench_C5_heat.json").read().strip().splitlines()[-1])
print("C5 + heat:", d["value"], d["repeats_s"])
PY
head -n 14 $OUT/profile_gpu.log
