#!/bin/bash
# round 4, GPU job 12: heat sweep / assembly / save-water with descriptor indices, float pair of link fluxes: parity, C5 + heat bench, trace + PMC
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job13
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_golden.py tests/test_gpu_heat.py tests/test_gpu_multirank.py -q -m gpu -k "heat or golden or vectors" --durations=5 > $OUT/heat.log 2>&1; echo "rc=$?" >> $OUT/heat.log
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
PMC_STEPS=1 bash scripts/profile_gpu.sh r04_job13/prof --workload C5 --heat --steps 1 --warmup 0 --reps 1 > $OUT/profile_gpu.log 2>&1
tail -n 6 $OUT/heat.log
python - <<'PY'
import json, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job13"
d = json.loads(open(out + "/bench_C5_heat.json").read().strip().splitlines()[-1])
print("C5 + heat:", d["value"], d["repeats_s"])
PY
head -n 30 $OUT/profile_gpu.log
