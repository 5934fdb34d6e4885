#!/bin/bash
# round 4, GPU job 3: the whole -m gpu suite with durations; masked strips; heat after the save-water restructure; paired sweep below the Infinity Cache
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job3
mkdir -p $OUT
cd $ROOT
( time python -m pytest tests -m gpu -x -q --durations=45 ) > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
for WL in C3 C4E C4Q; do
  for MODE in off w6 auto; do
    unset SF3D_PAIR_SWEEP SF3D_PAIR_W SF3D_PAIR_AUTO_COST
    case $MODE in off) export SF3D_PAIR_SWEEP=0;; w6) export SF3D_PAIR_SWEEP=1 SF3D_PAIR_W=6;; auto) export SF3D_PAIR_AUTO_COST=1.9;; esac
    python bench.py --workload $WL --steps 6 --warmup 1 --no-cpu-baseline --no-f60 --reps 3 > $OUT/bench_${WL}_$MODE.json 2> $OUT/bench_${WL}_$MODE.err
  done
done
unset SF3D_PAIR_SWEEP SF3D_PAIR_W SF3D_PAIR_AUTO_COST
python - <<'PY'
import json, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job3"
for f in sorted(glob.glob(out + "/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["roofline"]["kernels"]
        print(os.path.basename(f), "value", round(d["value"], 4), {n: (v["launches"], round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1)) for n, v in k.items() if v["launches"]})
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
tail -n 60 $OUT/suite.log
