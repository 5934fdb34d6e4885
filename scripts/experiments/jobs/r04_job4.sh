#!/bin/bash
# round 4, GPU job 4: why did value differ between --steps 20 and --steps 6 at C2; the whole -m gpu suite (no -x); C5 + heat kernel stats
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job4
mkdir -p $OUT
cd $ROOT
for K in "20 5" "6 1" "12 0"; do set -- $K
  python bench.py --workload C2 --steps $1 --warmup $2 --reps 3 --no-cpu-baseline > $OUT/bench_C2_$1.json 2> $OUT/bench_C2_$1.err
done
python - <<'PY'
import json, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job4"
for f in sorted(glob.glob(out + "/bench_C2_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), "value", round(d["value"], 2), "repeats", [round(x, 4) for x in d["repeats_s"]], "episodes", [round(x, 4) for x in (d["headline_6h"] or {}).get("episodes_s", [])],
          "per hour (last rep)", [round(x * 1e3, 2) for x in d["timed_region"]["per_hour_s_last_rep"]])
PY
( time python -m pytest tests -m gpu -q --durations=45 ) > $OUT/suite.log 2>&1; echo "rc=$?" >> $OUT/suite.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/heat_trace -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
cd $ROOT
f=$(ls $OUT/heat_trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -16 $f
tail -n 70 $OUT/suite.log
