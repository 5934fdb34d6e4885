#!/bin/bash
# round 4, GPU job 14 (A/B of a heat kernel variant): heat goldens, C5 + heat bench, kernel trace only
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04_job14}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_golden.py tests/test_gpu_heat.py -q -m gpu -k "heat and not full_size and not full_hour" --durations=3 > $OUT/heat.log 2>&1; echo "rc=$?" >> $OUT/heat.log
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o runc --output-format csv -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
cd $ROOT
tail -n 4 $OUT/heat.log
python - "$OUT" <<'PY'
import json, sys, glob, csv
out = sys.argv[1]
d = json.loads(open(out + "/bench_C5_heat.json").read().strip().splitlines()[-1])
print("C5 + heat:", d["value"], d["repeats_s"])
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:44]:44s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f} us total={float(r['TotalDurationNs'])/1e6:9.1f} ms max={float(r['MaxNs'])/1e3:9.1f}")
PY
