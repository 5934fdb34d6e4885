#!/bin/bash
# round 4, GPU job 32 (A/B of a product change): goldens + parity + heat subset, C5 + heat hour with a kernel trace, the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r04_job32}
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_golden.py tests/test_gpu_parity.py tests/test_gpu_heat.py tests/test_gpu_fuzz.py -q -m gpu -k "not full_size and not full_hour" --durations=3 > $OUT/tests.log 2>&1; echo "rc=$?" >> $OUT/tests.log
tail -n 3 $OUT/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o runc --output-format csv -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/trace.json 2> $OUT/trace.err
cd $ROOT
python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python - "$OUT" <<'PY'
import json, sys, glob, csv
out = sys.argv[1]
d = json.loads(open(out + "/trace.json").read().strip().splitlines()[-1]); print("C5 + heat (under the tracer):", d["value"])
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print(f"{r['Name'][:44]:44s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f} us total={float(r['TotalDurationNs'])/1e6:9.1f} ms")
d = json.loads(open(out + "/bench.json").read().strip().splitlines()[-1]); print("C4:", d["value"], d["roofline"]["avg_us"], d["f60_hour0"]["value"])
PY
rm -rf $OUT/trace
