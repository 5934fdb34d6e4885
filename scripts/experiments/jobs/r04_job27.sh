#!/bin/bash
# round 4, GPU job 27 (A/B): k_heat_save_water over a stack-ordered chunk list (SF3D_HEAT_STACK_ORDER=1) - parity with the switch on, C5 + heat hour off / on, kernel traces
# (the switch it drives - SF3D_HEAT_STACK_ORDER - measured no gain and was taken out again: scripts/experiments/retired/heat_stack_order.diff)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job27
mkdir -p $OUT
cd $ROOT
SF3D_HEAT_STACK_ORDER=1 python -m pytest tests/test_gpu_golden.py tests/test_gpu_heat.py -q -m gpu -k "heat and not full_size and not full_hour" --durations=3 > $OUT/heat.log 2>&1; echo "rc=$?" >> $OUT/heat.log
tail -n 3 $OUT/heat.log
cd /tmp && export TMPDIR=/tmp
for mode in 0 1 ${EXTRA_MODES:-}; do
  export SF3D_HEAT_STACK_ORDER=${mode%%:*}
  if [[ "$mode" == *:* ]]; then export SF3D_HEAT_STACK_BAND=${mode##*:}; else unset SF3D_HEAT_STACK_BAND; fi
  rocprofv3 --kernel-trace --stats -d $OUT/trace_$mode -o runc --output-format csv -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/trace_$mode.json 2> $OUT/trace_$mode.err
  python3 - "$OUT" "$mode" <<'PY'
import json, sys, glob, csv
out, mode = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{out}/trace_{mode}.json").read().strip().splitlines()[-1])
print(f"mode {mode}: C5 + heat {d['value']:.4f} sim-h/s")
for f in glob.glob(f"{out}/trace_{mode}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ("save_water<", "heat_assemble", "k_assemble")):
            print(f"   {r['Name'][:44]:44s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:9.1f} us")
PY
  rm -rf $OUT/trace_$mode
done
