#!/bin/bash
# round 4, GPU job 2: sharded paired sweep tests, heat tests after the save-water split, W sweep on a half strip, C5 + heat profile
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job2
mkdir -p $OUT
cd $ROOT
python -m pytest tests/test_gpu_multirank.py tests/test_partition_gloo.py -x -q --durations=15 > $OUT/multirank.log 2>&1; echo "rc=$?" >> $OUT/multirank.log
python -m pytest tests/test_gpu_heat.py tests/test_gpu_golden.py -q --durations=15 -s > $OUT/heat.log 2>&1; echo "rc=$?" >> $OUT/heat.log
for WL in C4H C4Q; do
  for W in 0 6 10 14; do
    if [ $W = 0 ]; then export SF3D_PAIR_SWEEP=0; unset SF3D_PAIR_W; else export SF3D_PAIR_SWEEP=1 SF3D_PAIR_W=$W; fi
    python bench.py --workload $WL --steps 6 --warmup 1 --no-cpu-baseline --no-f60 --reps 3 > $OUT/bench_${WL}_W$W.json 2> $OUT/bench_${WL}_W$W.err
  done
done
unset SF3D_PAIR_SWEEP SF3D_PAIR_W
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/heat_trace -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > $OUT/bench_C5_heat.json 2> $OUT/bench_C5_heat.err
cd $ROOT
python - <<'PY'
import json, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/r04_job2"
for f in sorted(glob.glob(out + "/bench_C4*_W*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d["roofline"]["kernels"]
        print(os.path.basename(f), "value", round(d["value"], 2), {n: (v["launches"], round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1)) for n, v in k.items() if v["launches"]})
    except Exception as e:
        print(os.path.basename(f), "failed", e)
PY
tail -3 $OUT/multirank.log $OUT/heat.log
f=$(ls $OUT/heat_trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 $f
cat $OUT/bench_C5_heat.json | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 heat value', d['value'], d['config']['work'])"
