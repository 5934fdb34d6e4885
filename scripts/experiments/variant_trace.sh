#!/bin/bash
# on the GPU box: kernel-trace durations of tuning builds (build_variants/lib<NAME>.so) over the launches of C4 hour 0 - also for builds whose
# results are wrong on purpose (ablations); usage: bash scripts/experiments/variant_trace.sh NAME...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for L in "$@"; do
  OUT=/tmp/variant_trace_$L; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && SF3D_PRODUCT_LIB=$ROOT/build_variants/lib$L.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --reps 1 --no-kernel-timing > /dev/null 2> $OUT/err.log )
  python3 - "$OUT" "$L" <<'PY'
import csv, glob, sys, collections, statistics
fs = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")
if not fs: print(sys.argv[2], "no trace"); sys.exit(0)
d = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    n = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = []
for n in ("k_sweep_pair", "k_assemble", "k_props"):
    v = sorted(x for x in d.get(n, []) if x > 20)[:200]
    if v: out.append("%s n=%d median=%.1f min=%.1f" % (n, len(v), statistics.median(v), min(v)))
print(sys.argv[2], "; ".join(out))
PY
done
