#!/bin/bash
# usage: bash scripts/pmc_probe.sh <tag> "<counter list>"   (one rocprofv3 --pmc pass of a 1-hour C4 bench; kernel k_sweep summarised)
set -u
TAG=$1; CNT=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-kernel-timing > /dev/null 2> $OUT/err.log
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")
if not f:
    print("no counter file", open(sys.argv[1] + "/err.log").read()[-800:]); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f[0])):
    k = re.split(r"[<(]", row["Kernel_Name"].replace("void ", ""))[0]
    agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in ("k_sweep_pair", "k_sweep", "k_assemble", "k_props", "k_post", "k_accept"):
    for c, v in agg.get(k, {}).items():
        big = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
        print(f"{k:12s} {c:28s} mean_active={sum(big)/max(len(big),1):14.4g}  max={max(v):14.4g} n={len(v)}")
PY
