#!/bin/bash
# on the GPU box: kernel trace of one C4 hour with the uniform soil rows in their own launch: durations of k_assemble_uniform and k_assemble
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/asm_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SF3D_ASM_UNIFORM=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --reps 1 --no-kernel-timing > /dev/null 2> $OUT/err.log
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
    d[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n in ("k_assemble_uniform", "k_assemble", "k_props", "k_sweep_pair"):
    v = [x for x in d.get(n, []) if x > 20]
    if v: print(n, "n=%d median=%.1f us min=%.1f max=%.1f" % (len(v), statistics.median(v), min(v), max(v)))
PY
