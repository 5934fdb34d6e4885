#!/bin/bash
# on the GPU box: which clock do the fp64-heavy kernels run at?  GRBM_GUI_ACTIVE (GPU-busy cycles) of every dispatch over its duration
# from the kernel trace of the same pass -> MHz per kernel; plus the VALU share of the wave cycles.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/clock
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --reps 1 --no-f60 --no-kernel-timing > /dev/null 2> $OUT/err.log
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
f = glob.glob(out + "/*/*counter_collection.csv")
t = glob.glob(out + "/*/*kernel_trace.csv")
if not f or not t:
    print("missing output", open(out + "/err.log").read()[-800:]); sys.exit(0)
dur = {}
for r in csv.DictReader(open(t[0])):
    dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = re.split(r"[<(]", r["Kernel_Name"].replace("void ", ""))[0]
    agg[k][r["Counter_Name"]].append((float(r["Counter_Value"]), dur.get(r["Dispatch_Id"], 0)))
for k in ("k_sweep_pair", "k_sweep", "k_assemble", "k_props", "k_post", "k_accept_links", "k_courant_probe"):
    c = agg.get(k)
    if not c: continue
    g = [(v, d) for v, d in c["GRBM_GUI_ACTIVE"] if d > 20000]          # launches that did work
    if not g: continue
    mhz = sum(v for v, d in g) / sum(d for v, d in g) * 1e3
    wc = sum(v for v, d in c["SQ_WAVE_CYCLES"] if d > 20000); va = sum(v for v, d in c["SQ_ACTIVE_INST_VALU"] if d > 20000)
    iv = sum(v for v, d in c["SQ_INSTS_VALU"] if d > 20000); n = len(g)
    print(f"{k:16s} launches {n:4d}  avg {sum(d for v, d in g)/n/1e3:7.1f} us  GRBM_GUI_ACTIVE/duration = {mhz:6.0f} MHz  VALU-active/wave-cycles {va/max(wc,1):.3f}  VALU insts/launch {iv/n/1e6:.1f} M")
PY
