#!/bin/bash
# round 4, GPU job 25: SQ counters of the C5 + heat hour (two --pmc passes, kernel trace only beside them): where the waves of the heat kernels spend their cycles
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r04_job25
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES SQ_WAVES"
B="SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
rocprofv3 --pmc $A --kernel-trace --output-format csv -d $OUT/passA -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/errA.log
rocprofv3 --pmc $B --kernel-trace --output-format csv -d $OUT/passB -- python3 $ROOT/bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/errB.log
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("passA", "passB"):
    f = glob.glob(f"{out}/{p}/*/*counter_collection.csv") + glob.glob(f"{out}/{p}/*counter_collection.csv")
    if not f:
        print("no counter file for", p, open(f"{out}/err{p[-1]}.log").read()[-600:]); continue
    for row in csv.DictReader(open(f[0])):
        k = re.split(r"[<(]", row["Kernel_Name"].replace("void ", ""))[0]
        agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for k, cs in agg.items():
    if "SQ_WAVE_CYCLES" not in cs: continue
    wc = cs["SQ_WAVE_CYCLES"]; thr = 0.25 * max(wc) if max(wc) > 0 else 0
    act = [i for i, x in enumerate(wc) if x > thr]
    def mean(c):
        v = cs.get(c)
        if not v: return None
        if len(v) == len(wc): return sum(v[i] for i in act) / max(len(act), 1)
        big = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
        return sum(big) / max(len(big), 1)
    r = {c: mean(c) for c in cs}
    r["active_launches"] = len(act)
    w = r["SQ_WAVE_CYCLES"]
    if w:
        r["frac_wait_any"] = r.get("SQ_WAIT_ANY", 0) / w; r["frac_wait_inst"] = r.get("SQ_WAIT_INST_ANY", 0) / w; r["frac_active_inst"] = r.get("SQ_ACTIVE_INST_ANY", 0) / w
        r["frac_active_valu"] = (r.get("SQ_ACTIVE_INST_VALU") or 0) / w
    res[k] = r
json.dump(res, open(out + "/sq_summary.json", "w"), indent=1)
for k in ("k_heat_save_water", "k_heat_assemble", "k_assemble", "k_heat_props", "k_heat_save_water_props", "k_props", "k_heat_sweep", "k_sweep_pair_masked", "k_heat_post"):
    r = res.get(k)
    if r: print(f"{k:26s} waves {r.get('SQ_WAVES',0):9.0f} wait_any {r.get('frac_wait_any',0):.2f} wait_inst {r.get('frac_wait_inst',0):.2f} active {r.get('frac_active_inst',0):.2f} valu {r.get('frac_active_valu',0):.2f} | insts/wave valu {(r.get('SQ_INSTS_VALU') or 0)/max(r.get('SQ_WAVES',1),1):8.0f} trans_f64 {(r.get('SQ_INSTS_VALU_TRANS_F64') or 0)/max(r.get('SQ_WAVES',1),1):6.0f} vmem_rd {(r.get('SQ_INSTS_VMEM_RD') or 0)/max(r.get('SQ_WAVES',1),1):6.0f}")
PY
rm -rf $OUT/passA $OUT/passB      # (the raw per-dispatch counter files of an hour of C5 + heat exceed what gpurun copies back)
