#!/bin/bash
# k_approx_patch against k_props + k_assemble: bitwise equality on C3/C4 F20 (scripts/run_case.py) and per-kernel times (bench --time-all-kernels)
out=gpurun_out/${1:-patch}; mkdir -p $out; big=/tmp/patch_probe_npz; mkdir -p $big
for case in c2f60 c3f20 c4f20; do
  SF3D_APPROX_PATCH=0 python scripts/run_case.py $case $big/${case}_off.npz
  for w in 6 10 14; do
    SF3D_APPROX_PATCH=1 SF3D_PATCH_W=$w python scripts/run_case.py $case $big/${case}_on$w.npz
    python - <<PY
import numpy as np
a=np.load("$big/${case}_off.npz"); b=np.load("$big/${case}_on$w.npz")
bad=[k for k in a.files if not np.array_equal(a[k],b[k])]
print("$case W=$w", "BITWISE EQUAL" if not bad else ("DIFFERENT: %s" % bad), flush=True)
for k in bad[:4]:
    d=np.abs(a[k]-b[k]); print("   ",k, d.max(), int((d>0).sum()), a[k].shape)
PY
  done
done
for mode in 0 1; do
  for w in 6 10; do
    [ $mode = 0 ] && [ $w = 10 ] && continue
    SF3D_APPROX_PATCH=$mode SF3D_PATCH_W=$w python bench.py --steps 6 --warmup 1 --time-all-kernels --reps 1 --no-cpu-baseline --no-f60 > $out/bench_timed_patch${mode}_w$w.json 2> $out/bench_timed_patch${mode}_w$w.err
    SF3D_APPROX_PATCH=$mode SF3D_PATCH_W=$w python bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-f60 > $out/bench_patch${mode}_w$w.json 2> $out/bench_patch${mode}_w$w.err
    python - <<PY
import json
for f in ("$out/bench_timed_patch${mode}_w$w.json","$out/bench_patch${mode}_w$w.json"):
    try:
        l=json.load(open(f)); r=l["roofline"]
        print(f, "value", round(l["value"],2), {k:(v["launches"], round(v["total_ms"]/max(1,v["launches"])*1e3,1)) for k,v in r["kernels"].items() if v["launches"]}, flush=True)
    except Exception as e: print(f, "FAILED", e)
PY
  done
done
