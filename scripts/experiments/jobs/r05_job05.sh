#!/bin/bash
# round 5 job 5: the whole GPU suite with the water bands at 1e-9 and the new tests; config 5 + heat after the dead link-flux work went;
# k_props split by approximation (durations of a 6-hour C4 episode under the kernel trace)
mkdir -p gpurun_out
python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r05_job05_suite.log 2>&1
tail -40 gpurun_out/r05_job05_suite.log
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline --time-all-kernels > gpurun_out/r05_job05_C5_heat.json 2> gpurun_out/r05_job05_C5_heat.err
python bench.py --workload C5 --heat --steps 1 --warmup 0 --reps 1 --no-cpu-baseline > gpurun_out/r05_job05_C5_heat_untimed.json 2> gpurun_out/r05_job05_C5_heat_untimed.err
python - <<'PY'
import json
for n in ("C5_heat","C5_heat_untimed"):
    d=json.loads(open(f"gpurun_out/r05_job05_{n}.json").read().strip().splitlines()[-1])
    print(n, d["value"], d["config"]["work"])
    if n=="C5_heat":
        for k,v in sorted(d["roofline"]["kernels"].items(), key=lambda kv:-kv[1]["total_ms"])[:14]: print("   ", k, v["launches"], round(v["total_ms"],1), "ms", round(v["total_ms"]/max(v["launches"],1)*1e3,1), "us")
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05_job05_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 0 --reps 1 --no-cpu-baseline --no-f60 --no-kernel-timing > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r05_job05_trace.err
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee gpurun_out/r05_job05_props_split.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r05_job05_trace/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda t: t[0])
# an approximation 0 is the k_props launch that follows a k_step_begin (or a refused attempt); classify by what precedes: walk the stream
first, later, last = [], [], None
seen_post_since_begin = False
for t0, d, name in rows:
    if "k_step_begin" in name: seen_post_since_begin = False; fresh = True
    if name.startswith("void k_props<0") and d > 20000:
        (later if seen_post_since_begin else first).append(d / 1e3)
    if name.startswith("void k_post") and d > 20000: seen_post_since_begin = True
import statistics as st
print(f"k_props launches that did work in the 6-hour C4 episode: first approximation of a computeStep {len(first)}: mean {st.mean(first):.1f} us (min {min(first):.1f}, max {max(first):.1f}); later approximations {len(later)}: mean {st.mean(later):.1f} us (min {min(later):.1f}, max {max(later):.1f})")
PY
rm -rf gpurun_out/r05_job05_trace
