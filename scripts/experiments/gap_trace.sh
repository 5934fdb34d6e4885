mkdir -p gpurun_out/gaps
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/gaps/trace -- python3 $ROOT/bench.py --no-cpu-baseline --no-kernel-timing --warmup 0 > $ROOT/gpurun_out/gaps/bench.json 2> $ROOT/gpurun_out/gaps/bench.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections, re, json
f = glob.glob("gpurun_out/gaps/trace/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.split(r"[<(]", r["Kernel_Name"].replace("void ", ""))[0]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region: from the first k_step_begin to the last kernel
i0 = next(i for i, r in enumerate(rows) if r[2] == "k_step_begin")
rows = rows[i0:]
busy = 0; gaps = collections.Counter(); gapn = collections.Counter(); cur_end = rows[0][0]
for s, e, n in rows:
    if s > cur_end:
        g = s - cur_end
        gaps[prev] += g; gapn[prev] += 1
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev = n
span = cur_end - rows[0][0]
print("span ms", span / 1e6, "busy ms", busy / 1e6, "idle ms", (span - busy) / 1e6, json.load(open("gpurun_out/gaps/bench.json"))["value"])
for k, v in gaps.most_common(10):
    print(f"  idle after {k:22s} {v/1e6:8.3f} ms in {gapn[k]:5d} gaps  (avg {v/gapn[k]/1e3:7.1f} us)")
PY
