"""Summarise a rocprofv3 kernel trace of a runoff-regime run: launches, busy time and wall time per computeStep."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in csv.DictReader(open(f))))
begins = [k for k, r in enumerate(rows) if r[2].startswith("k_step_begin")]
steps = begins[len(begins) // 2: len(begins) // 2 + 400]      # 400 steps from the middle of the run
n = collections.Counter(); busy = collections.Counter(); tiny = collections.Counter()
for a, b in zip(steps[:-1], steps[1:]):
    for s, e, name in rows[a:b]:
        n[name] += 1; busy[name] += e - s
        if e - s < 3000: tiny[name] += 1
S = len(steps) - 1
wall = (rows[steps[-1]][0] - rows[steps[0]][0]) / S
print(f"{S} steps: {wall/1e3:.1f} us wall per step, {sum(n.values())/S:.1f} launches, {sum(busy.values())/S/1e3:.1f} us busy")
for k in sorted(n, key=lambda k: -busy[k]):
    print(f"  {k:40s} {n[k]/S:6.2f} launches/step  {busy[k]/max(n[k],1)/1e3:6.2f} us each  {tiny[k]/S:5.2f} no-op (<3 us)/step")
gaps = [rows[k + 1][0] - rows[k][1] for k in range(steps[0], steps[-1])]
gaps.sort()
print("gaps between consecutive kernels: median %.2f us, p90 %.2f us, sum/step %.1f us" % (gaps[len(gaps)//2]/1e3, gaps[int(0.9*len(gaps))]/1e3, sum(g for g in gaps if g > 0)/S/1e3))
