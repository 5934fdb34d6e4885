import json, sys
d = json.load(open(sys.argv[1]))
print("value", round(d["value"], 3), "sim-h/s  ms/step", round(d["ms_per_step"], 2), "work", d["config"]["work"])
r = d["roofline"]
print("dominant", r["kernel"], "achieved", round(r["achieved"]), "GB/s frac", round(r["frac"], 3), "avg_us", round(r["avg_us"], 1))
tot = 0
for k, v in r["kernels"].items():
    if v["launches"]:
        print(f"  {k:11s} n={v['launches']:5d} total={v['total_ms']:8.2f} ms avg={v['total_ms']/v['launches']*1e3:8.1f} us  {v['GBps']:.0f} GB/s")
        tot += v["total_ms"]
print("  kernels total", round(tot, 1), "ms of", round(d["ms_per_step"] * d["steps"], 1))
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"])
