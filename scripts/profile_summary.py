"""Summarise a scripts/profile_gpu.sh run: per-kernel average duration of the launches that did work
(kernel trace) and HBM traffic per launch from the PMC passes, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 64 B per 128-B
request: x2; WRITE_SIZE exact; both in KiB)."""
import csv, glob, json, re, sys, collections

def base(name):
    """k_sweep<1>(DevView) / void k_sweep<1>(DevView) -> k_sweep"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    return re.split(r"[<(]", name)[0]

out = sys.argv[1]
res = {}
pmc = {}
f = glob.glob(f"{out}/trace/*/*kernel_trace.csv")
if f:
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        dur[base(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in dur.items():
        ref = sorted(v)[int(0.9 * (len(v) - 1))]        # 90th percentile: one slow outlier must not hide the real launches
        active = [x for x in v if x > 0.25 * ref] if k in ("k_sweep", "k_sweep_pair", "k_sweep_pair_masked", "k_props", "k_assemble", "k_post", "k_accept", "k_accept_links", "k_restore") else v
        res.setdefault(k, {})["launches"] = len(v)
        res[k]["active_launches"] = len(active)
        res[k]["avg_active_us"] = sum(active) / max(len(active), 1) / 1e3
        res[k]["total_ms"] = sum(v) / 1e6
for cnt, key, corr in (("FETCH_SIZE", "hbm_read_MB", 2.0), ("WRITE_SIZE", "hbm_write_MB", 1.0)):
    f = glob.glob(f"{out}/pmc_{cnt}/*/*counter_collection.csv")
    if not f: continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(f[0])):
        agg[base(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    whole = res.setdefault("whole_run", {})          # every dispatch of the pass: the traffic of the whole bench run (all repetitions of it)
    whole[key.replace("_MB", "_GB")] = corr * sum(sum(v) for v in agg.values()) * 1024 / 1e9
    pmc[cnt] = {k: {"dispatches": len(v), "max_KiB": max(v), "mean_of_active_KiB": (lambda b: sum(b) / max(len(b), 1))([x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v)}
                for k, v in agg.items()}
    for k, v in agg.items():
        big = [x for x in v if x > 0.25 * max(v)] if max(v) > 0 else v
        res.setdefault(k, {})[key] = corr * (sum(big) / max(len(big), 1)) * 1024 / 1e6
if "hbm_read_GB" in res.get("whole_run", {}) and "hbm_write_GB" in res["whole_run"]:
    res["whole_run"]["hbm_traffic_GB"] = res["whole_run"]["hbm_read_GB"] + res["whole_run"]["hbm_write_GB"]
    if len(sys.argv) > 2: res["whole_run"]["steps"] = int(sys.argv[2])
for k, v in res.items():
    if "hbm_read_MB" in v and "hbm_write_MB" in v:
        v["hbm_traffic_MB"] = v["hbm_read_MB"] + v["hbm_write_MB"]
        if v.get("avg_active_us"):
            v["hbm_GBps"] = v["hbm_traffic_MB"] / v["avg_active_us"] * 1e3   # MB/us = TB/s
# which kernels these numbers belong to: a fingerprint of the translation unit's sources (bench.py compares it with the sources it runs and
# says so on the line when a stored profile describes another version of the kernels - round 5's advisor)
try:
    import hashlib, pathlib
    root = pathlib.Path(__file__).resolve().parent.parent
    h = hashlib.sha256()
    for f in sorted((root / "criteria3d_amd" / "csrc").iterdir()):
        if f.suffix in (".inc", ".h", ".hip", ".cpp"):
            h.update(f.name.encode()); h.update(f.read_bytes())
    res["source_sha16"] = h.hexdigest()[:16]
except Exception:
    pass
json.dump(res, open(f"{out}/summary.json", "w"), indent=1, sort_keys=True)
json.dump(pmc, open(f"{out}/pmc_extract.json", "w"), indent=1, sort_keys=True)
for k in sorted((k for k in res if isinstance(res[k], dict)), key=lambda k: -res[k].get("total_ms", 0)):
    v = res[k]
    if k == "whole_run":
        print("whole run:", v); continue
    print(f"{k:22s} n={v.get('launches',0):5d} active={v.get('active_launches',0):5d} avg={v.get('avg_active_us',0):8.1f} us total={v.get('total_ms',0):8.2f} ms "
          f"traffic={v.get('hbm_traffic_MB', float('nan')):8.1f} MB  {v.get('hbm_GBps', float('nan')):7.0f} GB/s")
