"""CPU baselines on the GPU box's host (SURVEY.md 8d "CPU baseline timing"): the unmodified
reference (oracle/_ref, project flags -O2 -fopenmp) driven through the same harness, one fresh
process per measurement.  usage: python scripts/cpu_baselines.py [outfile]"""
import json, os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

WORKER = r'''
import sys, time, os
sys.path.insert(0, %r)
import numpy as np
from criteria3d_amd import capi, catchment as cm
from tests import checkers
backend, nx, ny, nz, forcing, hours, threads, budget = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7]), float(sys.argv[8])
sf = {"reference": checkers.load_reference, "reference_tuned": checkers.load_reference_tuned, "oracle": checkers.load_oracle}[backend]()
m = cm.catchment_model(nx, ny, nz)
cm.build(sf, m, threads=threads)
used = int(sf.lib.sf3d_set_threads_number(threads))
sim = wall = 0.0; steps = 0
for h in range(hours):
    sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(cm.FORCINGS[forcing](h), m.cell_area)))
    t = 0.0
    while t < 3600.0 and wall < budget:
        t0 = time.perf_counter(); dt = sf.lib.sf3d_compute_step(3600.0 - t); wall += time.perf_counter() - t0
        t += dt; sim += dt; steps += 1
    if wall >= budget: break
print("RESULT", used, steps, sim, wall)
''' % str(ROOT)

def run(backend, shape, forcing, hours, threads, budget=60.0):
    p = subprocess.run([sys.executable, "-c", WORKER, backend, *map(str, shape), forcing, str(hours), str(threads), str(budget)],
                       capture_output=True, text=True, timeout=1200)
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
    if not line:
        return {"error": (p.stdout + p.stderr)[-400:]}
    used, steps, sim, wall = line[0].split()[1:]
    return {"backend": backend, "grid": "x".join(map(str, shape)), "forcing": forcing, "threads": int(used), "steps": int(steps),
            "simulated_s": float(sim), "wall_s": float(wall), "sim_h_per_s": float(sim) / 3600.0 / float(wall)}

def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/cpu_baselines.json"
    ncpu = os.cpu_count() or 1
    res = {"host_threads": ncpu, "cpu_model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"), "runs": []}
    quick = len(sys.argv) > 2 and sys.argv[2] == "tuned"     # only the -O3 -march=x86-64-v3 build next to the project-flags build
    if len(sys.argv) > 2 and sys.argv[2] == "quota":
        # round 3: the GPU boxes give a container 16 CPUs' worth of time (cgroup cpu.max) of their 256 logical CPUs - thread counts around
        # that quota, both builds, and the runoff-regime sample (C4 F60 hour 0, as far as 25 s go)
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
            res["cgroup_cpu_quota"] = None if q == "max" else float(q) / float(per)
        except (OSError, ValueError):
            res["cgroup_cpu_quota"] = None
        for backend in ("reference", "reference_tuned"):
            for threads in (1, 8, 16):
                res["runs"].append(run(backend, (64, 64, 10), "F20", 6, threads))
            res["runs"].append(run(backend, (256, 256, 15), "F20", 2, 16, budget=40.0))
            for threads in (8, 16, 32):
                res["runs"].append(run(backend, (512, 512, 20), "F20", 1, threads, budget=25.0))
            res["runs"].append(run(backend, (512, 512, 20), "F60", 1, 16, budget=25.0))
        json.dump(res, open(out, "w"), indent=1)
        for r in res["runs"]:
            print(r)
        return
    for backend in (("reference", "reference_tuned") if quick else ("reference",)):
        for threads in ((1, 16) if quick else (1, 16, 64, ncpu)):
            res["runs"].append(run(backend, (64, 64, 10), "F20", 6, threads))
        for threads in ((16,) if quick else (16, 64, ncpu)):
            res["runs"].append(run(backend, (256, 256, 15), "F20", 2, threads, budget=40.0))
        for threads in ((32,) if quick else (32, 64, ncpu)):
            res["runs"].append(run(backend, (512, 512, 20), "F20", 1, threads, budget=25.0))
    json.dump(res, open(out, "w"), indent=1)
    for r in res["runs"]:
        print(r)

if __name__ == "__main__":
    main()
