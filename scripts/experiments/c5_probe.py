"""BASELINE config 5 (the Ravone project at full size) on the HIP product: work counters and wall time along hours 0-1 of a
25 mm hour, to see where the runoff regime (restore-best steps) starts and what a computeStep costs there.
usage: python scripts/experiments/c5_probe.py [mm] [hours] [max_steps_per_hour]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from criteria3d_amd import capi, catchment as cm, project3d as p3

mm = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
hours = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cap = int(sys.argv[3]) if len(sys.argv) > 3 else 100000
t0 = time.time()
m = p3.project_model(p3.load_project_fixture(Path(__file__).resolve().parents[2] / "tests" / "golden" / "ravone_project.npz"))
print(f"model {m.n} nodes, {m.ns} surface, built in {time.time() - t0:.1f}s", flush=True)
sf = capi.load_product()
sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
t0 = time.time(); cm.build(sf, m); print(f"pushed + uploaded in {time.time() - t0:.1f}s", flush=True)
k = 0
for h in range(hours):
    sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(mm if h == 0 else 0.0, m.cell_area)))
    t, n, w0 = 0.0, 0, time.time()
    while t < 3600.0 and n < cap:
        dt = sf.lib.sf3d_compute_step(3600.0 - t); t += dt; n += 1; k += 1
        if k % 100 == 0:
            c = sf.counters()
            print(k, f"h{h} t={t:.1f} dt={dt:.3f} wall={time.time() - w0:.2f}", {q: c[q] for q in ("attempts", "approximations", "sweeps", "courant_rejections", "restores")}, flush=True)
    sf.check(sf.lib.sf3d_synchronize(), "sync")
    c = sf.counters(); s = cm.snapshot(sf, m)
    print(f"hour {h}: {n} steps in {time.time() - w0:.2f}s", c, {q: s[q] for q in ("storage", "runoff", "drainage", "lateral")}, flush=True)
