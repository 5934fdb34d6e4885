"""Where does the HIP product leave the oracle on a window of the Ravone project (25 mm hour + dry hour)?  Steps both libraries in
lock step, compares H every `every` steps and reports the first node / step at which the relative difference exceeds 1e-8 ... 1e-4
with what kind of node it is.
usage: python scripts/experiments/c5_window_diverge.py [every] [threads] [r0 r1 c0 c1]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from criteria3d_amd import capi, catchment as cm
from tests import checkers
from tests.scenarios import ravone_project_model

every = int(sys.argv[1]) if len(sys.argv) > 1 else 100
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 32
win = tuple(int(v) for v in sys.argv[3:7]) if len(sys.argv) > 6 else (72, 200, 300, 428)
max_steps = int(sys.argv[7]) if len(sys.argv) > 7 else 10**9
fine_from = int(sys.argv[8]) if len(sys.argv) > 8 else 10**9      # from this step on: compare after every step
was_equal = True
m = ravone_project_model(win)
index = m.meta["index"]
pos = np.full((m.n, 3), -1)
L, R, C = np.nonzero(index >= 0)
pos[index[L, R, C]] = np.stack([L, R, C], 1)
g, o = capi.load_product(), checkers.load_oracle()
for sf in (g, o):
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset"); cm.build(sf, m, threads=threads)
levels = [1e-8, 1e-7, 1e-6, 1e-5, 1e-4]
k = 0
t0 = time.time()
for h, mm in enumerate((25.0, 0.0)):
    for sf in (g, o):
        sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(mm, m.cell_area)))
    t = 0.0
    while t < 3600.0:
        dg = g.lib.sf3d_compute_step(3600.0 - t); do = o.lib.sf3d_compute_step(3600.0 - t)
        if dg != do:
            print(f"step {k}: accepted dt differs {dg} vs {do}", flush=True); sys.exit(0)
        t += dg; k += 1
        if k >= max_steps: sys.exit(0)
        if k % every == 0 or t >= 3600.0 or k >= fine_from:
            Hg, Ho = g.total_potential(0, m.n), o.total_potential(0, m.n)
            rel = np.abs(Hg - Ho) / np.maximum(np.abs(Ho), 1e-9)
            i = int(np.argmax(rel))
            cg, co = g.counters(), o.counters()
            same = all(cg[q] == co[q] for q in ("attempts", "approximations", "sweeps", "courant_rejections", "restores"))
            if was_equal and not same:
                was_equal = False
                print(f"  !! counters first differ at step {k}: gpu {cg} oracle {co}", flush=True)
            print(f"step {k} h{h} t={t:.1f} dt={dg:.4f} max rel {rel[i]:.3e} at node {i} (layer,row,col)={pos[i].tolist()} btype={m.btype[i]} "
                  f"Hg={Hg[i]:.9f} Ho={Ho[i]:.9f} z={m.z[i]:.4f} counters_equal={same} sweeps={cg['sweeps']} restores={cg['restores']} wall={time.time()-t0:.0f}s", flush=True)
            while levels and rel[i] > levels[0]:
                lv = levels.pop(0)
                top = np.argsort(rel)[-5:][::-1]
                print(f"  >> first above {lv:g}: " + "; ".join(f"node {int(j)} {pos[j].tolist()} rel {rel[j]:.2e} psi_g {Hg[j]-m.z[j]:.6e} psi_o {Ho[j]-m.z[j]:.6e}" for j in top), flush=True)
