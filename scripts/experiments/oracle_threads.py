"""How fast is the oracle on this host?  C5 project at full size, 6 computeStep calls from the initial state per thread count, with and
without interleaved memory (SF3D_ORACLE_INTERLEAVE read at library load: one process per setting).
usage: python scripts/experiments/oracle_threads.py <threads> [window r0 r1 c0 c1]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
from criteria3d_amd import catchment as cm
from tests import checkers
from tests.scenarios import ravone_project_model
threads = int(sys.argv[1])
win = tuple(int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else None
m = ravone_project_model(win)
o = checkers.load_oracle()
o.lib.sf3d_reset_solver_state()
t0 = time.time(); cm.build(o, m, threads=threads); tb = time.time() - t0
o.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(25.0, m.cell_area)))
t0 = time.time()
for k in range(6):
    o.lib.sf3d_compute_step(3600.0)
print(f"threads {threads} nodes {m.n}: build {tb:.1f}s, 6 steps {time.time()-t0:.2f}s, sweeps {o.counters()['sweeps']}", flush=True)
