"""How sensitive is a trajectory to last-bit arithmetic?  The CPU restatement against ITSELF built with contracted multiply-adds
(make -C oracle oracle-fma: -ffp-contract=fast -march=x86-64-v3, the only difference), stepped in lock step on a window of the Ravone
project (25 mm hour + dry hour).  Where the two CPU builds separate, no two implementations with different rounding can stay within
1e-6 of each other: profiles/README.md "sensitivity".  CPU only.
usage: python scripts/experiments/oracle_fma_sensitivity.py r0 r1 c0 c1"""
import sys, time, numpy as np
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from criteria3d_amd import capi, catchment as cm
from tests import checkers
from tests.scenarios import ravone_project_model
win = tuple(int(v) for v in sys.argv[1:5])
m = ravone_project_model(win)
print("nodes", m.n, m.ns, flush=True)
a = checkers.load_oracle(); b = capi.SF3D(str(checkers.ROOT / 'oracle' / 'libsf3d_oracle_fma.so'))
for sf in (a,b):
    sf.lib.sf3d_reset_solver_state(); cm.build(sf, m, threads=4)
k=0; t0=time.time()
for h,mm in enumerate((25.0,0.0)):
    for sf in (a,b): sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(mm, m.cell_area)))
    t=0.0
    while t<3600:
        da=a.lib.sf3d_compute_step(3600-t); db=b.lib.sf3d_compute_step(3600-t)
        if da!=db: print("dt differs at step",k,da,db, flush=True); sys.exit()
        t+=da; k+=1
        if k%200==0 or t>=3600:
            Ha,Hb=a.total_potential(0,m.n),b.total_potential(0,m.n)
            rel=np.abs(Ha-Hb)/np.maximum(np.abs(Ha),1e-9); i=int(np.argmax(rel))
            print(f"step {k} h{h} t={t:.0f} dt={da:.4f} max rel {rel[i]:.3e} node {i} surf={i<m.ns} psi_a={Ha[i]-m.z[i]:.6e} psi_b={Hb[i]-m.z[i]:.6e} {time.time()-t0:.0f}s", flush=True)
