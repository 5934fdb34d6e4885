"""Debug helper (GPU box): run a scenario on the HIP product and the oracle side by side and
print per-hour differences.  usage: python scripts/gpu_compare.py c1|c2f20|c2f60|het [hours]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from criteria3d_amd import capi, catchment as cm
from tests import checkers

def compare(m, forcing, hours, max_steps=None):
    gpu, ora = capi.load_product(), checkers.load_oracle()
    for sf in (gpu, ora):
        sf.lib.sf3d_reset_solver_state()
        cm.build(sf, m, threads=(16 if m.n < 200000 else 32))
    for h in range(hours):
        out = []
        for sf in (gpu, ora):
            t0 = time.time()
            steps, dts = cm.run_hour(sf, m, cm.FORCINGS[forcing](h), max_steps=max_steps)
            s = cm.snapshot(sf, m); s['steps'] = steps; s['dts'] = dts; s['t'] = time.time() - t0
            out.append(s)
        a, b = out
        relH = np.max(np.abs(a['H'] - b['H']) / np.maximum(np.abs(b['H']), 1e-9))
        dSe = np.max(np.abs(a['Se'] - b['Se']))
        print(f"h{h} steps {a['steps']}/{b['steps']} dts_equal={a['dts']==b['dts']} rel|dH|={relH:.3e} max|dSe|={dSe:.3e} t={a['t']:.2f}/{b['t']:.2f}")
        for k in ('total_water', 'storage', 'mbr', 'runoff', 'drainage', 'lateral'):
            print(f"    {k:12s} {a[k]!r:26} {b[k]!r:26} rel={abs(a[k]-b[k])/max(abs(b[k]),1e-30):.2e}")
        if a['dts'] != b['dts'] and a['dts'] and b['dts']:
            for k, (x, y) in enumerate(zip(a['dts'], b['dts'])):
                if x != y:
                    print('    first dt mismatch at step', k, x, y); break
    print('gpu', gpu.counters()); print('ora', ora.counters())

which = sys.argv[1]
hours = int(sys.argv[2]) if len(sys.argv) > 2 else None
if which == 'c1': compare(cm.column_model(), 'R5', hours or 24)
if which == 'c2f20': compare(cm.catchment_model(64, 64, 10), 'F20', hours or 6)
if which == 'c2f60': compare(cm.catchment_model(64, 64, 10), 'F60', hours or 2, max_steps=500)
if which == 'c2f60full': compare(cm.catchment_model(64, 64, 10), 'F60', hours or 3)
if which == 'c3f20': compare(cm.catchment_model(256, 256, 15), 'F20', hours or 2)
if which == 'c4f20': compare(cm.catchment_model(512, 512, 20), 'F20', hours or 1)
if which == 'c5s': compare(cm.dem_model_fast(cm.synthetic_dem()), 'F20', 1, max_steps=int(sys.argv[2]) if len(sys.argv) > 2 else 40)
if which == 'het': compare(cm.catchment_model(32, 32, 6, heterogeneous=True), 'F20', hours or 2, max_steps=500)
