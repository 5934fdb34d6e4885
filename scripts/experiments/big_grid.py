"""One-off: 1024 x 1024 x 20 (21 M nodes, 4 x the headline grid) - index arithmetic beyond 2^31 bytes per array, timing."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from criteria3d_amd import capi, catchment as cm
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
t0 = time.time(); m = cm.catchment_model(nx, nx, 20); print("model", m.n, round(time.time() - t0, 1), "s", flush=True)
sf = capi.load_product(); sf.lib.sf3d_reset_solver_state()
t0 = time.time(); cm.build(sf, m); sf.lib.sf3d_synchronize(); print("build+upload", round(time.time() - t0, 1), "s", flush=True)
w0 = sf.lib.sf3d_get_total_water_content()
sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(20.0, m.cell_area))); sf.lib.sf3d_synchronize()
t0 = time.time(); t = 0.0; steps = 0
while t < 3600.0:
    t += sf.lib.sf3d_compute_step(3600.0 - t); steps += 1
sf.lib.sf3d_synchronize(); el = time.time() - t0
w1 = sf.lib.sf3d_get_total_water_content()
H = sf.total_potential(0, m.n)
rain = 20e-3 * m.ns * m.cell_area
out = sf.lib.sf3d_get_total_boundary_water_flow(capi.BND_RUNOFF) + sf.lib.sf3d_get_total_boundary_water_flow(capi.BND_FREE_DRAINAGE) + sf.lib.sf3d_get_total_boundary_water_flow(capi.BND_FREE_LATERAL_DRAINAGE)
print(f"hour 0: {steps} steps in {el:.2f} s = {1/el:.2f} sim-h/s; finite {np.isfinite(H).all()}; dW {w1 - w0:.3f} rain {rain:.3f} boundary {out:.3f} residual {(w1 - w0 - rain - out) / rain:.2e}; counters {sf.counters()}")
