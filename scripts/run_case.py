"""Run one scenario on the HIP product and save H/Se/dts (used by tests that compare library modes
selected through environment variables, which the library reads once per process).
usage: python scripts/run_case.py <case> <outfile>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from criteria3d_amd import capi, catchment as cm

case, out = sys.argv[1], sys.argv[2]
sf = capi.load_product()
if case == "c2f60":
    m, plan = cm.catchment_model(64, 64, 10), [(60.0, None), (0.0, 600)]
elif case == "c3f20":
    m, plan = cm.catchment_model(256, 256, 15), [(20.0, None), (0.0, None)]
elif case == "c4f20":
    m, plan = cm.catchment_model(512, 512, 20), [(20.0, None)]
else:
    raise SystemExit("unknown case")
sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
cm.build(sf, m)
res = {}
for h, (mm, mx) in enumerate(plan):
    _, dts = cm.run_hour(sf, m, mm, max_steps=mx)
    s = cm.snapshot(sf, m)
    res[f"H{h}"] = s["H"]; res[f"Se{h}"] = s["Se"]; res[f"dts{h}"] = np.array(dts); res[f"storage{h}"] = np.array(s["storage"])
# per-node lateral flow sums and boundary sums of the last hour: the link flow sums are added by their own kernel
res["lateral_in"] = np.array([sf.lib.sf3d_get_node_sum_lateral_water_flow_in(int(i)) for i in range(0, m.n, max(1, m.n // 4096))])
res["lateral_out"] = np.array([sf.lib.sf3d_get_node_sum_lateral_water_flow_out(int(i)) for i in range(0, m.n, max(1, m.n // 4096))])
res["down"] = np.array([sf.lib.sf3d_get_node_max_water_flow(int(i), capi.LINK_DOWN) for i in range(0, m.n, max(1, m.n // 4096))])
res["boundary"] = sf.boundary_water_flow(0, m.n)
c = sf.counters()
res["counters"] = np.array([c[k] for k in capi.COUNTER_NAMES[:7]], dtype=np.int64)
res["early_courant"] = np.array(c["early_courant_rejections"], dtype=np.int64)      # how the product got there: not compared between modes
np.savez(out, **res)
sf.lib.sf3d_clean()
