import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from criteria3d_amd import capi, catchment as cm
from tests import checkers
g, o = capi.load_product(), checkers.load_oracle()
m = cm.with_heat_surface(cm.dem_model(np.load('/root/repo/tests/golden/ravone_dem_window_72x72.npy')))
heat = cm.Heat(save_mode=0)
for sf in (g, o):
    sf.lib.sf3d_reset_solver_state(); cm.build(sf, m, threads=1, heat=heat)
    cm.apply_heat_forcing(sf, m, 0)
    sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(2.0, m.cell_area)))
    print(sf.backend, "dt", sf.lib.sf3d_compute_step(3600.0), sf.lib.sf3d_compute_step(3000.0))
Tg, To = g.temperature(0, m.n), o.temperature(0, m.n)
Hg, Ho = g.total_potential(0, m.n), o.total_potential(0, m.n)
d = np.abs(Tg - To); d[:m.ns] = 0
w = np.argsort(-d)[:12]
up = m.link_dir == capi.LINK_UP
upof = np.full(m.n, -1); upof[m.link_node[up]] = m.link_to[up]
print("max dT", d.max(), "max dH rel", np.max(np.abs(Hg - Ho) / np.maximum(np.abs(Ho), 1e-9)))
for i in w:
    print(i, "dT", d[i], "Tg", Tg[i], "To", To[i], "btype", m.btype[i], "up", upof[i], "up is surf", upof[i] < m.ns, "z", m.z[i], "size", m.size[i],
          "Hg-Ho", Hg[i] - Ho[i], "sens", g.lib.sf3d_get_node_boundary_sensible_flux(int(i)), o.lib.sf3d_get_node_boundary_sensible_flux(int(i)),
          "lat", g.lib.sf3d_get_node_boundary_latent_flux(int(i)), o.lib.sf3d_get_node_boundary_latent_flux(int(i)),
          "evap", g.lib.sf3d_get_node_boundary_water_flow(int(i)), o.lib.sf3d_get_node_boundary_water_flow(int(i)))
hs = np.flatnonzero(m.btype == capi.BND_HEAT_SURFACE)
print("n heat surface", len(hs), "frac with big dT", np.mean(d[hs] > 1e-5))
