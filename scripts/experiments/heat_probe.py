"""Exploration: drive a backend (reference by default) with heat enabled on a soil column."""
import sys, numpy as np
sys.path.insert(0, "/root/repo")
from criteria3d_amd import capi, catchment
from tests import checkers

which = sys.argv[1] if len(sys.argv) > 1 else "reference"
water = int(sys.argv[2]) if len(sys.argv) > 2 else 1
adv = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lat = int(sys.argv[4]) if len(sys.argv) > 4 else 1
sf = {"reference": checkers.load_reference, "oracle": checkers.load_oracle}[which]()
sf.lib.sf3d_reset_solver_state()
m = catchment.column_model(22, 0.05, 1.0)
L = sf.lib
n = m.n
m.btype[1] = 8  # HeatSurface on the first soil node
m.barea[1] = 1.0
sf.check(L.sf3d_initialize(n, m.ns, 8, water, 1, 0, 2), "init")
sf.check(L.sf3d_initialize_heat_flag(2, adv, lat), "heatflag")
sf.check(L.sf3d_set_surface_properties(0, m.roughness), "surf")
s = m.soils[0]
sf.check(L.sf3d_set_soil_properties(0, 0, s["alpha"], s["n"], 1 - 1 / s["n"], s["he"], s["theta_r"], s["theta_s"], s["ksat"], s["L"], s["organic_matter"], s["clay"]), "soil")
sf.set_nodes_bulk(0, m.x, m.y, m.z, m.size, m.is_surface, m.btype, m.bslope, m.barea)
sf.set_links_bulk(m.link_node, m.link_to, m.link_dir, m.link_area)
sf.set_surface_bulk(0, np.zeros(m.ns, np.uint16)); sf.set_pond_bulk(0, np.full(m.ns, m.pond))
sf.set_soil_bulk(m.ns, m.soil_index, np.zeros(n - m.ns, np.uint16))
sf.check(L.sf3d_set_hydraulic_properties(capi.WRC_MODIFIED_VG, capi.MEAN_LOGARITHMIC, 10.0), "hyd")
sf.check(L.sf3d_set_numerical_parameters(1.0, 3600.0, 150, 10, 10, 3), "num")
L.sf3d_set_threads_number(1)
psi = np.full(n, -3.0); psi[0] = 0
sf.set_matric_potential_bulk(0, psi)
for i in range(n):
    sf.check(L.sf3d_set_node_temperature(i, 288.15 + 0.1 * i), "T")
for name, v in (("height_wind", 2.0), ("height_temperature", 2.0), ("roughness", 0.01), ("temperature", 293.15), ("relative_humidity", 60.0), ("wind_speed", 2.0), ("net_irradiance", 150.0)):
    sf.check(getattr(L, "sf3d_set_node_boundary_" + name)(1, v), name)
sf.check(L.sf3d_set_node_boundary_fixed_temperature(n - 1, 285.15, 0.5), "fixedT")
sf.check(L.sf3d_initialize_balance(), "bal")
for h in range(3):
    L.sf3d_set_node_water_sink_source(0, float(sys.argv[5]) / 3600 if (len(sys.argv) > 5 and h == 0) else 0.0)
    t = 0; steps = 0
    while t < 3600:
        dt = L.sf3d_compute_step(3600 - t); t += dt; steps += 1
    T = [L.sf3d_get_node_temperature(i) for i in range(n)]
    print(h, steps, "T", np.array(T[1:6]), "H1", L.sf3d_get_node_total_potential(1), "heatMBR", L.sf3d_get_heat_mbr(),
          "sens", L.sf3d_get_node_boundary_sensible_flux(1), "lat", L.sf3d_get_node_boundary_latent_flux(1), "aero", L.sf3d_get_node_boundary_aerodynamic_conductance(1),
          "flux", L.sf3d_get_node_heat_max_flux(2, 1, 0), L.sf3d_get_node_heat_max_flux(2, 2, 1))
