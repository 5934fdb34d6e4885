"""Scenario definitions shared by the golden-vector generator and the tests.

Each scenario builds a model through the public ABI on a given backend and returns a flat dict
of numpy arrays (the "trace"): per-hour scalars, accepted-dt sequences and full H / Se arrays at
selected hours.  Reference-generated traces are committed under tests/golden/."""
import numpy as np

from criteria3d_amd import capi, catchment as cm

SCALARS = ("total_water", "storage", "mbr", "runoff", "drainage", "lateral")


def _hours(sf, m, plan, threads=1, use_period=False, wrc=None, mean=None, sinks_fn=None, pre=None):
    """plan: list of (rain_mm, max_steps or None, keep_arrays)"""
    sf.lib.sf3d_reset_solver_state()
    cm.build(sf, m, threads=threads)
    if wrc is not None or mean is not None:
        sf.check(sf.lib.sf3d_set_hydraulic_properties(capi.WRC_MODIFIED_VG if wrc is None else wrc,
                                                      capi.MEAN_LOGARITHMIC if mean is None else mean, m.lv_ratio), "hyd")
        # potentials were converted with the previous curve: impose them again, then rebalance
        psi = np.full(m.n, m.psi0_soil); psi[:m.ns] = m.psi0_surface
        sf.set_matric_potential_bulk(0, psi)
        sf.check(sf.lib.sf3d_initialize_balance(), "balance")
    if pre is not None:
        pre(sf, m)
    out = {}
    scal = {k: [] for k in SCALARS}
    nsteps, alldts = [], []
    for h, (mm, max_steps, keep) in enumerate(plan):
        if sinks_fn is not None:
            steps, dts = cm.run_hour_sinks(sf, m, sinks_fn(h, m), max_steps=max_steps)
        else:
            steps, dts = cm.run_hour(sf, m, mm, use_period=use_period, max_steps=max_steps)
        s = cm.snapshot(sf, m)
        for k in SCALARS:
            scal[k].append(s[k])
        if dts is not None:
            nsteps.append(steps); alldts.extend(dts)
        if keep:
            out[f"H_h{h}"] = s["H"]; out[f"Se_h{h}"] = s["Se"]
    for k in SCALARS:
        out[k] = np.array(scal[k])
    out["steps_per_hour"] = np.array(nsteps, np.int64)
    out["dts"] = np.array(alldts)
    return out


def c1_column(sf, threads=1):
    m = cm.column_model()
    return _hours(sf, m, [(5.0, None, h in (0, 23)) for h in range(24)], threads)


def c1_column_period(sf, threads=1):
    m = cm.column_model()
    return _hours(sf, m, [(5.0, None, h == 5) for h in range(6)], threads, use_period=True)


def c2_f20(sf, threads=1):
    m = cm.catchment_model(64, 64, 10)
    return _hours(sf, m, [(cm.FORCINGS["F20"](h), None, h in (0, 5)) for h in range(6)], threads)


def c2_f60(sf, threads=1):
    m = cm.catchment_model(64, 64, 10)
    return _hours(sf, m, [(60.0, None, True), (0.0, 400, True)], threads)


def het_patches(sf, threads=1):
    m = cm.catchment_model(32, 32, 6, heterogeneous=True)
    return _hours(sf, m, [(20.0, None, True), (0.0, 300, True)], threads)


def _ragged_sinks(h, m):
    s = np.zeros(m.n)
    if h == 0:
        s[:m.ns] = cm.rain_rate(10.0, m.cell_area)
    else:
        s[:m.ns] = -2.0e-7                      # evaporation demand on the surface (clamped by water.cpp:646-652)
        s[m.ns:m.ns + m.ns // 2] = -1.0e-8      # root uptake in part of the first soil layer
    return s


def _ragged_pre(sf, m):
    sf.check(sf.lib.sf3d_set_node_prescribed_total_potential(m.meta["prescribed_node"], m.meta["prescribed_H"]), "prescribed")


def ragged_edge_cases(sf, threads=1):
    """holes, short columns, 3 soils, prescribed-potential boundary, evaporation + uptake sinks"""
    m = cm.ragged_model()
    return _hours(sf, m, [(0, None, True), (0, None, True), (0, None, True)], threads, sinks_fn=_ragged_sinks, pre=_ragged_pre)


def ragged_arithmetic_vg(sf, threads=1):
    """same graph with the plain van Genuchten curve and the arithmetic conductivity mean"""
    m = cm.ragged_model()
    return _hours(sf, m, [(0, None, False), (0, None, True)], threads, wrc=capi.WRC_VG, mean=capi.MEAN_ARITHMETIC,
                  sinks_fn=_ragged_sinks, pre=_ragged_pre)


def ragged_geometric(sf, threads=1):
    m = cm.ragged_model()
    return _hours(sf, m, [(0, None, False), (0, None, True)], threads, mean=capi.MEAN_GEOMETRIC,
                  sinks_fn=_ragged_sinks, pre=_ragged_pre)


def ravone_window(sf, threads=1):
    """72x72 window of the Ravone DEM (DATA/DEM/DEM_Ravone.flt, rows 48-119, cols 444-515; 13 % NODATA,
    127 m of relief), graph built like the caller does (catchment.dem_model): 14 soil layers to 0.95 m,
    shallower soil on steep cells, float32-rounded geometry, real-terrain runoff with Courant
    rejections.  Parameters of DATA/PROJECT/Ravone/SETTINGS/parameters.ini (ratio 4, accuracy 2)."""
    from pathlib import Path
    dem = np.load(Path(__file__).resolve().parent / "golden" / "ravone_dem_window_72x72.npy")
    m = cm.dem_model(dem)
    return _hours(sf, m, [(15.0, None, True), (15.0, 300, False), (0.0, 300, True)], threads)


def surface_only(sf, threads=1):
    """no soil at all: 3 mm of ponded water runs off a tilted sheet, then 10 mm of rain"""
    m = cm.surface_only_model()
    return _hours(sf, m, [(0.0, None, True), (10.0, None, True)], threads)


def soil_only(sf, threads=1):
    """nrSurfaceNodes = 0: soil column wetted from a prescribed-potential top node"""
    m = cm.soil_only_column()
    return _hours(sf, m, [(0.0, None, True), (0.0, None, True)], threads, pre=_ragged_pre)


SCENARIOS = {
    "c1_column": c1_column,
    "c1_column_period": c1_column_period,
    "c2_f20": c2_f20,
    "c2_f60": c2_f60,
    "het_patches": het_patches,
    "ragged_edge_cases": ragged_edge_cases,
    "ragged_arithmetic_vg": ragged_arithmetic_vg,
    "ragged_geometric": ragged_geometric,
    "ravone_window": ravone_window,
    "surface_only": surface_only,
    "soil_only": soil_only,
}


def run_scenario(sf, name, threads=1):
    return SCENARIOS[name](sf, threads=threads)
