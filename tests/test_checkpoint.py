"""Resume on the exact trajectory: node potentials through the ordinary getters/setters (what the
application's WP_<depth>.flt state files hold, criteria3DProject.cpp:2260-2307, 2934-3123) plus the
adaptive time step through sf3d_get_time_step / sf3d_set_time_step."""
import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm


def continuous_and_resumed(sf, m, hours_before=1, hours_after=1, mm=(20.0, 0.0)):
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
    cm.build(sf, m)
    for h in range(hours_before + hours_after):
        cm.run_hour(sf, m, mm[min(h, len(mm) - 1)])
    cont = cm.snapshot(sf, m)

    sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
    cm.build(sf, m)
    for h in range(hours_before):
        cm.run_hour(sf, m, mm[min(h, len(mm) - 1)])
    H = sf.total_potential(0, m.n)
    dt = sf.lib.sf3d_get_time_step()
    # "new process": rebuild the model, impose the saved potentials and the saved time step
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
    cm.build(sf, m)
    sf.set_total_potential_bulk(0, H)
    sf.check(sf.lib.sf3d_set_time_step(dt), "set_time_step")
    sf.check(sf.lib.sf3d_initialize_balance(), "balance")
    for h in range(hours_before, hours_before + hours_after):
        cm.run_hour(sf, m, mm[min(h, len(mm) - 1)])
    return cont, cm.snapshot(sf, m)


def test_oracle_resumes_bit_for_bit(oracle):
    m = cm.catchment_model(24, 24, 6)
    cont, res = continuous_and_resumed(oracle, m)
    assert np.array_equal(cont["H"], res["H"]) and np.array_equal(cont["Se"], res["Se"])
    assert cont["total_water"] == res["total_water"]
    assert oracle.lib.sf3d_set_time_step(-1.0) == capi.PARAMETER_ERROR


@pytest.mark.gpu
def test_state_directory_written_by_the_product_resumes_product_and_oracle(product, oracle, tmp_path):
    """SURVEY.md 8f-3 on the device, against the oracle.  The product runs hour 0 of C2 F20 and writes the application's state
    directory - one WP_<depth cm>.flt of float32 matric potentials per layer (Crit3DProject::saveSoilWaterState,
    criteria3DProject.cpp:2260-2307) plus the adaptive time step; a FRESH product and a FRESH oracle are built, load that directory
    (loadWaterPotentialState, :2934-3123), call initializeBalance as the application does and run the dry hour: identical accepted dt,
    H / Se / balances within 1e-6 (measured: see the printed line).  The oracle run through the same first hour writes the same
    directory up to the last float32 bit of a handful of cells."""
    from criteria3d_amd import esri
    m = cm.catchment_model(64, 64, 10)
    hdr = dict(xllcorner=0.0, yllcorner=0.0, cellsize=10.0)
    written = {}
    for name, sf in (("product", product), ("oracle", oracle)):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=8)
        cm.run_hour(sf, m, 20.0)
        water = esri.save_water_state(sf, m, tmp_path / name, hdr)
        psi = sf.total_potential(0, m.n) - m.z
        grids = {f.name: esri.read_grid(f)[0] for f in sorted(water.glob("WP_*.flt"))}
        assert len(grids) == 10 and set(grids) == {f"WP_{c}.flt" for c in (0, 5, 15, 25, 35, 45, 55, 65, 75, 85)}
        for l, c in enumerate((0, 5, 15, 25, 35, 45, 55, 65, 75, 85)):        # the files hold exactly the float32 of the library's potentials
            assert np.array_equal(grids[f"WP_{c}.flt"].ravel(), psi[m.meta["index"][l].ravel()].astype(np.float32))
        written[name] = (grids, float((water / "deltaT.txt").read_text()), sf.lib.sf3d_get_time_step())
        sf.lib.sf3d_clean()
    assert written["product"][1] == written["product"][2] == written["oracle"][1]          # the adaptive time step, text round trip exact
    flips = 0
    for k, g in written["product"][0].items():
        o = written["oracle"][0][k]
        assert np.max(np.abs(g.astype(np.float64) - o.astype(np.float64)) / np.maximum(np.abs(o), 1e-6)) < 2.5e-7      # one float32 ulp
        flips += int(np.count_nonzero(g != o))
    assert flips < 0.001 * m.n, flips

    out = {}
    for name, sf in (("product", product), ("oracle", oracle)):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=8)
        levels = esri.load_water_state(sf, m, tmp_path / "product")
        assert levels == [0, 5, 15, 25, 35, 45, 55, 65, 75, 85]
        assert sf.lib.sf3d_get_time_step() == written["product"][1]
        sf.check(sf.lib.sf3d_initialize_balance(), "initialize_balance")
        loaded = sf.total_potential(0, m.n) - m.z
        _, dts = cm.run_hour(sf, m, 0.0)
        out[name] = (loaded, np.array(dts), cm.snapshot(sf, m), sf.counters())
        sf.lib.sf3d_clean()
    (gl, gd, g, gc), (ol, od, o, oc) = out["product"], out["oracle"]
    assert np.array_equal(gl, ol)                                    # both hold the float32 values of the files
    assert np.array_equal(gd, od) and len(gd) > 0
    r = float(np.max(np.abs(g["H"] - o["H"]) / np.maximum(np.abs(o["H"]), 1e-9)))
    print(f"resumed from the product's WP_*.flt directory: product vs oracle after the dry hour, {len(gd)} steps, max |dH|/H = {r:.2e}")
    assert r < 1e-6 and np.max(np.abs(g["Se"] - o["Se"])) < 1e-6
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        assert abs(g[k] - o[k]) <= 1e-6 * max(abs(o[k]), 1e-3), (k, g[k], o[k])
    for k in ("attempts", "accepted", "approximations", "sweeps", "courant_rejections", "linear_failures", "restores"):
        assert gc[k] == oc[k], (k, gc, oc)


@pytest.mark.gpu
def test_coarser_state_directory_loads_by_the_reference_rule_on_the_device(product, oracle, tmp_path):
    """a state directory with FEWER depth levels than the model has layers (written by a run with another layering): a layer between
    two levels takes the deeper level's value (the reference's integer division, criteria3DProject.cpp:3039-3043), a cell whose level
    holds NODATA the first valid level above.  Loaded into the product and into the oracle: the same potentials in both, and the same
    trajectory from them (an hour under 5 mm of rain, 1e-6, identical dt)."""
    from criteria3d_amd import esri
    m = cm.catchment_model(32, 24, 10)
    hdr = dict(xllcorner=0.0, yllcorner=0.0, cellsize=10.0, nodata=-9999.0)
    water = tmp_path / "state" / "water"
    water.mkdir(parents=True)
    rng = np.random.default_rng(5)
    levels = {0: 0.0, 25: -1.2, 65: -2.1}
    for cmv, val in levels.items():
        g = (val + 0.2 * rng.random((24, 32))).astype(np.float32) if cmv else np.zeros((24, 32), np.float32)
        if cmv == 65:
            g[3, 4] = -9999.0
        esri.write_grid(water / f"WP_{cmv}", g, hdr)
    (water / "deltaT.txt").write_text("37.5\n")
    out = {}
    for name, sf in (("product", product), ("oracle", oracle)):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=4)
        assert esri.load_water_state(sf, m, tmp_path / "state") == [0, 25, 65]
        assert sf.lib.sf3d_get_time_step() == 37.5
        sf.check(sf.lib.sf3d_initialize_balance(), "initialize_balance")
        psi = sf.total_potential(0, m.n) - m.z
        _, dts = cm.run_hour(sf, m, 5.0, max_steps=30)
        out[name] = (psi, np.array(dts), sf.total_potential(0, m.n))
        sf.lib.sf3d_clean()
    idx = m.meta["index"]
    lv = {c: esri.read_grid(water / f"WP_{c}")[0] for c in levels}
    psi = out["product"][0]
    assert np.allclose(psi[idx[2]], lv[25], atol=1e-6)                # depth 15 cm: between 0 and 25 -> the deeper level
    assert np.allclose(psi[idx[3]], lv[25], atol=1e-6)                # depth 25 cm: the level itself
    want = lv[65].copy(); want[3, 4] = lv[25][3, 4]                   # the hole takes the level above
    assert np.allclose(psi[idx[5]], want, atol=1e-6) and np.allclose(psi[idx[9]], want, atol=1e-6)
    assert np.array_equal(out["product"][0], out["oracle"][0])
    assert np.array_equal(out["product"][1], out["oracle"][1]) and len(out["product"][1]) >= 20
    assert np.max(np.abs(out["product"][2] - out["oracle"][2]) / np.maximum(np.abs(out["oracle"][2]), 1e-9)) < 1e-6
