"""BASELINE config 5 as specified - the Ravone PROJECT (DEM + soil map + soil database + land use through
criteria3d_amd.project3d, pinned in tests/test_project3d.py) - on the HIP product against the oracle, into the runoff regime."""
from pathlib import Path

import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm
from tests.scenarios import ravone_project_model

pytestmark = pytest.mark.gpu
COUNTERS = ("attempts", "accepted", "approximations", "sweeps", "courant_rejections", "linear_failures", "restores")


def _compare(product, oracle, m, what, base=None):
    """base: (product counters, oracle counters) at the hand-over - the work since then is compared"""
    g, o = cm.snapshot(product, m), cm.snapshot(oracle, m)
    rel = np.max(np.abs(g["H"] - o["H"]) / np.maximum(np.abs(o["H"]), 1e-9))
    assert rel < 1e-6, (what, rel)
    assert np.max(np.abs(g["Se"] - o["Se"])) < 1e-6, what
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        assert abs(g[k] - o[k]) <= 1e-6 * max(abs(o[k]), 1e-3), (what, k, g[k], o[k])
    gc, oc = product.counters(), oracle.counters()
    if base is not None:
        gc = {k: gc[k] - base[0][k] for k in gc}; oc = {k: oc[k] - base[1][k] for k in oc}
    for k in COUNTERS:
        assert gc[k] == oc[k], (what, k, gc, oc)
    return rel


@pytest.mark.parametrize("window,min_steps,need_restores", [((980, 1108, 300, 428), 1000, True), ((600, 728, 150, 278), 2000, False)])
def test_project_window_two_hours_match_oracle(product, oracle, window, min_steps, need_restores):
    """128 x 128 windows of the project, the 25 mm hour and the dry hour after it, both in full.  Rows 980:1108 / cols 300:428: the
    catchment's edge (36 % outside), four soils of the map incl. BSC (0.5 m: short columns), Courant rejections, restore-best steps;
    rows 600:728 / cols 150:278: three soils, twice as many steps at smaller dt.  H and the cumulative balances within 1e-6, every
    accepted dt and every work counter identical after each hour.  (Windows where the trajectory sits on the air-entry kink of the
    retention curve separate even between two CPU builds of the oracle - profiles/README.md "sensitivity" - and cannot be held to
    any band: scripts/experiments/c5_window_diverge.py, oracle_fma_sensitivity.py.)"""
    m = ravone_project_model(window)
    assert m.ns > 10000 and m.n > 100000
    for sf in (product, oracle):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=16)
    for h, mm in enumerate((25.0, 0.0)):
        _, gd = cm.run_hour(product, m, mm)
        _, od = cm.run_hour(oracle, m, mm)
        assert len(gd) == len(od), (h, len(gd), len(od))
        np.testing.assert_allclose(gd, od, rtol=1e-12)
        _compare(product, oracle, m, f"hour {h}")
    c = oracle.counters()
    assert c["accepted"] > min_steps and c["courant_rejections"] > 0 and (c["restores"] > 0 or not need_restores), c
    oracle.lib.sf3d_clean(); product.lib.sf3d_clean()


def _segment(sf, m, H0, dt0, steps, threads=16):
    """build, take over (H, dt) through the state setters, `steps` computeStep calls of the dry hour: what is compared afterwards"""
    sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
    cm.build(sf, m, threads=threads)
    sf.set_total_potential_bulk(0, H0)
    sf.check(sf.lib.sf3d_set_time_step(dt0), "set_time_step")
    sf.check(sf.lib.sf3d_initialize_balance(), "initialize_balance")
    base = sf.counters()
    _, dts = cm.run_hour(sf, m, 0.0, max_steps=steps)
    c = sf.counters()
    return {"dts": np.array(dts), "snap": cm.snapshot(sf, m), "work": {k: c[k] - base[k] for k in c}}


def test_project_full_size_runoff_regime_matches_oracle(product, oracle):
    """The whole project (5.85 M nodes, 422 282 columns).  The product alone runs the 25 mm hour (1 650 computeStep calls, down to
    dt = 1 s; 6 s of GPU time - the oracle would need an hour); its state at the end of that hour - H of every node and the adaptive
    time step - is then handed to BOTH libraries through the state setters (the application's own restart path,
    criteria3DProject.cpp:2934-3123), and both take the first 120 computeStep calls of the dry hour from there, where the time step
    falls to its minimum and restore-best steps occur.  The product then goes on alone for 280 steps, hands its state over a second
    time, and both take 180 more.  In each segment: H within 1e-6, identical accepted dt, identical work counters - 300 compared
    steps in all.  The oracle needs ~0.6 s per step at this size on 16 threads and 0.7 s on 8: its two segments run side by side on 8
    threads each, each on an instance of the oracle library of its own (the product has produced both hand-over states by then; the
    GPU boxes of this pool give a container 16 CPUs' worth of time - scripts/experiments/oracle_two_instances.py).  Two segments instead of one
    run of 300: a group of columns of this catchment crosses the air-entry kink of its retention curve ~60 steps into the dry hour
    and from there separates even CPU build from CPU build (profiles/README.md "sensitivity"); one uninterrupted run of
    300 steps from the hour boundary ends at 2.2e-4 there, 120 steps and any later stretch stay below 1e-6."""
    from concurrent.futures import ThreadPoolExecutor
    from tests import checkers
    m = ravone_project_model(None)
    assert m.ns == 422282 and m.n > 5_000_000
    product.check(product.lib.sf3d_reset_solver_state(), "reset")
    cm.build(product, m)
    n0, _ = cm.run_hour(product, m, 25.0)
    warm = product.counters()
    assert n0 > 1000 and warm["courant_rejections"] > 0
    plan = ((120, 280), (180, 0))
    states, got = [], []
    H0, dt0 = product.total_potential(0, m.n), product.lib.sf3d_get_time_step()
    for steps, alone in plan:
        assert np.all(np.isfinite(H0))
        states.append((H0, dt0))
        got.append(_segment(product, m, H0, dt0, steps))
        if alone:
            cm.run_hour(product, m, 0.0, max_steps=alone)
            H0, dt0 = product.total_potential(0, m.n), product.lib.sf3d_get_time_step()
    product.lib.sf3d_clean()
    second = checkers.load_oracle_copy("second")
    with ThreadPoolExecutor(2) as pool:          # (ctypes calls release the interpreter lock)
        jobs = [pool.submit(_segment, sf, m, st[0], st[1], steps, 8) for sf, st, (steps, _) in zip((oracle, second), states, plan)]
        want = [j.result() for j in jobs]
    restores = 0
    for k, (g, o) in enumerate(zip(got, want)):
        what = f"segment {k}: {plan[k][0]} steps"
        np.testing.assert_allclose(g["dts"], o["dts"], rtol=1e-12, err_msg=what)
        rel = np.max(np.abs(g["snap"]["H"] - o["snap"]["H"]) / np.maximum(np.abs(o["snap"]["H"]), 1e-9))
        assert rel < 1e-6, (what, rel)
        assert np.max(np.abs(g["snap"]["Se"] - o["snap"]["Se"])) < 1e-6, what
        for q in ("total_water", "storage", "runoff", "drainage", "lateral"):
            assert abs(g["snap"][q] - o["snap"][q]) <= 1e-6 * max(abs(o["snap"][q]), 1e-3), (what, q, g["snap"][q], o["snap"][q])
        for q in COUNTERS:
            assert g["work"][q] == o["work"][q], (what, q, g["work"], o["work"])
        restores += o["work"]["restores"]
    assert restores > 0
    oracle.lib.sf3d_clean(); second.lib.sf3d_clean()
