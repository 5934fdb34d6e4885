"""BASELINE config 5 as specified - the Ravone PROJECT (DEM + soil map + soil database + land use through
criteria3d_amd.project3d, pinned in tests/test_project3d.py) - on the HIP product against the oracle, into the runoff regime."""
from pathlib import Path

import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm
from tests.scenarios import ravone_project_model
from tests.tolerances import WATER_RTOL, assert_water_nodes

pytestmark = pytest.mark.gpu
COUNTERS = ("attempts", "accepted", "approximations", "sweeps", "courant_rejections", "linear_failures", "restores")


def _compare(product, oracle, m, what, base=None):
    """base: (product counters, oracle counters) at the hand-over - the work since then is compared"""
    g, o = cm.snapshot(product, m), cm.snapshot(oracle, m)
    rel = np.max(np.abs(g["H"] - o["H"]) / np.maximum(np.abs(o["H"]), 1e-9))
    assert_water_nodes(g["H"], o["H"], f"{what}: H")
    assert_water_nodes(g["Se"], o["Se"], f"{what}: Se")
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        assert abs(g[k] - o[k]) <= WATER_RTOL * max(abs(o[k]), 1e-3), (what, k, g[k], o[k])
    gc, oc = product.counters(), oracle.counters()
    if base is not None:
        gc = {k: gc[k] - base[0][k] for k in gc}; oc = {k: oc[k] - base[1][k] for k in oc}
    for k in COUNTERS:
        assert gc[k] == oc[k], (what, k, gc, oc)
    return rel


@pytest.mark.parametrize("window,min_steps,need_restores", [((980, 1108, 300, 428), 1000, True), ((600, 728, 150, 278), 2000, False)])
def test_project_window_two_hours_match_oracle(product, oracle, window, min_steps, need_restores):
    import os
    if not need_restores and os.environ.get("SF3D_LONG_TESTS") != "1":
        pytest.skip("the second 128 x 128 window (no restore-best steps; 2 000 steps, half a minute of oracle time): SF3D_LONG_TESTS=1 - the suite's budget (round 5's review, item 8)")
    """128 x 128 windows of the project, the 25 mm hour and the dry hour after it, both in full.  Rows 980:1108 / cols 300:428: the
    catchment's edge (36 % outside), four soils of the map incl. BSC (0.5 m: short columns), Courant rejections, restore-best steps;
    rows 600:728 / cols 150:278: three soils, twice as many steps at smaller dt.  H and the cumulative balances within 1e-9
    (tests/tolerances.py; north_star: 1e-6), every accepted dt and every work counter identical after each hour.  (The window where the
    trajectory sits on the air-entry kink of the retention curve - which rounds 3-4 could not hold to any band - is held to the same
    1e-9 in tests/test_gpu_sensitivity.py since the kernels evaluate the C library's own log / pow / cbrt.)"""
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


def _check_segment(seg, what, rtol, counters=COUNTERS):
    assert seg["dts_equal"], (what, "accepted dt sequences differ")
    assert seg["rel_H"] < rtol, (what, seg["rel_H"])
    assert seg["abs_Se"] < max(rtol, 1e-9), (what, seg["abs_Se"])
    from tests.tolerances import water_nodes_exact
    if water_nodes_exact() and "rel_T" not in seg:          # water path, default build: the checker's bits in every one of the 5.85 M nodes
        assert seg["H_bits_equal"] and seg["Se_bits_equal"], (what, seg["rel_H"], seg["abs_Se"])
    for q, (g, o) in seg["scalars"].items():
        assert abs(g - o) <= rtol * max(abs(o), 1e-3), (what, q, g, o)
    for q in counters:
        assert seg["work_product"][q] == seg["work_checker"][q], (what, q, seg["work_product"], seg["work_checker"])
    return seg["rel_H"]


@pytest.mark.fullsize_background("water")
def test_project_full_size_runoff_regime_matches_oracle(fullsize_results):
    """The whole project (5.85 M nodes, 422 282 columns).  The product alone runs the 25 mm hour (1 650 computeStep calls, down to
    dt = 1 s; 6 s of GPU time - the oracle would need an hour); its state at the end of that hour - H of every node and the adaptive
    time step - is handed to the checkers through the state setters (the application's own restart path,
    criteria3DProject.cpp:2934-3123).  From that hand-over the product takes 300 UNINTERRUPTED computeStep calls of the dry hour, where
    the time step falls to its minimum and a group of columns crosses the air-entry kink of its retention curve ~60 steps in; it is
    held against the glibc oracle - the pin -
      (a) after the first 50 of them: H within 1e-9, identical accepted dt, identical work counters;
      (b) after all 300: H within 1e-9, identical dt and counters.  (Rounds 3-4, whose log / pow / cbrt were 0.50-ulp routines of their
          own, were 1.2e-6 off at step 80 and 2.2e-4 at step 300 and could hold (b) only against a twin of the oracle built with those
          routines; the default build now evaluates the C library's functions bit for bit - tests/test_glibcmath.py.  With a
          -DSF3D_LIBM_GLIBC=0 build loaded, (b) is held against that twin as before.)
      (c) after 100 more steps alone the product hands its state over a second time and is held against the glibc oracle for 100
          steps there (restore-best steps at the minimum time step): 1e-9, identical dt and counters.
    The runs themselves - seconds of GPU time, minutes of oracle time (8 + 4 threads side by side) - are made by
    tests/fullsize_worker.py, a background process that tests/conftest.py starts right after collection: this test runs last and
    only reads the metrics."""
    assert "water" in fullsize_results, fullsize_results.get("_log")
    w = fullsize_results["water"]
    assert w["surface_nodes"] == 422282 and w["nodes"] > 5_000_000 and w["finite"]
    assert w["hour0_steps"] > 1000 and w["hour0_courant_rejections"] > 0
    assert w["pin"]["steps"] == 50 and w["uninterrupted"]["steps"] == 300 and w["late"]["steps"] == 100
    r_pin = _check_segment(w["pin"], "glibc oracle, steps 1-50 from the hour boundary", WATER_RTOL)
    r_late = _check_segment(w["late"], "glibc oracle, 100 steps from the second hand-over", WATER_RTOL)
    r_long = _check_segment(w["uninterrupted"], f"{w['uninterrupted_checker']}, 300 uninterrupted steps", 1e-9)
    print(f"full size: vs glibc oracle {r_pin:.2e} (50 steps), {r_late:.2e} (late 100); vs {w['uninterrupted_checker']} {r_long:.2e} (300 uninterrupted); "
          f"worker: {w['seconds_product']:.0f} s product, {w['seconds_total']:.0f} s in all")
    assert w["late"]["work_checker"]["restores"] + w["uninterrupted"]["work_checker"]["restores"] > 0
