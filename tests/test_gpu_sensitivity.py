"""Is the separation of the HIP path from the oracle on the ill-conditioned stretches of BASELINE config 5 arithmetic sensitivity of the
scheme, or a kernel effect?  (round 3's review, "weak": on the 128 x 128 project window rows 72:200 / cols 300:428 the product leaves
the glibc oracle at 7.7e-4 after ~700 steps while an FMA-contracted build of the oracle needs 2 400 steps to reach 3e-5.)

The instrument is a TWIN of the oracle built with the product's own elementary functions (`make -C oracle oracle-fm`: log / pow / exp /
cbrt of criteria3d_amd/csrc/sf3d_fastmath.inc - the host build of the text the kernels compile, device == host bit for bit in
tests/test_gpu_fastmath.py - and sqrt for Se^0.5 as in k_props).  Neither side contracts multiply-adds, so the HIP path and the twin
run the same operations on the same operands and differ ONLY in the order of their reductions (block-tree sums against index-order
sums of the Jacobi norm and the balance terms).  If the product stays on the twin through the stretches where it leaves the glibc
oracle, the separation is the last-ulp difference of the table routines against glibc amplified by the scheme (the air-entry branch
psi <= he of soilPhysics.cpp:68-279), not the kernels.  The twin is never the pin: results are pinned by the glibc oracle elsewhere."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from criteria3d_amd import catchment as cm
from tests import checkers
from tests.scenarios import ravone_project_model

pytestmark = pytest.mark.gpu
COUNTERS = ("attempts", "accepted", "approximations", "sweeps", "courant_rejections", "linear_failures", "restores")
TWIN_RTOL = 1e-9          # HIP vs twin: reduction order only


@pytest.fixture(scope="module")
def twin():
    return checkers.load_oracle_fastmath()


def rel_h(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-9)))


def test_twin_is_the_oracle_with_other_elementary_functions(oracle, twin):
    """sanity of the instrument on a well-conditioned case (C2 F20, hours 0-1): the twin follows the glibc oracle within the usual
    1e-6 with identical decisions - it is the same algorithm - but not bit for bit: its elementary functions are the product's"""
    m = cm.catchment_model(64, 64, 10)
    out = []
    for sf in (oracle, twin):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=8)
        dts = []
        for mm in (20.0, 0.0):
            dts += cm.run_hour(sf, m, mm)[1]
        out.append((np.array(dts), sf.total_potential(0, m.n), sf.counters()))
        sf.lib.sf3d_clean()
    assert np.array_equal(out[0][0], out[1][0]) and all(out[0][2][k] == out[1][2][k] for k in COUNTERS)
    assert 0 < rel_h(out[1][1], out[0][1]) < 1e-6


def test_kink_window_product_stays_on_the_twin(product, oracle, twin):
    """The window the suite cannot hold against the glibc oracle (rows 72:200 / cols 300:428: a group of nodes sits at the air-entry
    potential of its soil through the dry hour): the 25 mm hour and the dry hour, in lock step on the product, the twin and the glibc
    oracle.  Product vs twin: H within 1e-9 and every accepted dt and every work counter identical.  The glibc oracle, stepped
    alongside, is only reported: it leaves the 1e-6 band - and then the common dt sequence - between steps 600 and 700, ~100 steps
    into the dry hour.  By default the run stops at step 800, past that point (the product is then 7.7e-4 from the glibc oracle
    and still ON the twin); SF3D_LONG_TESTS=1 runs the whole two hours (7 262 steps, 12 minutes of oracle time:
    profiles/r04_sensitivity_product_vs_twin_kink_window.log - product vs twin 0.00e+00, bit for bit, to the end)."""
    import os
    limit = 10**9 if os.environ.get("SF3D_LONG_TESTS") == "1" else 800
    m = ravone_project_model((72, 200, 300, 428))
    libs = (product, twin, oracle)
    for sf in libs:
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=8)
    worst_twin, worst_glibc, steps, glibc_alive = 0.0, 0.0, 0, True
    with ThreadPoolExecutor(3) as pool:          # the three libraries step side by side (ctypes releases the interpreter lock)
        for h, mm in enumerate((25.0, 0.0)):
            for sf in libs:
                sf.set_sink_source_bulk(0, np.full(m.ns, cm.rain_rate(mm, m.cell_area)))
            t = 0.0
            while t < 3600.0 and steps < limit:
                live = libs if glibc_alive else libs[:2]
                dts = list(pool.map(lambda sf: sf.lib.sf3d_compute_step(3600.0 - t), live))
                assert dts[0] == dts[1] and dts[0] > 0, (h, steps, dts)
                if glibc_alive and dts[2] != dts[0]:
                    glibc_alive = False          # the glibc oracle has taken another decision: from here on it is another trajectory
                    print(f"glibc oracle leaves the common dt sequence at step {steps} (hour {h})")
                t += dts[0]; steps += 1
                if steps % 50 == 0 or t >= 3600.0 or steps == limit:
                    Hp, Ht = product.total_potential(0, m.n), twin.total_potential(0, m.n)
                    worst_twin = max(worst_twin, rel_h(Hp, Ht))
                    assert worst_twin < TWIN_RTOL, (h, steps, worst_twin)
                    cp, ct = product.counters(), twin.counters()
                    assert all(cp[k] == ct[k] for k in COUNTERS), (h, steps, cp, ct)
                    if glibc_alive:
                        worst_glibc = max(worst_glibc, rel_h(Hp, oracle.total_potential(0, m.n)))
    print(f"kink window: {steps} steps; product vs twin {worst_twin:.2e}; product vs glibc oracle {worst_glibc:.2e}"
          f"{'' if glibc_alive else ' (until it left the dt sequence)'}")
    assert steps >= min(limit, 1500) and worst_glibc > 1e-6          # (the glibc oracle HAS left the band by then: that is the point)
    gp, gt = cm.snapshot(product, m), cm.snapshot(twin, m)
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        assert abs(gp[k] - gt[k]) <= 1e-9 * max(abs(gt[k]), 1e-3), (k, gp[k], gt[k])
    for sf in libs:
        sf.lib.sf3d_clean()


@pytest.mark.parametrize("name", ["flows_c2_f20", "flows_c2_f60"])
def test_link_flow_sums_against_the_twin_are_tight(product, twin, name):
    """tests/test_gpu_flows.py holds the per-element link flow sums to 1e-3 of themselves against the glibc oracle (measured 2.6e-4 ...
    9.7e-4 on C2 F20: a flow sum is a conductance times a DIFFERENCE of two 100 m heads, the surface ones go with millimetres of water
    to the power 5/3).  Are those bands kernel error?  Against the twin - same elementary functions - the same sums agree to 1e-9 element
    by element (H itself bit for bit or at the last ulps of a reduction): the 1e-3 is the amplified last-ulp difference of glibc's
    log / pow / cbrt, like the config-5 separation, and the bands of test_gpu_flows.py are measurements of that, not tolerances for
    the kernels."""
    from tests import scenarios as sc
    g = sc.run_scenario(product, name)
    t = sc.run_scenario(twin, name)
    assert np.array_equal(g["dts"], t["dts"]) and list(g["steps_per_hour"]) == list(t["steps_per_hour"])
    for k in t:
        if k.startswith("H_h"):
            assert rel_h(g[k], t[k]) < TWIN_RTOL, (k, rel_h(g[k], t[k]))
    worst = 0.0
    for k, fname in enumerate(cm.LINK_FLOW_FIELDS):
        a, b = np.asarray(g["link_flows"][k]), np.asarray(t["link_flows"][k])
        scale = max(np.max(np.abs(b)), 1e-12)
        big = np.abs(b) > 1e-3 * scale
        assert np.max(np.abs(a - b)) <= 1e-9 * scale, (fname, np.max(np.abs(a - b)) / scale)
        if np.any(big):
            r = float(np.max(np.abs(a[big] - b[big]) / np.abs(b[big])))
            worst = max(worst, r)
            assert r < 1e-9, (fname, r)
    print(f"{name}: link flow sums, product vs twin, worst element-wise {worst:.2e}")
    product.lib.sf3d_clean(); twin.lib.sf3d_clean()


def test_headline_grid_hour0_on_the_twin(product, twin):
    """the headline grid (C4 512 x 512 x 20, F20) for its first hour - 22 steps, 5.24 M nodes, the paired sweep, norms and balance sums
    reduced over 2 048 blocks on the device and in index order on the CPU: against the twin H within 1e-9 (measured: see the printed
    line), identical accepted dt and counters.  What separates the product from the glibc oracle on this grid (2e-10 ... 6e-8) is the
    elementary functions, as everywhere else."""
    m = cm.catchment_model(512, 512, 20)
    out = []
    for sf in (product, twin):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=16)
        _, dts = cm.run_hour(sf, m, 20.0)
        out.append((np.array(dts), cm.snapshot(sf, m), sf.counters()))
        sf.lib.sf3d_clean()
    (gd, g, gc), (td, t, tc) = out
    assert np.array_equal(gd, td) and len(gd) == 22
    r = rel_h(g["H"], t["H"])
    print(f"C4 F20 hour 0: product vs twin max |dH|/H = {r:.2e}, bit-identical H: {np.array_equal(g['H'], t['H'])}, storage {g['storage']!r} vs {t['storage']!r}")
    assert r < TWIN_RTOL
    for k in COUNTERS:
        assert gc[k] == tc[k], (k, gc, tc)
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        assert abs(g[k] - t[k]) <= 1e-9 * max(abs(t[k]), 1e-3), (k, g[k], t[k])
