"""Full-size checks at BASELINE.json's headline configuration (C4 = 512x512x20, 5.24 M nodes, F20).
The GPU box's host is fast enough to run the oracle for hour 0 (22 steps) directly, so the first
test is a straight comparison; the others are size-independent properties of the path:
mass conservation, run-to-run bit-reproducibility, and idempotence of the state round trip."""
import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm
from tests.tolerances import WATER_RTOL, assert_water_nodes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c4():
    return cm.catchment_model(512, 512, 20)


def test_c4_hour0_matches_oracle(product, oracle, c4):
    from tests.scenarios import oracle_c4_f20
    m, ref = oracle_c4_f20(oracle, 1)
    od, o, oc = ref[0]
    product.check(product.lib.sf3d_reset_solver_state(), "reset")
    cm.build(product, m)
    gs, gd = cm.run_hour(product, m, 20.0)
    g = cm.snapshot(product, m)
    assert gs == len(od) == 22
    np.testing.assert_allclose(gd, od, rtol=1e-12)
    r = float(np.max(np.abs(g["H"] - o["H"]) / np.maximum(np.abs(o["H"]), 1e-9)))
    print(f"C4 F20 hour 0: product vs glibc oracle max |dH|/H = {r:.2e}, bit-identical H: {np.array_equal(g['H'], o['H'])}, storage {g['storage']!r} vs {o['storage']!r}")
    assert_water_nodes(g["H"], o["H"], "C4 F20 hour 0: H")
    assert_water_nodes(g["Se"], o["Se"], "C4 F20 hour 0: Se")
    for k in ("total_water", "storage", "runoff", "drainage", "lateral"):
        assert abs(g[k] - o[k]) <= WATER_RTOL * max(abs(o[k]), 1e-3), (k, g[k], o[k])
    gc = product.counters()
    for k in ("attempts", "accepted", "approximations", "sweeps", "courant_rejections"):
        assert gc[k] == oc[k], (k, gc, oc)


def test_c4_mass_conservation_and_reproducibility(product, c4):
    """storage change = rain - boundary outflow to within the solver's own mass-balance threshold,
    and two runs from the same state give bit-identical fields (deterministic reductions)."""
    m = c4
    runs = []
    for _ in range(2):
        product.check(product.lib.sf3d_reset_solver_state(), "reset")
        cm.build(product, m)
        w0 = product.get_total_water_content()
        cm.run_hour(product, m, 20.0)
        cm.run_hour(product, m, 0.0)
        s = cm.snapshot(product, m)
        rain = 20e-3 * m.cell_area * m.ns                      # m3 in hour 0
        out = s["runoff"] + s["drainage"] + s["lateral"]        # negative = leaving the domain
        err = (s["total_water"] - w0) - (rain + out)
        assert abs(err) <= 1e-3 * rain, (err, rain)             # MBR threshold 1e-3 per step, far smaller in sum
        runs.append(s)
    assert np.array_equal(runs[0]["H"], runs[1]["H"]) and np.array_equal(runs[0]["Se"], runs[1]["Se"])
    assert runs[0]["storage"] == runs[1]["storage"]


def test_c4_state_round_trip_is_idempotent(product, c4):
    """getNodeTotalPotential -> setNodeTotalPotential of every node changes nothing (host staging,
    lazy upload, pool-buffer indirection all keep H bit for bit)."""
    m = c4
    H = product.total_potential(0, m.n)
    product.set_total_potential_bulk(0, H)
    product.check(product.lib.sf3d_synchronize(), "synchronize")
    assert np.array_equal(product.total_potential(0, m.n), H)
    a = product.get_total_water_content()
    product.set_total_potential_bulk(0, H)
    assert product.get_total_water_content() == a
    product.lib.sf3d_clean()


def _launch_modes(full):
    """The reference point is the plain path: separate decision kernels, eager launches, link sums inside the step, single sweeps.
    Against it: the defaults a user gets (fused decisions, hipGraphs, overlapped link sums and - where the grid is large - the
    paired sweep), and every switchable form on its own."""
    base = dict(SF3D_FUSED_DECIDE="0", SF3D_GRAPHS="0", SF3D_OVERLAP_ACCEPT="0", SF3D_RESIDENT_GRIDS="1", SF3D_PAIR_SWEEP="0", SF3D_COURANT_PROBE="0", SF3D_RESIDENT_SWEEP="0")
    fast = dict(SF3D_FUSED_DECIDE="1", SF3D_GRAPHS="1", SF3D_OVERLAP_ACCEPT="1", SF3D_RESIDENT_GRIDS="1", SF3D_PAIR_SWEEP="0", SF3D_COURANT_PROBE="0", SF3D_RESIDENT_SWEEP="0")
    auto = {k: v for k, v in fast.items() if k not in ("SF3D_PAIR_SWEEP", "SF3D_COURANT_PROBE", "SF3D_RESIDENT_SWEEP")}      # the library picks sweep and launch form itself (early Courant check while the Courant number is high; the resident sweep loop where the rows of the grid fit on chip: C2)
    modes = [base, auto, fast,      # (fast: fused single sweeps from hipGraphs - what `auto` runs where neither the resident loop nor the paired pass applies)
             dict(fast, SF3D_PAIR_SWEEP="1", SF3D_PAIR_W="10", SF3D_COURANT_PROBE="always"),     # the paired sweep forced on (small grids too), the early Courant check before every approximation
             dict(fast, SF3D_PAIR_SWEEP="1", SF3D_PAIR_W="6", SF3D_OVERLAP_ACCEPT="0"),
             dict(auto, SF3D_SLAB_OVERLAP="3", SF3D_COURANT_PROBE="always")]      # an approximation queued in slabs: rows of one beside the node properties of the next
    if full:
        modes += [dict(fast, SF3D_OVERLAP_ACCEPT="0"), dict(fast, SF3D_RESIDENT_GRIDS="0"), dict(auto, SF3D_RESIDENT_SWEEP="1", SF3D_GRAPHS="0"),
                  dict(fast, SF3D_PAIR_SWEEP="1", SF3D_PAIR_W="14", SF3D_GRAPHS="0"),
                  dict(base, SF3D_COURANT_PROBE="always"), dict(fast, SF3D_SLAB_OVERLAP="2", SF3D_GRAPHS="0", SF3D_SLAB_FIRST="0.3")]
    return modes


def _run_modes(case, modes, tmp_path):
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    outs = []
    for k, mode in enumerate(modes):
        out = tmp_path / f"{case}_{k}.npz"
        env = {kk: vv for kk, vv in os.environ.items() if not kk.startswith(("SF3D_PAIR", "SF3D_ASM", "SF3D_RESIDENT", "SF3D_SLAB"))}
        env.update(mode)
        p = subprocess.run([sys.executable, str(root / "scripts" / "run_case.py"), case, str(out)], env=env,
                           capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout + p.stderr
        outs.append(np.load(out))
    a = outs[0]
    for mode, b in zip(modes[1:], outs[1:]):
        assert set(a.files) == set(b.files)
        for k in a.files:
            if k != "early_courant":
                assert np.array_equal(a[k], b[k]), (k, mode)
    if case == "c2f60":          # (43 Courant rejections in hour 0: the early check takes its share where it is on, none where it is off)
        always = next(k for k, md in enumerate(modes) if md.get("SF3D_COURANT_PROBE") == "always")
        assert int(outs[0]["early_courant"]) == 0 and int(outs[1]["early_courant"]) > 0 and int(outs[always]["early_courant"]) >= int(outs[1]["early_courant"])


@pytest.mark.parametrize("case", ["c2f60", "c3f20", "c4f20"])
def test_launch_modes_are_bitwise_equivalent(case, tmp_path):
    """The single-GPU fast paths (sweep + convergence decision fused through a last-block hand-off, batches replayed from hipGraphs,
    link flow sums on a second stream, two Jacobi iterations per pass through an LDS ring) must give exactly the bits of the plain path: same partial-sum order, same decisions, same sums.
    Four modes by default; SF3D_FULL_MATRIX=1 runs every switchable form on its own (test below)."""
    _run_modes(case, _launch_modes(False), tmp_path)


@pytest.mark.slow
@pytest.mark.parametrize("case", ["c2f60", "c3f20", "c4f20"])
def test_launch_modes_full_matrix(case, tmp_path):
    import os
    if os.environ.get("SF3D_FULL_MATRIX") != "1":
        pytest.skip("the full launch-mode matrix (8 modes x 3 cases, one process each) runs with SF3D_FULL_MATRIX=1")
    _run_modes(case, _launch_modes(True), tmp_path)


def test_ravone_project_coupled_heat_first_step(product, oracle):
    """BASELINE config 5 as specified - the Ravone project (DEM + soil map + soil database + land use, criteria3d_amd/project3d.py)
    with coupled heat transport - at full size: one computeStep, a 600 s water step and its heat steps, on 5.85 M nodes, every top
    soil cell an atmosphere boundary.  (The water-only trajectory at this size is tests/test_gpu_ravone_project.py.  The oracle runs
    its loops on 32 threads here; the first step has no ponded water, so the reference's racy ponded-evaporation write is not reached.)"""
    from tests.scenarios import ravone_project_model
    m = cm.with_heat_surface(ravone_project_model(None))
    assert m.n > 5_000_000
    heat = cm.Heat(water=True, latent=True, save_mode=0)
    for sf in (product, oracle):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=32, heat=heat)
        cm.apply_heat_forcing(sf, m, 0)
    _, gd = cm.run_hour(product, m, 2.0, max_steps=1)
    _, od = cm.run_hour(oracle, m, 2.0, max_steps=1)
    np.testing.assert_allclose(gd, od, rtol=1e-12)
    gT, oT = product.temperature(0, m.n)[m.ns:], oracle.temperature(0, m.n)[m.ns:]
    gH, oH = product.total_potential(0, m.n), oracle.total_potential(0, m.n)
    assert np.max(np.abs(gT - oT) / oT) < 1e-6
    assert np.max(np.abs(gH - oH) / np.maximum(np.abs(oH), 1e-9)) < 1e-6
    oracle.lib.sf3d_clean(); product.lib.sf3d_clean()


@pytest.mark.parametrize("shape,w", [((64, 6, 2), "6"), ((64, 7, 3), "6"), ((128, 9, 2), "6"), ((128, 11, 4), "10"), ((64, 23, 5), "10"), ((192, 14, 3), "14"),
                                     ((64, 64, 10), "14")])
def test_paired_sweep_on_awkward_grids(product, shape, w):
    """k_sweep_pair against k_sweep on grids at the edges of its patch logic: two layers only, fewer rows than two patches, a last
    patch that is shifted up instead of hanging over the edge, several column patches - same H, Se and accepted steps, bit for bit"""
    from tests.scenarios import env
    nx, ny, nz = shape
    m = cm.catchment_model(nx, ny, nz)
    res = []
    for pair in ("0", "1"):
        with env(SF3D_PAIR_SWEEP=pair, SF3D_PAIR_W=w, SF3D_RESIDENT_SWEEP="0"):      # (single sweeps as launches of their own: the resident loop has tests/test_gpu_resident.py)
            product.check(product.lib.sf3d_reset_solver_state(), "reset")
            cm.build(product, m)
            product.check(product.lib.sf3d_kernel_timing(1), "timing")       # event statistics tell which sweep kernel ran
            _, d0 = cm.run_hour(product, m, 30.0)
            _, d1 = cm.run_hour(product, m, 0.0, max_steps=40)
            stats = product.kernel_stats()
            product.lib.sf3d_kernel_timing(0)
            # (with the paired sweep on, an approximation expected to take an odd number of iterations gets one single sweep too)
            assert (stats["k_sweep_pair"][0] > 0) == (pair == "1") and (pair == "1" or stats["k_sweep"][0] > 0), stats
            res.append((np.array(d0 + d1), cm.snapshot(product, m), product.counters()))
        product.lib.sf3d_clean()
    (da, sa, ca), (db, sb, cb) = res
    assert np.array_equal(da, db)
    assert np.array_equal(sa["H"], sb["H"]) and np.array_equal(sa["Se"], sb["Se"])
    assert ca == cb


@pytest.mark.parametrize("which,w", [("dem_window", "10"), ("project_window", "6"), ("random_holes", "14"), ("random_holes_wide", "10"), ("full_box", "10")])
def test_paired_sweep_on_masked_grids(product, which, w):
    """k_sweep_pair_masked against k_sweep on layered MASKED grids - DEM outlines with holes, soil columns that end at different depths,
    a random subset of the lateral links, both row orientations: same H, Se, accepted steps and counters, bit for bit"""
    from pathlib import Path
    from tests.scenarios import env, ravone_project_model
    if which == "dem_window":
        m = cm.dem_model(np.load(Path(__file__).resolve().parent / "golden" / "ravone_dem_window_72x72.npy"))
    elif which == "project_window":
        m = ravone_project_model((980, 1060, 330, 420))
    elif which == "random_holes":
        m = cm.random_model(23, nx=70, ny=45, nz=5)
    elif which == "full_box":          # a regular grid sent through the masked kernel (SF3D_PAIR_FORCE_MASKED: the measurement switch)
        m = cm.catchment_model(128, 48, 8, heterogeneous=True)
    else:
        m = cm.random_model(5, nx=150, ny=20, nz=4)
    assert m.ns >= 64
    res = []
    for pair in ("0", "1"):
        with env(SF3D_PAIR_SWEEP=pair, SF3D_PAIR_W=w, SF3D_PAIR_FORCE_MASKED="1" if which == "full_box" else "0", SF3D_RESIDENT_SWEEP="0"):
            product.check(product.lib.sf3d_reset_solver_state(), "reset")
            cm.build(product, m)
            product.check(product.lib.sf3d_kernel_timing(1), "timing")       # event statistics tell which sweep kernel ran
            _, d0 = cm.run_hour(product, m, 30.0, max_steps=120)
            _, d1 = cm.run_hour(product, m, 0.0, max_steps=60)
            stats = product.kernel_stats()
            product.lib.sf3d_kernel_timing(0)
            assert (stats["k_sweep_pair"][0] > 0) == (pair == "1") and (pair == "1" or stats["k_sweep"][0] > 0), stats
            res.append((np.array(d0 + d1), cm.snapshot(product, m), product.counters()))
        product.lib.sf3d_clean()
    (da, sa, ca), (db, sb, cb) = res
    assert np.array_equal(da, db)
    assert np.array_equal(sa["H"], sb["H"]) and np.array_equal(sa["Se"], sb["Se"])
    assert ca == cb


def test_paired_sweep_with_heat_and_with_the_compat_rows(product):
    """the paired sweep next to the other users of the water system: the coupled heat step (thermal fluxes in the water rows, saved
    water fluxes read from the matrix) and the quirk-1 emulation (rows stored raw, normalised by k_compat_rows) - bitwise against
    single sweeps"""
    from tests.scenarios import env, run_scenario
    m = cm.with_heat_surface(cm.catchment_model(64, 40, 6, heterogeneous=True))
    heat = cm.Heat(water=True, latent=True, save_mode=1)
    res = []
    for pair in ("0", "1"):
        with env(SF3D_PAIR_SWEEP=pair, SF3D_PAIR_W="10"):
            product.check(product.lib.sf3d_reset_solver_state(), "reset")
            cm.build(product, m, heat=heat)
            dts = []
            for h, mm in enumerate((4.0, 0.0)):
                cm.apply_heat_forcing(product, m, h)
                dts += cm.run_hour(product, m, mm)[1]
            res.append((np.array(dts), product.total_potential(0, m.n), product.temperature(0, m.n)))
        product.lib.sf3d_clean()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2][m.ns:], res[1][2][m.ns:])
    out = []
    for pair in ("0", "1"):
        with env(SF3D_PAIR_SWEEP=pair, SF3D_PAIR_W="10", SF3D_COMPAT_STALE_LINK_FLOW="1"):
            out.append(run_scenario(product, "flows_c2_f60"))
        product.lib.sf3d_clean()
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), k


def test_runoff_link_in_an_up_slot_switches_the_early_courant_check_off(product, oracle):
    """setNodeLink accepts an Up link between two SURFACE nodes; the assembly treats it as a runoff link and counts its Courant term
    (water.cpp:308-324), but the early Courant check looks at the lateral slots only - its partial maximum would refuse such an attempt
    with another dt than checkCourant (cpusolver.cpp:248-281).  With such a link in the graph the check stays off: the run under the
    60 mm forcing (Courant refusals) matches the oracle step for step whatever SF3D_COURANT_PROBE asks for."""
    import dataclasses
    from tests.scenarios import env
    m0 = cm.catchment_model(64, 64, 6)
    # every 4th cell of rows 8 .. 40 gets one more runoff link, in its Up slot, to the cell one row down the slope
    r, c = np.meshgrid(np.arange(8, 40), np.arange(1, 64, 4), indexing="ij")
    a, b = (r * 64 + c).ravel().astype(np.uint32), ((r - 1) * 64 + c).ravel().astype(np.uint32)
    m = dataclasses.replace(m0, link_node=np.concatenate([m0.link_node, a]), link_to=np.concatenate([m0.link_to, b]),
                            link_dir=np.concatenate([m0.link_dir, np.full(a.size, capi.LINK_UP, m0.link_dir.dtype)]),
                            link_area=np.concatenate([m0.link_area, np.full(a.size, 5.0)]))
    oracle.lib.sf3d_reset_solver_state()
    cm.build(oracle, m, threads=4)
    _, od = cm.run_hour(oracle, m, 60.0, max_steps=150)
    o = cm.snapshot(oracle, m)
    assert oracle.counters()["courant_rejections"] >= 1
    for probe in ("always", "0"):
        with env(SF3D_COURANT_PROBE=probe):
            product.check(product.lib.sf3d_reset_solver_state(), "reset")
            cm.build(product, m)
            _, gd = cm.run_hour(product, m, 60.0, max_steps=150)
            g = cm.snapshot(product, m)
            gc = product.counters()
        np.testing.assert_allclose(gd, od, rtol=1e-12, err_msg=probe)
        assert np.max(np.abs(g["H"] - o["H"]) / np.maximum(np.abs(o["H"]), 1e-9)) < WATER_RTOL, probe
        assert gc["early_courant_rejections"] == 0 and gc["courant_rejections"] == oracle.counters()["courant_rejections"], (probe, gc)
    oracle.lib.sf3d_clean(); product.lib.sf3d_clean()
