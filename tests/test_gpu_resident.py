"""The resident-coefficient sweep loop (k_sweep_resident, csrc/sf3d_resident.inc): all Jacobi iterations of an approximation in ONE launch,
the normalised rows kept in registers, the iterate in an LDS tile, iterations separated by a grid barrier that also carries the norm.
It replaces solveLinearSystem's loop (cpusolver.cpp:672-703) around JacobiWaterCPU (water.cpp:565-601) on regular grids whose rows fit on
chip (one of eight strips of the 512 x 512 x 20 catchment: 0.66 M nodes).  Same operands in the same order: the bits of the single sweeps."""
import numpy as np
import pytest

from criteria3d_amd import catchment as cm
from tests.scenarios import env
from tests.tolerances import assert_water_nodes

pytestmark = pytest.mark.gpu


def _run(product, m, resident, hours, **kw):
    with env(SF3D_RESIDENT_SWEEP=resident, SF3D_PAIR_SWEEP="0", **kw):
        product.check(product.lib.sf3d_reset_solver_state(), "reset")
        cm.build(product, m)
        product.check(product.lib.sf3d_kernel_timing(1), "timing")       # event statistics tell which sweep kernel ran
        dts = []
        for mm, mx in hours:
            _, d = cm.run_hour(product, m, mm, max_steps=mx)
            dts += d
        stats = product.kernel_stats()
        product.lib.sf3d_kernel_timing(0)
        res = (np.array(dts), cm.snapshot(product, m), product.counters(), stats)
    product.lib.sf3d_clean()
    return res


@pytest.mark.parametrize("shape,pr", [((64, 6, 2), "1"), ((64, 10, 2), "5"), ((64, 8, 5), "4"), ((64, 64, 10), "4"), ((128, 12, 3), "4"), ((128, 30, 7), "2"),
                                     ((192, 20, 10), "4"), ((256, 40, 4), "8"), ((512, 64, 20), "2")])
def test_resident_sweep_loop_is_bitwise_the_single_sweeps(product, shape, pr):
    """k_sweep_resident against k_sweep: one patch column and several, one row per block and several, every compiled (K, NW) shape, the
    last shape one of eight strips of the headline grid (256 blocks of 512 threads, one per CU) - same accepted steps, H, Se, counters."""
    nx, ny, nz = shape
    m = cm.catchment_model(nx, ny, nz, heterogeneous=nz > 4)
    hours = [(30.0, 60 if m.n > 300_000 else None), (0.0, 40)]
    da, sa, ca, ta = _run(product, m, "0", hours)
    db, sb, cb, tb = _run(product, m, "1", hours, SF3D_RESIDENT_PR=pr)
    assert ta["k_sweep_resident"][0] == 0 and ta["k_sweep"][0] > 0, ta
    assert tb["k_sweep_resident"][0] > 0 and tb["k_sweep"][0] == 0, tb
    assert tb["k_sweep_resident"][0] == cb["approximations"] - cb["courant_rejections"], (tb, cb)      # one launch per linear system
    assert np.array_equal(da, db)
    assert np.array_equal(sa["H"], sb["H"]) and np.array_equal(sa["Se"], sb["Se"])
    assert ca == cb


def test_resident_sweep_loop_matches_oracle(product, oracle):
    """... and the checker itself, through the runoff regime of a small catchment (Courant refusals, restore-best steps)"""
    m = cm.catchment_model(64, 64, 10)
    with env(SF3D_RESIDENT_SWEEP="1"):
        for sf in (product, oracle):
            sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
            cm.build(sf, m)
        product.check(product.lib.sf3d_kernel_timing(1), "timing")
        _, gd = cm.run_hour(product, m, 60.0, max_steps=300)
        stats = product.kernel_stats()
        product.lib.sf3d_kernel_timing(0)
        _, od = cm.run_hour(oracle, m, 60.0, max_steps=300)
    assert stats["k_sweep_resident"][0] > 0, stats
    np.testing.assert_allclose(gd, od, rtol=1e-12)
    g, o = cm.snapshot(product, m), cm.snapshot(oracle, m)
    assert_water_nodes(g["H"], o["H"], "resident sweep loop, C2 F60: H")
    assert_water_nodes(g["Se"], o["Se"], "resident sweep loop, C2 F60: Se")
    gc, oc = product.counters(), oracle.counters()
    for k in ("attempts", "accepted", "approximations", "sweeps", "courant_rejections"):
        assert gc[k] == oc[k], (k, gc, oc)
    product.lib.sf3d_clean(); oracle.lib.sf3d_clean()


def test_a_loop_that_gives_up_costs_a_repeated_step_not_an_error(tmp_path):
    """All blocks of the loop must be on the device together; on a GPU shared with another process's long kernels they may not be, and a
    block then gives up after seconds of waiting.  A drop-in solver must not answer that with a solver error: the host puts the control
    block of the step's start back, turns the loop off for the model and takes the step again with the sweeps as separate launches
    (csrc/sf3d_host_step.inc).  SF3D_RESIDENT_FAIL_TEST=n makes the launches of the n-th computeStep give up in their second iteration:
    the run must give the bits of the undisturbed run - every field, every accepted dt, every counter - and say what happened."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    outs, errs = [], []
    for k, extra in enumerate(({}, {"SF3D_RESIDENT_FAIL_TEST": "5"})):
        out = tmp_path / f"c2f60_{k}.npz"
        env = {kk: vv for kk, vv in os.environ.items() if not kk.startswith(("SF3D_PAIR", "SF3D_RESIDENT"))}
        env.update(extra)
        p = subprocess.run([sys.executable, str(root / "scripts" / "run_case.py"), "c2f60", str(out)], env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout + p.stderr
        outs.append(np.load(out)); errs.append(p.stderr)
    assert "resident sweep loop: a wait for another block's records expired" in errs[1] and "expired" not in errs[0], errs[1][-800:]
    a, b = outs
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
