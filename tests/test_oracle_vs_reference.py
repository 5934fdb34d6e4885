"""Live cross-check of the oracle against the unmodified reference, when oracle/_ref is
loadable (in the build container; skipped on machines without the prebuilt library or Qt).
Each case runs in a fresh interpreter (tests/live_compare.py) - the reference cannot be reset."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.scenarios import run_scenario

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("name", ["c1_column", "ragged_edge_cases", "ragged_arithmetic_vg", "ragged_geometric", "het_patches",
                                  "heat_column_latent", "heat_column_conduction"])
def test_bitwise_equal(name):
    p = subprocess.run([sys.executable, str(ROOT / "tests" / "live_compare.py"), name], capture_output=True, text=True)
    if p.returncode == 77:
        pytest.skip(p.stdout.strip())
    assert p.returncode == 0, p.stdout + p.stderr


def test_thread_count_does_not_change_the_oracle(oracle):
    """The restatement sums in index order whatever the thread count (deterministic reductions)."""
    a, b = run_scenario(oracle, "ragged_edge_cases", threads=1), run_scenario(oracle, "ragged_edge_cases", threads=4)
    for k in a:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


def test_thread_count_does_not_change_the_oracles_heat_step(oracle):
    """The heat phase's serial pieces run in parallel when the checker has threads - the Gauss-Seidel sweep level by level
    (oracle/sf3d_oracle.cpp gaussSeidelHeat: every read sees the value the node order gives it), the storage sum as addends
    evaluated in parallel and added in index order - and must give the bits of the one-thread run: T, H, the accepted time steps,
    the heat sub-step and sweep counts, on a catchment large enough for the parallel forms (N >= 4096) with lateral heat links."""
    from criteria3d_amd import catchment as cm
    m = cm.with_heat_surface(cm.catchment_model(40, 36, 6, heterogeneous=True))
    assert m.n >= 4096
    hs = cm.Heat(water=True, latent=True, save_mode=0)
    runs = []
    from tests.scenarios import env
    # (the level-by-level sweep switches itself on only where a level holds ~1 000 nodes - the whole Ravone project; forced here)
    for threads, level_gs in ((1, None), (8, "1"), (3, "1"), (8, None)):
        with env(**({"SF3D_ORACLE_LEVEL_GS": level_gs} if level_gs else {})):
            oracle.check(oracle.lib.sf3d_reset_solver_state(), "reset")
            cm.build(oracle, m, threads=threads, heat=hs)
            base = oracle.heat_counters()
            out = []
            for h, mm in enumerate((5.0, 0.0)):
                cm.apply_heat_forcing(oracle, m, h)
                _, dts = cm.run_hour(oracle, m, mm, max_steps=25)
                out += [np.array(dts), oracle.temperature(0, m.n), oracle.total_potential(0, m.n)]
            hc = oracle.heat_counters()
            runs.append((out, {k: hc[k] - base[k] for k in hc}))
            oracle.lib.sf3d_clean()
    for out, work in runs[1:]:
        assert work == runs[0][1], (work, runs[0][1])
        for a, b in zip(out, runs[0][0]):
            assert np.array_equal(a, b)
    assert runs[0][1]["sweeps"] > 50 and runs[0][1]["accepted"] >= 10
