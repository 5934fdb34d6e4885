"""Randomised irregular graphs (catchment.random_model: random holes, column depths, partial lateral connectivity, 12 soil
classes, random boundary types, random relief) through the HIP product and the oracle: the chunk descriptors see every mix
of link kinds and index offsets.  Water alone (H within 1e-9: tests/tolerances.py) and water + heat (1e-6 on T and H); identical accepted dt."""
import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm
from tests.tolerances import WATER_RTOL, assert_water_nodes

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-9)))


@pytest.mark.parametrize("seed", [11, 12, 13, 14, 15, 16])
def test_random_graph_water(product, oracle, seed):
    m = cm.random_model(seed)
    for sf in (product, oracle):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=1)
    for h, mm in enumerate([12.0, 0.0]):
        res = []
        for sf in (product, oracle):
            _, dts = cm.run_hour(sf, m, mm, max_steps=40)
            res.append((dts, cm.snapshot(sf, m)))
        (gd, g), (od, o) = res
        assert len(gd) == len(od), (seed, h, len(gd), len(od))
        np.testing.assert_allclose(gd, od, rtol=1e-12)
        assert_water_nodes(g["H"], o["H"], f"random graph {seed}, hour {h}: H")
        assert abs(g["storage"] - o["storage"]) <= WATER_RTOL * abs(o["storage"])
        if len(gd) == 40:
            break                                   # the hour was cut short: the next one would start elsewhere in time


@pytest.mark.parametrize("seed", [21, 22, 23])
def test_random_graph_water_and_heat(product, oracle, seed):
    m = cm.with_heat_surface(cm.random_model(seed, nx=7, ny=7, nz=5))
    heat = cm.Heat(water=True, latent=True, save_mode=1)
    for sf in (product, oracle):
        sf.check(sf.lib.sf3d_reset_solver_state(), "reset")
        cm.build(sf, m, threads=1, heat=heat)
        cm.apply_heat_forcing(sf, m, 3)
    res = []
    for sf in (product, oracle):
        _, dts = cm.run_hour(sf, m, 2.0, max_steps=6)
        res.append((dts, sf.temperature(0, m.n)[m.ns:], sf.total_potential(0, m.n)))
    (gd, gT, gH), (od, oT, oH) = res
    np.testing.assert_allclose(gd, od, rtol=1e-12)
    assert rel(gT, oT) < 1e-6 and rel(gH, oH) < 1e-6, (seed, rel(gT, oT), rel(gH, oH))
