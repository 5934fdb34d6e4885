"""sf3d_get_regular_grid: host logic that recognises a layer-major NX x NY x NZ grid with the ten-link stencil (the structure
the two-iterations-per-pass sweep of DESIGN.md 10 needs) and refuses everything else.  No device call."""
import ctypes as C

import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm


@pytest.fixture(scope="module")
def staged():
    sf = capi.load_product()
    yield sf
    sf.lib.sf3d_clean()
    sf.lib.sf3d_reset_solver_state()


def query(sf):
    nx, ny, nz = C.c_uint32(), C.c_uint32(), C.c_uint32()
    dr, dc = (C.c_int8 * 8)(), (C.c_int8 * 8)()
    err = sf.lib.sf3d_get_regular_grid(C.byref(nx), C.byref(ny), C.byref(nz), dr, dc)
    return err, (nx.value, ny.value, nz.value), list(zip(list(dr), list(dc)))


@pytest.mark.parametrize("shape", [(64, 64, 10), (128, 32, 4), (16, 24, 3)])
def test_catchment_grids_are_recognised(staged, shape):
    m = cm.catchment_model(*shape)
    staged.check(staged.lib.sf3d_reset_solver_state(), "reset")
    cm.build(staged, m, finalize=False)                   # host staging only
    err, got, steps = query(staged)
    assert err == 0 and got == shape
    assert sorted(steps) == sorted((r, c) for r in (-1, 0, 1) for c in (-1, 0, 1) if (r, c) != (0, 0))
    # the steps reproduce the staged links of an interior node
    nx, ny, _ = shape
    i = (1 * ny + ny // 2) * nx + nx // 2
    lat = m.link_to[(m.link_node == i) & (m.link_dir == capi.LINK_LATERAL)]
    assert sorted(int(j) - i for j in lat) == sorted(r * nx + c for r, c in steps)


def test_irregular_graphs_are_refused(staged):
    for m in (cm.random_model(3), cm.dem_model_fast(cm.synthetic_dem(40, 36))):
        staged.check(staged.lib.sf3d_reset_solver_state(), "reset")
        cm.build(staged, m, finalize=False)
        assert query(staged)[0] == 5                      # SF3D_MISSING_DATA_ERROR


def test_column_is_refused(staged):
    staged.check(staged.lib.sf3d_reset_solver_state(), "reset")
    cm.build(staged, cm.column_model(20, 0.05, 1.0), finalize=False)
    assert query(staged)[0] == 5
