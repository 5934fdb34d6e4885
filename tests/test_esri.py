"""ESRI float grids on either side of the solver path: the project DEM is read without Qt and
the application's WP_<depth cm>.flt state directory is written / read back through the C ABI."""
import numpy as np

from criteria3d_amd import catchment as cm, esri
from pathlib import Path

GOLDEN = Path(__file__).resolve().parent / "golden"


def _ravone(tmp_path):
    """the Ravone DEM (values of DATA/DEM/DEM_Ravone.flt, kept as tests/golden/ravone_dem_519x1208.npz) written out as an
    ESRI float grid and read back through the reader under test"""
    dem, hdr = esri.load_dem_fixture(GOLDEN / "ravone_dem_519x1208.npz")
    esri.write_grid(tmp_path / "DEM_Ravone", dem, hdr)
    return esri.read_grid(tmp_path / "DEM_Ravone.flt")


def test_reads_the_ravone_dem(tmp_path):
    """the figures of SURVEY.md App. D"""
    dem, hdr = _ravone(tmp_path)
    assert dem.shape == (1208, 519) and hdr["cellsize"] == 4.0 and hdr["nodata"] == -9999.0
    valid = dem != hdr["nodata"]
    assert valid.sum() == 422282
    assert abs(dem[valid].min() - 70.3) < 0.05 and abs(dem[valid].max() - 358.1) < 0.05
    assert np.array_equal(np.load(GOLDEN / "ravone_dem_window_72x72.npy"), dem[48:120, 444:516])


def test_grid_round_trip(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4) * 0.25
    a[1, 2] = -9999.0
    esri.write_grid(tmp_path / "g", a, dict(xllcorner=682648, yllcorner=4923526.5, cellsize=4, nodata=-9999))
    b, hdr = esri.read_grid(tmp_path / "g.flt")
    assert np.array_equal(a, b)
    assert hdr["ncols"] == 4 and hdr["nrows"] == 3 and hdr["yllcorner"] == 4923526.5 and hdr["byteorder"] == "LSBFIRST"
    assert (tmp_path / "g.hdr").read_text().splitlines()[0] == "ncols         4"


def test_water_state_directory_round_trip(oracle, tmp_path):
    """saveSoilWaterState / loadWaterPotentialState: one WP_<cm> grid per layer, float32 matric potentials; a rebuilt
    model loaded from the directory holds exactly those values and the saved adaptive time step"""
    dem, hdr = _ravone(tmp_path)
    m = cm.dem_model(dem[48:72, 444:468])
    oracle.lib.sf3d_reset_solver_state()
    cm.build(oracle, m)
    cm.run_hour(oracle, m, 10.0)
    psi = oracle.total_potential(0, m.n) - m.z
    dt = oracle.lib.sf3d_get_time_step()
    water = esri.save_water_state(oracle, m, tmp_path / "state", hdr)
    names = sorted(p.name for p in water.glob("WP_*.flt"))
    depths = esri.layer_depths(m)
    assert len(names) == len(depths) == m.meta["index"].shape[0]
    assert "WP_0.flt" in names and f"WP_{int(round(depths[-1] * 100))}.flt" in names
    g, h = esri.read_grid(water / "WP_0")
    assert g.shape == (24, 24) and h["cellsize"] == 4.0
    assert np.array_equal(g == -9999.0, m.meta["index"][0] < 0)

    oracle.lib.sf3d_reset_solver_state()
    cm.build(oracle, m)
    levels = esri.load_water_state(oracle, m, tmp_path / "state")
    assert levels == sorted(int(round(d * 100)) for d in depths)
    back = oracle.total_potential(0, m.n) - m.z
    # the setter stores z + psi: compare through the same float32 rounding the files impose
    assert np.allclose(back, psi.astype(np.float32).astype(np.float64), rtol=0, atol=1e-9)
    assert oracle.lib.sf3d_get_time_step() == dt


def test_load_from_coarser_levels_follows_the_reference_rule(oracle, tmp_path):
    """a state directory with fewer depth levels than the model has layers (as written by a run with other layering):
    loadWaterPotentialState takes, for a layer between two levels, w0 = (depth - upper) / delta in INTEGER arithmetic
    (criteria3DProject.cpp:3039-3043): w0 = 0, w1 = 1, so the layer takes the DEEPER level's value - and for a cell
    whose level holds NODATA the first valid level above"""
    dem, hdr = _ravone(tmp_path)
    m = cm.dem_model(dem[48:60, 444:456])
    oracle.lib.sf3d_reset_solver_state()
    cm.build(oracle, m)
    index = m.meta["index"]
    depths_cm = [int(round(d * 100)) for d in esri.layer_depths(m)]
    water = tmp_path / "state" / "water"
    water.mkdir(parents=True)
    levels = {0: 0.0, depths_cm[3]: -1.5, depths_cm[8]: -2.5}
    for cmv, val in levels.items():
        g = np.full(index.shape[1:], val, np.float32)
        if cmv == depths_cm[8]:
            g[0, 0] = -9999.0                                   # a hole in the deepest level
        esri.write_grid(water / f"WP_{cmv}", g, dict(hdr, nodata=-9999.0))
    found = esri.load_water_state(oracle, m, tmp_path / "state", restore_time_step=False)
    assert found == sorted(levels)
    psi = oracle.total_potential(0, m.n) - m.z
    def layer_values(l):
        ok = index[l] >= 0
        return psi[index[l][ok]], ok
    v, _ = layer_values(0); assert np.allclose(v, 0.0, atol=1e-12)
    v, _ = layer_values(3); assert np.allclose(v, -1.5, atol=1e-6)
    v, _ = layer_values(1); assert np.allclose(v, -1.5, atol=1e-6)        # between level 0 and level 3: w0 = 0, w1 = 1 -> the deeper level
    v, ok5 = layer_values(5)                                              # between level 3 and level 8: the deeper level, except its hole
    for (r, c), val in zip(np.argwhere(ok5), v):
        assert abs(val - (-1.5 if (r, c) == (0, 0) else -2.5)) < 1e-6
    deep = len(depths_cm) - 1                                            # below the last level: the last level itself
    v, ok = layer_values(deep)
    cells = np.argwhere(ok)
    for (r, c), val in zip(cells, v):
        assert abs(val - (-1.5 if (r, c) == (0, 0) else -2.5)) < 1e-6    # the hole falls back to the level above
