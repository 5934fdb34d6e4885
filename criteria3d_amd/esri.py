"""ESRI float grids (.flt + .hdr) - the raster format on either side of the solver path in the reference application:
the DEM / soil / land-use inputs of a project (DATA/DEM/DEM_Ravone.flt, gis::readEsriGridFlt) and the water-potential
state files WP_<depth cm>.flt that `Crit3DProject::saveSoilWaterState` writes and `loadWaterPotentialState` reads
(bin/CRITERIA3D/criteria3DProject.cpp:2259-2307, 2934-3123).  SURVEY.md 8f-1 / 8f-3: with this a DEM project runs
through the C ABI without Qt, and a run can be stopped and resumed from the application's own state directory.

Host-side data plumbing only (numpy); nothing here is on the timed path."""
from __future__ import annotations

import os
import re
from pathlib import Path

import numpy as np

from . import capi

HEADER_KEYS = ("ncols", "nrows", "xllcorner", "yllcorner", "cellsize", "NODATA_value", "byteorder")


def _base(path) -> Path:
    p = Path(path)
    return p.with_suffix("") if p.suffix.lower() in (".flt", ".hdr") else p


def read_header(path) -> dict:
    hdr = {}
    for line in open(str(_base(path)) + ".hdr"):
        parts = line.split()
        if len(parts) >= 2:
            hdr[parts[0].lower()] = parts[1]
    out = dict(ncols=int(hdr["ncols"]), nrows=int(hdr["nrows"]), xllcorner=float(hdr.get("xllcorner", hdr.get("xllcenter", 0))),
               yllcorner=float(hdr.get("yllcorner", hdr.get("yllcenter", 0))), cellsize=float(hdr["cellsize"]),
               nodata=float(hdr.get("nodata_value", -9999)), byteorder=hdr.get("byteorder", "LSBFIRST").upper())
    return out


def read_grid(path):
    """-> (float32 array [nrows, ncols], row 0 = northern edge; header dict)"""
    hdr = read_header(path)
    dt = np.dtype("<f4") if hdr["byteorder"].startswith("LSB") else np.dtype(">f4")
    data = np.fromfile(str(_base(path)) + ".flt", dtype=dt)
    if data.size != hdr["nrows"] * hdr["ncols"]:
        raise ValueError(f"{path}: {data.size} values, header says {hdr['nrows']} x {hdr['ncols']}")
    return data.astype(np.float32).reshape(hdr["nrows"], hdr["ncols"]), hdr


def write_grid(path, array, hdr: dict):
    """same text layout as gis::writeEsriGridHeader (agrolib/gis/gisIO.cpp:1445-1507): little-endian float32 rows"""
    base = _base(path)
    a = np.asarray(array, dtype="<f4")
    with open(str(base) + ".hdr", "w") as f:
        def num(v):
            return str(int(v)) if float(v).is_integer() else repr(float(v))
        f.write(f"ncols         {a.shape[1]}\nnrows         {a.shape[0]}\n")
        f.write(f"xllcorner     {num(hdr.get('xllcorner', 0))}\nyllcorner     {num(hdr.get('yllcorner', 0))}\n")
        f.write(f"cellsize      {num(hdr.get('cellsize', 1))}\nNODATA_value  {num(hdr.get('nodata', -9999))}\n")
        f.write("byteorder     LSBFIRST\n")
    a.tofile(str(base) + ".flt")


def load_dem_fixture(path):
    """tests/golden/ravone_dem_519x1208.npz: the values and header fields of DATA/DEM/DEM_Ravone.flt kept as a compressed
    numpy fixture -> (float32 array, header dict as read_grid returns it)"""
    z = np.load(path)
    dem = z["dem"].astype(np.float32)
    hdr = dict(ncols=dem.shape[1], nrows=dem.shape[0], xllcorner=float(z["xllcorner"]), yllcorner=float(z["yllcorner"]),
               cellsize=float(z["cellsize"]), nodata=float(z["nodata"]), byteorder="LSBFIRST")
    return dem, hdr


def layer_depths(m) -> np.ndarray:
    """centre depth [m] of every layer of a DEM model; layer 0 = surface (depth 0)"""
    thick = np.asarray(m.meta["layers"], dtype=np.float64)
    top = np.concatenate([[0.0], np.cumsum(thick)[:-1]])
    return np.concatenate([[0.0], top + 0.5 * thick])


def save_water_state(sf: capi.SF3D, m, directory, hdr: dict | None = None, nodata: float = -9999.0):
    """write <directory>/water/WP_<depth cm>.{flt,hdr}: matric potential [m] of every layer, like saveSoilWaterState
    (the directory is recreated).  Extension: the adaptive time step, which the reference keeps only in memory
    (SURVEY.md 5), goes to <directory>/water/deltaT.txt so that a resumed run continues on the same step size."""
    index = m.meta["index"]
    water = Path(directory) / "water"
    if water.exists():
        for f in water.iterdir():
            f.unlink()
    water.mkdir(parents=True, exist_ok=True)
    hdr = dict(hdr or {}); hdr.setdefault("cellsize", m.meta.get("cell", 1.0)); hdr["nodata"] = nodata
    psi = sf.total_potential(0, m.n) - m.z
    depths = layer_depths(m)
    for l in range(index.shape[0]):
        grid = np.full(index.shape[1:], nodata, np.float32)
        ok = index[l] >= 0
        grid[ok] = psi[index[l][ok]].astype(np.float32)
        write_grid(water / f"WP_{int(round(depths[l] * 100))}", grid, hdr)
    (water / "deltaT.txt").write_text(repr(float(sf.lib.sf3d_get_time_step())) + "\n")
    return water


def load_water_state(sf: capi.SF3D, m, directory, restore_time_step: bool = True):
    """loadWaterPotentialState: every WP_<cm>.flt found is a depth level; each model layer takes the level of its own
    depth or, between two levels, the mix the reference computes - with its integer division
    `w0 = (currentDepthCm - depthList[layer0]) / delta` (criteria3DProject.cpp:3039-3043): w0 = 0 and w1 = 1 for a layer
    strictly between two levels, i.e. the deeper level's value (the upper one where the deeper holds NODATA); cells whose
    upper level holds NODATA take the first valid level above.
    Ends with initializeBalance-free state (call sf3d_initialize_balance afterwards, as the application does)."""
    water = Path(directory) / "water"
    levels = {}
    for f in os.listdir(water):
        mt = re.fullmatch(r"WP_(\d+)\.flt", f)
        if mt:
            levels[int(mt.group(1))] = read_grid(water / f)
    if not levels:
        raise FileNotFoundError(f"{water}: water directory is empty")
    depth_list = sorted(levels)
    index = m.meta["index"]
    depths_cm = [int(round(d * 100)) for d in layer_depths(m)]
    psi = sf.total_potential(0, m.n) - m.z
    last = len(depth_list) - 1
    for l, cur in enumerate(depths_cm):
        i = 0
        while cur > depth_list[i] and i < last:
            i += 1
        if cur == depth_list[i]:
            l0 = l1 = i
        elif cur > depth_list[i]:
            l0, l1 = i, min(i + 1, last)
        else:
            l0, l1 = max(0, i - 1), i
        delta = depth_list[l1] - depth_list[l0]
        w0 = 1 if delta == 0 else int((cur - depth_list[l0]) / delta)      # C++ int / int
        w1 = 0 if delta == 0 else 1 - w0
        g0, h0 = levels[depth_list[l0]]
        g1, _ = levels[depth_list[l1]]
        flag = np.float32(h0["nodata"])
        ok = index[l] >= 0
        wp0, wp1 = g0[ok], g1[ok]
        val = wp0.astype(np.float32).copy()
        mix = (wp0 != flag) & (wp1 != flag) & (w1 > 0)
        val[mix] = (w0 * wp0[mix].astype(np.float64) + w1 * wp1[mix].astype(np.float64)).astype(np.float32)
        missing = wp0 == flag
        k = l0 - 1
        while np.any(missing) and k > 0:                                  # first valid level above
            up = levels[depth_list[k]][0][ok]
            take = missing & (up != flag)
            val[take] = up[take]; missing &= ~take
            k -= 1
        good = val != flag
        nodes = index[l][ok][good]
        psi[nodes] = val[good].astype(np.float64)
    sf.set_matric_potential_bulk(0, psi)
    dt_file = water / "deltaT.txt"
    if restore_time_step and dt_file.exists():
        sf.check(sf.lib.sf3d_set_time_step(float(dt_file.read_text().split()[0])), "set_time_step")
    return depth_list
