"""Host-side mirror of the caller of the solver path for a REAL project (BASELINE config 5, the Ravone catchment):
what `Project3D::initialize3DModel` (src/project3D/project3D.cpp:456-616) does between the project files and the
soilFluxes3D API - soil database -> soil list, soil / land-use rasters -> per-cell soil and land-unit indices, layer
thicknesses, slope / aspect / runoff-boundary maps, node numbering, `setNode` / `setNodeLink` arguments with the
application's float roundings, `setSoilProperties` unit conversions, per-node (soil, horizon), surface classes and ponds.

Every function cites the reference lines it follows.  The numerically delicate parts are PINNED against the
unmodified reference code compiled in the build container (oracle/ref_project_prep.cpp over agrolib/gis and
agrolib/soil; outputs stored in tests/golden/ravone_project.npz by tests/golden/make_ravone_project.py):
`set_horizon` == soil::setHorizon for all 1 790 horizons of soil_ER_2021.db, `slope_aspect` == gis::computeSlopeAspectMaps
and `boundary_runoff` == gis::isBoundaryRunoff on every cell of the DEM, bit for bit (tests/test_project3d.py).
The loops of project3D.cpp itself (Qt application code; not buildable here) are restated below and not pinned.

Elementary functions go through python's `math` (the C library's pow / exp / atan / atan2 / tan, as in the reference
build), never through numpy's vector loops, so results are bit-identical to the reference's on the same libm.

Host-side data plumbing only; nothing here is on the timed path."""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import capi
from .catchment import LATERAL_OFFSETS, Model

NODATA = -9999.0                     # commonConstants.h
EPSILON = 0.00001                    # commonConstants.h:252
DEG_TO_RAD = 0.01745329252           # commonConstants.h:255
RAD_TO_DEG = 57.295779513            # commonConstants.h:256
GRAVITY = 9.80665
DAY_SECONDS = 86400.0
QUARTZ_DENSITY = 2.648               # commonConstants.h:216
MINIMUM_ORGANIC_MATTER = 0.005       # soil.h:11
LANDUSE_ROAD, LANDUSE_URBAN = "ROAD", "URBAN"


def is_equal(a: float, b: float) -> bool:
    """basicMath.h:25-29"""
    return abs(float(a) - float(b)) < EPSILON


def db_double(v) -> float:
    """QVariant -> double as `getValue(const QVariant&, double*)` does it (agrolib/utilities/utilities.cpp:113-130):
    NULL, '' or text that is not a number -> NODATA; numeric text ('25') -> its value."""
    if v is None:
        return NODATA
    if isinstance(v, (int, float)):
        return float(v)
    try:
        return float(str(v).strip()) if str(v).strip().lower() != "nan" else NODATA
    except ValueError:
        return NODATA


# ------------------------------------------------------------------------------------------------ soil database

def texture_classes(vg_rows):
    """loadVanGenuchtenParameters (agrolib/soil/soilDbTools.cpp:83-197): rows of table van_genuchten
    (id_texture, texture, alpha [1/kPa], n, he [kPa], theta_r, theta_s, k_sat [cm/d], l) -> list indexed by class id"""
    out = [None] * 13
    for (tid, name, alpha, n, he, thr, ths, ksat, l) in vg_rows:
        alpha, n, he = float(alpha), float(n), float(he)
        m = 1.0 - 1.0 / n                                                    # :180
        sc = math.pow(1.0 + math.pow(alpha * he, n), -m)                     # :183
        out[int(tid)] = dict(name=name, alpha=alpha, n=n, he=he, m=m, sc=sc, theta_r=float(thr), ref_theta_s=float(ths),
                             theta_s=float(ths), ksat=float(ksat), l=float(l))
    return out


def usda_texture_class(sand: float, silt: float, clay: float) -> int:
    """soil::getUSDATextureClass (agrolib/soil/soil.cpp:252-288); -9999 when undefined"""
    if int(sand) == int(NODATA) or int(silt) == int(NODATA) or int(clay) == int(NODATA):
        return int(NODATA)
    if abs((sand + clay + silt) - 100) > 2.0:
        return int(NODATA)
    c = int(NODATA)
    if clay >= 40: c = 12
    if silt >= 40 and clay >= 40: c = 11
    if clay >= 35 and sand >= 45: c = 10
    if (clay < 27.5 and silt >= 50 and silt <= 80) or (clay >= 12.5 and silt >= 80): c = 4
    if clay < 12.5 and silt >= 80: c = 6
    if clay < 40 and sand < 20 and clay >= 27.5: c = 8
    if (clay < 20 and sand >= 52.5) or (clay < 7.5 and silt < 50 and sand >= 42.5 and sand <= 52.5): c = 3
    if sand >= 70 and clay <= (sand - 70): c = 2
    if sand >= 85 and clay <= (2 * sand - 170): c = 1
    if clay >= 20 and clay < 35 and sand >= 45 and silt < 27.5: c = 7
    if clay >= 7.5 and clay < 27.5 and sand < 52.5 and silt >= 27.5 and silt < 50: c = 5
    if clay >= 27.5 and clay < 40 and sand >= 20 and sand < 45: c = 9
    return c


def _specific_density(om: float) -> float:
    """soil::estimateSpecificDensity, Ruehlmann et al. 2006 (soil.cpp:400-412)"""
    if is_equal(om, NODATA):
        om = MINIMUM_ORGANIC_MATTER
    return 1.0 / ((1.0 - om) / QUARTZ_DENSITY + om / (1.127 + 0.373 * om))


def _bulk_density(h: dict, total_porosity: float, increase_with_depth: bool) -> float:
    """soil::estimateBulkDensity (soil.cpp:416-445)"""
    if is_equal(total_porosity, NODATA):
        total_porosity = h["ref_theta_s"]
    total_porosity = min(max(total_porosity, 0.0), 1.0)
    sd = _specific_density(h["organic_matter"])
    if is_equal(sd, NODATA) or sd <= 0.0:
        return NODATA
    bd = (1 - total_porosity) * sd
    if increase_with_depth:
        depth = (h["upper"] + h["lower"]) * 0.5
        coeff = (depth - 0.3) * 0.05
        bd *= max(0.0, 1.0 + coeff)
    return bd


def _saturated_conductivity(h: dict, bd: float) -> float:
    """soil::estimateSaturatedConductivity (soil.cpp:469-486)"""
    if is_equal(bd, NODATA):
        return NODATA
    sd = _specific_density(h["organic_matter"])
    ref_bd = (1.0 - h["ref_theta_s"]) * sd
    if bd <= ref_bd:
        return h["ksat"]
    ratio = 1 - (bd / ref_bd)
    return h["ksat"] * math.exp(10.0 * ratio)


def field_capacity_kpa(clay: float) -> float:
    """soil::getFieldCapacity(.., KPA) (soil.cpp:522-556)"""
    if clay <= 20:
        return -10.0
    if clay >= 50:
        return -33.0
    return -10.0 + (-33.0 - -10.0) * ((clay - 20.0) / (50.0 - 20.0))


def set_horizon(row: dict, textures) -> tuple:
    """soil::setHorizon (agrolib/soil/soil.cpp:849-1019) for one row of table `horizons` already converted with
    db_double (keys upper_depth, lower_depth [cm], sand, silt, clay, coarse_fragment, organic_matter [%], bulk_density,
    theta_sat, k_sat).  -> (ok, horizon dict).  The water_retention table of the project database is empty, so the
    curve-fitting branch (:947-956) never runs; it is not restated."""
    h = dict(upper=NODATA, lower=NODATA, class_usda=int(NODATA), coarse=NODATA, organic_matter=NODATA, bulk_density=NODATA,
             alpha=NODATA, n=NODATA, m=NODATA, he=NODATA, theta_r=NODATA, theta_s=NODATA, ksat=NODATA, l=NODATA,
             clay=NODATA, field_capacity=NODATA, ref_theta_s=NODATA, sc=NODATA)
    if row["upper_depth"] != NODATA and row["lower_depth"] != NODATA:                              # :856-865
        h["upper"] = row["upper_depth"] / 100
        h["lower"] = row["lower_depth"] / 100
    else:
        return False, h
    sand, silt, clay = row["sand"], row["silt"], row["clay"]                                        # :868-877
    if (not is_equal(sand, NODATA) and not is_equal(silt, NODATA) and not is_equal(clay, NODATA)
            and (sand + silt + clay) <= 1.01):
        sand *= 100; silt *= 100; clay *= 100
    h["clay"] = clay
    cls = usda_texture_class(sand, silt, clay)                                                     # :880-888
    h["class_usda"] = cls
    if cls == int(NODATA):
        return False, h
    cf = row["coarse_fragment"]                                                                     # :894-904
    h["coarse"] = cf / 100 if (cf != NODATA and cf >= 0 and cf < 100) else 0.0
    om = row["organic_matter"]                                                                      # :907-916
    if om != NODATA and om > 0 and om < 100:
        h["organic_matter"] = om / 100
    else:                                                                                           # estimateOrganicMatter :833-841
        up = h["upper"]
        h["organic_matter"] = 0.02 if up == 0.0 else (0.01 if (up > 0 and up < 0.4) else MINIMUM_ORGANIC_MATTER)
    t = textures[cls]                                                                               # :919-920
    for k in ("alpha", "n", "m", "he", "sc", "theta_r", "theta_s", "ref_theta_s", "ksat", "l"):
        h[k] = t[k]
    ts = row["theta_sat"]                                                                           # :924-927
    if ts != NODATA and ts > 0 and ts < 1:
        h["theta_s"] = ts
    bd = row["bulk_density"]                                                                        # :930-938
    if bd != NODATA and bd > 0 and bd < QUARTZ_DENSITY:
        h["bulk_density"] = bd
    else:
        h["bulk_density"] = _bulk_density(h, h["theta_s"], True)
    if ts == NODATA:                                                                                # :941-944, estimateThetaSat :448-466
        h["theta_s"] = NODATA if is_equal(h["bulk_density"], NODATA) else 1.0 - (h["bulk_density"] / _specific_density(h["organic_matter"]))
    ks = row["k_sat"]                                                                               # :959-981
    if ks != NODATA and ks > 0:
        ref = _saturated_conductivity(h, h["bulk_density"])
        if ks < ref / 100.:
            h["ksat"] = ref / 100.
        elif ks > ref * 100.:
            h["ksat"] = ref * 100.
        else:
            h["ksat"] = ks
    else:
        h["ksat"] = _saturated_conductivity(h, h["bulk_density"])
    h["field_capacity"] = field_capacity_kpa(clay)                                                  # :1009
    return True, h


HORIZON_KEYS = ("horizon_nr", "upper_depth", "lower_depth", "sand", "silt", "clay", "coarse_fragment", "organic_matter",
                "bulk_density", "theta_sat", "k_sat")


def convert_horizon_row(raw: dict) -> dict:
    """loadSoilData's reads (soilDbTools.cpp:316-355): every numeric field through getValue(.., double*)"""
    out = {k: db_double(raw.get(k)) for k in HORIZON_KEYS[1:]}
    out["horizon_nr"] = int(db_double(raw.get("horizon_nr")))
    s, si, c = out["sand"], out["silt"], out["clay"]                                                # :329-334
    if not is_equal(s, NODATA) and not is_equal(si, NODATA) and not is_equal(c, NODATA) and (s + si + c) <= 1.01:
        out["sand"], out["silt"], out["clay"] = s * 100, si * 100, c * 100
    return out


def load_all_soils(soil_rows, horizon_rows, vg_rows):
    """loadAllSoils + loadSoil (soilDbTools.cpp:817-882, 406-475).  soil_rows: (id_soil, soil_code) in table order;
    horizon_rows: dict soil_code -> list of raw row dicts ORDER BY horizon_nr.  A soil whose first horizon is wrong
    (or that has no horizons) is not in the list; a later wrong horizon truncates the profile (bedrock tolerated,
    :451-466).  -> list of dict(id, code, horizons=[...], total_depth): the list index is the solver's soil index."""
    textures = texture_classes(vg_rows)
    soils = []
    for id_soil, code in soil_rows:
        if id_soil is None or code in (None, ""):
            continue
        rows = horizon_rows.get(code, [])
        if not rows:                                                                                # :289-303
            continue
        hz, first_wrong = [], None
        for i, raw in enumerate(rows):
            ok, h = set_horizon(convert_horizon_row(raw), textures)
            hz.append(h)
            if not ok and first_wrong is None:
                first_wrong = i
        if first_wrong is not None:
            if len(hz) == 1 or first_wrong == 0:
                continue
            hz = hz[:first_wrong]
        soils.append(dict(id=int(id_soil), code=code, horizons=hz, total_depth=hz[-1]["lower"]))
    return soils


def soil_property_args(h: dict) -> tuple:
    """arguments of setSoilProperties as Project3D::setCrit3DSoils converts them (project3D.cpp:915-925):
    alpha [1/kPa] -> [1/m], he [kPa] -> [m], theta scaled by the fine-earth fraction, Ksat [cm/d] -> [m/s]"""
    frac = 1.0 - h["coarse"]
    return (h["alpha"] * GRAVITY, h["n"], h["m"], h["he"] / GRAVITY, h["theta_r"] * frac, h["theta_s"] * frac,
            (h["ksat"] * 0.01) / DAY_SECONDS, h["l"], h["organic_matter"], float(h["clay"]))


def horizon_index(soil: dict, depth: float) -> int:
    """soil::getHorizonIndex (soil.cpp:489-498); -9999 when no horizon holds the depth"""
    for i, h in enumerate(soil["horizons"]):
        if depth >= h["upper"] and depth <= (h["lower"] + EPSILON):
            return i
    return int(NODATA)


# ------------------------------------------------------------------------------------------------ layers

def soil_layers(depth: float, min_thickness: float = 0.02, max_thickness: float = 0.10, max_thickness_depth: float = 0.40):
    """Project3D::setSoilLayers + setLayersDepth (project3D.cpp:1568-1661) -> (thickness[nrLayers], centre depth[nrLayers]),
    layer 0 = surface (0, 0).  Defaults: WaterFluxesParameters::initialize (:66-69)."""
    if depth <= 0:
        return [0.0], [0.0]
    if min_thickness == max_thickness:
        growth = 1.0
    else:
        factor, growth, best = 1.01, 1.01, 99.0
        while factor <= 2.0:
            upper, cur = 0.0, min_thickness
            cur_depth = upper + cur * 0.5
            while cur < max_thickness:
                upper += cur
                cur = min(cur * factor, max_thickness)
                cur_depth = upper + cur * 0.5
            err = abs(cur_depth - max_thickness_depth)
            if err < best:
                best, growth = err, factor
            factor += 0.01
    n = 2
    cur_t, lower = min_thickness, min_thickness
    while (depth - lower) > min_thickness:
        n += 1
        nxt = min(cur_t * growth, max_thickness)
        lower += nxt
        cur_t = nxt
    thick, centre = [0.0] * n, [0.0] * n
    thick[1], centre[1] = min_thickness, min_thickness * 0.5
    cur = min_thickness
    for i in range(2, n):
        thick[i] = (depth - cur) if i == n - 1 else min(max_thickness, thick[i - 1] * growth)
        centre[i] = cur + thick[i] * 0.5
        cur += thick[i]
    return thick, centre


# ------------------------------------------------------------------------------------------------ gis

def _neighbour(a: np.ndarray, dr: int, dc: int, fill) -> np.ndarray:
    """a shifted so that out[r, c] = a[r + dr, c + dc]; cells outside the grid hold `fill` (getValueFromRowCol, gis.cpp:520-530)"""
    p = np.pad(a, 1, constant_values=fill)
    return p[1 + dr:1 + dr + a.shape[0], 1 + dc:1 + dc + a.shape[1]]


def is_boundary(dem: np.ndarray, flag: float) -> np.ndarray:
    """gis::isBoundary (gis.cpp:1494-1510): valid cell with at least one missing neighbour (the grid's edge counts)"""
    valid = np.abs(dem.astype(np.float64) - flag) >= EPSILON
    out = np.zeros(dem.shape, bool)
    for dr, dc in LATERAL_OFFSETS:
        out |= ~_neighbour(valid, dr, dc, False)
    return out & valid


def slope_aspect(dem: np.ndarray, cell: float, flag: float = NODATA):
    """gis::computeSlopeAspectMaps (gis.cpp:1190-1267; Horn 3x3 inside, computeSlopeAspectBoundary :1113-1186 on cells with
    a missing neighbour) -> (slope [deg], aspect [deg from north, clockwise]) as float32 maps, `flag` where the DEM has none."""
    dem = np.asarray(dem, np.float32)
    valid = np.abs(dem.astype(np.float64) - flag) >= EPSILON
    bnd = is_boundary(dem, flag)
    slope = np.full(dem.shape, np.float32(flag), np.float32)
    aspect = np.full(dem.shape, np.float32(flag), np.float32)
    z = {(dr, dc): _neighbour(dem, dr, dc, np.float32(flag)) for dr in (-1, 0, 1) for dc in (-1, 0, 1)}

    # interior: Horn derivatives in double (:1219-1256)
    zi = {k: v.astype(np.float64) for k, v in z.items()}
    dzdx = ((zi[(-1, 1)] + 2 * zi[(0, 1)] + zi[(1, 1)]) - (zi[(-1, -1)] + 2 * zi[(0, -1)] + zi[(1, -1)])) / (8.0 * cell)
    dzdy = ((zi[(1, -1)] + 2 * zi[(1, 0)] + zi[(1, 1)]) - (zi[(-1, -1)] + 2 * zi[(-1, 0)] + zi[(-1, 1)])) / (8.0 * cell)
    inner = valid & ~bnd
    flat = inner & (np.abs(dzdx) < EPSILON) & (np.abs(dzdy) < EPSILON)
    slope[flat] = 0.0; aspect[flat] = 0.0
    rr, cc = np.nonzero(inner & ~flat)
    gx, gy = dzdx[rr, cc], dzdy[rr, cc]
    mag = np.sqrt(gx * gx + gy * gy)                      # IEEE-exact in numpy; atan / atan2 through libm below
    sl = [math.atan(v) * RAD_TO_DEG for v in mag.tolist()]
    asp = []
    for a, b in zip(gy.tolist(), gx.tolist()):
        v = 90.0 - math.atan2(a, -b) * RAD_TO_DEG
        asp.append(v + 360.0 if v < 0 else v)
    slope[rr, cc] = np.array(sl, np.float64).astype(np.float32)
    aspect[rr, cc] = np.array(asp, np.float64).astype(np.float32)

    # boundary cells: one-sided differences over the neighbours that exist; (z - z1) is a float subtraction (:1136-1160)
    nb_ok = {k: np.abs(v.astype(np.float64) - flag) >= EPSILON for k, v in z.items()}
    dz_y = np.zeros(dem.shape); dy = np.zeros(dem.shape)
    for i in (-1, 1):
        for j in (-1, 0, 1):
            d = (np.float32(i) * (dem - z[(i, j)])).astype(np.float64)
            dz_y = np.where(nb_ok[(i, j)], dz_y + d, dz_y)
            dy = np.where(nb_ok[(i, j)], dy + cell, dy)
    dz_x = np.zeros(dem.shape); dx = np.zeros(dem.shape)
    for j in (-1, 1):
        for i in (-1, 0, 1):
            d = (np.float32(j) * (dem - z[(i, j)])).astype(np.float64)
            dz_x = np.where(nb_ok[(i, j)], dz_x + d, dz_x)
            dx = np.where(nb_ok[(i, j)], dx + cell, dx)
    with np.errstate(invalid="ignore", over="ignore"):
        g_y = dz_y / np.maximum(dy, EPSILON)
        g_x = dz_x / np.maximum(dx, EPSILON)
        magb = np.sqrt(g_x * g_x + g_y * g_y)
    rr, cc = np.nonzero(bnd)
    sl = [math.atan(v) * RAD_TO_DEG for v in magb[rr, cc].tolist()]
    asp = []
    for a, b in zip(g_y[rr, cc].tolist(), g_x[rr, cc].tolist()):
        v = 90.0 - math.atan2(-a, b) * RAD_TO_DEG
        asp.append(v + 360 if v < 0 else v)
    slope[rr, cc] = np.array(sl, np.float64).astype(np.float32)
    aspect[rr, cc] = np.array(asp, np.float64).astype(np.float32)
    return slope, aspect


def boundary_runoff(has_node: np.ndarray, dem: np.ndarray, aspect: np.ndarray, flag: float = NODATA) -> np.ndarray:
    """gis::isBoundaryRunoff over the grid (gis.cpp:1452-1488; Project3D::setLateralBoundary, project3D.cpp:851-873):
    an edge cell that holds a surface node and is a strict minimum, or whose aspect points at a cell without a node."""
    dem = np.asarray(dem, np.float32)
    bnd = is_boundary(dem, flag) & has_node
    valid = np.abs(dem.astype(np.float64) - flag) >= EPSILON
    strict_min = valid.copy()                                      # isMinimum(dtm, true, ..) :1395-1427
    for dr, dc in LATERAL_OFFSETS:
        zn = _neighbour(dem, dr, dc, np.float32(flag))
        ok = np.abs(zn.astype(np.float64) - flag) >= EPSILON
        strict_min &= ~(ok & (dem >= zn))
    a = aspect
    a_ok = np.abs(a.astype(np.float64) - flag) >= EPSILON
    r = np.where((a >= 135) & (a <= 225), 1, np.where((a <= 45) | (a >= 315), -1, 0))
    c = np.where((a >= 45) & (a <= 135), 1, np.where((a >= 225) & (a <= 315), -1, 0))
    ny, nx = dem.shape
    R, C = np.mgrid[0:ny, 0:nx]
    rr, cc = R + r, C + c
    inside = (rr >= 0) & (rr < ny) & (cc >= 0) & (cc < nx)
    target_has = np.zeros(dem.shape, bool)
    target_has[inside] = has_node[rr[inside], cc[inside]]
    return bnd & (strict_min | (a_ok & ~target_has))


# ------------------------------------------------------------------------------------------------ the model

@dataclass
class ProjectParameters:
    """[soilWaterFluxes] of SETTINGS/parameters.ini over the defaults of WaterFluxesParameters::initialize
    (project3D.cpp:51-81); the values below are the Ravone project's"""
    initial_water_potential: float = -3.0
    is_initial_water_potential: bool = True
    initial_degree_of_saturation: float = 0.8
    compute_all_soil_depth: bool = False
    imposed_computation_depth: float = 0.95
    conductivity_horiz_vert_ratio: float = 4.0
    free_catchment_runoff: bool = True
    free_bottom_drainage: bool = True
    free_lateral_drainage: bool = True
    model_accuracy: int = 2
    min_layer_thickness: float = 0.02
    max_layer_thickness: float = 0.10
    max_layer_thickness_depth: float = 0.40
    compute_crop: bool = False           # Crit3DProcesses::computeCrop: off (water [+ heat] only)


@dataclass
class ProjectInputs:
    dem: np.ndarray                      # float32 [rows, cols], row 0 = north
    soil_map: np.ndarray                 # soil ids on the DEM grid (flag where none)
    land_use: np.ndarray                 # land-unit ids on the DEM grid
    header: dict                         # esri header of the DEM (xllcorner, yllcorner, cellsize, nodata)
    soils: list                          # load_all_soils(...)
    land_units: list                     # rows of land_units: dict(id, id_landuse, roughness, pond)
    meta: dict = field(default_factory=dict)


def window(inp: ProjectInputs, r0: int, r1: int, c0: int, c1: int) -> ProjectInputs:
    """the same project cut to DEM rows r0:r1, columns c0:c1 (cells outside become the catchment's outside)"""
    hdr = dict(inp.header)
    hdr["xllcorner"] = inp.header["xllcorner"] + c0 * inp.header["cellsize"]
    hdr["yllcorner"] = inp.header["yllcorner"] + (inp.dem.shape[0] - r1) * inp.header["cellsize"]
    hdr["nrows"], hdr["ncols"] = r1 - r0, c1 - c0
    return ProjectInputs(dem=inp.dem[r0:r1, c0:c1].copy(), soil_map=inp.soil_map[r0:r1, c0:c1].copy(),
                         land_use=inp.land_use[r0:r1, c0:c1].copy(), header=hdr, soils=inp.soils, land_units=inp.land_units,
                         meta=dict(inp.meta, window=(r0, r1, c0, c1)))


def project_model(inp: ProjectInputs, par: ProjectParameters | None = None) -> Model:
    """The solver model `Project3D::initialize3DModel` builds (project3D.cpp:456-616), as arrays for the bulk ABI:
    setSoilIndexMap :708-755, computation depth :494-515, setSoilLayers / setLayersDepth :1568-1661, setIndexMaps :758-818,
    setLateralBoundary :851-873, setCrit3DSurfaces :876-899, setCrit3DSoils :903-938, setCrit3DTopography :941-1103,
    setCrit3DNodeSoil :1164-1238 (+ computeCurrentPond :1779-1816), setAccuracy :619-652, initializeWaterContent :1106-1160."""
    par = par or ProjectParameters()
    dem = np.asarray(inp.dem, np.float32)
    ny, nx = dem.shape
    flag = float(inp.header.get("nodata", NODATA))
    cell = float(inp.header["cellsize"])
    valid = np.abs(dem.astype(np.float64) - flag) >= EPSILON
    soils, units = inp.soils, inp.land_units

    # setSoilIndexMap: soil-map id -> index in the soil list (getSoilListIndex :1682-1698); NODATA where no such soil
    id_to_index = {}
    for i, s in enumerate(soils):
        id_to_index.setdefault(s["id"], i)
    soil_index = np.full(dem.shape, -1, np.int64)
    sm = np.asarray(inp.soil_map)
    for sid in np.unique(sm[valid]):
        if abs(float(sid) - flag) < EPSILON:
            continue
        k = id_to_index.get(int(sid))
        if k is not None:
            soil_index[valid & (sm == sid)] = k

    # land-unit index per cell (getLandUnitIndexRowCol :1476-1492)
    if len(units) <= 1:
        unit_index = np.where(valid, 0, -1)
    else:
        unit_index = np.full(dem.shape, -1, np.int64)
        lu = np.asarray(inp.land_use)
        for k, u in enumerate(units):
            m = valid & (lu.astype(np.int64) == int(u["id"])) & (unit_index < 0)
            unit_index[m] = k

    # computation depth :494-515
    if par.compute_all_soil_depth:
        depth = 0.0
        for k in np.unique(soil_index[soil_index >= 0]):
            depth = max(depth, soils[int(k)]["total_depth"])
    else:
        depth = par.imposed_computation_depth
    thick, centre = soil_layers(depth, par.min_layer_thickness, par.max_layer_thickness, par.max_layer_thickness_depth)
    nz = len(thick)

    # setIndexMaps :758-818: surface node where a land unit exists; soil node where also the layer's centre lies within the
    # soil profile (isWithinSoil :1737-1747: depth <= lower depth of the last horizon) and the unit is not a road
    total_depth = np.array([s["total_depth"] for s in soils] + [-np.inf])
    road = np.array([str(u.get("id_landuse", "")).upper() == LANDUSE_ROAD for u in units] + [False])
    cell_depth = total_depth[soil_index]                      # -1 -> the -inf sentinel
    exists = np.zeros((nz, ny, nx), bool)
    exists[0] = valid & (unit_index >= 0)
    for l in range(1, nz):
        exists[l] = exists[0] & ~road[unit_index] & (centre[l] <= cell_depth)
    index = np.where(exists, np.cumsum(exists.ravel()).reshape(exists.shape) - 1, -1)
    n, ns = int(exists.sum()), int(exists[0].sum())
    L, R, C = np.nonzero(exists)

    # slope / aspect (Crit3DRadiationMaps, solarRadiation.cpp:65) and the runoff boundary
    slope_deg, aspect = slope_aspect(dem, cell, flag)
    bmap = boundary_runoff(exists[0], dem, aspect, flag)
    # float boundarySlope = tan(slopeDegree * DEG_TO_RAD) (:965): double tan of float x double, stored as float
    tan_slope = np.zeros(dem.shape)
    rr, cc = np.nonzero(valid)
    tan_slope[rr, cc] = [math.tan(v * DEG_TO_RAD) for v in slope_deg[rr, cc].astype(np.float64).tolist()]
    bslope2d = tan_slope.astype(np.float32).astype(np.float64)

    # setCrit3DTopography :941-1103
    area = cell * cell
    thick_a, centre_a = np.array(thick), np.array(centre)
    x = inp.header["xllcorner"] + cell * (C + 0.5)                                   # DEM.getXY, gis.cpp:473-477
    y = inp.header["yllcorner"] + cell * ((ny - R) - 0.5)
    z = (dem[R, C] - centre_a.astype(np.float32)[L]).astype(np.float64)              # float z = DEM - float(layerDepth) :966
    soil = L > 0
    size = np.where(soil, area * thick_a[L], area)                                   # volume = area * thickness :948
    lat_area32 = np.where(soil, (cell * thick_a[L]).astype(np.float32), np.float32(cell)).astype(np.float32)   # :952
    below_depth = np.append(centre_a[1:], np.inf)[L]                                 # layerDepth[layer + 1]
    within_next = below_depth <= cell_depth[R, C]
    last = soil & ((L == nz - 1) | ~within_next)                                     # :988
    on_b = bmap[R, C]
    btype = np.zeros(n, np.uint8); bslope = np.zeros(n); barea = np.zeros(n)
    m = ~soil & on_b & par.free_catchment_runoff                                     # :974-979
    btype[m] = capi.BND_RUNOFF; bslope[m] = bslope2d[R, C][m]; barea[m] = float(np.float32(cell))
    if par.free_bottom_drainage:                                                     # :990-995
        btype[last] = capi.BND_FREE_DRAINAGE; bslope[last] = 0.0; barea[last] = float(np.float32(area))
    m = soil & ~last & on_b & par.free_lateral_drainage                              # :1004-1010
    btype[m] = capi.BND_FREE_LATERAL_DRAINAGE; bslope[m] = bslope2d[R, C][m]; barea[m] = lat_area32.astype(np.float64)[m]
    landuse_type = np.array([str(u.get("id_landuse", "")).upper() for u in units] + [""])
    m = soil & ~last & ~(on_b & par.free_lateral_drainage) & (L == 1)                # :1013-1027
    ut = landuse_type[unit_index[R, C]]
    btype[m & (ut == LANDUSE_ROAD)] = capi.BND_ROAD
    btype[m & (ut == LANDUSE_URBAN)] = capi.BND_URBAN

    # links :1041-1097: Up (when the node above exists), Down (when the next layer is within the soil and the node exists),
    # the eight laterals of the same layer in (i, j) order, interface area lateralArea * 0.5 (float x double)
    ipad = np.pad(index, ((1, 1), (1, 1), (1, 1)), constant_values=-1)
    cand_to = np.full((n, 10), -1, np.int64)
    cand_dir = np.zeros((n, 10), np.uint8)
    cand_area = np.zeros((n, 10))
    cand_to[:, 0] = ipad[L, R + 1, C + 1]; cand_dir[:, 0] = capi.LINK_UP; cand_area[:, 0] = area
    cand_to[:, 1] = np.where((L < nz - 1) & within_next, ipad[L + 2, R + 1, C + 1], -1)
    cand_dir[:, 1] = capi.LINK_DOWN; cand_area[:, 1] = area
    lat = lat_area32.astype(np.float64) * 0.5
    for k, (dr, dc) in enumerate(LATERAL_OFFSETS):
        cand_to[:, 2 + k] = ipad[L + 1, R + 1 + dr, C + 1 + dc]
        cand_dir[:, 2 + k] = capi.LINK_LATERAL
        cand_area[:, 2 + k] = lat
    # the `continue` statements of the link section (:1044-1046, :1058-1062) leave the node's loop body: a node whose Up target is
    # missing gets no link at all, and a node whose next layer is within the soil profile but holds no node there - the surface cell of
    # a ROAD land unit over a valid soil (setIndexMaps gives roads no soil nodes) - gets no LATERAL links: its neighbours still link to
    # it, one way.  (The Ravone land-use map holds a single non-road unit: no node of BASELINE config 5 takes either branch.)
    no_up = (L > 0) & (cand_to[:, 0] < 0)
    cand_to[no_up, 1:] = -1
    no_down = (L < nz - 1) & within_next & (ipad[np.minimum(L + 2, nz + 1), R + 1, C + 1] < 0)
    cand_to[no_down, 2:] = -1
    mask = cand_to >= 0
    idx = np.arange(n, dtype=np.int64)

    # setCrit3DSoils :903-938: every horizon of every soil of the database (texture classes 1..12)
    soil_table = []
    for si, s in enumerate(soils):
        for hi, h in enumerate(s["horizons"]):
            if h["class_usda"] <= 0 or h["class_usda"] > 12:
                continue
            soil_table.append((si, hi, soil_property_args(h)))

    # setCrit3DNodeSoil :1164-1238: horizon holding the layer's centre depth; surface: land unit + current pond
    hz_of = np.full((len(soils) + 1, nz), -1, np.int64)
    for k in np.unique(soil_index[soil_index >= 0]):
        for l in range(1, nz):
            hz_of[int(k), l] = horizon_index(soils[int(k)], centre[l])
    node_soil = soil_index[R, C][soil]
    node_hz = hz_of[node_soil, L[soil]]
    if np.any(node_hz < 0):
        raise ValueError("setCrit3DNodeSoil: no horizon definition at some layer depth (check soil totalDepth)")
    su = unit_index[R, C][~soil]
    max_pond = np.array([float(u["pond"]) for u in units])[su]
    soil_max_pond = max_pond / (tan_slope[R, C][~soil] + 1.)                         # computeCurrentPond :1789-1796
    pond = soil_max_pond.astype(np.float32).astype(np.float64)                       # return float(..) :1815

    # setAccuracy :619-636
    vmax = 5 + 5 * par.model_accuracy
    min_dt = min(6.0, cell / vmax)
    numerics = (min_dt, 3600.0, 150, 10, 7 + par.model_accuracy, par.model_accuracy)

    # initializeWaterContent :1106-1160
    psi_surface = par.initial_water_potential if (par.is_initial_water_potential and par.initial_water_potential > 0) else 0.0
    if not par.is_initial_water_potential:
        raise NotImplementedError("initial degree of saturation: not used by the Ravone project")

    mdl = Model(n=n, ns=ns, x=x.astype(float), y=y.astype(float), z=z, size=size, is_surface=(~soil).astype(np.uint8), btype=btype,
                bslope=bslope, barea=barea, link_node=np.broadcast_to(idx[:, None], (n, 10))[mask].astype(np.uint32),
                link_to=cand_to[mask].astype(np.uint32), link_dir=cand_dir[mask], link_area=cand_area[mask],
                soil_index=node_soil.astype(np.uint16), soils=[], psi0_surface=psi_surface, psi0_soil=par.initial_water_potential,
                lv_ratio=par.conductivity_horiz_vert_ratio, numerics=numerics, cell_area=area, shape=(nx, ny, nz),
                meta=dict(kind="project", layers=thick[1:], index=index, cell=cell, depth=depth, header=dict(inp.header)),
                horizon_index=node_hz.astype(np.uint16), soil_table=soil_table, surface_index=su.astype(np.uint16),
                surface_roughness=[float(u["roughness"]) for u in units], pond_node=pond)
    return mdl


def load_project_fixture(path) -> ProjectInputs:
    """tests/golden/ravone_project.npz (+ the DEM fixture beside it): rasters and database tables of the Ravone project
    as data -> ProjectInputs with the soil list computed HERE by load_all_soils"""
    import json
    from pathlib import Path
    from . import esri
    path = Path(path)
    z = np.load(path, allow_pickle=False)
    dem, hdr = esri.load_dem_fixture(path.parent / "ravone_dem_519x1208.npz")
    tables = json.loads(str(z["tables_json"]))
    soils = load_all_soils([tuple(r) for r in tables["soils"]], tables["horizons"], [tuple(r) for r in tables["van_genuchten"]])
    units = [dict(id=r[0], id_landuse=r[3], id_crop=r[4], roughness=r[5], pond=r[6]) for r in tables["land_units"]]
    return ProjectInputs(dem=dem, soil_map=z["soil_map"].astype(np.float32), land_use=z["land_use"].astype(np.float32),
                         header=hdr, soils=soils, land_units=units, meta=dict(name="Ravone"))
