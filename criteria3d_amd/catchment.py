"""Synthetic model builders and scenario runner for the sf3d C ABI (host-side harness).

The graph-construction rules restate what the reference's caller does
(src/project3D/project3D.cpp:941-1103 `setCrit3DTopography`):
layer-major / row-major node numbering with the surface layer first, Up/Down links with the
cell area, eight lateral links per node in the (row, col) offset order
(-1,-1),(-1,0),(-1,+1),(0,-1),(0,+1),(+1,-1),(+1,0),(+1,+1) with interface area 0.5*lateralArea,
Runoff / FreeLateralDrainage on the outlet edge and FreeDrainage under the last layer.
All calls go through the public ABI, so the same builder feeds the reference, the CPU
restatement and the HIP product.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import capi

# loam-like soil used by SURVEY.md 8c/8d (alpha [1/m], n, he [m], thetaR, thetaS, Ksat [m/s], L)
LOAM = dict(alpha=3.6, n=1.56, he=0.1, theta_r=0.078, theta_s=0.43, ksat=2.9e-6, L=0.5,
            organic_matter=0.01, clay=0.2)

# van_genuchten table of DATA/PROJECT/Ravone/SOIL/soil_ER_2021.db:
# texture, alpha [1/kPa], n, he [kPa], thetaR, thetaS, Ksat [cm/d], L
USDA_VG = [
    ("sand", 0.40, 1.70, 0.7, 0.01, 0.38, 192.0, 0.5), ("loamy sand", 0.35, 1.50, 1.0, 0.02, 0.39, 96.0, 0.5),
    ("sandy loam", 0.29, 1.35, 1.5, 0.03, 0.40, 48.0, 0.5), ("silt loam", 0.14, 1.25, 2.6, 0.03, 0.44, 4.8, 0.5),
    ("loam", 0.16, 1.25, 2.3, 0.03, 0.43, 9.6, 0.5), ("silt", 0.10, 1.24, 2.7, 0.03, 0.44, 2.4, 0.5),
    ("sandy clayloam", 0.22, 1.24, 2.1, 0.03, 0.42, 12.0, 0.5), ("silty clayloam", 0.13, 1.22, 3.1, 0.03, 0.46, 2.4, 0.5),
    ("clayloam", 0.18, 1.19, 2.7, 0.04, 0.45, 3.6, 0.5), ("sandy clay", 0.21, 1.20, 2.5, 0.04, 0.44, 4.8, 0.5),
    ("silty clay", 0.17, 1.19, 3.3, 0.05, 0.48, 2.4, 0.5), ("clay", 0.16, 1.18, 3.7, 0.05, 0.49, 1.2, 0.5),
]
GRAVITY = 9.80665

LATERAL_OFFSETS = [(-1, -1), (-1, 0), (-1, 1), (0, -1), (0, 1), (1, -1), (1, 0), (1, 1)]


@dataclass
class Model:
    """Arrays describing one model, ready to be pushed through the ABI."""
    n: int
    ns: int
    x: np.ndarray
    y: np.ndarray
    z: np.ndarray
    size: np.ndarray
    is_surface: np.ndarray
    btype: np.ndarray
    bslope: np.ndarray
    barea: np.ndarray
    link_node: np.ndarray
    link_to: np.ndarray
    link_dir: np.ndarray
    link_area: np.ndarray
    soil_index: np.ndarray            # per soil node (n - ns): index into `soils`
    soils: list                        # list of dicts (LOAM-like keys)
    roughness: float = 0.05
    pond: float = 0.002
    psi0_surface: float = 0.0
    psi0_soil: float = -2.0
    lv_ratio: float = 10.0
    numerics: tuple = (0.5, 3600.0, 150, 10, 10, 3)
    cell_area: float = 100.0
    shape: tuple = ()
    meta: dict = field(default_factory=dict)
    # project models (criteria3d_amd/project3d.py): several horizons per soil, several surface classes, a pond per cell
    horizon_index: np.ndarray | None = None     # per soil node: horizon of `soil_index` (None: horizon 0)
    soil_table: list | None = None              # [(soil, horizon, setSoilProperties arguments)] (None: `soils`, horizon 0)
    surface_index: np.ndarray | None = None     # per surface node: surface class (None: class 0)
    surface_roughness: list | None = None       # Manning roughness per surface class (None: [roughness])
    pond_node: np.ndarray | None = None         # per surface node: pond [m] (None: `pond` everywhere)


def splitmix64(v: np.ndarray) -> np.ndarray:
    v = (v + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    v = (v ^ (v >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    v = (v ^ (v >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return v ^ (v >> np.uint64(31))


def usda_soils():
    """The 12 USDA classes converted as the caller does (project3D.cpp:915-925)."""
    out = []
    for _, a, n, he, tr, ts, ks, L in USDA_VG:
        out.append(dict(alpha=a * GRAVITY, n=n, he=he / GRAVITY, theta_r=tr, theta_s=ts,
                        ksat=ks * 0.01 / 86400.0, L=L, organic_matter=0.01, clay=0.2))
    return out


def column_model(n_nodes: int = 100, dz: float = 0.02, area: float = 1.0) -> Model:
    """C1 of SURVEY.md 8c: node 0 surface, 99 soil nodes, FreeDrainage bottom."""
    n = n_nodes
    i = np.arange(n)
    z = np.where(i == 0, 0.0, -(dz * (i - 0.5)))
    size = np.where(i == 0, area, area * dz)
    btype = np.zeros(n, np.uint8)
    btype[n - 1] = capi.BND_FREE_DRAINAGE
    barea = np.zeros(n)
    barea[n - 1] = area
    ln, lt, ld = [], [], []
    for k in range(n):
        if k > 0:
            ln.append(k); lt.append(k - 1); ld.append(capi.LINK_UP)
        if k < n - 1:
            ln.append(k); lt.append(k + 1); ld.append(capi.LINK_DOWN)
    return Model(n=n, ns=1, x=np.zeros(n), y=np.zeros(n), z=z, size=size,
                 is_surface=(i == 0).astype(np.uint8), btype=btype, bslope=np.zeros(n), barea=barea,
                 link_node=np.array(ln, np.uint32), link_to=np.array(lt, np.uint32),
                 link_dir=np.array(ld, np.uint8), link_area=np.full(len(ln), area),
                 soil_index=np.zeros(n - 1, np.uint16), soils=[LOAM], psi0_soil=-3.0,
                 numerics=(1.0, 3600.0, 150, 10, 10, 3), cell_area=area, shape=(1, 1, n),
                 meta=dict(kind="column"))


def catchment_model(nx: int, ny: int, nz: int, heterogeneous: bool = False, cell: float = 10.0,
                    thickness: float = 0.10) -> Model:
    """Tilted-plane catchment of SURVEY.md 8d (C2-C4).  nz counts the surface layer."""
    ns, n = nx * ny, nx * ny * nz
    area = cell * cell
    l, r, c = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    l, r, c = l.ravel(), r.ravel(), c.ravel()
    x, y = c * cell, r * cell
    zs = 100.0 + 0.05 * x + 0.02 * y
    depth = 0.05 + 0.1 * (l - 1)
    z = np.where(l == 0, zs, zs - depth)
    size = np.where(l == 0, area, area * thickness)
    outlet = (r == 0) | (c == 0)
    slope = 0.0538
    btype = np.zeros(n, np.uint8)
    bslope = np.zeros(n)
    barea = np.zeros(n)
    surf_out = (l == 0) & outlet
    btype[surf_out] = capi.BND_RUNOFF; bslope[surf_out] = slope; barea[surf_out] = cell
    lat_out = (l > 0) & outlet
    btype[lat_out] = capi.BND_FREE_LATERAL_DRAINAGE; bslope[lat_out] = slope; barea[lat_out] = cell * thickness
    bottom = (l == nz - 1) & (nz > 1)
    btype[bottom] = capi.BND_FREE_DRAINAGE; bslope[bottom] = 0.0; barea[bottom] = area

    idx = np.arange(n, dtype=np.int64)
    # candidate links per node in call order: Up, Down, 8 laterals
    cand_to = np.full((n, 10), -1, dtype=np.int64)
    cand_dir = np.zeros((n, 10), np.uint8)
    cand_area = np.zeros((n, 10))
    up = l > 0
    cand_to[up, 0] = idx[up] - ns; cand_dir[:, 0] = capi.LINK_UP; cand_area[:, 0] = area
    dn = l < nz - 1
    cand_to[dn, 1] = idx[dn] + ns; cand_dir[:, 1] = capi.LINK_DOWN; cand_area[:, 1] = area
    lat_area = np.where(l == 0, cell, cell * thickness) * 0.5
    for k, (dr, dc) in enumerate(LATERAL_OFFSETS):
        rr, cc = r + dr, c + dc
        ok = (rr >= 0) & (rr < ny) & (cc >= 0) & (cc < nx)
        cand_to[ok, 2 + k] = (l[ok] * ny + rr[ok]) * nx + cc[ok]
        cand_dir[:, 2 + k] = capi.LINK_LATERAL
        cand_area[:, 2 + k] = lat_area
    mask = cand_to >= 0
    link_node = np.broadcast_to(idx[:, None], (n, 10))[mask].astype(np.uint32)
    link_to = cand_to[mask].astype(np.uint32)
    link_dir = cand_dir[mask]
    link_area = cand_area[mask]

    if heterogeneous:
        soils = usda_soils()
        rs, cs = r[ns:] // 8, c[ns:] // 8
        h = splitmix64(np.uint64(1234) ^ (rs.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ cs.astype(np.uint64))
        soil_index = (h % np.uint64(12)).astype(np.uint16)
    else:
        soils = [LOAM]
        soil_index = np.zeros(n - ns, np.uint16)
    return Model(n=n, ns=ns, x=x.astype(float), y=y.astype(float), z=z, size=size,
                 is_surface=(l == 0).astype(np.uint8), btype=btype, bslope=bslope, barea=barea,
                 link_node=link_node, link_to=link_to, link_dir=link_dir, link_area=link_area,
                 soil_index=soil_index, soils=soils, numerics=(min(6.0, cell / 20.0), 3600.0, 150, 10, 10, 3),
                 cell_area=area, shape=(nx, ny, nz),
                 # layers / index / cell: what criteria3d_amd/esri.py needs to write and read the application's WP_<depth cm>.flt state directory
                 meta=dict(kind="catchment", heterogeneous=heterogeneous, layers=[thickness] * (nz - 1), cell=cell,
                           index=np.arange(n, dtype=np.int64).reshape(nz, ny, nx)))


@dataclass
class Heat:
    """Coupled heat transport set-up of a Model (heat.cpp): which processes run, the initial soil
    temperature, the HeatSurface boundary nodes with their static atmosphere geometry and the
    fixed-temperature bottom boundary."""
    water: bool = True                 # isComputeWater
    advection: bool = False            # initializeHeatFlag(.., isComputeAdvectiveFlux, ..)
    latent: bool = True                # initializeHeatFlag(.., .., isComputeLatentHeat)
    save_mode: int = 2                 # heatFluxSaveMode_t: 0 None, 1 Total, 2 All
    t0_surface: float | None = 288.15  # initial temperature at the top soil node [K] (None: never call setNodeTemperature - setNode's 20 C default stays) ...
    t0_gradient: float = -2.0          # ... plus this many K per metre of depth below the surface
    height_wind: float = 2.0
    height_temperature: float = 2.0
    roughness_height: float = 0.01
    fixed_temperature: float = 285.15  # bottom (FreeDrainage) boundary
    fixed_depth: float = 0.5


def heat_surface_nodes(m: Model) -> np.ndarray:
    """first soil node below every surface node (the nodes that carry the atmosphere boundary)"""
    down = m.link_dir == capi.LINK_DOWN
    src = m.link_node[down]
    top = src < m.ns
    return m.link_to[down][top].astype(np.uint32)


def with_heat_surface(m: Model) -> Model:
    """copy of the model whose top soil nodes without another boundary are HeatSurface nodes"""
    import copy
    m = copy.deepcopy(m)
    top = heat_surface_nodes(m)
    top = top[m.btype[top] == capi.BND_NONE]
    m.btype[top] = capi.BND_HEAT_SURFACE
    m.bslope[top] = 0.0
    m.barea[top] = m.cell_area
    return m


def renumber_soil_nodes(m: Model, new_of_old: np.ndarray) -> Model:
    """copy of the model with its SOIL nodes renumbered (surface nodes keep [0, ns): the library requires that):
    `new_of_old[k]` = new index of soil node ns + k, a permutation of ns .. n - 1.  Links keep their order (a node's k-th lateral stays
    its k-th lateral), so sums over a row's links are taken in the same order as before."""
    import copy
    m = copy.deepcopy(m)
    mp = np.arange(m.n)
    mp[m.ns:] = np.asarray(new_of_old)
    assert np.array_equal(np.sort(mp), np.arange(m.n))
    inv = np.empty(m.n, dtype=np.int64)
    inv[mp] = np.arange(m.n)
    for name in ("x", "y", "z", "size", "is_surface", "btype", "bslope", "barea"):
        setattr(m, name, np.ascontiguousarray(getattr(m, name)[inv]))
    m.soil_index = np.ascontiguousarray(m.soil_index[inv[m.ns:] - m.ns])
    if m.horizon_index is not None:
        m.horizon_index = np.ascontiguousarray(m.horizon_index[inv[m.ns:] - m.ns])
    m.link_node = mp[m.link_node].astype(m.link_node.dtype)
    m.link_to = mp[m.link_to].astype(m.link_to.dtype)
    return m


def bottom_up(m: Model) -> Model:
    """the same layered model with its soil layers numbered from the BOTTOM up (every Up neighbour of a soil node below the first
    layer has the larger index): a numbering the API accepts and the reference's serial sweeps handle like any other"""
    ncol = m.ns
    nl = (m.n - m.ns) // ncol
    assert m.ns + nl * ncol == m.n, "needs whole layers"
    k = np.arange(m.n - m.ns)
    layer, col = k // ncol, k % ncol
    return renumber_soil_nodes(m, m.ns + (nl - 1 - layer) * ncol + col)


def heat_forcing(h: int) -> dict:
    """synthetic hourly atmosphere at the HeatSurface nodes: diurnal air temperature, humidity, wind, net irradiance"""
    ph = 2.0 * np.pi * ((h + 8) % 24) / 24.0
    return dict(temperature=288.15 + 6.0 * np.sin(ph - np.pi / 2), relative_humidity=65.0 - 20.0 * np.sin(ph - np.pi / 2),
                wind_speed=2.0 + 1.0 * np.cos(ph), net_irradiance=max(0.0, 350.0 * np.sin(ph - np.pi / 2)) - 40.0)


def apply_heat_forcing(sf: capi.SF3D, m: Model, h: int, staged=None):
    if staged is None:
        staged = m.meta.get("staged")
    sel = m.btype == capi.BND_HEAT_SURFACE
    nodes = np.flatnonzero(sel if staged is None else (sel & staged)).astype(np.uint32)
    for k, v in heat_forcing(h).items():
        sf.set_boundary_heat_bulk(k, nodes, v)


def column_of(m: Model) -> np.ndarray:
    """surface ancestor of every node through the Up links (the column a node belongs to); a soil node without an Up link counts as
    column 0 - the library's rule for orphans (sf3d_compute_partition)"""
    up = m.link_dir == capi.LINK_UP
    parent = np.arange(m.n)
    parent[m.link_node[up]] = m.link_to[up]
    root = parent.copy()
    for _ in range(64):
        nxt = parent[root]
        if np.array_equal(nxt, root):
            break
        root = nxt
    return np.where(root < m.ns, root, 0)


def strip_nodes(sf: capi.SF3D, m: Model, rank: int, world: int) -> np.ndarray:
    """STRIP-LOCAL BUILD (include/sf3d.h: sf3d_dist_bounds): boolean mask of the nodes rank `rank` of `world` has to stage - the columns
    whose surface node lies in its index range and the one-cell ring of columns around them, every column whole"""
    bounds = sf.dist_bounds(m.ns, world)
    col = column_of(m)
    owned = (col >= bounds[rank]) & (col < bounds[rank + 1])
    ring = np.unique(col[m.link_to[owned[m.link_node]]])
    need = np.zeros(m.ns, dtype=bool)
    need[np.unique(col[owned])] = True
    need[ring] = True
    return need[col]


def _runs(mask: np.ndarray, lo: int, hi: int):
    """maximal runs [a, b) of True inside mask[lo:hi]"""
    w = np.flatnonzero(np.diff(np.concatenate(([False], mask[lo:hi], [False])).astype(np.int8)))
    return [(lo + int(a), lo + int(b)) for a, b in zip(w[0::2], w[1::2])]


def build(sf: capi.SF3D, m: Model, threads: int = 1, dist=None, finalize: bool = True, heat: Heat | None = None, sparse: bool = False):
    """Push a Model through the ABI in the caller's order.

    dist = (rank, world, allgather) shards the model over `world` ranks (HIP product only):
    every rank pushes the same global model - or, with sparse=True, only the nodes its strip touches (strip_nodes: the strip-local
    build of include/sf3d.h); `allgather(bytes) -> list[bytes]` is the launcher's
    control-plane exchange (torch.distributed all_gather_object in bench.py and the tests)."""
    keep = None
    if dist is not None:
        rank, world, allgather = dist
        sf.check(sf.lib.sf3d_dist_prepare(rank, world), "dist_prepare")
        if sparse and world > 1:
            keep = strip_nodes(sf, m, rank, world)
    if heat is None:
        sf.check(sf.lib.sf3d_initialize(m.n, m.ns, 8, 1, 0, 0, 0), "initialize")
    else:
        sf.check(sf.lib.sf3d_initialize(m.n, m.ns, 8, int(heat.water), 1, 0, heat.save_mode), "initialize")
        sf.check(sf.lib.sf3d_initialize_heat_flag(heat.save_mode, int(heat.advection), int(heat.latent)), "initialize_heat_flag")
    for k, rough in enumerate(m.surface_roughness if m.surface_roughness is not None else [m.roughness]):
        sf.check(sf.lib.sf3d_set_surface_properties(k, rough), "set_surface_properties")
    if m.soil_table is not None:
        for soil, horizon, args in m.soil_table:
            sf.check(sf.lib.sf3d_set_soil_properties(soil, horizon, *args), f"set_soil_properties({soil}, {horizon})")
    for k, s in enumerate(m.soils):
        sf.check(sf.lib.sf3d_set_soil_properties(k, 0, s["alpha"], s["n"], 1.0 - 1.0 / s["n"], s["he"],
                                                 s["theta_r"], s["theta_s"], s["ksat"], s["L"],
                                                 s["organic_matter"], s["clay"]), "set_soil_properties")
    if keep is None:
        sf.set_nodes_bulk(0, m.x, m.y, m.z, m.size, m.is_surface, m.btype, m.bslope, m.barea)
        sf.set_links_bulk(m.link_node, m.link_to, m.link_dir, m.link_area)
        if m.ns > 0:
            sf.set_surface_bulk(0, m.surface_index if m.surface_index is not None else np.zeros(m.ns, np.uint16))
            sf.set_pond_bulk(0, m.pond_node if m.pond_node is not None else np.full(m.ns, m.pond))
        if m.n > m.ns:
            sf.set_soil_bulk(m.ns, m.soil_index, m.horizon_index if m.horizon_index is not None else np.zeros(m.n - m.ns, np.uint16))
    else:
        # the same calls for the runs of nodes this rank stages (global indices; in layer-major numbering one run per layer)
        for a, b in _runs(keep, 0, m.n):
            sf.set_nodes_bulk(a, m.x[a:b], m.y[a:b], m.z[a:b], m.size[a:b], m.is_surface[a:b], m.btype[a:b], m.bslope[a:b], m.barea[a:b])
        lk = keep[m.link_node]
        sf.set_links_bulk(m.link_node[lk], m.link_to[lk], m.link_dir[lk], m.link_area[lk])
        surf = m.surface_index if m.surface_index is not None else np.zeros(m.ns, np.uint16)
        pond = m.pond_node if m.pond_node is not None else np.full(m.ns, m.pond)
        for a, b in _runs(keep, 0, m.ns):
            sf.set_surface_bulk(a, surf[a:b])
            sf.set_pond_bulk(a, pond[a:b])
        hor = m.horizon_index if m.horizon_index is not None else np.zeros(m.n - m.ns, np.uint16)
        for a, b in _runs(keep, m.ns, m.n):
            sf.set_soil_bulk(a, m.soil_index[a - m.ns:b - m.ns], hor[a - m.ns:b - m.ns])
    sf.check(sf.lib.sf3d_set_hydraulic_properties(capi.WRC_MODIFIED_VG, capi.MEAN_LOGARITHMIC, m.lv_ratio),
             "set_hydraulic_properties")
    sf.check(sf.lib.sf3d_set_numerical_parameters(*m.numerics), "set_numerical_parameters")
    sf.lib.sf3d_set_threads_number(threads)
    if heat is not None and heat.t0_surface is not None:
        # temperatures first: with latent heat the conductivity set by the potential setters has a vapour term
        up = m.link_dir == capi.LINK_UP
        parent = np.arange(m.n)
        parent[m.link_node[up]] = m.link_to[up]
        root = parent.copy()
        for _ in range(64):
            nxt = parent[root]
            if np.array_equal(nxt, root):
                break
            root = nxt
        depth = m.z[root] - m.z
        t0 = heat.t0_surface + heat.t0_gradient * depth
        for a, b in ([(0, m.n)] if keep is None else _runs(keep, 0, m.n)):
            sf.set_temperature_bulk(a, t0[a:b])
    psi = np.full(m.n, m.psi0_soil)
    psi[:m.ns] = m.psi0_surface
    for a, b in ([(0, m.n)] if keep is None else _runs(keep, 0, m.n)):
        sf.set_matric_potential_bulk(a, psi[a:b])
    if heat is not None:
        staged = np.ones(m.n, dtype=bool) if keep is None else keep
        hs = np.flatnonzero((m.btype == capi.BND_HEAT_SURFACE) & staged).astype(np.uint32)
        sf.set_boundary_heat_bulk("height_wind", hs, heat.height_wind)
        sf.set_boundary_heat_bulk("height_temperature", hs, heat.height_temperature)
        sf.set_boundary_heat_bulk("roughness", hs, heat.roughness_height)
        apply_heat_forcing(sf, m, 0, staged)
        for i in np.flatnonzero((m.btype == capi.BND_FREE_DRAINAGE) & staged):
            sf.check(sf.lib.sf3d_set_node_boundary_fixed_temperature(int(i), heat.fixed_temperature, heat.fixed_depth), "fixed_temperature")
    if keep is not None:
        m.meta["staged"] = keep          # (apply_heat_forcing of this rank touches only what it staged)
    else:
        m.meta.pop("staged", None)
    if not finalize:          # host-side staging only (partition queries on a machine without a GPU)
        return
    if dist is not None:
        sf.dist_connect(dist[0], dist[1], dist[2])
    sf.check(sf.lib.sf3d_initialize_balance(), "initialize_balance")


def rain_rate(mm_per_hour: float, cell_area: float) -> float:
    """surface source [m3/s] per cell, as assignPrecipitation does (criteria3DProject.cpp:914-968)."""
    return mm_per_hour * 1e-3 / 3600.0 * cell_area


FORCINGS = {
    "F20": lambda h: 20.0 if h == 0 else 0.0,      # infiltration regime
    "F60": lambda h: 60.0 if h == 0 else 0.0,      # runoff regime
    "R5": lambda h: 5.0,                            # C1: constant 5 mm/h
}


def run_hour(sf: capi.SF3D, m: Model, mm: float, use_period: bool = False, max_steps: int | None = None):
    """One simulated hour: hourly sinks then the caller's computeStep loop
    (project3D.cpp:1330-1359).  Returns (steps, list of accepted dt)."""
    sf.set_sink_source_bulk(0, np.full(m.ns, rain_rate(mm, m.cell_area)))
    if use_period:
        sf.lib.sf3d_compute_period(3600.0)
        return None, None
    t, dts = 0.0, []
    while t < 3600.0:
        dt = sf.lib.sf3d_compute_step(3600.0 - t)
        if not (dt > 0.0):
            raise capi.SF3DError(f"{sf.backend}: compute_step returned {dt}")
        dts.append(dt)
        t += dt
        if max_steps is not None and len(dts) >= max_steps:
            break
    return len(dts), dts


def snapshot(sf: capi.SF3D, m: Model) -> dict:
    return dict(
        H=sf.total_potential(0, m.n),
        Se=sf.degree_of_saturation(0, m.n),
        total_water=sf.lib.sf3d_get_total_water_content(),
        storage=sf.lib.sf3d_get_water_storage(),
        mbr=sf.lib.sf3d_get_water_mbr(),
        runoff=sf.lib.sf3d_get_total_boundary_water_flow(capi.BND_RUNOFF),
        drainage=sf.lib.sf3d_get_total_boundary_water_flow(capi.BND_FREE_DRAINAGE),
        lateral=sf.lib.sf3d_get_total_boundary_water_flow(capi.BND_FREE_LATERAL_DRAINAGE),
    )


LINK_FLOW_FIELDS = ("up", "down", "lateral_max", "lateral_sum", "lateral_in", "lateral_out")


def link_flows(sf: capi.SF3D, m: Model, nodes=None) -> np.ndarray:
    """[6][len(nodes)] per-link flow sums through the reference's getters (soilFluxes3D.cpp:1130-1230):
    getNodeMaxWaterFlow(Up / Down / Lateral), getNodeSumLateralWaterFlow, ...In, ...Out."""
    nodes = np.arange(m.n) if nodes is None else np.asarray(nodes)
    L = sf.lib
    out = np.empty((6, len(nodes)))
    for k, i in enumerate(nodes):
        i = int(i)
        out[0, k] = L.sf3d_get_node_max_water_flow(i, capi.LINK_UP)
        out[1, k] = L.sf3d_get_node_max_water_flow(i, capi.LINK_DOWN)
        out[2, k] = L.sf3d_get_node_max_water_flow(i, capi.LINK_LATERAL)
        out[3, k] = L.sf3d_get_node_sum_lateral_water_flow(i)
        out[4, k] = L.sf3d_get_node_sum_lateral_water_flow_in(i)
        out[5, k] = L.sf3d_get_node_sum_lateral_water_flow_out(i)
    return out


def urban_road_model(nx: int = 16, ny: int = 16, nz: int = 5) -> Model:
    """Tilted catchment whose top soil layer carries Urban (infiltration x 0.33) and Road (no infiltration) boundary
    types in two patches (water.cpp:504-513); their boundary flow is 0 (water.cpp:796-799 built -DNDEBUG)."""
    m = catchment_model(nx, ny, nz)
    ns = m.ns
    r, c = np.divmod(np.arange(ns), nx)
    top = ns + np.arange(ns)
    free = m.btype[top] == capi.BND_NONE
    urban = free & (r >= 2) & (r < ny // 2) & (c >= 2) & (c < nx // 2 + 2)
    road = free & (r >= ny // 2 + 1) & (r < ny - 2) & (c >= nx // 3) & (c < nx - 2)
    m.btype[top[urban]] = capi.BND_URBAN
    m.btype[top[road]] = capi.BND_ROAD
    m.bslope[top[urban | road]] = 0.0
    m.barea[top[urban | road]] = m.cell_area
    m.meta = dict(kind="urban_road", urban=int(urban.sum()), road=int(road.sum()))
    return m


def ragged_model(nx: int = 7, ny: int = 6, nz: int = 4, cell: float = 5.0) -> Model:
    """Edge-case graph: NODATA holes in the DEM, columns whose soil ends early (deeper layers have
    fewer nodes, project3D.cpp:791-800), three soil classes, a prescribed-potential node.
    Numbering stays layer-major / row-major over the EXISTING nodes (surface first)."""
    rng = np.random.RandomState(7)
    valid = np.ones((ny, nx), bool)
    valid[0, nx - 1] = valid[2, 3] = valid[ny - 1, 0] = False          # holes
    depth_layers = rng.randint(2, nz, size=(ny, nx))                     # soil layers per column: 2..nz-1
    depth_layers[1, 1] = nz - 1
    thick = 0.08
    area = cell * cell
    index = -np.ones((nz, ny, nx), np.int64)
    n = 0
    for l in range(nz):
        for r in range(ny):
            for c in range(nx):
                if valid[r, c] and (l == 0 or l <= depth_layers[r, c]):
                    index[l, r, c] = n
                    n += 1
    ns = int(valid.sum())
    x = np.zeros(n); y = np.zeros(n); z = np.zeros(n); size = np.zeros(n)
    surf = np.zeros(n, np.uint8); btype = np.zeros(n, np.uint8); bslope = np.zeros(n); barea = np.zeros(n)
    soil_index = np.zeros(n - ns, np.uint16)
    ln, lt, ld, la = [], [], [], []
    for l in range(nz):
        for r in range(ny):
            for c in range(nx):
                i = index[l, r, c]
                if i < 0:
                    continue
                x[i], y[i] = c * cell, r * cell
                zs = 50.0 + 0.03 * x[i] + 0.04 * y[i] + 0.2 * np.sin(1.3 * c + 0.7 * r)
                outlet = (r == 0) or (c == 0)
                if l == 0:
                    z[i], size[i], surf[i] = zs, area, 1
                    if outlet:
                        btype[i], bslope[i], barea[i] = capi.BND_RUNOFF, 0.05, cell
                else:
                    z[i], size[i] = zs - (thick * (l - 0.5)), area * thick
                    soil_index[i - ns] = (r // 2 + c // 3) % 3
                    last = (l == depth_layers[r, c])
                    if last:
                        btype[i], barea[i] = capi.BND_FREE_DRAINAGE, area
                    elif outlet:
                        btype[i], bslope[i], barea[i] = capi.BND_FREE_LATERAL_DRAINAGE, 0.05, cell * thick
                if l > 0:
                    ln.append(i); lt.append(index[l - 1, r, c]); ld.append(capi.LINK_UP); la.append(area)
                if l + 1 < nz and index[l + 1, r, c] >= 0:
                    ln.append(i); lt.append(index[l + 1, r, c]); ld.append(capi.LINK_DOWN); la.append(area)
                for dr, dc in LATERAL_OFFSETS:
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < ny and 0 <= cc < nx and index[l, rr, cc] >= 0:
                        ln.append(i); lt.append(index[l, rr, cc]); ld.append(capi.LINK_LATERAL)
                        la.append((cell if l == 0 else cell * thick) * 0.5)
    # one prescribed-total-potential node in the middle of the deepest complete column
    pnode = int(index[2, 1, 1])
    btype[pnode], bslope[pnode], barea[pnode] = capi.BND_PRESCRIBED, 0.0, area
    soils = usda_soils()
    soils = [soils[2], soils[4], soils[8]]       # sandy loam, loam, clay loam
    return Model(n=n, ns=ns, x=x, y=y, z=z, size=size, is_surface=surf, btype=btype, bslope=bslope, barea=barea,
                 link_node=np.array(ln, np.uint32), link_to=np.array(lt, np.uint32), link_dir=np.array(ld, np.uint8),
                 link_area=np.array(la), soil_index=soil_index, soils=soils, psi0_soil=-1.5, lv_ratio=4.0,
                 numerics=(0.25, 3600.0, 150, 10, 10, 3), cell_area=area, shape=(nx, ny, nz),
                 meta=dict(kind="ragged", prescribed_node=pnode, prescribed_H=float(z[pnode] - 0.3)))


def run_hour_sinks(sf: capi.SF3D, m: Model, sinks: np.ndarray, max_steps: int | None = None):
    """One simulated hour with an explicit per-node sink/source array [m3/s] (all n nodes)."""
    sf.set_sink_source_bulk(0, sinks)
    t, dts = 0.0, []
    while t < 3600.0:
        dt = sf.lib.sf3d_compute_step(3600.0 - t)
        if not (dt > 0.0):
            raise capi.SF3DError(f"{sf.backend}: compute_step returned {dt}")
        dts.append(dt)
        t += dt
        if max_steps is not None and len(dts) >= max_steps:
            break
    return len(dts), dts


def dem_layer_thicknesses(depth: float = 0.95, first: float = 0.02, maximum: float = 0.10, reach: float = 0.40):
    """Soil layer thicknesses as the caller builds them (src/project3D/project3D.cpp:1568-1661):
    geometric growth from `first` so that `maximum` is reached at depth `reach`, then constant,
    the last layer taking the remainder."""
    best = None
    for k in range(101, 201):
        g = k / 100.0
        t, z = first, 0.0
        while t < maximum:                  # depth at which the growing thickness reaches the maximum
            z += t; t *= g
        if best is None or abs(z - reach) < best[0]:
            best = (abs(z - reach), g)
    g = best[1]
    out, t, z = [], first, 0.0
    while z < depth - 1e-9:
        t_eff = min(t, maximum, depth - z)
        if depth - (z + t_eff) < 0.5 * first:       # the last layer takes the remainder
            t_eff = depth - z
        out.append(t_eff); z += t_eff; t = min(t * g, maximum)
    return out


def dem_model(dem: np.ndarray, cell: float = 4.0, nodata: float = -9999.0, depth: float = 0.95, lv_ratio: float = 4.0,
              n_soils: int = 3) -> Model:
    """Graph of a real DEM window the way the caller builds it: surface layer first,
    nodes only where the DEM is valid, soil columns cut where the local soil depth ends, z and the
    boundary slope rounded through float32 like `setCrit3DTopography` (project3D.cpp:949-1010),
    runoff outlets on DEM-edge cells that are local minima or slope outwards, FreeDrainage under the
    last soil node of each column, FreeLateralDrainage on the edge columns."""
    ny, nx = dem.shape
    valid = dem != nodata
    thick = dem_layer_thicknesses(depth)
    nz = len(thick) + 1
    centre = np.cumsum(thick) - 0.5 * np.array(thick)
    bottom = np.cumsum(thick)
    # local soil depth: full depth on gentle slopes, shallower on steep ones (deterministic from the DEM)
    gy, gx = np.gradient(np.where(valid, dem, np.nan).astype(np.float64), cell)
    slope = np.sqrt(np.nan_to_num(gx) ** 2 + np.nan_to_num(gy) ** 2)
    soil_depth = np.clip(depth * (1.0 - 0.8 * np.clip(slope, 0, 1)), 0.25, depth)
    index = -np.ones((nz, ny, nx), np.int64)
    n = 0
    for l in range(nz):
        for r in range(ny):
            for c in range(nx):
                if valid[r, c] and (l == 0 or bottom[l - 1] <= soil_depth[r, c] + 1e-9 or l == 1):
                    index[l, r, c] = n; n += 1
    ns = int(valid.sum())
    area = cell * cell
    x = np.zeros(n); y = np.zeros(n); z = np.zeros(n); size = np.zeros(n)
    surf = np.zeros(n, np.uint8); btype = np.zeros(n, np.uint8); bslope = np.zeros(n); barea = np.zeros(n)
    soil_index = np.zeros(n - ns, np.uint16)
    edge = np.zeros((ny, nx), bool)
    for r in range(ny):
        for c in range(nx):
            if not valid[r, c]:
                continue
            for dr, dc in LATERAL_OFFSETS:
                rr, cc = r + dr, c + dc
                if rr < 0 or rr >= ny or cc < 0 or cc >= nx or not valid[rr, cc]:
                    edge[r, c] = True
    ln, lt, ld, la = [], [], [], []
    for l in range(nz):
        for r in range(ny):
            for c in range(nx):
                i = index[l, r, c]
                if i < 0:
                    continue
                x[i], y[i] = c * cell, (ny - 1 - r) * cell
                zs = float(dem[r, c])
                nbr = [float(dem[r + dr, c + dc]) for dr, dc in LATERAL_OFFSETS
                       if 0 <= r + dr < ny and 0 <= c + dc < nx and valid[r + dr, c + dc]]
                is_min = all(zs < v for v in nbr) if nbr else True
                outward = edge[r, c] and (is_min or (nbr and zs <= min(nbr) + 0.5))
                slp = float(np.float32(max(float(slope[r, c]), 0.001)))
                if l == 0:
                    z[i], size[i], surf[i] = float(np.float32(zs)), area, 1
                    if outward:
                        btype[i], bslope[i], barea[i] = capi.BND_RUNOFF, slp, cell
                else:
                    z[i] = float(np.float32(zs - centre[l - 1]))
                    size[i] = area * thick[l - 1]
                    soil_index[i - ns] = (int(zs) // 7 + l // 4) % n_soils
                    last = (l + 1 >= nz) or index[l + 1, r, c] < 0
                    if last:
                        btype[i], barea[i] = capi.BND_FREE_DRAINAGE, area
                    elif outward:
                        btype[i], bslope[i], barea[i] = capi.BND_FREE_LATERAL_DRAINAGE, slp, float(np.float32(cell * thick[l - 1]))
                if l > 0:
                    ln.append(i); lt.append(index[l - 1, r, c]); ld.append(capi.LINK_UP); la.append(area)
                if l + 1 < nz and index[l + 1, r, c] >= 0:
                    ln.append(i); lt.append(index[l + 1, r, c]); ld.append(capi.LINK_DOWN); la.append(area)
                lat = cell if l == 0 else float(np.float32(cell * thick[l - 1]))
                for dr, dc in LATERAL_OFFSETS:
                    rr, cc = r + dr, c + dc
                    if 0 <= rr < ny and 0 <= cc < nx and index[l, rr, cc] >= 0:
                        ln.append(i); lt.append(index[l, rr, cc]); ld.append(capi.LINK_LATERAL); la.append(lat * 0.5)
    soils = usda_soils()
    soils = [soils[3], soils[4], soils[8]][:n_soils]       # silt loam, loam, clay loam
    dtmin = min(6.0, cell / 20.0)
    return Model(n=n, ns=ns, x=x, y=y, z=z, size=size, is_surface=surf, btype=btype, bslope=bslope, barea=barea,
                 link_node=np.array(ln, np.uint32), link_to=np.array(lt, np.uint32), link_dir=np.array(ld, np.uint8),
                 link_area=np.array(la), soil_index=soil_index, soils=soils, psi0_soil=-3.0, lv_ratio=lv_ratio,
                 numerics=(dtmin, 3600.0, 150, 10, 9, 2), cell_area=area, shape=(nx, ny, nz),
                 meta=dict(kind="dem", layers=thick, index=index, cell=cell))


def dem_model_fast(dem: np.ndarray, cell: float = 4.0, nodata: float = -9999.0, depth: float = 0.95,
                   lv_ratio: float = 4.0, n_soils: int = 3) -> Model:
    """Vectorised twin of dem_model (same rules, same node/link order, identical arrays - checked by
    tests/test_abi.py on the Ravone window) for catchments of millions of nodes."""
    dem = np.asarray(dem, np.float32)
    ny, nx = dem.shape
    valid = dem != nodata
    thick = np.array(dem_layer_thicknesses(depth))
    nz = len(thick) + 1
    centre = np.cumsum(thick) - 0.5 * thick
    bottom = np.cumsum(thick)
    gy, gx = np.gradient(np.where(valid, dem, np.nan).astype(np.float64), cell)
    slope = np.sqrt(np.nan_to_num(gx) ** 2 + np.nan_to_num(gy) ** 2)
    soil_depth = np.clip(depth * (1.0 - 0.8 * np.clip(slope, 0, 1)), 0.25, depth)
    exists = np.zeros((nz, ny, nx), bool)
    exists[0] = valid
    for l in range(1, nz):
        exists[l] = valid & ((bottom[l - 1] <= soil_depth + 1e-9) | (l == 1))
    index = np.where(exists, np.cumsum(exists.ravel()).reshape(exists.shape) - 1, -1)
    n, ns, area = int(exists.sum()), int(valid.sum()), cell * cell
    L, R, C = np.nonzero(exists)                                  # layer-major, row-major order = node order
    zs64 = dem.astype(np.float64)

    pad = np.pad(valid, 1, constant_values=False)
    zpad = np.pad(np.where(valid, zs64, np.inf), 1, constant_values=np.inf)
    edge = np.zeros((ny, nx), bool)
    nmin = np.full((ny, nx), np.inf)
    for dr, dc in LATERAL_OFFSETS:
        nb_valid = pad[1 + dr:1 + dr + ny, 1 + dc:1 + dc + nx]
        edge |= valid & ~nb_valid
        nmin = np.minimum(nmin, zpad[1 + dr:1 + dr + ny, 1 + dc:1 + dc + nx])
    has_nbr = np.isfinite(nmin)
    is_min = np.where(has_nbr, zs64 < nmin, True)
    outward2d = edge & (is_min | (has_nbr & (zs64 <= nmin + 0.5)))
    slp2d = np.maximum(slope, 0.001).astype(np.float32).astype(np.float64)

    zsn, outward = zs64[R, C], outward2d[R, C]
    soil = L > 0
    lc = np.maximum(L - 1, 0)
    x = C * cell
    y = (ny - 1 - R) * cell
    z = np.where(soil, (zsn - centre[lc]).astype(np.float32).astype(np.float64), zsn)
    size = np.where(soil, area * thick[lc], area)
    surf = (~soil).astype(np.uint8)
    below = np.zeros_like(exists)
    below[:-1] = exists[1:]
    last = soil & ~below[L, R, C]
    btype = np.zeros(n, np.uint8); bslope = np.zeros(n); barea = np.zeros(n)
    m = ~soil & outward
    btype[m] = capi.BND_RUNOFF; bslope[m] = slp2d[R, C][m]; barea[m] = cell
    m = soil & ~last & outward
    btype[m] = capi.BND_FREE_LATERAL_DRAINAGE; bslope[m] = slp2d[R, C][m]
    barea[m] = (cell * thick[lc]).astype(np.float32).astype(np.float64)[m]
    btype[last] = capi.BND_FREE_DRAINAGE; bslope[last] = 0.0; barea[last] = area
    soil_index = ((zsn[soil].astype(np.int64) // 7 + L[soil] // 4) % n_soils).astype(np.uint16)

    ipad = np.pad(index, ((1, 1), (1, 1), (1, 1)), constant_values=-1)
    cand_to = np.full((n, 10), -1, np.int64)
    cand_dir = np.zeros((n, 10), np.uint8)
    cand_area = np.zeros((n, 10))
    cand_to[:, 0] = ipad[L, R + 1, C + 1]; cand_dir[:, 0] = capi.LINK_UP; cand_area[:, 0] = area
    cand_to[:, 1] = ipad[L + 2, R + 1, C + 1]; cand_dir[:, 1] = capi.LINK_DOWN; cand_area[:, 1] = area
    lat = np.where(soil, (cell * thick[lc]).astype(np.float32).astype(np.float64), cell) * 0.5
    for k, (dr, dc) in enumerate(LATERAL_OFFSETS):
        cand_to[:, 2 + k] = ipad[L + 1, R + 1 + dr, C + 1 + dc]
        cand_dir[:, 2 + k] = capi.LINK_LATERAL
        cand_area[:, 2 + k] = lat
    mask = cand_to >= 0
    idx = np.arange(n, dtype=np.int64)
    soils = usda_soils()
    soils = [soils[3], soils[4], soils[8]][:n_soils]
    return Model(n=n, ns=ns, x=x.astype(float), y=y.astype(float), z=z, size=size, is_surface=surf, btype=btype, bslope=bslope,
                 barea=barea, link_node=np.broadcast_to(idx[:, None], (n, 10))[mask].astype(np.uint32),
                 link_to=cand_to[mask].astype(np.uint32), link_dir=cand_dir[mask], link_area=cand_area[mask],
                 soil_index=soil_index, soils=soils, psi0_soil=-3.0, lv_ratio=lv_ratio,
                 numerics=(min(6.0, cell / 20.0), 3600.0, 150, 10, 9, 2), cell_area=area, shape=(nx, ny, nz),
                 meta=dict(kind="dem", layers=list(thick), index=index, cell=cell))


def synthetic_dem(ny: int = 1208, nx: int = 519, seed: int = 2021, nodata: float = -9999.0) -> np.ndarray:
    """Deterministic fractal hill-slope terrain with an irregular catchment outline, shaped like the
    Ravone DEM (DATA/DEM/DEM_Ravone.hdr: 519 x 1208 cells of 4 m, 70-358 m a.s.l., two thirds valid)."""
    rng = np.random.RandomState(seed)
    ky = np.fft.fftfreq(ny)[:, None]; kx = np.fft.fftfreq(nx)[None, :]
    k = np.sqrt(kx * kx + ky * ky); k[0, 0] = 1.0
    spec = (rng.normal(size=(ny, nx)) + 1j * rng.normal(size=(ny, nx))) / k ** 2.1
    rough = np.real(np.fft.ifft2(spec)); rough = (rough - rough.min()) / (rough.max() - rough.min())
    yy, xx = np.mgrid[0:ny, 0:nx]
    axis = nx * (0.5 + 0.18 * np.sin(yy / ny * 5.0) + 0.07 * np.sin(yy / ny * 17.0))      # meandering valley axis
    dist = np.abs(xx - axis) / nx
    elev = 70.0 + 200.0 * (1.0 - yy / ny) * 0.6 + 380.0 * dist + 60.0 * rough
    half_width = 0.30 + 0.10 * np.sin(yy / ny * 9.0) + 0.05 * (rough - 0.5)
    dem = np.where(dist < half_width, elev, nodata).astype(np.float32)
    return dem


def surface_only_model(nx: int = 12, ny: int = 10, cell: float = 5.0) -> Model:
    """Edge case: a sheet of surface nodes only (no soil, no vertical links): pure St-Venant runoff."""
    full = catchment_model(nx, ny, 1, cell=cell)
    full.psi0_surface = 0.003           # 3 mm of water everywhere to start with
    full.meta = dict(kind="surface_only")
    return full


def soil_only_column(n_nodes: int = 40, dz: float = 0.05) -> Model:
    """Edge case: nrSurfaceNodes = 0 - a soil column with a prescribed-potential top node and free
    drainage at the bottom (no surface node at all)."""
    n = n_nodes
    i = np.arange(n)
    z = -(dz * (i + 0.5))
    btype = np.zeros(n, np.uint8); barea = np.zeros(n)
    btype[n - 1] = capi.BND_FREE_DRAINAGE; barea[n - 1] = 1.0
    btype[0] = capi.BND_PRESCRIBED; barea[0] = 1.0
    ln, lt, ld = [], [], []
    for k in range(n):
        if k > 0:
            ln.append(k); lt.append(k - 1); ld.append(capi.LINK_UP)
        if k < n - 1:
            ln.append(k); lt.append(k + 1); ld.append(capi.LINK_DOWN)
    return Model(n=n, ns=0, x=np.zeros(n), y=np.zeros(n), z=z, size=np.full(n, dz), is_surface=np.zeros(n, np.uint8),
                 btype=btype, bslope=np.zeros(n), barea=barea, link_node=np.array(ln, np.uint32),
                 link_to=np.array(lt, np.uint32), link_dir=np.array(ld, np.uint8), link_area=np.ones(len(ln)),
                 soil_index=np.zeros(n, np.uint16), soils=[LOAM], psi0_soil=-2.5, numerics=(1.0, 3600.0, 150, 10, 10, 3),
                 cell_area=1.0, shape=(1, 1, n), meta=dict(kind="soil_only", prescribed_node=0, prescribed_H=-0.2))


def random_model(seed: int, nx: int = 9, ny: int = 8, nz: int = 5) -> Model:
    """Randomised irregular graph for fuzz tests: random holes, random column depths, random layer thickness per
    column, a random subset of the eight lateral neighbours linked (so chunks mix link kinds and index offsets),
    random soils out of the 12 USDA classes, random boundary types on edge and bottom nodes, random relief."""
    rng = np.random.RandomState(seed)
    cell = float(rng.choice([2.0, 5.0, 10.0]))
    area = cell * cell
    valid = rng.rand(ny, nx) > 0.12
    valid[ny // 2, nx // 2] = True
    layers = rng.randint(1, nz, size=(ny, nx))                  # soil layers per column: 1 .. nz-1
    thick = rng.choice([0.05, 0.1, 0.2], size=nz - 1)
    relief = rng.rand(ny, nx) * 1.5 + 0.05 * np.arange(nx)[None, :] * cell + 0.03 * np.arange(ny)[:, None] * cell + 50.0
    index = -np.ones((nz, ny, nx), np.int64)
    n = 0
    for l in range(nz):
        for r in range(ny):
            for c in range(nx):
                if valid[r, c] and (l == 0 or l <= layers[r, c]):
                    index[l, r, c] = n; n += 1
    ns = int((index[0] >= 0).sum())
    x = np.zeros(n); y = np.zeros(n); z = np.zeros(n); size = np.zeros(n)
    surf = np.zeros(n, np.uint8); btype = np.zeros(n, np.uint8); bslope = np.zeros(n); barea = np.zeros(n)
    soil_index = np.zeros(n - ns, np.uint16)
    ln, lt, ld, la = [], [], [], []
    top = np.concatenate([[0.0], np.cumsum(thick)[:-1]])
    for l in range(nz):
        for r in range(ny):
            for c in range(nx):
                i = index[l, r, c]
                if i < 0:
                    continue
                x[i], y[i] = c * cell, r * cell
                edge = r in (0, ny - 1) or c in (0, nx - 1)
                if l == 0:
                    z[i] = relief[r, c]; size[i] = area; surf[i] = 1
                    if edge and rng.rand() < 0.5:
                        btype[i] = capi.BND_RUNOFF; bslope[i] = 0.02 + 0.05 * rng.rand(); barea[i] = cell
                else:
                    z[i] = relief[r, c] - (top[l - 1] + 0.5 * thick[l - 1]); size[i] = area * thick[l - 1]
                    soil_index[i - ns] = rng.randint(0, 12)
                    last = (l + 1 >= nz) or index[l + 1, r, c] < 0
                    if last:
                        btype[i] = rng.choice([capi.BND_FREE_DRAINAGE, capi.BND_NONE], p=[0.7, 0.3]); barea[i] = area
                    elif edge and rng.rand() < 0.5:
                        btype[i] = capi.BND_FREE_LATERAL_DRAINAGE; bslope[i] = 0.03; barea[i] = cell * thick[l - 1]
                if l > 0:
                    ln.append(i); lt.append(index[l - 1, r, c]); ld.append(capi.LINK_UP); la.append(area)
                if l + 1 < nz and index[l + 1, r, c] >= 0:
                    ln.append(i); lt.append(index[l + 1, r, c]); ld.append(capi.LINK_DOWN); la.append(area)
                lat = cell if l == 0 else cell * thick[l - 1]
                for dr, dc in LATERAL_OFFSETS:
                    rr, cc = r + dr, c + dc
                    # the same coin for both directions of a link: keyed on the unordered cell pair
                    key = hash((seed, l, min((r, c), (rr, cc)), max((r, c), (rr, cc)))) % 100
                    if 0 <= rr < ny and 0 <= cc < nx and index[l, rr, cc] >= 0 and key < 80:
                        ln.append(i); lt.append(index[l, rr, cc]); ld.append(capi.LINK_LATERAL); la.append(lat * 0.5)
    return Model(n=n, ns=ns, x=x, y=y, z=z, size=size, is_surface=surf, btype=btype, bslope=bslope, barea=barea,
                 link_node=np.array(ln, np.uint32), link_to=np.array(lt, np.uint32), link_dir=np.array(ld, np.uint8),
                 link_area=np.array(la), soil_index=soil_index, soils=usda_soils(), psi0_soil=float(-0.5 - 3.0 * rng.rand()),
                 lv_ratio=float(rng.choice([1.0, 4.0, 10.0])), numerics=(0.5, 3600.0, 150, 10, 10, 3), cell_area=area,
                 shape=(nx, ny, nz), meta=dict(kind="random", seed=seed))
