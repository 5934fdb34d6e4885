"""N>1 path on CPU: world_size 2 and 3 over gloo (tests/partition_worker.py)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world,case,port", [(2, "grid", 29731), (3, "grid", 29732), (2, "ragged", 29733)])
def test_sharded_jacobi_over_gloo(product, world, case, port):
    procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "partition_worker.py"), str(r), str(world), str(port), case],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)


def test_strips_are_row_blocks_cut_at_chunk_boundaries(product):
    """512-wide grid, 8 ranks: every rank owns 64 full rows of every layer; halos are one row."""
    m = cm.catchment_model(128, 64, 4)
    cm.build(product, m, finalize=False)
    assert 250 * m.n < int(product.lib.sf3d_host_bytes()) < 600 * m.n          # the resident staging model: ~330 B per node, counted page by page
    owner = product.owner_map(4, m.n).reshape(4, 64, 128)
    for r in range(4):
        assert (owner[:, 16 * r:16 * (r + 1), :] == r).all()           # 64 rows / 4 ranks, all layers (columns stay whole)
    up = product.halo_list(1, 4, 0, 1)                                   # rank 1 receives from rank 0: row 15 of each layer
    assert len(up) == 128 * 4 and (np.sort(up) == up).all()
    assert len(product.halo_list(1, 4, 3, 1)) == 0                       # strips only talk to adjacent strips
    assert np.array_equal(product.halo_list(0, 4, 1, 0), up)             # what 0 sends to 1 is what 1 receives from 0
    assert product.lib.sf3d_dist_prepare(5, 4) == capi.PARAMETER_ERROR
    product.lib.sf3d_clean()


def test_edge_rows_of_a_regular_strip_lie_in_the_lists_as_layer_times_nx_plus_column(product):
    """What the in-launch record hand-overs count on (sf3d_host_build.inc edge_rows_direct): on a regular grid a rank's edge row
    lies in the neighbour's send / receive lists as base + layer * NX + column - the kernels of the paired pass and of the resident loop then
    store and poll their records without a walk through a chunk's send list.  A change of the partition's list order would not break a
    result (the host checks the layout and falls back to the two-launch form / the list walk), it would silently cost the speed: held here."""
    nx, ny, nz, world = 128, 64, 5, 4
    m = cm.catchment_model(nx, ny, nz)
    cm.build(product, m, finalize=False)
    ns, rows = nx * ny, ny // world
    for r in range(world):
        for peer, r_edge, r_halo in ((r - 1, r * rows, r * rows - 1), (r + 1, (r + 1) * rows - 1, (r + 1) * rows)):
            if peer < 0 or peer >= world:
                continue
            layer, col = np.meshgrid(np.arange(nz), np.arange(nx), indexing="ij")
            assert np.array_equal(product.halo_list(r, world, peer, 0), (layer * ns + r_edge * nx + col).ravel()), (r, peer, "send")
            assert np.array_equal(product.halo_list(r, world, peer, 1), (layer * ns + r_halo * nx + col).ravel()), (r, peer, "receive")
    product.lib.sf3d_clean()
    product.check(product.lib.sf3d_dist_prepare(0, 1), "dist_prepare")


def _models():
    from tests.scenarios import ravone_project_model
    return {"grid": lambda: cm.catchment_model(128, 48, 5), "het": lambda: cm.catchment_model(64, 40, 6, heterogeneous=True),
            "ragged": lambda: cm.ragged_model(9, 24, 4), "holes": lambda: cm.random_model(23, nx=70, ny=45, nz=5),
            "project window": lambda: ravone_project_model((980, 1060, 330, 420))}


@pytest.mark.parametrize("name,world", [("grid", 2), ("grid", 8), ("het", 3), ("ragged", 2), ("holes", 3), ("project window", 2), ("project window", 4)])
def test_strip_local_build_gives_the_partition_of_the_global_build(product, name, world):
    """STRIP-LOCAL BUILD (include/sf3d.h: sf3d_dist_bounds).  A rank that stages only the columns of its strip and the one-cell ring
    of columns around them - found from sf3d_dist_bounds and the model's links alone (catchment.strip_nodes), global indices - must
    arrive at the partition the global build gives it: the same owner for every node it staged (-1 for the others), the same halo
    lists towards every peer in both directions.  Host logic only (no device); what it buys is measured too: the resident staging
    memory of a rank (sf3d_host_bytes) shrinks to its strip."""
    m = _models()[name]()
    for rank in range(world):
        cm.build(product, m, dist=(rank, world, None), finalize=False)
        g_owner = product.owner_map(world, m.n)
        g_lists = {(p, d): product.halo_list(rank, world, p, d) for p in range(world) for d in (0, 1)}
        g_bytes = int(product.lib.sf3d_host_bytes())
        product.lib.sf3d_clean()
        cm.build(product, m, dist=(rank, world, None), finalize=False, sparse=True)
        keep = m.meta["staged"]
        s_owner = product.owner_map(world, m.n)
        assert np.array_equal(s_owner[keep], g_owner[keep]) and (s_owner[~keep] == -1).all(), (name, world, rank)
        assert (g_owner[keep] == rank).sum() == (g_owner == rank).sum()          # every node of the strip was staged
        for (p, d), want in g_lists.items():
            assert np.array_equal(product.halo_list(rank, world, p, d), want), (name, world, rank, p, d)
        s_bytes = int(product.lib.sf3d_host_bytes())
        if world >= 4 and name == "grid":
            assert s_bytes < 0.6 * g_bytes, (s_bytes, g_bytes)
        product.lib.sf3d_clean()
    product.check(product.lib.sf3d_dist_prepare(0, 1), "dist_prepare")


def test_strip_local_build_without_its_halo_columns_is_refused(product):
    """a rank that stages its own columns but forgets the ring around them: one of its nodes links to a node that never got a class -
    MissingDataError from the partition (with the two node numbers on stderr) instead of a wrong halo later"""
    m = cm.catchment_model(64, 32, 4)
    bounds = product.dist_bounds(m.ns, 2)
    assert list(bounds) == [0, 1024, 2048]                                          # 32 rows of 64 cells cut in two at a multiple of 64
    col = cm.column_of(m)
    product.check(product.lib.sf3d_dist_prepare(1, 2), "dist_prepare")
    product.check(product.lib.sf3d_initialize(m.n, m.ns, 8, 1, 0, 0, 0), "initialize")
    product.check(product.lib.sf3d_set_surface_properties(0, 0.05), "surface")
    s = m.soils[0]
    product.check(product.lib.sf3d_set_soil_properties(0, 0, s["alpha"], s["n"], 1.0 - 1.0 / s["n"], s["he"], s["theta_r"], s["theta_s"], s["ksat"], s["L"],
                                                        s["organic_matter"], s["clay"]), "soil")
    own = col >= bounds[1]
    for a, b in cm._runs(own, 0, m.n):
        product.set_nodes_bulk(a, m.x[a:b], m.y[a:b], m.z[a:b], m.size[a:b], m.is_surface[a:b], m.btype[a:b], m.bslope[a:b], m.barea[a:b])
    lk = own[m.link_node]
    product.set_links_bulk(m.link_node[lk], m.link_to[lk], m.link_dir[lk], m.link_area[lk])
    for a, b in cm._runs(own, 0, m.ns):
        product.set_surface_bulk(a, np.zeros(b - a, np.uint16))
    for a, b in cm._runs(own, m.ns, m.n):
        product.set_soil_bulk(a, m.soil_index[a - m.ns:b - m.ns], np.zeros(b - a, np.uint16))
    cnt = capi.u32(0)
    assert product.lib.sf3d_dist_halo(1, 2, 0, 1, 0, None, capi.C.byref(cnt)) == capi.MISSING_DATA_ERROR
    product.lib.sf3d_clean()
    product.check(product.lib.sf3d_dist_prepare(0, 1), "dist_prepare")
