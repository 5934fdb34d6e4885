"""Multi-rank path of the HIP product on real hardware: N processes (one per rank) share the GPU of
the test box - the exchange (HIP-IPC windows, device-side flags, halo puts, all-gather of partial
sums) is the same code that runs one rank per GPU.  Each rank owns a row strip; the union of the
owned nodes must match the oracle like the single-rank run does."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from criteria3d_amd import capi, catchment as cm
from tests.tolerances import HEAT_RTOL, WATER_RTOL, assert_water_nodes

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
RTOL = WATER_RTOL          # 1e-9 (tests/tolerances.py); north_star: 1e-6


def _run_ranks_once(world, case, tmp_path, port, env):
    import os
    procs, outs = [], []
    for r in range(world):
        out = tmp_path / f"rank{r}_{port}.npz"
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "scripts" / "multirank_worker.py"), str(r), str(world),
                                       str(port), case, str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                      env={**os.environ, "SF3D_DIST_VERBOSE": "1", "SF3D_DIST_TIMEOUT_S": os.environ.get("SF3D_DIST_TIMEOUT_S", "60"),      # (ranks taking turns on one GPU: the exchange's 10 s bound is for ranks with a GPU each)
                                           **(env or {})}))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    return all(p.returncode == 0 for p in procs), logs, outs


SETUP_FAILURES = ("did not answer within the bounded wait", "no answer from rank", "no answer through the window", "self-check kernel failed")


def run_ranks(world, case, tmp_path, port, env=None):
    """`world` rank processes sharing the one GPU of the box.  Several processes' spinning kernels on one device depend on the
    hardware scheduler running them side by side; once in a dozen suite runs (round 4, job 33: the first contact of a 2-rank case
    while other GPU processes of the suite were alive) a rank's start-up self-check timed out.  That is the box, not the exchange
    (one process per GPU is the product's layout): such a SET-UP failure is retried once, loudly; anything else fails at once."""
    ok, logs, outs = _run_ranks_once(world, case, tmp_path, port, env)
    if not ok and any(k in log for log in logs for k in SETUP_FAILURES):
        import warnings
        warnings.warn(f"multi-rank case {case} x{world}: exchange set-up failed on the shared GPU, retrying once:\n" + "\n".join(l[-1500:] for l in logs))
        ok, logs, outs = _run_ranks_once(world, case, tmp_path, port + 400, env)
    assert ok, "\n".join(logs)
    return [np.load(o) for o in outs]


def oracle_reference(oracle, case):
    if case == "c2f20":
        m, plan = cm.catchment_model(64, 64, 10), [20.0, 0.0]
    elif case == "c2f60":
        m, plan = cm.catchment_model(64, 64, 10), [60.0, (0.0, 150)]
    elif case == "het":
        m, plan = cm.catchment_model(48, 40, 6, heterogeneous=True), [20.0, (0.0, 150)]
    elif case == "ravone":
        from tests.scenarios import ravone_project_model
        m, plan = ravone_project_model(None), [(20.0, 3)]
    elif case == "random":
        m, plan = cm.random_model(17, nx=12, ny=40, nz=5), [12.0, (0.0, 30)]
    elif case == "c4f20h0":
        from tests.scenarios import oracle_c4_f20
        m, ref = oracle_c4_f20(oracle, 1)          # (shared with the single-GPU full-size tests: one oracle run per session)
        return m, [(ref[0][0], ref[0][1])]
    else:
        m, plan = cm.ragged_model(9, 24, 4), [10.0, 0.0]
    oracle.lib.sf3d_reset_solver_state()
    cm.build(oracle, m, threads=16 if case in ("ravone", "c4f20h0") else 1)
    out = []
    for item in plan:
        mm, mx = item if isinstance(item, tuple) else (item, None)
        _, dts = cm.run_hour(oracle, m, mm, max_steps=mx)
        out.append((dts, cm.snapshot(oracle, m)))
    return m, out


@pytest.mark.parametrize("world,case,port", [(2, "c2f20", 29611), (3, "c2f60", 29612), (4, "het", 29613), (2, "ragged", 29614), (3, "random", 29615),
                                             (4, "ravone", 29616),       # the Ravone project (5.85 M nodes, irregular outline) cut into four strips
                                             (2, "c4f20h0", 29618),      # C4 in two strips: 2.6 M nodes per rank, above the Infinity Cache - the paired sweep's case
                                             (8, "c4f20h0", 29617)])     # BASELINE config 4's cut: C4 in eight strips, hour 0 of F20
def test_sharded_run_matches_oracle(oracle, tmp_path, world, case, port):
    # C4: with the paired sweep on every strip (k_sweep_pair<DIST> + k_sweep_bnd; forced: eight strips of C4 are cache-resident and
    # would keep single sweeps), held against the oracle AND, bit for bit, against the same sharded run with single sweeps
    pair_env = {"SF3D_PAIR_SWEEP": "1"} if case in ("c4f20h0", "ravone") else None      # (ravone: k_sweep_pair_masked<DIST> on every strip)
    ranks = run_ranks(world, case, tmp_path, port, env=pair_env)
    if pair_env:
        assert all(int(res["sweep_launches"][1]) > 0 for res in ranks), [res["sweep_launches"] for res in ranks]
    if case == "c4f20h0" and world == 2:      # (eight strips: against the oracle only - sixteen rank processes of C4 are 25 s of model building)
        single = run_ranks(world, case, tmp_path, port + 20, env={"SF3D_PAIR_SWEEP": "0", "SF3D_RESIDENT_SWEEP": "0"})
        assert all(int(res["sweep_launches"][1]) == 0 for res in single)
        own = ranks[0]["owner"]
        for r in range(world):
            mine = own == r
            assert np.array_equal(ranks[r]["H_h0"][mine], single[r]["H_h0"][mine]) and np.array_equal(ranks[r]["Se_h0"][mine], single[r]["Se_h0"][mine]), r
            assert np.array_equal(ranks[r]["dts_h0"], single[r]["dts_h0"]) and np.array_equal(ranks[r]["counters"], single[r]["counters"]), r
    m, ref = oracle_reference(oracle, case)
    owner = ranks[0]["owner"]
    assert set(np.unique(owner)) == set(range(world))
    for h, (dts, snap) in enumerate(ref):
        H = np.empty(m.n); Se = np.empty(m.n)
        for r, res in enumerate(ranks):
            mine = owner == r
            H[mine] = res[f"H_h{h}"][mine]; Se[mine] = res[f"Se_h{h}"][mine]
            np.testing.assert_allclose(res[f"dts_h{h}"], dts, rtol=1e-12)            # identical decisions on every rank
            assert abs(float(res[f"storage_h{h}"]) - snap["storage"]) <= RTOL * abs(snap["storage"])
        assert_water_nodes(H, snap["H"], f"{case} in {world} strips, hour {h}: H")      # the strips' union holds the single-GPU oracle's bits
        assert_water_nodes(Se, snap["Se"], f"{case} in {world} strips, hour {h}: Se")
        # boundary sums are reported per rank for its own nodes: they add up to the global ones
        for k in ("runoff", "drainage", "lateral"):
            tot = sum(float(res[f"{k}_h{h}"]) for res in ranks)
            assert abs(tot - snap[k]) <= RTOL * max(abs(snap[k]), 1e-3), (k, tot, snap[k])
    assert all((res["counters"] == ranks[0]["counters"]).all() for res in ranks)
    if case == "c4f20h0":
        # strip-local device models: a rank holds its 64 rows + one halo row on either side of 512, not the whole grid
        product = capi.load_product()
        product.check(product.lib.sf3d_reset_solver_state(), "reset")
        cm.build(product, m)
        whole = int(product.lib.sf3d_device_bytes())
        whole_host = int(product.lib.sf3d_host_bytes())
        product.lib.sf3d_clean()
        per_rank = [int(res["device_bytes"]) for res in ranks]
        assert whole > 3e9 and max(per_rank) < (0.2 if world == 8 else 0.6) * whole, (whole, per_rank)
        # ... and on the host: once connected, a rank keeps the pages of the staging model that hold ITS nodes (sf3d_host_bytes: resident
        # pages of the global copy + the strip-local copy) - at eight ranks below a quarter of what one rank holds for the whole grid,
        # at the start and at the end of the run (bulk setters and getters leave the other ranks' pages alone)
        host = [max(int(res["host_bytes"]), int(res["host_bytes_end"])) for res in ranks]
        assert whole_host > 1.5e9 and max(host) < (0.25 if world == 8 else 0.8) * whole_host, (whole_host, host)      # (two strips: half of the global copy + a strip-local copy without its graph-build arrays: measured 0.75)


@pytest.mark.parametrize("world,case,port,local", [(2, "c2f20", 29651, "1"), (3, "c2f60", 29653, "1"), (3, "c2f60", 29655, "0"),
                                                   (3, "holes", 29657, "1"), (2, "holes", 29659, "0"), (2, "projwin", 29661, "1")])      # masked grids: k_sweep_pair_masked<DIST>
def test_paired_sweep_on_strips_is_bitwise_the_single_sweeps(tmp_path, world, case, port, local):
    """k_sweep_pair<DIST> / k_sweep_pair_masked<DIST> + k_sweep_bnd on small grids (forced: they are cache-resident), infiltration and
    runoff regime (Courant refusals, restore-best steps, approximations of odd and even length), full boxes and masked grids (random
    holes, a window of the Ravone project: strips cut mid-row), strip-local models and the global-index checker mode: every
    owned node's H and Se, every accepted dt and every work counter equal to the run with single sweeps; paired passes on every rank"""
    pair = run_ranks(world, case, tmp_path, port, env={"SF3D_PAIR_SWEEP": "1", "SF3D_PAIR_W": "6", "SF3D_DIST_LOCAL": local})
    single = run_ranks(world, case, tmp_path, port + 1, env={"SF3D_PAIR_SWEEP": "0", "SF3D_RESIDENT_SWEEP": "0", "SF3D_DIST_LOCAL": local})
    runs = [pair]
    if (case, local) in (("c2f60", "1"), ("holes", "1")):
        # the default pass hands the edge rows' first iterate over as tagged records while it runs (one launch, ONE exchange per pass);
        # SF3D_PAIR_RECORDS=0 is the two-launch form (k_sweep_bnd, two exchanges per pass): same bits, more mailbox rounds
        plain = run_ranks(world, case, tmp_path, port + 40, env={"SF3D_PAIR_SWEEP": "1", "SF3D_PAIR_W": "6", "SF3D_DIST_LOCAL": local, "SF3D_PAIR_RECORDS": "0"})
        runs.append(plain)
        e_rec, e_plain, passes = int(pair[0]["epochs"]), int(plain[0]["epochs"]), int(pair[0]["sweep_launches"][1])
        assert e_rec > 0 and 0 < e_plain - e_rec <= passes, (e_rec, e_plain, passes)      # one exchange less per pass that did not end on its first iterate
    owner = pair[0]["owner"]
    for run in runs:
        for r in range(world):
            assert int(run[r]["sweep_launches"][1]) > 0 and int(single[r]["sweep_launches"][1]) == 0, (r, run[r]["sweep_launches"])
            mine = owner == r
            for k in run[r].files:
                if k.startswith(("H_h", "Se_h")):
                    assert np.array_equal(run[r][k][mine], single[r][k][mine]), (r, k)
                elif k.startswith("dts_h") or k == "counters":
                    assert np.array_equal(run[r][k], single[r][k]), (r, k)


@pytest.mark.parametrize("world,case,port,local", [(2, "c2f20", 29671, "1"), (4, "c2f60", 29673, "1"), (2, "c2f60", 29675, "0"), (4, "het64", 29677, "1"),
                                                   (2, "tall", 29679, "1")])       # twenty layers, two rows per block: the kernel shape of one of eight strips of C4 (five chunks per wave, eight waves)
# (that strip itself - C4 in eight strips, a 256-block persistent kernel per rank - cannot be tested with the ranks on ONE GPU: the kernels do not fit on it together, and
# persistent kernels that wait for each other's time slices starve; the library turns the loop off in that layout, sf3d_dist_connect: "resident sweep loop off")
def test_resident_sweep_loop_on_strips_is_bitwise_the_single_sweeps(tmp_path, world, case, port, local):
    """k_sweep_resident<DIST> (csrc/sf3d_resident.inc: all Jacobi iterations of an approximation in one launch; the edge rows hand their
    new iterate to the neighbouring rank as tagged records in its window, block 0 all-gathers the norm once per iteration) against
    single sweeps on the same strips: infiltration and runoff regime, strip-local models and the global-index checker mode, two and
    four ranks - every owned node's H and Se, every accepted dt, every work counter and the number of exchange epochs"""
    res = run_ranks(world, case, tmp_path, port, env={"SF3D_RESIDENT_SWEEP": "1", "SF3D_PAIR_SWEEP": "0", "SF3D_DIST_LOCAL": local})
    single = run_ranks(world, case, tmp_path, port + 1, env={"SF3D_RESIDENT_SWEEP": "0", "SF3D_PAIR_SWEEP": "0", "SF3D_DIST_LOCAL": local})
    owner = res[0]["owner"]
    for r in range(world):
        assert int(res[r]["sweep_launches"][2]) > 0 and int(res[r]["sweep_launches"][0]) == 0, (r, res[r]["sweep_launches"])
        assert int(single[r]["sweep_launches"][2]) == 0 and int(single[r]["sweep_launches"][0]) > 0, (r, single[r]["sweep_launches"])
        mine = owner == r
        for k in res[r].files:
            if k.startswith(("H_h", "Se_h")):
                assert np.array_equal(res[r][k][mine], single[r][k][mine]), (r, k)
            elif k.startswith("dts_h") or k == "counters":
                assert np.array_equal(res[r][k], single[r][k]), (r, k)


@pytest.mark.parametrize("world,case,port", [(3, "c2f60", 29641), (2, "ragged", 29643), (3, "random", 29645)])
def test_strip_local_models_equal_the_global_index_path(tmp_path, world, case, port):
    """the default for world > 1 - every rank uploads only its strip + halo, renumbered locally - against the checker mode
    SF3D_DIST_LOCAL=0 (every rank uploads the whole global model, indices global): the same bits in H and Se of every owned node,
    the same accepted steps and counters, the balance sums equal to rounding; and less device memory"""
    local = run_ranks(world, case, tmp_path, port)
    glob = run_ranks(world, case, tmp_path, port + 1, env={"SF3D_DIST_LOCAL": "0"})
    owner = glob[0]["owner"]
    for r in range(world):
        mine = owner == r
        for k in local[r].files:
            if k.startswith(("H_h", "Se_h")):
                assert np.array_equal(local[r][k][mine], glob[r][k][mine]), (r, k)
            elif k.startswith("dts_h") or k == "counters":
                assert np.array_equal(local[r][k], glob[r][k]), (r, k)
            elif k.startswith(("storage_h", "total_water_h", "runoff_h", "drainage_h", "lateral_h")):
                # sums over the rank's nodes: the local numbering groups the nodes into other blocks, so the same terms are added in
                # another association (identical on regular grids, last bits on irregular ones)
                np.testing.assert_allclose(local[r][k], glob[r][k], rtol=1e-12, atol=1e-300)
        assert int(local[r]["device_bytes"]) < int(glob[r]["device_bytes"])


@pytest.mark.parametrize("world,port", [(3, 29622)])
def test_sharded_heat_matches_oracle(oracle, tmp_path, world, port):
    """coupled water + heat (latent heat, atmosphere boundary on every column) sharded by row strips: halo temperatures
    travel with every heat sweep, halo conductivities are recomputed locally, decisions are all-gathered"""
    ranks = run_ranks(world, "heat", tmp_path, port)
    m = cm.with_heat_surface(cm.catchment_model(40, 48, 6, heterogeneous=True))
    oracle.lib.sf3d_reset_solver_state()
    cm.build(oracle, m, threads=1, heat=cm.Heat(water=True, latent=True, save_mode=0))
    owner = ranks[0]["owner"]
    soil = np.arange(m.n) >= m.ns
    for h, mm in enumerate([4.0, 0.0]):
        cm.apply_heat_forcing(oracle, m, h)
        _, dts = cm.run_hour(oracle, m, mm)
        To, Ho = oracle.temperature(0, m.n), oracle.total_potential(0, m.n)
        T = np.empty(m.n); H = np.empty(m.n)
        for r, res in enumerate(ranks):
            mine = owner == r
            T[mine] = res[f"T_h{h}"][mine]; H[mine] = res[f"H_h{h}"][mine]
            np.testing.assert_allclose(res[f"dts_h{h}"], dts, rtol=1e-12)
        assert np.max(np.abs(T[soil] - To[soil]) / To[soil]) < HEAT_RTOL
        assert np.max(np.abs(H - Ho) / np.maximum(np.abs(Ho), 1e-9)) < HEAT_RTOL


@pytest.mark.parametrize("world,case,port", [(2, "c2f20", 29671), (3, "c2f60", 29673), (2, "projwin", 29675)])
def test_host_memory_windows_give_the_same_bits(tmp_path, world, case, port):
    """the fall-back transport (sf3d_dist_finalize(2): every rank's window a second time in POSIX shared memory, registered with HIP,
    reached by the kernels with the same system-scope loads and stores) forced with SF3D_EXCHANGE=host: the launcher's status round
    decides for it on every rank, the run gives the bits of the device windows - H, Se, accepted dt, counters - with the paired sweep
    on the strips"""
    dev = run_ranks(world, case, tmp_path, port, env={"SF3D_PAIR_SWEEP": "1", "SF3D_PAIR_W": "6"})
    host = run_ranks(world, case, tmp_path, port + 1, env={"SF3D_PAIR_SWEEP": "1", "SF3D_PAIR_W": "6", "SF3D_EXCHANGE": "host"})
    owner = dev[0]["owner"]
    for r in range(world):
        assert int(dev[r]["transport"]) == 1 and int(host[r]["transport"]) == 2, (r, dev[r]["transport"], host[r]["transport"])
        assert int(host[r]["sweep_launches"][1]) > 0
        mine = owner == r
        for k in dev[r].files:
            if k.startswith(("H_h", "Se_h")):
                assert np.array_equal(dev[r][k][mine], host[r][k][mine]), (r, k)
            elif k.startswith("dts_h") or k == "counters":
                assert np.array_equal(dev[r][k], host[r][k]), (r, k)


def test_forced_rccl_on_a_shared_gpu_fails_loudly(tmp_path, monkeypatch):
    """SF3D_EXCHANGE=rccl asks for the ncclSend/ncclRecv exchange; with two ranks on ONE GPU there is no communicator to be had
    (RCCL refuses two ranks on a device): every rank reports it at connect time - nothing hangs, nothing falls back silently"""
    monkeypatch.setenv("SF3D_EXCHANGE", "rccl")
    procs = [subprocess.Popen([sys.executable, str(ROOT / "scripts" / "multirank_worker.py"), str(r), "2", "29631", "c2f20", str(tmp_path / f"r{r}.npz")],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode != 0 for p in procs), logs
    assert all("RCCL" in o and "one GPU per rank" in o for o in logs), logs


def test_connect_without_finalize_is_refused(tmp_path):
    """sf3d_dist_connect alone does not connect a multi-rank model: the ranks' common decision comes with sf3d_dist_finalize (a rank
    whose windows failed while the others' passed would otherwise leave the others spinning in the in-kernel exchange): the first
    call that needs the exchange fails with a message on every rank"""
    procs = [subprocess.Popen([sys.executable, str(ROOT / "scripts" / "multirank_worker.py"), str(r), "2", "29632", "c2f20_nofinalize", str(tmp_path / f"r{r}.npz")],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode != 0 for p in procs), logs
    assert all("sf3d_dist_finalize" in o for o in logs), logs


@pytest.mark.parametrize("world,case,port", [(2, "c2f20", 29641), (3, "projwin", 29642)])
def test_topology_edit_after_connect_is_a_topography_error_on_every_rank(tmp_path, world, case, port):
    """A setter that changes the graph (here setSurfaceProperties) after sf3d_dist_connect / finalize: the strip is frozen and its
    build arrays have been released (trimHostStaging), so the next device call must answer SF3D_TOPOGRAPHY_ERROR on every rank
    (round 4's advice: it read the released arrays and crashed, leaving the peers in their all-gather time-outs) and computeStep
    returns NaN instead of stepping a model the peers no longer agree on."""
    res = run_ranks(world, case + "_editafter", tmp_path, port)
    for r in res:
        assert int(r["set"]) == capi.OK
        assert int(r["balance"]) == capi.TOPOGRAPHY_ERROR, int(r["balance"])
        assert np.isnan(float(r["dt"]))


@pytest.mark.parametrize("world,case,port", [(2, "c2f20", 29681), (3, "c2f60", 29683), (2, "projwin", 29685)])
def test_rccl_sequencing_over_mock(tmp_path, world, case, port):
    """SF3D_EXCHANGE=rccl - the exchange in the form north_star sketches: halos as grouped ncclSend / ncclRecv of packed buffers
    (k_push_* packs, dist_unpack scatters), partial sums as an ncclAllGather, all queued by the host BETWEEN the kernels, separate
    decision kernels - had never executed a call: RCCL refuses two ranks on one device and the test box has one GPU.  Here the nine
    entry points are resolved from tests/librccl_mock.so (SF3D_RCCL_LIB; shared memory + stream-ordered copies, tests/rccl_mock.cpp)
    so that the product's side of that path - loader, communicator set-up from rank 0's id in the blob, pack / unpack kernels, the host
    sequencing of every exchange in a look-ahead batch, k_local_reduce + the rank-ordered combination, tear-down and re-connect after
    a re-initialisation - runs with the ranks sharing the GPU.  Held bit for bit against the window transport: H, Se, every accepted
    dt, every work counter (infiltration regime, runoff regime with Courant refusals and restore-best steps, a DEM outline cut
    mid-row).  What the mock cannot tell: anything about xGMI, RCCL's own launch costs, or its progress guarantees."""
    from criteria3d_amd import build
    mock = build.build_rccl_mock()
    win = run_ranks(world, case, tmp_path, port, env={"SF3D_PAIR_SWEEP": "0"})
    rccl = run_ranks(world, case, tmp_path, port + 1, env={"SF3D_EXCHANGE": "rccl", "SF3D_RCCL_LIB": str(mock), "SF3D_RCCL_SHARED_GPU": "1"})
    owner = win[0]["owner"]
    for r in range(world):
        assert int(win[r]["transport"]) == 1 and int(rccl[r]["transport"]) == 3, (r, win[r]["transport"], rccl[r]["transport"])
        mine = owner == r
        for k in win[r].files:
            if k.startswith(("H_h", "Se_h")):
                assert np.array_equal(win[r][k][mine], rccl[r][k][mine]), (r, k)
            elif k.startswith("dts_h") or k == "counters":
                assert np.array_equal(win[r][k], rccl[r][k]), (r, k)
            elif k.startswith(("storage_h", "total_water_h", "runoff_h", "drainage_h", "lateral_h")):
                np.testing.assert_allclose(win[r][k], rccl[r][k], rtol=1e-12, atol=1e-300)


@pytest.mark.parametrize("world,case,port", [(2, "c2f20", 29691), (3, "projwin", 29693), (3, "heat", 29695), (8, "c4f20h0", 29697)])
def test_strip_local_build_runs_to_the_bits_of_the_global_build(tmp_path, world, case, port):
    """STRIP-LOCAL BUILD on the device: every rank stages only its strip and the one-cell ring of columns around it
    (catchment.build(sparse=True): sf3d_dist_bounds + the model's links; global indices) and the run gives the bits of the run in which
    every rank staged the whole model - H, Se (T with heat) of every owned node, every accepted dt, every counter - on a regular grid, a
    DEM outline cut mid-row (masked paired sweep), coupled heat (whose two-colour decision now travels in the blobs) and C4 in eight
    strips, where the resident staging memory of a rank during the WHOLE run is a fraction of the global build's."""
    glob = run_ranks(world, case, tmp_path, port)
    loc = run_ranks(world, case, tmp_path, port + 1, env={"SF3D_TEST_SPARSE_BUILD": "1"})
    owner = glob[0]["owner"]
    for r in range(world):
        mine = owner == r
        assert np.array_equal(loc[r]["owner"][mine], owner[mine]) and int((loc[r]["owner"] == -1).sum()) > 0
        for k in glob[r].files:
            if k.startswith(("H_h", "Se_h", "T_h")):
                assert np.array_equal(loc[r][k][mine], glob[r][k][mine]), (r, k)
            elif k.startswith("dts_h") or k in ("counters", "sweep_launches"):
                assert np.array_equal(loc[r][k], glob[r][k]), (r, k)
            elif k.startswith(("storage_h", "total_water_h", "runoff_h", "drainage_h", "lateral_h")):
                np.testing.assert_allclose(loc[r][k], glob[r][k], rtol=1e-12, atol=1e-300)
    if case == "c4f20h0":
        print("build seconds per rank, global:", [round(float(x["build_seconds"]), 2) for x in glob], "strip-local:", [round(float(x["build_seconds"]), 2) for x in loc],
              "| peak resident set of a rank process [MB] (the caller's own model arrays included), global:", [int(x["maxrss_mb"]) for x in glob], "strip-local:", [int(x["maxrss_mb"]) for x in loc])
        # the library's global staging copy is ~330 B per node (1.7 GB at C4): a strip-local rank never touches seven eighths of it
        assert max(int(x["maxrss_mb"]) for x in loc) < min(int(x["maxrss_mb"]) for x in glob) - 1000
