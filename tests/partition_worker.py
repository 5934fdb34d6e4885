"""One gloo rank of the CPU test of the N>1 path: the product library's host-side partition plan
(row strips + halo lists, no GPU involved) drives a partitioned Jacobi iteration in numpy whose halo
exchange and partial-sum all-gather follow exactly the plan and protocol the device code uses
(owner computes its rows; after every sweep each rank sends the values on its send lists and
receives its recv lists; partial norms are combined in rank order).  Exit code 0 = the sharded
iteration reproduces the global one."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from criteria3d_amd import capi, catchment as cm  # noqa: E402


def jacobi_rows(rows, A, J, b, x):
    """x_new[i] = b[i] - sum_s A[s, i] * x[J[s, i]] for i in rows (slot order as the solver)."""
    xn = b[rows].copy()
    for s in (0, 2, 3, 4, 5, 6, 7, 8, 9, 1):
        xn -= A[s, rows] * x[J[s, rows]]
    return xn


def main():
    rank, world, port, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = cm.catchment_model(24, 40, 5) if case == "grid" else cm.ragged_model(9, 30, 4)
    sf = capi.load_product()
    cm.build(sf, m, finalize=False)                       # host staging only: no device call
    owner = sf.owner_map(world, m.n)
    send = [sf.halo_list(rank, world, p, 0) for p in range(world)]
    recv = [sf.halo_list(rank, world, p, 1) for p in range(world)]
    sf.lib.sf3d_clean()

    # the same random, diagonally dominant system on every rank (static ELL over the model's links)
    rng = np.random.RandomState(123)
    J = np.tile(np.arange(m.n), (10, 1))
    A = np.zeros((10, m.n))
    nlat = np.zeros(m.n, np.int64)
    for node, to, d in zip(m.link_node, m.link_to, m.link_dir):
        if d == capi.LINK_UP: s = 0
        elif d == capi.LINK_DOWN: s = 1
        else:
            s = 2 + nlat[node]; nlat[node] += 1
        J[s, node] = to
        A[s, node] = -rng.uniform(0.01, 0.09)
    b = rng.uniform(-1, 1, m.n)
    x0 = rng.uniform(-1, 1, m.n)

    # plan checks: strips cover every node once; my recv from p == p's send to me
    assert set(np.unique(owner)) == set(range(world))
    lists = [None] * world
    dist.all_gather_object(lists, {"send": [l.tolist() for l in send], "recv": [l.tolist() for l in recv]})
    for p in range(world):
        assert lists[p]["send"][rank] == recv[p].tolist(), "recv list != peer's send list"
        assert (owner[recv[p]] == p).all() and (owner[send[p]] == rank).all()
    mine = np.flatnonzero(owner == rank)
    needed = np.unique(J[:, mine][A[:, mine] != 0])
    halo = np.concatenate([r for r in recv]) if world > 1 else np.array([], np.int64)
    assert set(needed[owner[needed] != rank]) <= set(halo.tolist()), "a neighbour value is missing from the halo lists"

    # global iteration (reference) and sharded iteration
    xg = x0.copy()
    xs = x0.copy()
    xs[owner != rank] = np.nan                            # a rank never holds other strips' values ...
    for p in range(world):
        xs[recv[p]] = x0[recv[p]]                         # ... except its one-cell halo
    for it in range(12):
        new = jacobi_rows(np.arange(m.n), A, J, b, xg)
        gnorm_parts = [np.abs(new[owner == r] - xg[owner == r]).sum() for r in range(world)]
        xg = new
        mynew = jacobi_rows(mine, A, J, b, xs)
        part = np.abs(mynew - xs[mine]).sum()
        xs[mine] = mynew
        # halo exchange following the plan (device: puts into the peer's window)
        reqs, bufs = [], {}
        for p in range(world):
            if p == rank: continue
            if len(send[p]):
                reqs.append(dist.isend(torch.from_numpy(xs[send[p]].copy()), p))
            if len(recv[p]):
                bufs[p] = torch.empty(len(recv[p]), dtype=torch.float64)
                reqs.append(dist.irecv(bufs[p], p))
        for r in reqs: r.wait()
        for p, t in bufs.items(): xs[recv[p]] = t.numpy()
        # all-gather of partial sums, combined in rank order (device: mailboxes in the windows)
        parts = [None] * world
        dist.all_gather_object(parts, float(part))
        total, gtotal = 0.0, 0.0
        for r in range(world):
            total += parts[r]; gtotal += gnorm_parts[r]
        assert total == gtotal, (it, total, gtotal)
        assert np.array_equal(xs[mine], xg[mine]), f"sweep {it}: owned rows differ from the global iteration"
        assert np.array_equal(xs[halo], xg[halo]) if len(halo) else True
    # ---- the window protocol of the fused device path: every rank owns a window with, per source rank,
    # [2 epoch parities][3 fields][count] slots.  A sweep at epoch e puts the new values of its send lists into field 0,
    # parity e & 1, of the readers' windows; the all-gather that closes the sweep is the barrier and advances the epoch.
    # From the second sweep of an approximation on, a row reads its FOREIGN neighbours from its own window, parity
    # (e - 1) & 1, and never from x; the halo part of x is refreshed once, after the last sweep (k_post / k_halo_copy).
    # K puts (field 1) of a rank that is already one approximation ahead go to the same parity as the last iterate and
    # must not disturb it.
    FIELDS = 3
    window = {p: np.full((2, FIELDS, len(recv[p])), np.nan) for p in range(world)}
    pos = {}                                              # foreign node -> (source rank, position in its list)
    for p in range(world):
        for k, node in enumerate(recv[p]):
            pos[int(node)] = (p, k)
    foreign = np.array([[(int(J[s, i]) in pos) and A[s, i] != 0 for i in mine] for s in range(10)])
    xg = x0.copy(); xs = x0.copy(); xs[owner != rank] = np.nan
    for p in range(world):
        xs[recv[p]] = x0[recv[p]]
    epoch = 7                                             # any starting epoch
    def put(field, values_by_peer):
        par = epoch & 1
        reqs, bufs = [], {}
        for p in range(world):
            if p == rank: continue
            if len(send[p]): reqs.append(dist.isend(torch.from_numpy(values_by_peer(p).copy()), p))
            if len(recv[p]):
                bufs[p] = torch.empty(len(recv[p]), dtype=torch.float64); reqs.append(dist.irecv(bufs[p], p))
        for r in reqs: r.wait()
        for p, t in bufs.items(): window[p][par, field] = t.numpy()
    for approx in range(3):
        nsweeps = 4 + approx
        for it in range(nsweeps):
            xg = jacobi_rows(np.arange(m.n), A, J, b, xg)
            xin = xs.copy()
            if it > 0:
                xin[halo] = np.nan                        # the halo part of x is stale during the sweeps: it must not be read
            mynew = b[mine].copy()
            for s in (0, 2, 3, 4, 5, 6, 7, 8, 9, 1):
                xj = xin[J[s, mine]]
                if it > 0:
                    for c in np.flatnonzero(foreign[s]):
                        p, k = pos[int(J[s, mine[c]])]
                        xj[c] = window[p][(epoch - 1) & 1, 0, k]
                mynew -= A[s, mine] * xj
            xs[mine] = mynew
            put(0, lambda p: xs[send[p]])
            dist.barrier(); epoch += 1                    # the sweep's all-gather
            assert np.array_equal(xs[mine], xg[mine]), f"window protocol, approximation {approx} sweep {it}"
        par_last = (epoch - 1) & 1
        dist.barrier(); epoch += 1                        # k_post's all-gather (no puts)
        put(1, lambda p: -xs[send[p]])                    # a neighbour's next k_props may already put K (field 1) ...
        for p in range(world):
            if len(recv[p]): xs[recv[p]] = window[p][par_last, 0]   # ... while the halo of the final iterate is still being copied
        dist.barrier(); epoch += 1                        # k_props' all-gather
        dist.barrier(); epoch += 1                        # k_assemble's all-gather
        assert np.array_equal(xs[halo], xg[halo]) if len(halo) else True
    # ---- the PAIRED sweep on a strip (sf3d_pair.inc, DIST): two Jacobi iterations per pass over the coefficients.  The first iterate
    # of a halo node is its owner's to compute, so the pass is cut in two launches, each closed by an all-gather (= one epoch):
    #   k_sweep_pair<DIST>  x' of every owned node (foreign neighbours from the window, parity (e - 1) & 1, from the second pass of an
    #                       approximation on), put at parity e & 1; x'' - from x' of OWNED nodes only - for the nodes of 64-node chunks
    #                       without a foreign link; both norm partials all-gathered;
    #   k_sweep_bnd         x'' of the nodes in chunks WITH a foreign link (ChunkDesc::pad0), from x' in memory and the neighbours' x'
    #                       in the window at parity (e - 1) & 1 (what they put one epoch ago), put at parity e & 1, norm all-gathered
    #                       and added to the part k_sweep_pair left behind.
    # An approximation that takes an odd number of iterations ends with a single k_sweep<2>.  Held against the global iteration:
    # iterate AND both norms, bit for bit (norms: the same terms in rank order).
    chunk_of = mine // 64
    has_foreign = foreign.any(axis=0)
    bnd_chunks = np.unique(chunk_of[has_foreign])
    in_bnd = np.isin(chunk_of, bnd_chunks)               # k_sweep_bnd's nodes: every owned node of a flagged chunk
    window = {p: np.full((2, FIELDS, len(recv[p])), np.nan) for p in range(world)}
    xg = x0.copy(); xs = x0.copy(); xs[owner != rank] = np.nan
    for p in range(world):
        xs[recv[p]] = x0[recv[p]]
    def sweep_rows(sel, xin, from_window):
        """one Jacobi iteration of the owned nodes `sel` (indices into mine) from xin, foreign neighbours from the window when asked"""
        rows = mine[sel]
        out = b[rows].copy()
        for s in (0, 2, 3, 4, 5, 6, 7, 8, 9, 1):
            xj = xin[J[s, rows]]
            if from_window:
                for c in np.flatnonzero(foreign[s][sel]):
                    p, k = pos[int(J[s, rows[c]])]
                    xj[c] = window[p][(epoch - 1) & 1, 0, k]
            out -= A[s, rows] * xj
        return out
    def allsum(v):
        parts = [None] * world
        dist.all_gather_object(parts, float(v))
        t = 0.0
        for r in range(world): t += parts[r]
        return t
    def gnorm(new, old):
        t = 0.0
        for r in range(world): t += np.abs(new[owner == r] - old[owner == r]).sum()
        return t
    # the norms as the device adds them since round 4: every partial a double-double (hi = the terms' sum rounded, lo = what that
    # rounding lost - sf3d_physics.inc "the norm of a Jacobi sweep"), all-gathered with both words, rounded ONCE at the end: the same
    # double however the nodes are dealt to launches and ranks, i.e. the exactly rounded sum of all terms
    import math
    def dd(terms):
        t = [float(x) for x in terms]
        hi = math.fsum(t)
        return hi, math.fsum(t + [-hi])
    def allsum_dd(*pairs):
        parts = [None] * world
        dist.all_gather_object(parts, [w for p in pairs for w in p])
        return math.fsum(w for r in range(world) for w in parts[r])
    def gnorm_exact(new, old):
        return math.fsum(np.abs(new - old).tolist())
    everyone = np.arange(len(mine))
    for approx, niter in enumerate((6, 5, 1, 4)):
        it = 0
        while it < niter:
            stale = it > 0                                # the halo part of x has not been refreshed since the approximation began
            xin = xs.copy()
            if stale: xin[halo] = np.nan
            if niter - it >= 2:
                g1 = jacobi_rows(np.arange(m.n), A, J, b, xg); g2 = jacobi_rows(np.arange(m.n), A, J, b, g1)
                n1g, n2g = gnorm(g1, xg), gnorm(g2, g1)
                # launch 1: x' everywhere, x'' away from the neighbours
                x1 = xs.copy(); x1[halo] = np.nan        # x' of a halo node is NOT available inside this launch
                x1[mine] = sweep_rows(everyone, xin, stale)
                n1 = np.abs(x1[mine] - xs[mine]).sum()
                x2 = np.full(m.n, np.nan)
                inner = np.flatnonzero(~in_bnd)
                x2[mine[inner]] = sweep_rows(inner, x1, False)      # (NaN would surface if an inner node read a halo node)
                n2_inner = np.abs(x2[mine[inner]] - x1[mine[inner]]).sum()
                put(0, lambda p: x1[send[p]])
                n1t, n2t = allsum(n1), allsum(n2_inner)
                n1dd = allsum_dd(dd(np.abs(x1[mine] - xs[mine])))
                dd_inner = dd(np.abs(x2[mine[inner]] - x1[mine[inner]]))
                dist.barrier(); epoch += 1
                assert n1t == n1g and np.array_equal(x1[mine], g1[mine]), f"paired sweep, approximation {approx} iteration {it}: x'"
                assert n1dd == gnorm_exact(g1, xg), "first norm as double-doubles: the exactly rounded sum, whatever the partition"
                # launch 2: the nodes next to the neighbours, x' of the neighbours from the window
                edge = np.flatnonzero(in_bnd)
                x2[mine[edge]] = sweep_rows(edge, x1, True)
                n2_edge = np.abs(x2[mine[edge]] - x1[mine[edge]]).sum()
                xs[mine] = x2[mine]
                put(0, lambda p: xs[send[p]])
                n2t = n2t + allsum(n2_edge)
                n2dd = allsum_dd(dd_inner, dd(np.abs(x2[mine[edge]] - x1[mine[edge]])))      # (k_sweep_bnd adds the part the pass left in Ctrl::pairNorm2 / pairNorm2Lo)
                dist.barrier(); epoch += 1
                assert np.array_equal(xs[mine], g2[mine]), f"paired sweep, approximation {approx} iteration {it}: x''"
                assert abs(n2t - n2g) <= 4e-16 * n2g       # (plain doubles: the second norm is the same terms in another association - inner + edge)
                assert n2dd == gnorm_exact(g2, g1), "second norm as double-doubles: inner + edge parts of every rank round to the global sum"
                xg = g2; it += 2
            else:                                         # the odd iteration: a single sweep
                g1 = jacobi_rows(np.arange(m.n), A, J, b, xg)
                xs[mine] = sweep_rows(everyone, xin, stale)
                put(0, lambda p: xs[send[p]])
                dist.barrier(); epoch += 1
                assert np.array_equal(xs[mine], g1[mine]), f"single sweep after pairs, approximation {approx}"
                xg = g1; it += 1
        par_last = (epoch - 1) & 1
        dist.barrier(); epoch += 1                        # k_post's all-gather
        for p in range(world):
            if len(recv[p]): xs[recv[p]] = window[p][par_last, 0]   # k_halo_copy<1>: the halo of the final iterate, once per approximation
        assert np.array_equal(xs[halo], xg[halo]) if len(halo) else True
        dist.barrier(); epoch += 1                        # k_props' all-gather
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
