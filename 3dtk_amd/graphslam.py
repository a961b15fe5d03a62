"""Graph-SLAM simultaneous matching (lum6DEuler, -G 1) with the link loop sharded over GPUs.

Reference: lum6DEuler::FillGB3D / doGraphSlam6D (src/slam6d/lum6Deuler.cc:265-477).  The
reference parallelises FillGB3D with `omp parallel for` over graph links and adds each link's
6x6 C and 6-vector CD into the global system under `omp critical`.  Here the independent unit
is the same -- one link = one whole-scan correspondence pass -- but links are dealt
round-robin to the ranks of one node (one process per MI355X, every scan and tree replicated
in each GPU's HBM), each rank accumulates its links into a local dense (G | B), and ONE
all-reduce (sum, fp64) per LUM iteration over RCCL/xGMI combines them: (6(n-1))^2 + 6(n-1)
doubles, 1.1 MB for 64 scans -- latency-bound, so it is a single flat buffer, not bucketed.
Every rank then solves the small SPD system redundantly and moves its own replicas.
"""
import numpy as np

from . import slam6d as _s


def shard_links(nlinks, rank, world):
    """Round-robin (links have equal cost when scans have equal size)."""
    return list(range(rank, nlinks, world))


def fill_GB(gr, allScans, max_dist_match2, link_fn, rank=0, world=1):
    """This rank's share of FillGB3D (lum6Deuler.cc:265-303) into dense G (6n x 6n), B (6n)."""
    n = gr.getNrScans() - 1
    G = np.zeros((6 * n, 6 * n))
    B = np.zeros(6 * n)
    for i in shard_links(gr.getNrLinks(), rank, world):
        fa, fb = gr.getLink(i, 0), gr.getLink(i, 1)
        a, b = fa - 1, fb - 1
        Cab, CDab = link_fn(allScans[fa], allScans[fb], max_dist_match2)[:2]
        if a >= 0:
            B[a * 6:a * 6 + 6] += CDab
            G[a * 6:a * 6 + 6, a * 6:a * 6 + 6] += Cab
        if b >= 0:
            B[b * 6:b * 6 + 6] -= CDab
            G[b * 6:b * 6 + 6, b * 6:b * 6 + 6] += Cab
        if a >= 0 and b >= 0:
            G[a * 6:a * 6 + 6, b * 6:b * 6 + 6] -= Cab
            G[b * 6:b * 6 + 6, a * 6:a * 6 + 6] -= Cab
    return G, B


def allreduce_GB(G, B, group=None, device=None):
    """One flat fp64 all-reduce of (G | B).  `device` = torch device for the NCCL(RCCL) backend,
    None for gloo/CPU."""
    import torch
    import torch.distributed as dist
    flat = torch.from_numpy(np.concatenate([G.reshape(-1), B]))
    if device is not None:
        flat = flat.to(device)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat = flat.cpu().numpy()
    n = len(B)
    return flat[:n * n].reshape(n, n).copy(), flat[n * n:].copy()


def lum_iteration(gr, allScans, max_dist_match2, group=None, link_fn=None, device=None,
                  solve_fn=None):
    """One iteration of lum6DEuler::doGraphSlam6D (lum6Deuler.cc:351-474).  Returns `ret`."""
    link_fn = link_fn or _s.covarianceEuler
    solve_fn = solve_fn or _s.solveSparseCholesky
    rank, world = 0, 1
    if group is not None or _dist_ready():
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    G, B = fill_GB(gr, allScans, max_dist_match2, link_fn, rank, world)
    if group is not None or _dist_ready():
        G, B = allreduce_GB(G, B, group, device)
    X = solve_fn(G, B)
    sum_position_diff = 0.0
    nscans = gr.getNrScans()
    for i in range(1, nscans):
        rPos, rPosTheta, dlen = _s.lum_pose_update(allScans[i], X[(i - 1) * 6:(i - 1) * 6 + 6])
        allScans[i].transformToEuler(rPos, rPosTheta, "LUM", 1 if i != nscans - 1 else 2)
        sum_position_diff += dlen
    return sum_position_diff / nscans


def lum_iteration_native(gr, allScans, max_dist_match2, group=None, device=None):
    """Same iteration with the host work in the library: all of this rank's links in ONE batched
    call (tdtk_lum_links: kernels enqueued back to back, one sync), the all-reduce, the SPD solve,
    and the native pose update (tdtk_lum_update_poses) that also moves the resident scans."""
    import ctypes as C
    from ._capi import lib, check, dptr
    rank, world = 0, 1
    if group is not None or _dist_ready():
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    nscans = gr.getNrScans()
    n = nscans - 1
    mine = shard_links(gr.getNrLinks(), rank, world)
    nl = len(mine)
    G = np.zeros((6 * n, 6 * n))
    B = np.zeros(6 * n)
    if nl:
        first = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 1)].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([allScans[gr.getLink(i, 0)].dalignxf for i in mine]))
        Cm = np.empty((nl, 36)); CD = np.empty((nl, 6))
        m = (C.c_uint64 * nl)(); ss = np.empty(nl)
        check(lib().tdtk_lum_links(nl, first, dptr(dal), second, float(max_dist_match2), dptr(Cm), dptr(CD),
                                   m, dptr(ss)))
        for k, i in enumerate(mine):
            a, b = gr.getLink(i, 0) - 1, gr.getLink(i, 1) - 1
            Cab = Cm[k].reshape(6, 6)
            if a >= 0:
                B[a * 6:a * 6 + 6] += CD[k]
                G[a * 6:a * 6 + 6, a * 6:a * 6 + 6] += Cab
            if b >= 0:
                B[b * 6:b * 6 + 6] -= CD[k]
                G[b * 6:b * 6 + 6, b * 6:b * 6 + 6] += Cab
            if a >= 0 and b >= 0:
                G[a * 6:a * 6 + 6, b * 6:b * 6 + 6] -= Cab
                G[b * 6:b * 6 + 6, a * 6:a * 6 + 6] -= Cab
    if group is not None or _dist_ready():
        G, B = allreduce_GB(G, B, group, device)
    X = _s.solveSparseCholesky(G, B)
    tm = np.ascontiguousarray(np.stack([s.transMat for s in allScans[:nscans]]))
    da = np.ascontiguousarray(np.stack([s.dalignxf for s in allScans[:nscans]]))
    rp = np.ascontiguousarray(np.stack([s.rPos for s in allScans[:nscans]]))
    rt = np.ascontiguousarray(np.stack([s.rPosTheta for s in allScans[:nscans]]))
    hs = (C.c_void_p * nscans)(*[s._h for s in allScans[:nscans]])
    xf = np.zeros((nscans, 32))
    ret = C.c_double(0.0)
    check(lib().tdtk_lum_update_poses(nscans, dptr(X), dptr(tm), dptr(da), dptr(rp), dptr(rt), hs, dptr(xf),
                                      C.byref(ret)))
    for i in range(1, nscans):
        s = allScans[i]
        s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i].copy(), da[i].copy(), rp[i].copy(), rt[i].copy()
        if s._h is None:                       # not resident on this rank: replay later, in order
            s._queue.append(xf[i, :16].copy())
            s._queue.append(xf[i, 16:].copy())
        s.frames.append((s.transMat.copy(), "LUM"))
    return ret.value


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:  # torch absent: single process
        return False
