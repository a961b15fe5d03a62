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
    if world > 1:
        G, B = allreduce_GB(G, B, group, device)
    X = solve_fn(G, B)
    sum_position_diff = 0.0
    nscans = gr.getNrScans()
    for i in range(1, nscans):
        rPos, rPosTheta, dlen = _s.lum_pose_update(allScans[i], X[(i - 1) * 6:(i - 1) * 6 + 6])
        allScans[i].transformToEuler(rPos, rPosTheta, "LUM", 1 if i != nscans - 1 else 2)
        sum_position_diff += dlen
    return sum_position_diff / nscans


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:  # torch absent: single process
        return False
