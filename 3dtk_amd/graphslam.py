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
import math
import os

import numpy as np

from . import slam6d as _s


def shard_links(gr, rank, world):
    """Which links of `gr` this rank evaluates.  Links cost the same when scans have equal size, so
    the odometry chain (links 0..n-2, always present) is dealt round-robin; loop-closure links are
    keyed by their end points, (from + to) % world, so that a closure appearing or disappearing
    between LUM rounds (the Graph is rebuilt from the current poses every round,
    src/slam6d/slam6D.cc:501-532) does not reshuffle every other link -- and with it every
    resident tree and scan -- across the ranks."""
    chain = gr.getNrScans() - 1
    mine = []
    for i in range(gr.getNrLinks()):
        f, t = gr.getLink(i, 0), gr.getLink(i, 1)
        owner = (i % world) if (i < chain and t == f + 1) else ((f + t) % world)
        if owner == rank:
            mine.append(i)
    return mine


def fill_GB(gr, allScans, max_dist_match2, link_fn, rank=0, world=1):
    """This rank's share of FillGB3D (lum6Deuler.cc:265-303) into dense G (6n x 6n), B (6n)."""
    n = gr.getNrScans() - 1
    G = np.zeros((6 * n, 6 * n))
    B = np.zeros(6 * n)
    for i in shard_links(gr, rank, world):
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


def lum_reduce_solve(gr, mine, Cm, CD, world=1, group=None, device=None):
    """The exchange step of the sharded FillGB3D: this rank's per-link blocks (Cm [len(mine), 36],
    CD [len(mine), 6]) go into a zero [nlinks, 42] buffer, ONE all-reduce makes every link's block
    known everywhere (each link has exactly one owner, so the sum is exact), then every rank runs the
    same scatter-in-link-order + SPD solve (tdtk_lum_assemble_solve).  Returns X [6(n-1)]."""
    from ._capi import lib, check, dptr, iptr
    nscans, nlinks = gr.getNrScans(), gr.getNrLinks()
    blocks = np.zeros((nlinks, 42))                 # [C 36 | CD 6] per link
    if len(mine):
        blocks[mine, :36] = Cm
        blocks[mine, 36:] = CD
    if world > 1 or (os.environ.get("TDTK_FORCE_ALLREDUCE") and _dist_ready()):   # the env knob: 1-rank RCCL smoke test
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(blocks)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        blocks = t.cpu().numpy()
    frm = np.ascontiguousarray([gr.getLink(i, 0) for i in range(nlinks)], dtype=np.int32)
    to = np.ascontiguousarray([gr.getLink(i, 1) for i in range(nlinks)], dtype=np.int32)
    Call = np.ascontiguousarray(blocks[:, :36])
    CDall = np.ascontiguousarray(blocks[:, 36:])
    X = np.empty(6 * (nscans - 1))
    check(lib().tdtk_lum_assemble_solve(nlinks, iptr(frm), iptr(to), dptr(Call), dptr(CDall), nscans, dptr(X),
                                        None, None))
    return X


def lum_iteration_native(gr, allScans, max_dist_match2, group=None, device=None):
    """Same iteration with the host work in the library: all of this rank's links in ONE batched
    call (tdtk_lum_links: kernels enqueued back to back, one sync); ONE all-reduce of the per-link
    blocks (42 doubles per link, zeros for links other ranks own, so the sum is exact and the result
    does not depend on the number of ranks); scatter into G / B in link order + SPD solve
    (tdtk_lum_assemble_solve); native pose update (tdtk_lum_update_poses) that also moves the
    resident scans."""
    import ctypes as C
    from ._capi import lib, check, dptr
    rank, world = 0, 1
    if group is not None or _dist_ready():
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    nscans = gr.getNrScans()
    n = nscans - 1
    nlinks = gr.getNrLinks()
    mine = shard_links(gr, rank, world)
    nl = len(mine)
    Cm = np.empty((nl, 36)); CD = np.empty((nl, 6))
    if nl:
        first = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 1)].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([allScans[gr.getLink(i, 0)].dalignxf for i in mine]))
        m = (C.c_uint64 * nl)(); ss = np.empty(nl)
        check(lib().tdtk_lum_links(nl, first, dptr(dal), second, float(max_dist_match2), dptr(Cm), dptr(CD),
                                   m, dptr(ss)))
    X = lum_reduce_solve(gr, mine, Cm, CD, world, group, device)
    tm = np.ascontiguousarray(np.stack([s.transMat for s in allScans[:nscans]]))
    da = np.ascontiguousarray(np.stack([s.dalignxf for s in allScans[:nscans]]))
    rp = np.ascontiguousarray(np.stack([s.rPos for s in allScans[:nscans]]))
    rt = np.ascontiguousarray(np.stack([s.rPosTheta for s in allScans[:nscans]]))
    hs = (C.c_void_p * nscans)(*[s._h for s in allScans[:nscans]])
    xf = np.zeros((nscans, 32))
    ret = C.c_double(0.0)
    check(lib().tdtk_lum_update_poses(nscans, dptr(X), dptr(tm), dptr(da), dptr(rp), dptr(rt), hs, dptr(xf),
                                      C.byref(ret)))
    # tm / da / rp / rt / xf are fresh arrays of this call and nothing below writes into them again (every
    # later pose update makes new arrays), so the scans keep row views instead of copies
    for i in range(1, nscans):
        s = allScans[i]
        s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i], da[i], rp[i], rt[i]
        if s._h is None:                       # not resident on this rank: replay later, in order
            s._queue.append(xf[i, :16])
            s._queue.append(xf[i, 16:])
        s.frames.append((tm[i], "LUM"))
    return ret.value


def _rank_world(group):
    if group is not None or _dist_ready():
        import torch.distributed as dist
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _link_sums(gr, allScans, mine, max_dist_match2, want):
    """whole-scan pair sums of this rank's links, one batched call"""
    import ctypes as C
    from ._capi import lib, check, dptr, PairSums
    nl = len(mine)
    sums = (PairSums * max(nl, 1))()
    if nl:
        first = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 1)].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([allScans[gr.getLink(i, 0)].dalignxf for i in mine]))
        check(lib().tdtk_links_pair_sums(nl, first, dptr(dal), second, float(max_dist_match2), int(want), sums))
    return sums


def _exchange_blocks(blocks, world, group, device):
    """one all-reduce of the per-link blocks; each link has one owner, the others hold zeros"""
    if world <= 1:
        return blocks
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(blocks)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


def lumquat_iteration(gr, allScans, max_dist_match2, group=None, device=None):
    """One iteration of lum6DQuat::doGraphSlam6D (src/slam6d/lum6Dquat.cc:319-481).  Per link the device
    delivers the LUM sums (u = (p1+p2)/2, delta = p1-p2) of covarianceQuat (:143-171); MM (7x7), MZ, D and
    ss = (sum|delta|^2 - D.MZ) / (2m-3) (the residual of :196-207 for MM D = MZ) follow per link; blocks
    are exchanged once; FillGB3D (:248-276) ASSIGNS the off-diagonal blocks."""
    from . import slam6d as _sl
    from ._capi import WANT_LUM
    rank, world = _rank_world(group)
    nscans, nlinks = gr.getNrScans(), gr.getNrLinks()
    n = nscans - 1
    mine = shard_links(gr, rank, world)
    sums = _link_sums(gr, allScans, mine, max_dist_match2, WANT_LUM)
    blocks = np.zeros((nlinks, 56))                       # [C 49 | CD 7]
    for k, i in enumerate(mine):
        s = sums[k]
        m = int(s.n)
        if m <= 2:
            continue                                      # "Error calculating covariance matrix": zeros
        L = s.lum
        sx, sy, sz, xpy, xpz, ypz, xy, xz, yz = [L[j] for j in range(9)]
        MZ = np.array([L[9], L[10], L[11], s.lum_udot, -L[12], -L[14], -L[13]])
        MM = np.zeros((7, 7))
        MM[0, 0] = MM[1, 1] = MM[2, 2] = m
        MM[3, 3] = (xpy + xpz + ypz) / 2.0; MM[4, 4] = ypz; MM[5, 5] = xpz; MM[6, 6] = xpy
        MM[0, 3] = MM[3, 0] = sx; MM[0, 5] = MM[5, 0] = -sz; MM[0, 6] = MM[6, 0] = sy
        MM[1, 3] = MM[3, 1] = sy; MM[1, 4] = MM[4, 1] = sz;  MM[1, 6] = MM[6, 1] = -sx
        MM[2, 3] = MM[3, 2] = sz; MM[2, 4] = MM[4, 2] = -sy; MM[2, 5] = MM[5, 2] = sx
        MM[4, 5] = MM[5, 4] = -xy; MM[4, 6] = MM[6, 4] = -xz; MM[5, 6] = MM[6, 5] = -yz
        D = np.linalg.solve(MM, MZ)
        ss = (s.sum - float(D @ MZ)) / (2 * m - 3)
        blocks[i, :49] = (MM / ss).reshape(49)
        blocks[i, 49:] = MZ / ss
    blocks = _exchange_blocks(blocks, world, group, device)
    G = np.zeros((7 * n, 7 * n)); B = np.zeros(7 * n)
    for i in range(nlinks):
        a, b = gr.getLink(i, 0) - 1, gr.getLink(i, 1) - 1
        Cab, CDab = blocks[i, :49].reshape(7, 7), blocks[i, 49:]
        if a >= 0:
            B[a * 7:a * 7 + 7] += CDab; G[a * 7:a * 7 + 7, a * 7:a * 7 + 7] += Cab
        if b >= 0:
            B[b * 7:b * 7 + 7] -= CDab; G[b * 7:b * 7 + 7, b * 7:b * 7 + 7] += Cab
        if a >= 0 and b >= 0:
            G[a * 7:a * 7 + 7, b * 7:b * 7 + 7] = -Cab; G[b * 7:b * 7 + 7, a * 7:a * 7 + 7] = -Cab
    X = _sl.solveSparseCholesky(G, B)
    A1, A2 = [], []
    tot = 0.0
    for i in range(1, nscans):
        sc = allScans[i]
        xa, ya, za = sc.get_rPos()
        p, q, r, s_ = sc.get_rPosQuat()
        px, py, pz, qx, qy, qz = p * xa, p * ya, p * za, q * xa, q * ya, q * za
        rx, ry, rz, sx, sy, sz = r * xa, r * ya, r * za, s_ * xa, s_ * ya, s_ * za
        Ha = np.eye(7)
        Ha[3, 3] = 2 * p; Ha[4, 3] = 2 * q; Ha[5, 3] = 2 * r; Ha[6, 3] = 2 * s_
        Ha[3, 4] = 2 * q; Ha[4, 4] = -2 * p; Ha[5, 4] = -2 * s_; Ha[6, 4] = 2 * r
        Ha[3, 5] = 2 * r; Ha[4, 5] = 2 * s_; Ha[5, 5] = -2 * p; Ha[6, 5] = -2 * q
        Ha[3, 6] = 2 * s_; Ha[4, 6] = -2 * r; Ha[5, 6] = 2 * q; Ha[6, 6] = -2 * p
        Ha[0, 3] = -2 * (px + sy - rz); Ha[1, 3] = -2 * (-sx + py + qz); Ha[2, 3] = -2 * (rx - qy + pz)
        Ha[0, 4] = -2 * (qx + ry + sz); Ha[1, 4] = -2 * (-rx + qy - pz); Ha[2, 4] = -2 * (-sx + py + qz)
        Ha[0, 5] = -2 * (rx - qy + pz); Ha[1, 5] = -2 * (qx + ry + sz);  Ha[2, 5] = -2 * (-px - sy + rz)
        Ha[0, 6] = -2 * (sx - py - qz); Ha[1, 6] = -2 * (px + sy - rz);  Ha[2, 6] = -2 * (qx + ry + sz)
        Ainv = np.empty((7, 7))
        from ._capi import lib, check, dptr
        check(lib().tdtk_invert(dptr(np.ascontiguousarray(Ha)), 7, dptr(Ainv)))
        result = Ainv @ X[(i - 1) * 7:(i - 1) * 7 + 7]
        rPos = sc.get_rPos() - result[:3]
        quat = np.array([p, q, r, s_]) - result[3:]
        quat = quat / math.sqrt(float(quat @ quat))          # Normalize4
        A1.append(_sl.M4inv(sc.transMat))                     # transformToQuat (scan.cc:1093-1104)
        A2.append(_sl.QuatToMatrix4(quat, rPos))
        tot += math.sqrt(result[0] ** 2 + result[1] ** 2 + result[2] ** 2)
    _sl.transform_many(allScans[1:nscans], A1, A2, "LUM")
    return tot / nscans


def helix_compute_rt(ccs):
    """icp6D_HELIX::computeRt (src/slam6d/icp6Dhelix.cc:144-206) for one scan's 6 unknowns."""
    c, cs = -ccs[:3], -ccs[3:]
    CLength = math.sqrt(float(c @ c))
    rotationCheck = float(c @ cs)
    angle = math.atan(CLength)
    g = c / CLength
    sinA = math.sin(-angle / 2)
    b0, b1, b2, b3 = math.cos(-angle / 2), g[0] * sinA, g[1] * sinA, g[2] * sinA
    R = np.array([[b0 * b0 + b1 * b1 - b2 * b2 - b3 * b3, 2 * (b1 * b2 + b0 * b3), 2 * (b1 * b3 - b0 * b2)],
                  [2 * (b1 * b2 - b0 * b3), b0 * b0 - b1 * b1 + b2 * b2 - b3 * b3, 2 * (b2 * b3 + b0 * b1)],
                  [2 * (b1 * b3 + b0 * b2), 2 * (b2 * b3 - b0 * b1), b0 * b0 - b1 * b1 - b2 * b2 + b3 * b3]])
    R = R / (b0 * b0 + b1 * b1 + b2 * b2 + b3 * b3)
    skew = rotationCheck / (CLength * CLength)
    gs = (cs - c * skew) / CLength
    pT = np.cross(g, gs)
    t = R @ -pT + g * (skew * angle) + pT
    a = np.zeros(16)
    for r_ in range(3):
        for c_ in range(3):
            a[c_ * 4 + r_] = R[r_, c_]
    a[12:15] = t
    a[15] = 1.0
    return a


def ghelix_iteration(gr, allScans, max_dist_match2, state, group=None, device=None):
    """One iteration of ghelix6DQ2::doGraphSlam6D (src/slam6d/ghelix6DQ2.cc:330-449).  Per link the sums
    of genBBdForLinkedPair (:88-150) are raw first / second moments of p2 and the antisymmetric part of
    sum p1 p2^T; they come from the second-moment block of the device pass.  state = (B, bd), which the
    reference zeroes once per call and keeps adding to over the iterations of that call."""
    from . import slam6d as _sl
    from ._capi import WANT_MOM2
    rank, world = _rank_world(group)
    nscans, nlinks = gr.getNrScans(), gr.getNrLinks()
    mine = shard_links(gr, rank, world)
    sums = _link_sums(gr, allScans, mine, max_dist_match2, WANT_MOM2)
    blocks = np.zeros((nlinks, 43))                       # [flag | Blk 36 | bd1 6]
    for k, i in enumerate(mine):
        s = sums[k]
        m = int(s.n)
        if m <= 1:
            continue                                      # "Error: Link ... is empty" (:395-401)
        cm, cd = np.array(s.centroid_m), np.array(s.centroid_d)
        Si = np.array(s.Si).reshape(3, 3)
        dd = s.mom_dd
        DD = np.array([[dd[0], dd[1], dd[2]], [dd[1], dd[3], dd[4]], [dd[2], dd[4], dd[5]]]) + m * np.outer(cd, cd)
        X = Si + m * np.outer(cm, cd)                     # sum p1 p2^T
        s2 = m * cd                                       # sum p2
        Blk = np.zeros((6, 6))
        Blk[3, 3] = Blk[4, 4] = Blk[5, 5] = m
        Blk[0, 4] = Blk[4, 0] = -s2[2]; Blk[1, 3] = Blk[3, 1] = s2[2]
        Blk[0, 5] = Blk[5, 0] = s2[1];  Blk[2, 3] = Blk[3, 2] = -s2[1]
        Blk[2, 4] = Blk[4, 2] = s2[0];  Blk[1, 5] = Blk[5, 1] = -s2[0]
        Blk[0, 1] = Blk[1, 0] = -DD[0, 1]; Blk[0, 2] = Blk[2, 0] = -DD[0, 2]; Blk[1, 2] = Blk[2, 1] = -DD[1, 2]
        Blk[0, 0] = DD[2, 2] + DD[1, 1]; Blk[1, 1] = DD[2, 2] + DD[0, 0]; Blk[2, 2] = DD[0, 0] + DD[1, 1]
        sd = m * (cm - cd)                                # sum (p1 - p2)
        bd1 = np.array([X[2, 1] - X[1, 2], X[0, 2] - X[2, 0], X[1, 0] - X[0, 1], sd[0], sd[1], sd[2]])
        blocks[i, 0] = 1.0
        blocks[i, 1:37] = Blk.reshape(36)
        blocks[i, 37:] = bd1
    blocks = _exchange_blocks(blocks, world, group, device)
    B, bd = state
    for i in range(nlinks):
        if blocks[i, 0] == 0.0:
            continue
        fa, fb = gr.getLink(i, 0), gr.getLink(i, 1)
        if fb == 0:
            raise ValueError("ghelix6DQ2: a link must not end at the fixed scan 0")
        Blk, bd1 = blocks[i, 1:37].reshape(6, 6), blocks[i, 37:]
        a, b = (fa - 1) * 6, (fb - 1) * 6
        if fa != 0:
            B[a:a + 6, a:a + 6] += Blk; bd[a:a + 6] += bd1
        B[b:b + 6, b:b + 6] += Blk; bd[b:b + 6] -= bd1      # bd2 == -bd1 term by term (:126-138)
        if fa != 0:
            B[a:a + 6, b:b + 6] -= Blk; B[b:b + 6, a:a + 6] -= Blk
    ccs = _sl.solveSparseCholesky(B, bd)
    A1 = []
    tot = 0.0
    for i in range(1, nscans):
        axf = helix_compute_rt(ccs[(i - 1) * 6:(i - 1) * 6 + 6])
        A1.append(axf)
        tot += math.sqrt(axf[12] ** 2 + axf[13] ** 2 + axf[14] ** 2)
    _sl.transform_many(allScans[1:nscans], A1, None, "LUM")
    return tot / nscans


def compute_rt(x, dx):
    """icp6D_APX::computeRt (src/slam6d/icp6Dapx.cc:310-335): small-angle rotation from the three
    solved sines + translation dx, column-major."""
    import math
    sx, sy, sz = x[0], x[1], x[2]
    cx, cy, cz = math.sqrt(1.0 - sx * sx), math.sqrt(1.0 - sy * sy), math.sqrt(1.0 - sz * sz)
    a = np.zeros(16)
    a[0] = cy * cz; a[1] = sx * sy * cz + cx * sz; a[2] = -cx * sy * cz + sx * sz
    a[4] = -cy * sz; a[5] = -sx * sy * sz + cx * cz; a[6] = cx * sy * sz + sx * cz
    a[8] = sy; a[9] = -sx * cy; a[10] = cx * cy
    a[12], a[13], a[14] = dx[0], dx[1], dx[2]
    a[15] = 1
    return a


def gapx_iteration(gr, allScans, max_dist_match2, T, group=None, device=None):
    """One iteration of gapx6D::doGraphSlam6D (src/slam6d/gapx6D.cc:323-542), links sharded like
    FillGB3D: each rank evaluates its links' genBArotForLinkedPair blocks on the GPU (one batched
    call), a single all-reduce combines the dense rotation system (B | A) together with the per-link
    centroids the translation step needs, then every rank solves redundantly.  T is the reference's
    translation vector, which keeps accumulating over the iterations of one call."""
    import ctypes as C
    from ._capi import lib, check, dptr, PairSums, WANT_GAPX
    rank, world = 0, 1
    if group is not None or _dist_ready():
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    nscans = gr.getNrScans()
    n = nscans - 1
    nlinks = gr.getNrLinks()
    mine = shard_links(gr, rank, world)
    nl = len(mine)
    B = np.zeros((3 * n, 3 * n)); A = np.zeros(3 * n)
    cent = np.zeros((nlinks, 7))            # cm[3], cd[3], non-empty flag
    if nl:
        first = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 1)].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([allScans[gr.getLink(i, 0)].dalignxf for i in mine]))
        sums = (PairSums * nl)()
        check(lib().tdtk_links_pair_sums(nl, first, dptr(dal), second, float(max_dist_match2), WANT_GAPX, sums))
        for k, i in enumerate(mine):
            s = sums[k]
            f, sx = gr.getLink(i, 0), gr.getLink(i, 1)
            cent[i, :3] = s.centroid_m; cent[i, 3:6] = s.centroid_d
            if s.n <= 1:                      # "Error: Link ... is empty" (gapx6D.cc:424-431)
                continue
            cent[i, 6] = 1.0
            a, b = f - 1, sx - 1
            if f != 0:
                A[a * 3:a * 3 + 3] += np.array(s.gapx_Ak1)
                B[a * 3:a * 3 + 3, a * 3:a * 3 + 3] += np.array(s.gapx_MkMkt).reshape(3, 3)
                B[a * 3:a * 3 + 3, b * 3:b * 3 + 3] += np.array(s.gapx_DkMkt).reshape(3, 3)
                B[b * 3:b * 3 + 3, a * 3:a * 3 + 3] += np.array(s.gapx_MkDkt).reshape(3, 3)
            A[b * 3:b * 3 + 3] += np.array(s.gapx_Ak2)
            B[b * 3:b * 3 + 3, b * 3:b * 3 + 3] += np.array(s.gapx_DkDkt).reshape(3, 3)
    if group is not None or _dist_ready():
        import torch
        import torch.distributed as dist
        flat = torch.from_numpy(np.concatenate([B.reshape(-1), A, cent.reshape(-1)]))
        if device is not None:
            flat = flat.to(device)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat = flat.cpu().numpy()
        m = 3 * n
        B = flat[:m * m].reshape(m, m).copy(); A = flat[m * m:m * m + m].copy()
        cent = flat[m * m + m:].reshape(nlinks, 7).copy()
    sum_position_diff = float(cent[:, 6].sum())          # genBArotForLinkedPair returns 1.0 per link (sic)
    X = np.empty(3 * n)
    check(lib().tdtk_solve_chol_upper(dptr(np.ascontiguousarray(B)), dptr(A), 3 * n, dptr(X)))
    # translation system (genBAtransForLinkedPair, gapx6D.cc:76-137)
    Bt = np.zeros((n, n)); At = np.zeros(3 * n)

    def rot_apply(x, p):
        a = compute_rt(x, (0.0, 0.0, 0.0))
        xn = p[0] * a[0] + p[1] * a[4] + p[2] * a[8]
        yn = p[0] * a[1] + p[1] * a[5] + p[2] * a[9]
        zn = p[0] * a[2] + p[1] * a[6] + p[2] * a[10]
        return np.array([xn + a[12], yn + a[13], zn + a[14]])           # Point::transform
    for i in range(nlinks):
        f, sx = gr.getLink(i, 0), gr.getLink(i, 1)
        x = X[(f - 1) * 3:(f - 1) * 3 + 3] if f != 0 else np.zeros(3)
        Ak1 = rot_apply(x, cent[i, :3]) - rot_apply(X[(sx - 1) * 3:(sx - 1) * 3 + 3], cent[i, 3:6])
        if f != 0:
            At[(f - 1) * 3:(f - 1) * 3 + 3] -= Ak1
            Bt[f - 1, f - 1] += 1
            Bt[f - 1, sx - 1] -= 1; Bt[sx - 1, f - 1] -= 1
        At[(sx - 1) * 3:(sx - 1) * 3 + 3] += Ak1
        Bt[sx - 1, sx - 1] += 1
    Bti = np.empty((n, n))
    check(lib().tdtk_invert(dptr(np.ascontiguousarray(Bt)), n, dptr(Bti)))
    T += (Bti @ At.reshape(n, 3)).reshape(-1)
    for i in range(1, nscans):
        dx = T[(i - 1) * 3:(i - 1) * 3 + 3]
        allScans[i].transform(compute_rt(X[(i - 1) * 3:(i - 1) * 3 + 3], dx), "LUM", 1 if i < nscans - 1 else 2)
        sum_position_diff += float(np.sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]))
    return sum_position_diff / nscans


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:  # torch absent: single process
        return False
