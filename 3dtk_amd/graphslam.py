"""Graph-SLAM simultaneous matching (lum6DEuler, -G 1) with the link loop sharded over GPUs.

Reference: lum6DEuler::FillGB3D / doGraphSlam6D (src/slam6d/lum6Deuler.cc:265-477).  The
reference parallelises FillGB3D with `omp parallel for` over graph links and adds each link's
6x6 C and 6-vector CD into the global system under `omp critical`.  Here the independent unit
is the same -- one link = one whole-scan correspondence pass -- but links are dealt to the ranks of
one node (one process per MI355X; a rank keeps only the trees / scans its links touch), each rank
computes the blocks of its links, and ONE all-reduce (sum, fp64) per iteration over RCCL/xGMI makes
every link's block known everywhere: 42 doubles per link, 28 KB for the 84 links of 64 scans --
latency-bound, one flat buffer.  Every rank then solves the small SPD system redundantly and moves
its own replicas.  graph_iteration_comm runs all of that inside the library (tdtk_graph_iteration with
the library's own RCCL communicator); the torch.distributed variants below serve the gloo test rig.
"""
import math
import os

import numpy as np

from . import slam6d as _s


def _link_arrays(gr):
    """(from, to) of every link as int32 arrays -- straight from the Graph's lists when it is our Graph"""
    if hasattr(gr, "frm") and hasattr(gr, "to"):
        a = getattr(gr, "_arrays", None)          # (kept by Graph when the library listed the links; trusted only while the
        if a is not None and a[0].tolist() == gr.frm and a[1].tolist() == gr.to:      #  lists still say the same: they are public)
            return a
        return np.array(gr.frm, dtype=np.int32), np.array(gr.to, dtype=np.int32)
    nl = gr.getNrLinks()
    return (np.ascontiguousarray([gr.getLink(i, 0) for i in range(nl)], dtype=np.int32),
            np.ascontiguousarray([gr.getLink(i, 1) for i in range(nl)], dtype=np.int32))


def link_owners(gr, world, scans=None):
    """owner[i] = the rank that evaluates link i (tdtk_graph_deal_links).  A link costs one whole-scan pass over its
    second scan.  Scans of equal size: the odometry chain (links 0..n-2, always present) is dealt round-robin and
    loop closures are keyed by their end points, (from + to) % world, so that a closure appearing or disappearing
    between LUM rounds (the Graph is rebuilt from the current poses every round, src/slam6d/slam6D.cc:501-532) does
    not reshuffle every other link -- and with it every resident tree and scan -- across the ranks.  Scans of
    different size (real data): longest-processing-time-first by the point count of the second scan."""
    import ctypes as C
    from ._capi import lib, check, iptr
    nl, ns = gr.getNrLinks(), gr.getNrScans()
    frm, to = _link_arrays(gr)
    owner = np.zeros(nl, np.int32)
    pts = None
    if scans is not None:
        pts = np.ascontiguousarray([s.n for s in scans[:ns]], dtype=np.uint64)
    check(lib().tdtk_graph_deal_links(nl, iptr(frm), iptr(to), pts.ctypes.data_as(C.POINTER(C.c_uint64)) if pts is not None else None,
                                      ns, int(world), iptr(owner)))
    return owner


def shard_links(gr, rank, world, scans=None):
    """Which links of `gr` this rank evaluates (see link_owners)."""
    return [int(i) for i in np.flatnonzero(link_owners(gr, world, scans) == rank)]


class NativeComm:
    """The library's own RCCL communicator (tdtk_comm_*): the all-reduce of the link blocks runs inside
    tdtk_graph_iteration, C++ end to end.  The 128-byte unique id travels from rank 0 to the others through
    `bcast` (a callable bytes -> bytes; torch.distributed in bench.py, anything else elsewhere)."""

    def __init__(self, rank, world, device, bcast=None):
        import ctypes as C
        from ._capi import lib, check
        buf = C.create_string_buffer(128)
        if rank == 0:
            check(lib().tdtk_comm_unique_id(buf))
        ident = bytes(buf.raw)
        if world > 1:
            if bcast is None:
                raise ValueError("a multi-rank communicator needs a way to hand out the unique id")
            ident = bcast(ident)
        h = C.c_void_p()
        check(lib().tdtk_comm_create(ident, int(rank), int(world), int(device), C.byref(h)))
        self._h, self.rank, self.world = h, rank, world
        # what RCCL itself reports (ncclCommCount); tdtk_comm_create has already refused a mismatch
        self.rccl_world = int(lib().tdtk_comm_rccl_world(h))
        if self.rccl_world != world:
            raise RuntimeError("RCCL communicator has %d ranks, expected %d" % (self.rccl_world, world))

    def n_allreduce(self):
        import ctypes as C
        from ._capi import lib
        n = C.c_uint64(0)
        lib().tdtk_comm_info(self._h, None, None, C.byref(n))
        return int(n.value)

    def close(self):
        from ._capi import lib
        if self._h:
            lib().tdtk_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def torch_id_bcast(device=None):
    """unique-id hand-out over an initialised torch.distributed group (rank 0 is the source)"""
    def bcast(ident):
        import torch
        import torch.distributed as dist
        t = torch.tensor(list(ident), dtype=torch.uint8, device=device)
        dist.broadcast(t, 0)
        return bytes(t.cpu().tolist())
    return bcast


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
        s._addFrames("LUM", 2 if i == nscans - 1 else 1)     # lum6Deuler.cc:451-455: the last scan with islum == 2
    return ret.value


def _rank_world(group):
    if group is not None or _dist_ready():
        import torch.distributed as dist
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _exchange_blocks(blocks, world, group, device):
    """one all-reduce of the per-link blocks; each link has one owner, the others hold zeros"""
    if world <= 1 and not (os.environ.get("TDTK_FORCE_ALLREDUCE") and _dist_ready()):
        return blocks
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(blocks)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.cpu().numpy()


GRAPH_LUMEULER, GRAPH_LUMQUAT, GRAPH_GHELIX, GRAPH_GAPX = 1, 2, 3, 4      # the -G ids


def graph_state(backend, nscans):
    """the vector a back-end carries over the iterations of one doGraphSlam6D call: ghelix6DQ2's B | bd
    (ghelix6DQ2.cc:329-330), gapx6D's T (gapx6D.cc:356); None for the LUM back-ends"""
    n = nscans - 1
    if backend == GRAPH_GHELIX:
        return np.zeros((6 * n) * (6 * n) + 6 * n)
    if backend == GRAPH_GAPX:
        return np.zeros(3 * n)
    return None


def graph_iteration_comm(backend, gr, allScans, max_dist_match2, comm, state=None):
    """One iteration of doGraphSlam6D of back-end `backend` with the library's own communicator: link passes of this
    rank, ncclAllReduce of the blocks, solve and pose update all inside tdtk_graph_iteration -- Python only
    marshals the handles and poses in and the poses out.  comm = NativeComm or None (single process)."""
    import ctypes as C
    from ._capi import lib, check, dptr, iptr
    rank, world = (comm.rank, comm.world) if comm is not None else (0, 1)
    nscans, nlinks = gr.getNrScans(), gr.getNrLinks()
    sc = allScans[:nscans]
    frm, to = _link_arrays(gr)
    mine_a = np.flatnonzero(link_owners(gr, world, sc) == rank).astype(np.int32) if world > 1 else np.arange(nlinks, dtype=np.int32)
    nl = len(mine_a)
    fl, tl = frm[mine_a].tolist(), to[mine_a].tolist()
    first = (C.c_void_p * max(1, nl))(*[sc[a].getSearchTree()._h for a in fl])
    second = (C.c_void_p * max(1, nl))(*[sc[b].handle for b in tl])
    tm = np.array([s.transMat for s in sc], dtype=np.float64).reshape(nscans, 16)
    da = np.array([s.dalignxf for s in sc], dtype=np.float64).reshape(nscans, 16)
    rp = np.array([s.rPos for s in sc], dtype=np.float64).reshape(nscans, 3)
    rt = np.array([s.rPosTheta for s in sc], dtype=np.float64).reshape(nscans, 3)
    dal = np.ascontiguousarray(da[fl]) if nl else np.zeros((1, 16))
    if not nl:
        mine_a = np.zeros(1, np.int32)
    hs = (C.c_void_p * nscans)(*[s._h for s in sc])
    xf = np.zeros((nscans, 32))
    ret = C.c_double(0.0)
    check(lib().tdtk_graph_iteration(int(backend), comm._h if comm is not None else None, nlinks, iptr(frm), iptr(to), nl,
                                     iptr(mine_a), first, dptr(dal), second, float(max_dist_match2), nscans, dptr(tm), dptr(da),
                                     dptr(rp), dptr(rt), hs, dptr(state) if state is not None else None, dptr(xf),
                                     C.byref(ret)))
    two = backend in (GRAPH_LUMEULER, GRAPH_LUMQUAT)
    # (the rows as lists of views, the frames' copies in one go: this loop runs once per scan and round on the host, and
    # sixty-three rounds of numpy indexing + method calls were a quarter of a millisecond of an 84-link round's 0.45 ms
    # outside the link launch)
    r_tm, r_da, r_rp, r_rt, r_fr = list(tm), list(da), list(rp), list(rt), list(tm.copy())
    last = nscans - 1
    for i in range(1, nscans):
        s = sc[i]
        s.transMat = r_tm[i]; s.dalignxf = r_da[i]; s.rPos = r_rp[i]; s.rPosTheta = r_rt[i]
        if s._h is None:
            s._queue.append(xf[i, :16])
            if two:
                s._queue.append(xf[i, 16:])
        if i != last:
            s.frames.append((r_fr[i], "LUM"))            # Scan::addFrame: what _addFrames("LUM", 1) does
        else:
            s._addFrames("LUM", 2)                       # lum6Deuler.cc:451-455: the last scan with islum == 2
    return ret.value


def graph_iteration(backend, gr, allScans, max_dist_match2, state=None, group=None, device=None):
    """One iteration of doGraphSlam6D of the back-end `backend` (-G id), everything numeric in the library:
    this rank's links in one batched call (tdtk_graph_link_blocks), ONE all-reduce of the per-link blocks
    (each link has exactly one owner, so the sum is exact and the result does not depend on the number of
    ranks), then scatter in link order + solve + pose update on every rank (tdtk_graph_solve_update), which
    also moves the resident scans.  Returns `ret`.  (The exchange here goes through torch.distributed -- the CPU
    test rig's gloo groups; with RCCL use graph_iteration_comm.)"""
    import ctypes as C
    from ._capi import lib, check, dptr, iptr
    rank, world = _rank_world(group)
    nscans, nlinks = gr.getNrScans(), gr.getNrLinks()
    mine = shard_links(gr, rank, world)
    nl = len(mine)
    Bn = lib().tdtk_graph_block_doubles(int(backend))
    blocks = np.zeros((nlinks, Bn))
    if nl:
        first = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[allScans[gr.getLink(i, 1)].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([allScans[gr.getLink(i, 0)].dalignxf for i in mine]))
        mb = np.empty((nl, Bn))
        check(lib().tdtk_graph_link_blocks(int(backend), nl, first, dptr(dal), second, float(max_dist_match2), dptr(mb)))
        blocks[mine] = mb
    blocks = np.ascontiguousarray(_exchange_blocks(blocks, world, group, device))
    frm = np.ascontiguousarray([gr.getLink(i, 0) for i in range(nlinks)], dtype=np.int32)
    to = np.ascontiguousarray([gr.getLink(i, 1) for i in range(nlinks)], dtype=np.int32)
    sc = allScans[:nscans]
    tm = np.ascontiguousarray(np.stack([s.transMat for s in sc]))
    da = np.ascontiguousarray(np.stack([s.dalignxf for s in sc]))
    rp = np.ascontiguousarray(np.stack([s.rPos for s in sc]))
    rt = np.ascontiguousarray(np.stack([s.rPosTheta for s in sc]))
    hs = (C.c_void_p * nscans)(*[s._h for s in sc])
    xf = np.zeros((nscans, 32))
    ret = C.c_double(0.0)
    check(lib().tdtk_graph_solve_update(int(backend), nlinks, iptr(frm), iptr(to), dptr(blocks), nscans, dptr(tm),
                                        dptr(da), dptr(rp), dptr(rt), hs, dptr(state) if state is not None else None,
                                        dptr(xf), C.byref(ret)))
    two = backend in (GRAPH_LUMEULER, GRAPH_LUMQUAT)       # transformToEuler / transformToQuat: two transforms
    for i in range(1, nscans):
        s = sc[i]
        s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i], da[i], rp[i], rt[i]
        if s._h is None:                       # not resident on this rank: replay later, in order
            s._queue.append(xf[i, :16])
            if two:
                s._queue.append(xf[i, 16:])
        s._addFrames("LUM", 2 if i == nscans - 1 else 1)     # lum6Deuler.cc:451-455: the last scan with islum == 2
    return ret.value


def lumquat_iteration(gr, allScans, max_dist_match2, group=None, device=None):
    """lum6DQuat::doGraphSlam6D, one iteration (src/slam6d/lum6Dquat.cc:319-481)"""
    return graph_iteration(GRAPH_LUMQUAT, gr, allScans, max_dist_match2, None, group, device)


def ghelix_iteration(gr, allScans, max_dist_match2, state, group=None, device=None):
    """ghelix6DQ2::doGraphSlam6D, one iteration (src/slam6d/ghelix6DQ2.cc:330-449); state = graph_state(3, n)"""
    return graph_iteration(GRAPH_GHELIX, gr, allScans, max_dist_match2, state, group, device)


def gapx_iteration(gr, allScans, max_dist_match2, T, group=None, device=None):
    """gapx6D::doGraphSlam6D, one iteration (src/slam6d/gapx6D.cc:323-542); T = graph_state(4, n)"""
    return graph_iteration(GRAPH_GAPX, gr, allScans, max_dist_match2, T, group, device)


def _dist_ready():
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:  # torch absent: single process
        return False
