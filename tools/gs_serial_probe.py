"""GPU box: phases of one native LUM iteration (64 x 1M, 84 links) as rank 0 of a simulated world of W ranks
(only rank 0's links are computed and only its scans are resident; the exchange is skipped), to see what stays
serial when the links are dealt over more GPUs.  usage: python tools/gs_serial_probe.py [W ...]"""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
L = capi.lib(); dptr = capi.dptr; iptr = capi.iptr; check = capi.check
ns, npts = 64, 1000000
raw = bench.make_graphslam_scans(ns, npts)
for W in [int(a) for a in sys.argv[1:]] or [1, 8]:
    scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
    g0 = t.Graph(ns, 500.0 ** 2, 20, scans)
    mine = gs.shard_links(g0, 0, W)
    need_tree = sorted({g0.getLink(i, 0) for i in mine}); need_pts = sorted({g0.getLink(i, 1) for i in mine} - set(need_tree))
    t.prepare_scans([scans[k] for k in need_tree], trees=True, threads=8)
    t.prepare_scans([scans[k] for k in need_pts], trees=False, threads=8)
    nres = sum(1 for s in scans if s._h is not None)
    for rep in range(5):
        T = [time.perf_counter()]
        gr = t.Graph(ns, 500.0 ** 2, 20, scans); T.append(time.perf_counter())
        nlinks = gr.getNrLinks(); nl = len(mine)
        first = (C.c_void_p * nl)(*[scans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[scans[gr.getLink(i, 1)].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([scans[gr.getLink(i, 0)].dalignxf for i in mine]))
        Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
        T.append(time.perf_counter())
        check(L.tdtk_lum_links(nl, first, dptr(dal), second, 625.0, dptr(Cm), dptr(CD), m, dptr(ss))); T.append(time.perf_counter())
        blocks = np.zeros((nlinks, 42)); blocks[mine, :36] = Cm; blocks[mine, 36:] = CD
        if W > 1:     # stand-in for the other ranks' blocks so that the system is solvable: reuse this rank's
            for i in range(nlinks):
                if i not in mine: blocks[i] = blocks[mine[i % nl]]
        frm = np.ascontiguousarray([gr.getLink(i, 0) for i in range(nlinks)], dtype=np.int32)
        to = np.ascontiguousarray([gr.getLink(i, 1) for i in range(nlinks)], dtype=np.int32)
        Call = np.ascontiguousarray(blocks[:, :36]); CDall = np.ascontiguousarray(blocks[:, 36:]); X = np.empty(6 * (ns - 1))
        T.append(time.perf_counter())
        check(L.tdtk_lum_assemble_solve(nlinks, iptr(frm), iptr(to), dptr(Call), dptr(CDall), ns, dptr(X), None, None)); T.append(time.perf_counter())
        X *= 0.0      # keep the scene where it is from repetition to repetition
        tm = np.ascontiguousarray(np.stack([s.transMat for s in scans])); da = np.ascontiguousarray(np.stack([s.dalignxf for s in scans]))
        rp = np.ascontiguousarray(np.stack([s.rPos for s in scans])); rt = np.ascontiguousarray(np.stack([s.rPosTheta for s in scans]))
        hs = (C.c_void_p * ns)(*[s._h for s in scans]); xf = np.zeros((ns, 32)); ret = C.c_double(0.0)
        T.append(time.perf_counter())
        check(L.tdtk_lum_update_poses(ns, dptr(X), dptr(tm), dptr(da), dptr(rp), dptr(rt), hs, dptr(xf), C.byref(ret))); T.append(time.perf_counter())
        for i in range(1, ns):
            s = scans[i]
            s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i], da[i], rp[i], rt[i]
            s.frames.append((tm[i], "LUM"))
        T.append(time.perf_counter())
        names = ["graph", "marshal", "lum_links(%d)" % nl, "blocks", "assemble+solve", "marshal poses", "update_poses(%d resident)" % nres, "bookkeeping"]
        if rep >= 2:
            print("W=%d | " % W + " | ".join("%s %.3f" % (nm, (T[i + 1] - T[i]) * 1e3) for i, nm in enumerate(names)), "| total %.2f" % ((T[-1] - T[0]) * 1e3), flush=True)
    del scans
