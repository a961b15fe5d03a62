"""GPU box: where the time of one native graph-SLAM iteration (C4: 64 x 1M, 84 links) goes outside the link passes --
graph build, link dealing, marshalling, link passes, solve + pose update (with and without moving the resident scans).
usage: python tools/gs_breakdown_probe.py [world]"""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
world = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ns, npts = 64, 1000000
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
t.prepare_scans(scans, trees=True, threads=8)
L = capi.lib()
for _ in range(3):
    gs.graph_iteration_comm(1, t.Graph(ns, 500.0 ** 2, 20, scans), scans, 625.0, None)
T = {}
def tick(k, t0):
    T.setdefault(k, []).append(time.perf_counter() - t0)
for rep in range(6):
    t0 = time.perf_counter(); gr = t.Graph(ns, 500.0 ** 2, 20, scans); tick("Graph()", t0)
    t0 = time.perf_counter(); own = gs.link_owners(gr, world, scans); mine = [i for i in range(gr.getNrLinks()) if own[i] == 0]; tick("deal links", t0)
    t0 = time.perf_counter()
    nl, NL = len(mine), gr.getNrLinks()
    frm = np.ascontiguousarray([gr.getLink(i, 0) for i in range(NL)], dtype=np.int32)
    to = np.ascontiguousarray([gr.getLink(i, 1) for i in range(NL)], dtype=np.int32)
    first = (C.c_void_p * nl)(*[scans[frm[i]].getSearchTree()._h for i in mine])
    second = (C.c_void_p * nl)(*[scans[to[i]].handle for i in mine])
    dal = np.ascontiguousarray(np.stack([scans[frm[i]].dalignxf for i in mine]))
    tm = np.ascontiguousarray(np.stack([s.transMat for s in scans])); da = np.ascontiguousarray(np.stack([s.dalignxf for s in scans]))
    rp = np.ascontiguousarray(np.stack([s.rPos for s in scans])); rt = np.ascontiguousarray(np.stack([s.rPosTheta for s in scans]))
    hs = (C.c_void_p * ns)(*[s._h for s in scans]); xf = np.zeros((ns, 32)); ret = C.c_double(0.0)
    blocks = np.zeros((NL, 42)); mb = np.empty((nl, 42))
    tick("marshal in", t0)
    t0 = time.perf_counter()
    capi.check(L.tdtk_graph_link_blocks(1, nl, first, capi.dptr(dal), second, 625.0, capi.dptr(mb))); tick("link passes (%d links)" % nl, t0)
    blocks[mine] = mb
    tm2, da2, rp2, rt2 = tm.copy(), da.copy(), rp.copy(), rt.copy()
    t0 = time.perf_counter()
    capi.check(L.tdtk_graph_solve_update(1, NL, capi.iptr(frm), capi.iptr(to), capi.dptr(blocks), ns, capi.dptr(tm2), capi.dptr(da2),
                                         capi.dptr(rp2), capi.dptr(rt2), None, None, capi.dptr(xf), C.byref(ret))); tick("solve + poses, scans not moved", t0)
    t0 = time.perf_counter()
    capi.check(L.tdtk_graph_solve_update(1, NL, capi.iptr(frm), capi.iptr(to), capi.dptr(blocks), ns, capi.dptr(tm), capi.dptr(da),
                                         capi.dptr(rp), capi.dptr(rt), hs, None, capi.dptr(xf), C.byref(ret))); tick("solve + poses + %d resident scans moved" % (ns - 1), t0)
    t0 = time.perf_counter()
    for i in range(1, ns):
        s = scans[i]
        s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i], da[i], rp[i], rt[i]
        s.frames.append((tm[i], "LUM"))
    tick("marshal out", t0)
    t0 = time.perf_counter(); gs.graph_iteration_comm(1, t.Graph(ns, 500.0 ** 2, 20, scans), scans, 625.0, None); tick("whole iteration (world 1)", t0)
for k, v in T.items():
    print("%-52s min %.3f  median %.3f ms" % (k, min(v) * 1e3, float(np.median(v)) * 1e3))
