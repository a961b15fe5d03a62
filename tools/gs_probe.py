"""GPU box: time the pieces of one native LUM iteration on the C4 workload."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam"); sl = importlib.import_module("3dtk_amd.slam6d")
from importlib import import_module
capi = import_module("3dtk_amd._capi")
ns, npts = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
g = t.Graph(ns, 500.0 ** 2, 20, scans)
for i in range(g.getNrLinks()):
    scans[g.getLink(i, 0)].getSearchTree(); _ = scans[g.getLink(i, 1)].handle
L = capi.lib()
for rep in range(3):
    t0 = time.perf_counter(); gr = t.Graph(ns, 500.0 ** 2, 20, scans); t1 = time.perf_counter()
    nl = gr.getNrLinks()
    first = (C.c_void_p * nl)(*[scans[gr.getLink(i, 0)].getSearchTree()._h for i in range(nl)])
    second = (C.c_void_p * nl)(*[scans[gr.getLink(i, 1)].handle for i in range(nl)])
    dal = np.ascontiguousarray(np.stack([scans[gr.getLink(i, 0)].dalignxf for i in range(nl)]))
    Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
    t2 = time.perf_counter()
    capi.check(L.tdtk_lum_links(nl, first, capi.dptr(dal), second, 625.0, capi.dptr(Cm), capi.dptr(CD), m, capi.dptr(ss)))
    t3 = time.perf_counter()
    # per-link path for comparison
    for i in range(nl):
        sl.covarianceEuler(scans[gr.getLink(i, 0)], scans[gr.getLink(i, 1)], 625.0)
    t4 = time.perf_counter()
    ret = gs.lum_iteration_native(gr, scans, 625.0)
    t5 = time.perf_counter()
    print("graph %.2f ms | marshal %.2f | lum_links(batched) %.2f ms (%d links, %.3f ms/link) | per-link 2-pass path %.2f ms | full native iteration %.2f ms ret %.4f"
          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, nl, (t3 - t2) * 1e3 / nl, (t4 - t3) * 1e3, (t5 - t4) * 1e3, ret))
