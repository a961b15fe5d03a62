"""GPU box: one native graph-SLAM iteration on the C4 workload (64 x 1M, 84 links) for different numbers of
concurrent link streams (TDTK_LINK_LANES) -- link passes alone and the whole iteration.
usage: python tools/gs_lanes_probe.py"""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
ns, npts = 64, 1000000
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
t.prepare_scans(scans, trees=True, threads=8)
L = capi.lib()
for _ in range(3):
    gs.graph_iteration_comm(1, t.Graph(ns, 500.0 ** 2, 20, scans), scans, 625.0, None)
for lanes in ("1", "2", "3", "4", "6", "8"):
    os.environ["TDTK_LINK_LANES"] = lanes
    tl, ti = [], []
    for rep in range(4):
        gr = t.Graph(ns, 500.0 ** 2, 20, scans)
        nl = gr.getNrLinks()
        first = (C.c_void_p * nl)(*[scans[gr.getLink(i, 0)].getSearchTree()._h for i in range(nl)])
        second = (C.c_void_p * nl)(*[scans[gr.getLink(i, 1)].handle for i in range(nl)])
        dal = np.ascontiguousarray(np.stack([scans[gr.getLink(i, 0)].dalignxf for i in range(nl)]))
        blocks = np.empty((nl, 42))
        t0 = time.perf_counter()
        capi.check(L.tdtk_graph_link_blocks(1, nl, first, capi.dptr(dal), second, 625.0, capi.dptr(blocks)))
        tl.append(time.perf_counter() - t0)
        t0 = time.perf_counter()
        gs.graph_iteration_comm(1, t.Graph(ns, 500.0 ** 2, 20, scans), scans, 625.0, None)
        ti.append(time.perf_counter() - t0)
    print("lanes %s: link passes %.2f ms (%.3f ms / link), whole iteration %.2f ms" % (lanes, min(tl) * 1e3, min(tl) * 1e3 / nl, min(ti) * 1e3))
