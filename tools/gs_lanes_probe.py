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
gr = t.Graph(ns, 500.0 ** 2, 20, scans)
NL = gr.getNrLinks()
res = {}
for rnd in range(3):
    for lanes in ("1", "2", "3", "4", "5", "6", "8"):
        os.environ["TDTK_LINK_LANES"] = lanes
        for world in (1, 2, 4, 8):
            mine = [i for i in range(NL) if gs.link_owners(gr, world, scans)[i] == 0]
            nl = len(mine)
            first = (C.c_void_p * nl)(*[scans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
            second = (C.c_void_p * nl)(*[scans[gr.getLink(i, 1)].handle for i in mine])
            dal = np.ascontiguousarray(np.stack([scans[gr.getLink(i, 0)].dalignxf for i in mine]))
            blocks = np.empty((nl, 42))
            for rep in range(3):
                t0 = time.perf_counter()
                capi.check(L.tdtk_graph_link_blocks(1, nl, first, capi.dptr(dal), second, 625.0, capi.dptr(blocks)))
                dt = time.perf_counter() - t0
                res.setdefault((lanes, world), []).append(dt)
for lanes in ("1", "2", "3", "4", "5", "6", "8"):
    print("lanes %s: " % lanes + "  ".join("world %d (%2d links) %.2f/%.2f ms" % (w, len(res[(lanes, w)]) and sum(1 for i in range(NL) if gs.link_owners(gr, w, scans)[i] == 0), min(res[(lanes, w)]) * 1e3, float(np.median(res[(lanes, w)])) * 1e3) for w in (1, 2, 4, 8)))
