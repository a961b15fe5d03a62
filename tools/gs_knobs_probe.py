"""GPU box: link passes of the C4 workload (64 x 1M, 84 links, converged poses) under different search-kernel knobs
(slab length, refill threshold, pieces per slab) -- the knobs are read per launch.
usage: python tools/gs_knobs_probe.py"""
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
nl = gr.getNrLinks()
first = (C.c_void_p * nl)(*[scans[gr.getLink(i, 0)].getSearchTree()._h for i in range(nl)])
second = (C.c_void_p * nl)(*[scans[gr.getLink(i, 1)].handle for i in range(nl)])
dal = np.ascontiguousarray(np.stack([scans[gr.getLink(i, 0)].dalignxf for i in range(nl)]))
blocks = np.empty((nl, 42))
ref = None
knobs = [{}] + [{"TDTK_REFILL_QPW": q, "TDTK_REFILL_PHASES": "1", "TDTK_LINK_LANES": l} for q in ("256", "320", "384", "448", "512", "640") for l in ("2", "3", "4")]
if len(sys.argv) > 1:
    knobs = [dict(kv.split("=") for kv in a.split(",")) if a != "default" else {} for a in sys.argv[1:]]
for kn in knobs * 2:
    for k in ("TDTK_REFILL_QPW", "TDTK_REFILL_THRESH", "TDTK_REFILL_PHASES", "TDTK_SEARCH_VARIANT", "TDTK_LINK_LANES", "TDTK_FUSE_LUM", "TDTK_LINK_BATCH", "TDTK_LINK_PHASES"):
        os.environ.pop(k, None)
    os.environ.update(kn)
    ts = []
    for rep in range(4):
        t0 = time.perf_counter()
        capi.check(L.tdtk_graph_link_blocks(1, nl, first, capi.dptr(dal), second, 625.0, capi.dptr(blocks)))
        ts.append(time.perf_counter() - t0)
    if ref is None:
        ref = blocks.copy()
    print("%-60s %.2f ms  (%.4f ms / link)  blocks identical: %s" % (kn or "default", min(ts) * 1e3, min(ts) * 1e3 / nl, np.array_equal(ref[:, :36] != 0, blocks[:, :36] != 0) and np.allclose(ref, blocks, rtol=1e-9, atol=1e-9)))
