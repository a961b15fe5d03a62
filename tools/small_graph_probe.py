"""GPU box: one LUM iteration on a graph of small scans, links on 1 vs several streams."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
ns, npts = int(sys.argv[1]) if len(sys.argv) > 1 else 32, int(sys.argv[2]) if len(sys.argv) > 2 else 60000
raw = bench.make_graphslam_scans(ns, npts, seed=4)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
t.prepare_scans(scans, threads=8)
g = t.Graph(ns, 700.0 ** 2, 8, scans)
best = 1e9
for rep in range(6):
    t0 = time.perf_counter(); ret = gs.lum_iteration_native(g, scans, 625.0); dt = time.perf_counter() - t0
    best = min(best, dt)
print("lanes=%s: %d scans x %d pts, %d links: LUM iteration %.2f ms (best of 6), ret %.5f" % (os.environ.get("TDTK_LINK_LANES", "8"), ns, npts, g.getNrLinks(), best * 1e3, ret))
