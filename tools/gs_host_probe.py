"""GPU box: where a graph-SLAM bench step spends its time on the host side (Graph construction, marshalling, the library call,
writing the poses back) -- 64 scans x N points, one lum6DEuler round per step."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
gs = importlib.import_module("3dtk_amd.graphslam")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
scans = [t.Scan(p, th, loc) for (p, th, loc) in bench.make_graphslam_scans(64, npts)]
t.prepare_scans(scans, trees=True, threads=8)
import cProfile, pstats
def step():
    gr = t.Graph(64, 500.0 ** 2, 20, scans)
    return gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 625.0, None)
for _ in range(3): step()
t0 = time.perf_counter()
for _ in range(10): step()
print("step %.3f ms" % ((time.perf_counter() - t0) * 100))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
