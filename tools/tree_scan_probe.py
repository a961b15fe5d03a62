import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t = importlib.import_module("3dtk_amd")
import bench
m, d, T = bench.make_icp_pair(1000000)
for rep in range(4):
    s = t.Scan([0, 0, 0], [0, 0, 0], m); _ = s.handle
    t0 = time.perf_counter(); tr = s.getSearchTree(); dt = time.perf_counter() - t0
    print("scan-ordered 1M: getSearchTree %.2f ms build_ms %.2f" % (dt * 1e3, tr.info()["build_ms"]))
    del s, tr
