"""GPU box (optionally under rocprofv3 --kernel-trace): build the tree of an N-point scan a few times; wall per build."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 81000
m, d, T = bench.make_icp_pair(n)
ts = []
for rep in range(8):
    S = t.Scan([0, 0, 0], [0, 0, 0], m); _ = S.handle
    t0 = time.perf_counter(); S.getSearchTree(); ts.append(time.perf_counter() - t0)
    S.release()
print("tree of %d points: %s us (median %.0f)" % (n, " ".join("%.0f" % (x * 1e6) for x in ts), 1e6 * sorted(ts)[4]))
