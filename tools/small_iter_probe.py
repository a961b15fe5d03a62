"""GPU box: wall time of an ICP iteration on synthetic pairs of 20K / 81K / 97K / 200K points (the size classes of the
small-batch kernels): 40 iterations of one icp6D::match, median of 7 matches."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
for n in (20000, 81000, 97000, 200000):
    m, d, T = bench.make_icp_pair(n)
    model = t.Scan([0, 0, 0], [0, 0, 0], m); model.getSearchTree()
    ts = []
    for rep in range(7):
        data = t.Scan([0, 0, 0], [0, 0, 0], d); _ = data.handle
        icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 40, quiet=True, epsilonICP=-1.0)
        t0 = time.perf_counter(); icp.match(model, data); ts.append(time.perf_counter() - t0)
        data.release()
    print("%6d points: %.1f us per iteration (40 iterations, median of 7; rms %.6f, pairs %d)" % (n, 1e6 * sorted(ts)[3] / 40, icp.last["rms"], icp.last["pairs"]))
    model.release()
