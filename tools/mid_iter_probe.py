"""GPU box: where an ICP iteration over a mid-size scan (97K .. 400K points: the persistent-lane kernel below one resident
generation) spends its time -- wall per iteration, and the library's HIP events around the search and the pair-sum launches."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
L = t.lib()
for n in (97000, 200000, 400000):
    m, d, T = bench.make_icp_pair(n)
    model = t.Scan([0, 0, 0], [0, 0, 0], m); model.getSearchTree()
    for timing in (0, 1):
        L.tdtk_kernel_timing(timing)
        ts = []
        for rep in range(5):
            data = t.Scan([0, 0, 0], [0, 0, 0], d); _ = data.handle
            icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 40, quiet=True, epsilonICP=-1.0)
            t0 = time.perf_counter(); icp.match(model, data); ts.append(time.perf_counter() - t0)
            data.release()
        print("%6d points, events %d: %.1f us per iteration wall; search %.1f us, sums %.1f us per iteration" %
              (n, timing, 1e6 * sorted(ts)[2] / 40, 1e3 * icp.last["nn_ms"] / 40, 1e3 * icp.last["sums_ms"] / 40), flush=True)
    L.tdtk_kernel_timing(0)
    model.release()
