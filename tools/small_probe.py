"""GPU box: per-iteration cost of the resident ICP loop on small scans (the bundled 81K-point dat/ pair)."""
import importlib, os, sys, time
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
for rep in range(3):
    S = [t.Scan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(2)]
    S[0].getSearchTree(); _ = S[1].handle
    S[1].mergeCoordinatesWithRoboterPosition(S[0])
    icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 50, quiet=True, epsilonICP=1e-5)
    t0 = time.perf_counter(); it = icp.match(S[0], S[1]); dt = time.perf_counter() - t0
    print("dat pair: %d iterations, wall %.3f ms -> %.1f us / iteration; kernel (k_search) total %.3f ms -> %.1f us / iteration"
          % (it + 1, dt * 1e3, dt * 1e6 / (it + 1), icp.last["nn_ms"], icp.last["nn_ms"] * 1e3 / (it + 1)))
