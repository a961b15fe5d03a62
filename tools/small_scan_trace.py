"""GPU box, under rocprofv3 --kernel-trace: one small pair, 3 x 40 ICP iterations; the trace's timestamps give kernel durations and gaps."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
raw = bench.make_small_scans(2)
red = [t.calcReducedPoints(loc, 10.0, device=0) for _, _, loc in raw]
S = [t.Scan(p, th, r) for (p, th, _), r in zip(raw[:2], red[:2])]
S[0].getSearchTree(); _ = S[1].handle
for rep in range(3):
    icp = t.icp6D(t.icp6D_QUAT(True), 75.0, 40, quiet=True, epsilonICP=-1.0)
    t0 = time.perf_counter(); icp.match(S[0], S[1]); dt = time.perf_counter() - t0
    print("40 iterations: %.1f us" % (dt * 1e6))
