"""GPU box: sequential ICP over a chain of scans (the C3 shape), with and without prefetching the next
scan's upload + tree build on a second host thread."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
ns, npts = int(sys.argv[1]) if len(sys.argv) > 1 else 10, int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
raw = bench.make_graphslam_scans(ns, npts)
for pf in (False, True, False, True):
    scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
    icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 30, quiet=True, epsilonICP=1e-5)
    t0 = time.perf_counter()
    icp.doICP(scans, prefetch=pf)
    dt = time.perf_counter() - t0
    print("prefetch %-5s: %d scans x %d pts, doICP %.1f ms (%.1f ms / scan); last pose %s" % (pf, ns, npts, dt * 1e3, dt * 1e3 / ns, np.round(scans[-1].get_rPos(), 4)))
