"""GPU box: where does a scan's time go in doICP over small (~15K-point) scans?  Per scan: waiting for the prepared scan,
mergeCoordinatesWithRoboterPosition, the match call (wall), inside it tdtk_icp_match's own total / search / sums clock."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
L = t.lib()
raw = bench.make_small_scans(16)
red = [t.calcReducedPoints(loc, 10.0, device=0) for _, _, loc in raw]
print("reduced points", [len(r) for r in red][:6])
for timing in (0, 1):
    for rep in range(3):
        S = [t.Scan(p, th, r) for (p, th, _), r in zip(raw, red)]
        icp = t.icp6D(t.icp6D_QUAT(True), 75.0, 100, quiet=True, epsilonICP=1e-5)
        L.tdtk_kernel_timing(timing)
        rec = []
        orig = icp.match
        def m(a, b, pm=0):
            t0 = time.perf_counter(); it = orig(a, b, pm); dt = time.perf_counter() - t0
            rec.append((it, dt, icp.last["total_ms"], icp.last["nn_ms"], icp.last["sums_ms"])); return it
        icp.match = m
        t0 = time.perf_counter(); icp.doICP(S, prefetch=True); wall = time.perf_counter() - t0
        L.tdtk_kernel_timing(0)
        its = np.array([r[0] + 1 for r in rec]); dts = np.array([r[1] for r in rec]) * 1e3
        tot = np.array([r[2] for r in rec]); nn = np.array([r[3] for r in rec]); sm = np.array([r[4] for r in rec])
        print("timing=%d rep %d: doICP %.3f ms per scan | match wall mean %.3f ms, library total %.3f, iterations %.1f -> %.1f us per iteration; kernels: search %.3f sums %.3f ms per match" %
              (timing, rep, wall * 1e3 / 15, dts.mean(), tot.mean(), its.mean(), 1e3 * tot.sum() / its.sum(), nn.mean(), sm.mean()))
        for s in S: s.release()
# one pair, fixed iteration counts: fixed cost of a match vs its per-iteration cost
S = [t.Scan(p, th, r) for (p, th, _), r in zip(raw[:2], red[:2])]
S[0].getSearchTree(); _ = S[1].handle
for n in (1, 2, 5, 10, 20, 40):
    ts = []
    for rep in range(7):
        icp = t.icp6D(t.icp6D_QUAT(True), 75.0, n, quiet=True, epsilonICP=-1.0)
        t0 = time.perf_counter(); icp.match(S[0], S[1]); ts.append(time.perf_counter() - t0)
    print("match with %2d iterations: %.1f us (median of 7), library %.1f us" % (n, 1e6 * sorted(ts)[3], 1e3 * icp.last["total_ms"]))
