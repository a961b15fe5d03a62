import importlib, os, sys, ctypes as C
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
S = [t.Scan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(2)]
kd = S[0].getSearchTree(); print(kd.info())
S[1].mergeCoordinatesWithRoboterPosition(S[0])
q = S[1].get_xyz_reduced()
c = kd.count_visits(q, 625.0); print("visits/query: internal %.1f leaves %.2f points %.1f" % tuple(x / len(q) for x in c))
idx, d2 = kd.FindClosestBatch(q, 625.0)
print("found", (idx >= 0).sum(), "of", len(q))
for rep in range(3):
    r = t.Scan.getPtPairs(S[0], S[1], max_dist_match2=625.0)
    ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms)); print("k_search %.4f ms" % ms.value)
pts = z["scan000"]; print("extent", pts.min(0), pts.max(0))
