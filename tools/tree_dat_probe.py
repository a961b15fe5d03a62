"""GPU box: tdtk_tree_create of the bundled dat/ scan (81K points, a real scan: unbalanced), median of 8 builds."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
for k in range(2):
    ts = []
    for rep in range(8):
        S = t.Scan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]); _ = S.handle
        t0 = time.perf_counter(); S.getSearchTree(); ts.append(time.perf_counter() - t0)
        S.release()
    print("dat scan %d (%d points): %s us (median %.0f)" % (k, len(z["scan%03d" % k]), " ".join("%.0f" % (x * 1e6) for x in ts), 1e6 * sorted(ts)[4]))
