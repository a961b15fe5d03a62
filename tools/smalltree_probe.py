import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
pts = z["scan000"]
for rep in range(5):
    t0 = time.perf_counter(); kd = t.KDtree(pts, 20); dt = time.perf_counter() - t0
    print("tree_create %.2f ms" % (dt * 1e3), {k: round(v, 2) for k, v in kd.info().items() if k.endswith("_ms") or k == "max_depth"})
