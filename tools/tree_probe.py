"""GPU box: search-tree build time (tdtk_tree_create from host points, device builder) for the bundled scan and 1M."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
from oracle import orc
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
for name, pts in (("dat 81K", z["scan000"]), ("uniform 1M", orc.gen_mt64_uniform(42, 3000000, -1000, 1000).reshape(-1, 3))):
    v = []
    for rep in range(6):
        t0 = time.perf_counter(); kd = t.KDtree(pts, 20); v.append((time.perf_counter() - t0) * 1e3)
    print("%s: tree_create min %.2f ms median %.2f ms (build_ms %.2f), verify %s" % (name, min(v), np.median(v), kd.info()["build_ms"], kd.verify()))
