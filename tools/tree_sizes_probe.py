"""GPU box: tree build time (build_ms, warm, best of 8) of uniform clouds from 15K to 1M points and of the dat/ scan, verified."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
rng = np.random.default_rng(9)
out = []
for name, p in [("15K", rng.uniform(-500, 500, (15000, 3))), ("40K", rng.uniform(-500, 500, (40000, 3))), ("81K", rng.uniform(-500, 500, (81000, 3))),
                ("dat 81K", z["scan000"]), ("300K", rng.uniform(-500, 500, (300000, 3))), ("1M", rng.uniform(-1000, 1000, (1000000, 3)))]:
    b = []
    for r in range(8):
        kd = t.KDtree(np.ascontiguousarray(p), 20); b.append(kd.info()["build_ms"])
    out.append("%s %.3f (%s)" % (name, min(b), kd.verify() == [0, 0, 0, 0]))
print("build_ms: " + " | ".join(out))
