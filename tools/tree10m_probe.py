"""GPU box: search-tree build time at 1M / 10M uniform points (device builder, verified against the host twin once)."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
sizes = [int(a) for a in sys.argv[1:]] or [1000000, 10000000]
for n in sizes:
    rng = np.random.default_rng(7 + n)
    pts = rng.uniform(-1000.0, 1000.0, (n, 3))
    v, b = [], []
    kd = None
    for rep in range(6):
        t0 = time.perf_counter(); kd = t.KDtree(pts, 20); v.append((time.perf_counter() - t0) * 1e3); b.append(kd.info()["build_ms"])
    print("%d points: tree_create min %.2f ms, build_ms min %.3f median %.3f, verify %s" % (n, min(v), min(b), np.median(b), kd.verify()), flush=True)
