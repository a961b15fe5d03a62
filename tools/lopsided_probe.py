import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
t = importlib.import_module("3dtk_amd")
rng = np.random.default_rng(5); n = 4000000
p = np.concatenate([rng.uniform(-1000, 1000, (n - n // 3, 3)), rng.normal(0, 3, (n // 3, 3)) + rng.uniform(-500, 500, 3)])[rng.permutation(n)]
for r in range(3):
    kd = t.KDtree(np.ascontiguousarray(p), 20); print("build_ms %.2f depth %d" % (kd.info()["build_ms"], kd.info()["max_depth"]), flush=True)
