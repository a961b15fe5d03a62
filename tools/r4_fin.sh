#!/bin/bash
cd "$GRAFT_REPO_ROOT"
TDTK_LIB=lab TDTK_BUILD_TRACE=2 python /dev/stdin <<'PY' 2>&1 | tail -14
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "dat_scans.npz"))
pts = z["scan000"]
for rep in range(3):
    t0 = time.perf_counter(); kd = t.KDtree(pts, 20); print("dat 81K tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), kd.verify(), flush=True)
u = np.random.default_rng(1).uniform(-100, 100, (40000, 3))
for rep in range(2):
    t0 = time.perf_counter(); kd = t.KDtree(u, 20); print("uniform 40K tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), kd.verify(), flush=True)
u = np.random.default_rng(2).uniform(-100, 100, (1000000, 3))
for rep in range(2):
    t0 = time.perf_counter(); kd = t.KDtree(u, 20); print("uniform 1M tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), kd.verify(), flush=True)
PY
