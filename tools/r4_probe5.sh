#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4e
TDTK_BUILD_TRACE=1 python /dev/stdin <<'PY' 2>&1 | tail -12 | tee gpurun_out/r4e/levels.log
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "dat_scans.npz"))
pts = z["scan000"]
for rep in range(4):
    t0 = time.perf_counter(); kd = t.KDtree(pts, 20); print("tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), flush=True)
u = np.random.default_rng(1).uniform(-100, 100, (40000, 3))
for rep in range(3):
    t0 = time.perf_counter(); kd = t.KDtree(u, 20); print("uniform 40K tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), flush=True)
PY
