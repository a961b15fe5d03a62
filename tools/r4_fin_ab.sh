#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for fin in 1 0; do
TDTK_LIB=lab TDTK_BUILD_FINISH=$fin python /dev/stdin <<'PY' 2>&1 | tail -4
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "dat_scans.npz"))
clouds = {"dat 81K": z["scan000"], "uniform 40K": np.random.default_rng(1).uniform(-100, 100, (40000, 3)),
          "uniform 300K": np.random.default_rng(3).uniform(-100, 100, (300000, 3)), "uniform 1M": np.random.default_rng(2).uniform(-100, 100, (1000000, 3))}
out = []
for name, pts in clouds.items():
    ts = []
    for rep in range(8):
        t0 = time.perf_counter(); kd = t.KDtree(pts, 20); ts.append((time.perf_counter() - t0) * 1e3)
        if rep == 7: v = kd.verify()
        del kd
    out.append("%s: min %.3f med %.3f ms %s" % (name, min(ts), sorted(ts)[4], v))
print("FINISH=%s | " % os.environ["TDTK_BUILD_FINISH"] + " | ".join(out))
PY
done
