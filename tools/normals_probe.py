"""Times tdtk_normals_apx_knn (host arrays in, host arrays out) and, with --cpu, the oracle restatement of
calculateNormalsApxKNN on a bounded sample.  usage: python tools/normals_probe.py [--n 1000000] [--reps 5] [--cpu N]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module  # noqa: E402

tdtk = import_module("3dtk_amd")

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1000000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cpu", type=int, default=0, help="also time the CPU oracle on this many points")
ap.add_argument("--shape", default="plane")
a = ap.parse_args()
rng = np.random.default_rng(21)
p = rng.uniform(-1000, 1000, (a.n, 3))
if a.shape == "plane":
    p[:, 2] = 0.05 * p[:, 0] + rng.normal(0, 1.0, a.n)
rp = [0.0, 0.0, 500.0]
tdtk.calculateNormalsApxKNN(p[:1000], 10, rp, 1.0)
for r in range(a.reps):
    t = time.perf_counter()
    n = tdtk.calculateNormalsApxKNN(p, 10, rp, 1.0)
    dt = time.perf_counter() - t
    print("n=%d %s: %.2f ms  (%.3g points/s)" % (a.n, a.shape, 1e3 * dt, a.n / dt), flush=True)
sc = tdtk.Scan([0, 0, 0], [0, 0, 0], p)
_ = sc.handle
sc.rPos = np.array(rp)
for r in range(a.reps):
    t = time.perf_counter()
    sc.calcNormals()
    dt = time.perf_counter() - t
    print("resident scan n=%d: %.2f ms  (%.3g points/s)" % (a.n, 1e3 * dt, a.n / dt), flush=True)
if a.cpu:
    from oracle import orc
    t = time.perf_counter()
    w = orc.normals_apx_knn(p[:a.cpu], 10, rp, 1.0)
    dt = time.perf_counter() - t
    print("cpu oracle n=%d: %.1f ms (%.3g points/s, 1 thread)" % (a.cpu, 1e3 * dt, a.cpu / dt))
    if orc.have_ref():
        t = time.perf_counter()
        orc.normals_apx_knn(p[:a.cpu], 10, rp, 1.0, "ref")
        dt = time.perf_counter() - t
        print("vendored ANN + newmat n=%d: %.1f ms (%.3g points/s, 1 thread)" % (a.cpu, 1e3 * dt, a.cpu / dt))
