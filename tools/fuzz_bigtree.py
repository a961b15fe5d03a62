"""GPU box: random clouds of 2-6M points (the sizes at which the tree build hands over by subtree size: k_fin_wave /
k_fin_subtrees_half / k_fin_subtrees; two-pass partition over many tiles) built and verified record for record against
the host builder, for --seconds.  usage: python tools/fuzz_bigtree.py [--seconds 90] [--seed 1]"""
import argparse, importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
ap = argparse.ArgumentParser(); ap.add_argument("--seconds", type=float, default=90); ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time(); runs = bad = 0
while time.time() - t0 < a.seconds:
    n = int(rng.integers(2000000, 6000000)); kind = int(rng.integers(0, 6)); bucket = int(rng.integers(6, 41))
    if kind == 0: p = rng.uniform(-1000, 1000, (n, 3))
    elif kind == 1:
        k = int(rng.integers(3, 40)); c = rng.uniform(-1000, 1000, (k, 3)); s = 10.0 ** rng.uniform(-1, 2.5, k)
        w = rng.integers(0, k, n); p = c[w] + rng.normal(0, 1, (n, 3)) * s[w, None]
    elif kind == 2: p = rng.uniform(-1000, 1000, (n, 3)); p[:, 1] = 0.02 * p[:, 0] + rng.normal(0, 0.5, n)
    elif kind == 3: p = np.concatenate([rng.uniform(-1000, 1000, (n - n // 3, 3)), rng.normal(0, 3, (n // 3, 3)) + rng.uniform(-500, 500, 3)])[rng.permutation(n)]
    elif kind == 4: p = np.round(rng.uniform(-300, 300, (n, 3)), int(rng.integers(0, 2)))
    else:
        r = 1000 * rng.uniform(0.01, 1, n) ** 1.7; ang = rng.uniform(0, 2 * np.pi, n)
        p = np.stack([r * np.cos(ang), rng.normal(0, 2, n) + (rng.uniform(0, 1, n) < 0.3) * rng.uniform(0, 60, n), r * np.sin(ang)], 1)
    kd = t.KDtree(np.ascontiguousarray(p), bucket)
    v = kd.verify(); runs += 1
    if v != [0, 0, 0, 0]: bad += 1; print("MISMATCH kind %d n %d bucket %d: %s" % (kind, n, bucket, v), flush=True)
    print("run %d kind %d n %d bucket %d build_ms %.2f depth %d %s" % (runs, kind, n, bucket, kd.info()["build_ms"], kd.info()["max_depth"], v), flush=True)
    del kd
print("fuzz_bigtree: %d clouds, %d mismatches, seed %d" % (runs, bad, a.seed))
