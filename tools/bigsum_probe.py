"""GPU box: the piecewise big-node sum of the tree build (build.hip, k_big_*) against the host builder on clouds chosen
to stress it -- sums that wander through zero, exact rounding ties (integer / dyadic coordinates), huge dynamic range,
repeated points, one-sided clouds -- plus the build time.  TDTK_BUILD_CHAIN=1 selects the plain chain for comparison.
usage: python tools/bigsum_probe.py [points]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
rng = np.random.default_rng(12)
clouds = {
    "uniform +-1000": rng.uniform(-1000, 1000, (n, 3)),
    "positive 0..2000": rng.uniform(0, 2000, (n, 3)),
    "integers +-500 (ties)": rng.integers(-500, 501, (n, 3)).astype(float),
    "half-integers (ties)": rng.integers(-2000, 2001, (n, 3)) * 0.5,
    "dyadic 2^-k steps": rng.integers(-1 << 20, 1 << 20, (n, 3)) * (2.0 ** -7),
    "wide range 1e-6..1e6": rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.uniform(-6, 6, (n, 1)),
    "gaussian clusters": np.concatenate([rng.normal(c, 3.0, (n // 10, 3)) for c in rng.uniform(-800, 800, (10, 3))]),
    "plane z = 0.25 x": (lambda p: np.stack([p[:, 0], p[:, 1], 0.25 * p[:, 0]], 1))(rng.uniform(-700, 700, (n, 2))),
    "tiny values 1e-300": rng.uniform(-1, 1, (n, 3)) * 1e-300,
    "huge values 1e300": rng.uniform(-1, 1, (n, 3)) * 1e300,
    "repeated points": np.repeat(rng.uniform(-100, 100, (n // 50, 3)), 50, axis=0),
    "sum returns to zero": np.concatenate([rng.uniform(0, 1000, (n // 2, 3)), -rng.uniform(0, 1000, (n // 2, 3))])[rng.permutation(n // 2 * 2)],
    "alternating +-1e8 and small": np.where((np.arange(n) % 2 == 0)[:, None], 1e8, -1e8) + rng.uniform(-1, 1, (n, 3)),
}
bad = 0
for name, pts in clouds.items():
    pts = np.ascontiguousarray(pts)
    for bucket in (20, 3):
        t0 = time.perf_counter(); kd = t.KDtree(pts, bucket); dt = (time.perf_counter() - t0) * 1e3
        v = kd.verify()
        bad += v != [0, 0, 0, 0]
        print("%-32s bucket %2d  %8d pts  build %.2f ms  verify %s" % (name, bucket, len(pts), kd.info()["build_ms"], v), flush=True)
print("MISMATCHES:", bad)
