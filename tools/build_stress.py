"""GPU box: two host threads building (and verifying) trees of various sizes at the same time -- the speculative build's
second stream, the handle pool and the arenas under concurrency."""
import importlib, os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
capi = importlib.import_module("3dtk_amd._capi")
bad = []


def work(seed, reps):
    rng = np.random.default_rng(seed)
    for r in range(reps):
        n = int(rng.choice([9000, 20000, 81360, 150000, 400000, 1000000]))
        kind = rng.integers(0, 3)
        if kind == 0:
            p = rng.uniform(-500, 500, (n, 3))
        elif kind == 1:
            p = np.concatenate([rng.normal(0, 0.05, (n // 2, 3)), rng.uniform(-300, 300, (n - n // 2, 3))])
        else:
            p = np.round(rng.uniform(-200, 200, (n, 3)), 1)      # many duplicates, ties in the sums
        kd = t.KDtree(p, int(rng.choice([3, 20])))
        v = kd.verify()
        if v != [0, 0, 0, 0]:
            bad.append((seed, r, n, kind, v))
        del kd


th = [threading.Thread(target=work, args=(s, int(sys.argv[1]) if len(sys.argv) > 1 else 25)) for s in (1, 2)]
for x in th: x.start()
for x in th: x.join()
print("build stress: mismatching trees", bad, "rebuilt in order", capi.build_respeculated())
