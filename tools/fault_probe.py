"""GPU box: the speculative tree build with its check forced to fail (TDTK_BUILD_SPEC_FAULT=1): the in-order build must
take over and the tree must come out right; prints how many builds were redone."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
capi = importlib.import_module("3dtk_amd._capi")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
rng = np.random.default_rng(3)
clouds = [("dat", z["scan000"]), ("uniform 9000", rng.uniform(-100, 100, (9000, 3))), ("uniform 300000", rng.uniform(-100, 100, (300000, 3))),
          ("clump", np.concatenate([rng.normal(0, 0.01, (200000, 3)), rng.uniform(-100, 100, (100000, 3))]))]
for name, pts in clouds:
    print(name, len(pts), flush=True)
    kd = t.KDtree(pts, 20)
    print("   ", kd.verify(), "respeculated", capi.build_respeculated(), kd.info()["max_depth"], flush=True)
