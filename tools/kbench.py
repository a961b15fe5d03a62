"""Kernel micro-bench (GPU box): repeated whole-scan correspondence passes on the 1M pair.
usage: python tools/kbench.py [reps] [mode]   mode: icp (data near model) | rand (independent queries)"""
import importlib, os, sys, time, ctypes as C
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
from oracle import orc
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "icp"
M = int(os.environ.get("KB_M", 1000000))
s = orc.gen_mt64_uniform(42, 6 * M, -1000, 1000)
m, q = s[:3 * M].reshape(M, 3).copy(), s[3 * M:].reshape(M, 3).copy()
if mode == "icp":
    q = m[np.random.default_rng(1).permutation(M)] + np.random.default_rng(2).normal(0, 1, (M, 3))
ms_ = t.Scan([0, 0, 0], [0, 0, 0], m); qs = t.Scan([0, 0, 0], [0, 0, 0], q)
ms_.getSearchTree(); _ = qs.handle
tms = []
for r in range(reps):
    res = t.Scan.getPtPairs(ms_, qs, max_dist_match2=625.0)
    ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms)); tms.append(ms.value)
tms = np.array(tms[2:]) if reps > 4 else np.array(tms)
print("mode %s n=%d pairs=%d  k_search ms: min %.4f med %.4f  -> %.3e NN/s (med)" % (mode, M, res["n"], tms.min(), np.median(tms), M / (np.median(tms) * 1e-3)))
