"""GPU box: is k_search sensitive to where the tree is served from?  1M queries against models of 125K .. 8M points at
the SAME point density (box scaled), so the traversal statistics differ only by the tree depth while the working set
goes from "fits every XCD's L2" to "Infinity Cache only".  usage: python tools/l2_probe.py"""
import importlib, os, sys, ctypes as C
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
rng = np.random.default_rng(3)
NQ = 1000000
for M in (125000, 250000, 500000, 1000000, 2000000, 4000000, 8000000):
    side = 1000.0 * (M / 1e6) ** (1.0 / 3.0)
    m = rng.uniform(-side, side, (M, 3))
    q = m[rng.integers(0, M, NQ)] + rng.normal(0, 1.0, (NQ, 3))
    ms_ = t.Scan([0, 0, 0], [0, 0, 0], m); qs = t.Scan([0, 0, 0], [0, 0, 0], q)
    info = ms_.getSearchTree().info(); _ = qs.handle
    tms = []
    for r in range(12):
        res = t.Scan.getPtPairs(ms_, qs, max_dist_match2=625.0)
        ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms)); tms.append(ms.value)
    t.lib().tdtk_visit_counting(0, 1)
    t.Scan.getPtPairs(ms_, qs, max_dist_match2=625.0)
    c = (C.c_uint64 * 8)(); t.lib().tdtk_visit_counters(0, c); t.lib().tdtk_visit_counting(0, 0)
    print("model %8d pts (%5.1f MB tree, depth %2d): k_search %.4f ms  visits/query: %.2f nodes %.2f buckets %.2f points"
          % (M, (info["n_internal"] * 64 + M * 32) / 1e6, info["max_depth"], float(np.median(tms[2:])), c[0] / NQ, c[1] / NQ, c[2] / NQ))
    del ms_, qs
