"""Scratch probe (GPU box): correctness vs oracle + first timings on 1M-vs-1M."""
import importlib, os, sys, time
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
from oracle import orc
import ctypes as C

M = int(os.environ.get("PROBE_M", 1000000))
rng = np.random.default_rng(42)
m = rng.uniform(-1000, 1000, (M, 3))
q = rng.uniform(-1000, 1000, (M, 3))
t0 = time.time(); kd = t.KDtree(m, 20); print("tree create s", time.time() - t0, kd.info())
t0 = time.time(); ot = orc.Tree(m, 20); print("oracle build s", time.time() - t0, ot.stats())
for md2 in (625.0, 1e18):
    t0 = time.time(); idx, d2 = kd.FindClosestBatch(q, md2); t1 = time.time() - t0
    ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms))
    nth = orc.lib().orc_max_threads()
    t0 = time.time(); oi, od2, cnt = ot.find_closest(q, md2, nth, True); t2 = time.time() - t0
    print("maxd2", md2, "idx equal", np.array_equal(idx, oi), "d2 equal", np.array_equal(d2, od2),
          "found", int((idx >= 0).sum()), "gpu wall s %.4f kernel ms %.3f -> %.3e NN/s" % (t1, ms.value, M / (ms.value * 1e-3)),
          "oracle %d thr s %.3f" % (nth, t2), "visits/q", np.array(cnt) / M)
    print("gpu counters", np.array(kd.count_visits(q[:200000], md2)) / 200000)
# ICP-shaped pass: resident sorted scan
s = t.Scan([0, 0, 0], [0, 0, 0], q)
ms_ = t.Scan([0, 0, 0], [0, 0, 0], m)
for rep in range(3):
    t0 = time.time(); r = t.Scan.getPtPairs(ms_, s, max_dist_match2=625.0); t1 = time.time() - t0
    ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms))
    print("scan_pairs n", r["n"], "wall s %.4f search kernel ms %.3f -> %.3e NN/s" % (t1, ms.value, M / (ms.value * 1e-3)))
ref = ot.get_pt_pairs(np.eye(4).reshape(16), q, maxdist2=625.0)
print("pairs n equal", r["n"] == ref["n"], "sum rel", abs(r["sum"] - ref["sum"]) / ref["sum"],
      "cm", np.abs(r["centroid_m"] - ref["centroid_m"] / ref["n"]).max())
