"""GPU box: k_search of ONE cold whole-scan pass, 1M-vs-1M uniform pair, by the pair-sum block asked for: base block (the sums
inside the search launch, FUSE 3) and base + APX block (k_accum behind the search: the FUSE 0 instantiation) -- HIP events."""
import ctypes as C, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
L = t.lib()
m, d, T = bench.make_icp_pair(1000000)
a = t.Scan([0, 0, 0], [0, 0, 0], m); b = t.Scan([0, 0, 0], [0, 0, 0], d)
a.getSearchTree(); _ = b.handle
tm = (C.c_double * 4)()
L.tdtk_kernel_timing(1)
for want, name in ((0, "base block (FUSE 3)"), (1, "base + APX (FUSE 0 + k_accum)")):
    ks = []
    for _ in range(8):
        t.Scan.getPtPairs(a, b, 0, 0, 625.0, want=want)
        L.tdtk_last_timings(tm); ks.append(tm[0])
    print("1M cold pass, %-32s k_search %.4f ms (min of 8: %.4f)" % (name, float(np.median(ks)), min(ks)))
L.tdtk_kernel_timing(0)
