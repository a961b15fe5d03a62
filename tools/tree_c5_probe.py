"""GPU box: the tree build of a configs[4]-shaped scan (bench.make_c5_scans) and of the bench's 1M ICP model scan, as the bench
measures them (KDtree.from_scan on the resident scan, build_ms of a warm process)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
pos, th, p = bench.make_c5_scans(1, npts)[0]
s = t.Scan(pos, th, p, device=0); _ = s.handle
b = []
for _ in range(5):
    k = t.KDtree.from_scan(s.handle, s.n, 20, 0); b.append(k.info()["build_ms"]); ok = k.verify() if len(b) == 1 else None; del k
    if ok is not None: print("verify", ok)
print("c5-shaped scan of %d points: build_ms %s" % (npts, " ".join("%.2f" % x for x in b)), flush=True)
m, d, T = bench.make_icp_pair(1000000)
S = t.Scan([0, 0, 0], [0, 0, 0], m); _ = S.handle
b = []
for _ in range(6):
    k = t.KDtree.from_scan(S.handle, S.n, 20, 0); b.append(k.info()["build_ms"]); del k
print("bench model scan 1M: build_ms %s" % " ".join("%.2f" % x for x in b), flush=True)
