"""GPU box: 8 host threads building the tree of the SAME 1M-point cloud (separate Scan objects) at the same time."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
raw = bench.make_graphslam_scans(2, 1000000)
p, th, loc = raw[1]
scans = [t.Scan(p, th, loc.copy()) for _ in range(16)]
try:
    t.prepare_scans(scans, trees=True, threads=thr)
    print("same cloud x16, threads", thr, "ok;", [s.getSearchTree().verify() == [0, 0, 0, 0] for s in scans].count(False), "bad verifies")
except Exception as e:
    print("same cloud x16, threads", thr, "FAILED", e)
