"""GPU box: several host threads building search trees of 1M-point scans at the same time, each verified against the
host builder.  usage: python tools/tree_threads_probe.py [threads] [scans]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 16
raw = bench.make_graphslam_scans(ns, 1000000)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
try:
    t.prepare_scans(scans, trees=True, threads=thr)
    print("threads", thr, "ok;", [s.getSearchTree().verify() == [0, 0, 0, 0] for s in scans].count(False), "bad verifies")
except Exception as e:
    print("threads", thr, "FAILED", e)
