"""GPU box: build the search tree of every scan of the C4 workload one by one (device builder) and verify it against the
host builder.  usage: python tools/gs_tree_probe.py [nscans] [npts]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
ns = int(sys.argv[1]) if len(sys.argv) > 1 else 64
npts = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
raw = bench.make_graphslam_scans(ns, npts)
bad = 0
for k, (p, th, loc) in enumerate(raw):
    s = t.Scan(p, th, loc)
    try:
        t0 = time.perf_counter(); tr = s.getSearchTree(); dt = (time.perf_counter() - t0) * 1e3
        v = tr.verify()
        if v != [0, 0, 0, 0]: bad += 1
        print("scan %2d: build %.2f ms verify %s" % (k, tr.info()["build_ms"], v), flush=True)
    except Exception as e:
        bad += 1
        print("scan %2d: FAILED %s" % (k, e), flush=True)
        np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "bad_scan_%d.npy" % k), np.asarray(loc)[:0])
    del s
print("BAD:", bad)
