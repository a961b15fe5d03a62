"""GPU box: does what ran earlier in the process change the speed of the graph-SLAM link passes?  The C4 workload is
timed (a) in a fresh process, (b) again after freeing it, (c) after a 1M-vs-1M ICP + tree builds + bandwidth
measurement have allocated and freed their buffers (what bench.py does before its graphslam_1gpu leg).
usage: python tools/gs_after_probe.py [prelude ...]   prelude in {icp, bw, normals, trees}"""
import gc, importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
ns, npts = 64, 1000000
raw = bench.make_graphslam_scans(ns, npts)

def run(tag):
    scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
    t.prepare_scans(scans, trees=True, threads=8)
    for _ in range(3):
        gs.graph_iteration_comm(1, t.Graph(ns, 500.0 ** 2, 20, scans), scans, 625.0, None)
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); gs.graph_iteration_comm(1, t.Graph(ns, 500.0 ** 2, 20, scans), scans, 625.0, None); ts.append(time.perf_counter() - t0)
    print("%-40s whole iteration min %.2f median %.2f ms" % (tag, min(ts) * 1e3, float(np.median(ts)) * 1e3), flush=True)
    del scans; gc.collect()

run("fresh process")
run("again (scans freed and re-created)")
for what in sys.argv[1:] or ["icp", "bw", "trees", "normals"]:
    if what == "icp":
        m, d, T = bench.make_icp_pair(1000000)
        model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
        model.getSearchTree(); _ = data.handle
        t.icp6D(t.icp6D_QUAT(True), 25.0, 50, quiet=True, epsilonICP=-1.0).match(model, data)
        del model, data; gc.collect()
    elif what == "bw":
        print("   measured bandwidth", bench.measured_bandwidth(0) if hasattr(bench, "measured_bandwidth") else None)
    elif what == "trees":
        m, d, T = bench.make_icp_pair(1000000)
        for _ in range(3):
            s = t.Scan([0, 0, 0], [0, 0, 0], m); s.getSearchTree(); del s
        gc.collect()
    elif what == "normals":
        m, d, T = bench.make_icp_pair(1000000)
        s = t.Scan([0, 0, 0], [0, 0, 0], m); _ = s.handle
        if hasattr(s, "calcNormals"): s.calcNormals(10)
        del s; gc.collect()
    run("after " + what)
