"""GPU box (1 GPU): rank 0 of 8's share of the 64 x 1M graph (11 links) as tdtk_graph_iteration runs it -- link passes with the
previous round's scan moves queued --, a dozen rounds; for rocprofv3 (tools/r4_profile.sh) and as a plain timing."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
ns, npts = 64, 1000000
world, rank = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 0
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
g = t.Graph(ns, 500.0 ** 2, 20, scans)
idx = gs.shard_links(g, rank, world, scans)
t.prepare_scans([scans[k] for k in sorted({g.getLink(i, 0) for i in idx})], trees=True, threads=8)
t.prepare_scans([scans[k] for k in sorted({g.getLink(i, 1) for i in idx})], trees=False, threads=8)
L = capi.lib()
nl = len(idx)
first = (C.c_void_p * nl)(*[scans[g.getLink(i, 0)].getSearchTree()._h for i in idx])
second = (C.c_void_p * nl)(*[scans[g.getLink(i, 1)].handle for i in idx])
moved = sorted({g.getLink(i, 1) for i in idx})
hs = (C.c_void_p * len(moved))(*[scans[k].handle for k in moved])
wig = np.ascontiguousarray(np.tile(t.EulerToMatrix4([1e-4, -1e-4, 1e-4], [1e-7, -1e-7, 1e-7]), (len(moved), 1)))
wig_inv = np.ascontiguousarray(np.stack([t.M4inv(m) for m in wig]))
dal = np.ascontiguousarray(np.stack([scans[g.getLink(i, 0)].dalignxf for i in idx]))
Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
ts = []
for rep in range(13):
    capi.check(L.tdtk_scans_transform2(len(moved), hs, capi.dptr(wig), capi.dptr(wig_inv)))
    t0 = time.perf_counter()
    capi.check(L.tdtk_lum_links(nl, first, capi.dptr(dal), second, 625.0, capi.dptr(Cm), capi.dptr(CD), m, capi.dptr(ss)))
    ts.append((time.perf_counter() - t0) * 1e3)
print('{"share": "rank %d of %d", "links": %d, "ms_per_round_min": %.4f, "ms_per_round_median": %.4f}' % (rank, world, nl, min(ts[3:]), sorted(ts[3:])[5]))
