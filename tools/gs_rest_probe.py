"""GPU box: break down the non-link part of one native LUM iteration (64 x 1M)."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
import torch
import torch.distributed as dist
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam"); sl = importlib.import_module("3dtk_amd.slam6d")
capi = importlib.import_module("3dtk_amd._capi")
from importlib import import_module
ns, npts = 64, 1000000
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
dev = torch.device("cuda", 0)
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
g = t.Graph(ns, 500.0 ** 2, 20, scans)
for i in range(g.getNrLinks()):
    scans[g.getLink(i, 0)].getSearchTree(); _ = scans[g.getLink(i, 1)].handle
L = capi.lib(); dptr = capi.dptr; check = capi.check
for rep in range(4):
    T = [time.perf_counter()]
    gr = t.Graph(ns, 500.0 ** 2, 20, scans); T.append(time.perf_counter())
    n = ns - 1
    mine = gs.shard_links(gr, 0, 1); nl = len(mine)
    G = np.zeros((6 * n, 6 * n)); B = np.zeros(6 * n)
    first = (C.c_void_p * nl)(*[scans[gr.getLink(i, 0)].getSearchTree()._h for i in mine])
    second = (C.c_void_p * nl)(*[scans[gr.getLink(i, 1)].handle for i in mine])
    dal = np.ascontiguousarray(np.stack([scans[gr.getLink(i, 0)].dalignxf for i in mine]))
    Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
    T.append(time.perf_counter())
    check(L.tdtk_lum_links(nl, first, dptr(dal), second, 625.0, dptr(Cm), dptr(CD), m, dptr(ss))); T.append(time.perf_counter())
    for k, i in enumerate(mine):
        a, b = gr.getLink(i, 0) - 1, gr.getLink(i, 1) - 1
        Cab = Cm[k].reshape(6, 6)
        if a >= 0:
            B[a * 6:a * 6 + 6] += CD[k]; G[a * 6:a * 6 + 6, a * 6:a * 6 + 6] += Cab
        if b >= 0:
            B[b * 6:b * 6 + 6] -= CD[k]; G[b * 6:b * 6 + 6, b * 6:b * 6 + 6] += Cab
        if a >= 0 and b >= 0:
            G[a * 6:a * 6 + 6, b * 6:b * 6 + 6] -= Cab; G[b * 6:b * 6 + 6, a * 6:a * 6 + 6] -= Cab
    T.append(time.perf_counter())
    G, B = gs.allreduce_GB(G, B, None, dev); T.append(time.perf_counter())
    X = sl.solveSparseCholesky(G, B); T.append(time.perf_counter())
    tm = np.ascontiguousarray(np.stack([s.transMat for s in scans])); da = np.ascontiguousarray(np.stack([s.dalignxf for s in scans]))
    rp = np.ascontiguousarray(np.stack([s.rPos for s in scans])); rt = np.ascontiguousarray(np.stack([s.rPosTheta for s in scans]))
    hs = (C.c_void_p * ns)(*[s._h for s in scans]); xf = np.zeros((ns, 32)); ret = C.c_double(0.0)
    T.append(time.perf_counter())
    check(L.tdtk_lum_update_poses(ns, dptr(X), dptr(tm), dptr(da), dptr(rp), dptr(rt), hs, dptr(xf), C.byref(ret))); T.append(time.perf_counter())
    for i in range(1, ns):
        s = scans[i]
        s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i].copy(), da[i].copy(), rp[i].copy(), rt[i].copy()
        s.frames.append((s.transMat.copy(), "LUM"))
    T.append(time.perf_counter())
    names = ["graph", "marshal", "lum_links", "fill G/B", "allreduce(H2D+nccl+D2H)", "solve", "marshal poses", "update_poses(+transforms)", "python bookkeeping"]
    print(" | ".join("%s %.3f" % (nm, (T[i + 1] - T[i]) * 1e3) for i, nm in enumerate(names)), "| total %.2f" % ((T[-1] - T[0]) * 1e3))
dist.destroy_process_group()
