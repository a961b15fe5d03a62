"""GPU box (1 GPU): where one graph-SLAM round of bench.py goes -- Graph construction, marshalling, the library call
(tdtk_graph_iteration: link passes + solve + pose update), the Python bookkeeping behind it."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
from importlib import import_module
ns, npts = 64, 1000000
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
g0 = t.Graph(ns, 500.0 ** 2, 20, scans)
t.prepare_scans(scans, trees=True, threads=8)
lib, check, dptr, iptr = capi.lib(), capi.check, capi.dptr, capi.iptr
for rep in range(8):
    T = [time.perf_counter()]
    gr = t.Graph(ns, 500.0 ** 2, 20, scans); T.append(time.perf_counter())
    nscans, nlinks = gr.getNrScans(), gr.getNrLinks()
    sc = scans[:nscans]
    frm, to = gs._link_arrays(gr)
    mine_a = np.arange(nlinks, dtype=np.int32)
    nl = len(mine_a)
    fl, tl = frm[mine_a].tolist(), to[mine_a].tolist()
    first = (C.c_void_p * max(1, nl))(*[sc[a].getSearchTree()._h for a in fl])
    second = (C.c_void_p * max(1, nl))(*[sc[b].handle for b in tl])
    tm = np.array([s.transMat for s in sc], dtype=np.float64).reshape(nscans, 16)
    da = np.array([s.dalignxf for s in sc], dtype=np.float64).reshape(nscans, 16)
    rp = np.array([s.rPos for s in sc], dtype=np.float64).reshape(nscans, 3)
    rt = np.array([s.rPosTheta for s in sc], dtype=np.float64).reshape(nscans, 3)
    dal = np.ascontiguousarray(da[fl])
    hs = (C.c_void_p * nscans)(*[s._h for s in sc])
    xf = np.zeros((nscans, 32)); ret = C.c_double(0.0)
    T.append(time.perf_counter())
    check(lib.tdtk_graph_iteration(1, None, nlinks, iptr(frm), iptr(to), nl, iptr(mine_a), first, dptr(dal), second, 625.0, nscans,
                                   dptr(tm), dptr(da), dptr(rp), dptr(rt), hs, None, dptr(xf), C.byref(ret)))
    T.append(time.perf_counter())
    for i in range(1, nscans):
        s = sc[i]
        s.transMat, s.dalignxf, s.rPos, s.rPosTheta = tm[i], da[i], rp[i], rt[i]
        s._addFrames("LUM", 2 if i == nscans - 1 else 1)
    T.append(time.perf_counter())
    ms = C.c_double(0.0); lib.tdtk_kernel_timing(1); lib.tdtk_last_kernel_ms(C.byref(ms))
    names = ["Graph()", "marshal", "tdtk_graph_iteration", "bookkeeping"]
    print(" | ".join("%s %.3f" % (nm, (T[i + 1] - T[i]) * 1e3) for i, nm in enumerate(names)), "| total %.3f ms | last search launch %.3f ms" % ((T[-1] - T[0]) * 1e3, ms.value))
