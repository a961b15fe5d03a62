"""GPU box: one lum6DEuler round of the bench's graph (64 x 1M, 84 links): wall time of graph_iteration_comm, of the C call
inside it (tdtk_graph_iteration) and of the link launch (HIP events) -- what of a round's 'rest' is the Python mirror."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam"); capi = importlib.import_module("3dtk_amd._capi")
raw = bench.make_graphslam_scans(64, 1000000)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
t.prepare_scans(scans, trees=True, threads=2)
gr = t.Graph(64, 500.0 ** 2, 20, scans)
L = capi.lib()
orig = L.tdtk_graph_iteration
acc = {"c": 0.0}
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); acc["c"] += time.perf_counter() - t0; return r
for _ in range(3): gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 625.0, None)
L.tdtk_kernel_timing(1)
import types
gs_lib = capi.lib
w = Wrap()
class LibProxy:
    def __getattr__(self, k): return w if k == "tdtk_graph_iteration" else getattr(L, k)
capi_lib_orig = capi.lib
import importlib as _il
mod = _il.import_module("3dtk_amd._capi")
mod.lib = lambda: LibProxy()
N = 10; tot = 0.0; kms = 0.0
for _ in range(N):
    t0 = time.perf_counter(); gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 625.0, None); tot += time.perf_counter() - t0
    ms = C.c_double(0); L.tdtk_last_kernel_ms(C.byref(ms)); kms += ms.value
mod.lib = capi_lib_orig
print("per round: python+C %.3f ms | C call %.3f ms | link launch %.3f ms | python mirror %.3f ms | C outside the launch %.3f ms" % (
    tot / N * 1e3, acc["c"] / N * 1e3, kms / N, (tot - acc["c"]) / N * 1e3, acc["c"] / N * 1e3 - kms / N))
