"""GPU box: time the per-scan preparation (tree create, scan create) on 1M points."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = importlib.import_module("3dtk_amd")
capi = importlib.import_module("3dtk_amd._capi")
rng = np.random.default_rng(0)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m = rng.uniform(-1000, 1000, (M, 3))
L = capi.lib()
for rep in range(4):
    h = C.c_void_p(); t0 = time.perf_counter()
    capi.check(L.tdtk_scan_create(capi.dptr(m), None, M, 0, C.byref(h))); t1 = time.perf_counter()
    L.tdtk_scan_destroy(h)
    t2 = time.perf_counter(); kd = t.KDtree(m, 20); t3 = time.perf_counter()
    print("scan_create %.2f ms | tree_create %.2f ms %s" % ((t1 - t0) * 1e3, (t3 - t2) * 1e3, {k: round(v, 2) for k, v in kd.info().items() if k.endswith("_ms")}))
    del kd
