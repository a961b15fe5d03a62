"""GPU box: k_search time and pair statistics per ICP iteration on the bench pair."""
import importlib, os, sys, ctypes as C
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
m, d, T = bench.make_icp_pair(1000000)
model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
model.getSearchTree(); _ = data.handle
icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 1, quiet=True, epsilonICP=-1.0)
out = []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 120):
    icp.match(model, data)
    ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms))
    out.append((it, ms.value, icp.last["pairs"], icp.last["rms"]))
for r in out[::10] + out[-3:]:
    print("iter %3d  k_search %.4f ms  pairs %7d  rms %.4f" % r)
