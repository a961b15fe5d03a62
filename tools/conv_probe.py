"""GPU box: why does a converged ICP iteration's k_search take longer than tools/kbench.py icp?  Same pair as bench.py."""
import importlib, os, sys, ctypes as C
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
n = 1000000
m, d, T = bench.make_icp_pair(n)
model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
model.getSearchTree(); _ = data.handle
def kms():
    ms = C.c_double(); t.lib().tdtk_last_kernel_ms(C.byref(ms)); return ms.value
def pairs_ms(a, b, reps=8):
    v = []
    for _ in range(reps):
        r = t.Scan.getPtPairs(a, b, max_dist_match2=625.0); v.append(kms())
    return np.median(v[2:]), r["n"]
print("before ICP (initial pose): getPtPairs k_search %.4f ms, pairs %d" % pairs_ms(model, data))
icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 30, quiet=True, epsilonICP=-1.0)
icp.match(model, data)
print("ICP loop: nn_ms per iteration %.4f (30 iterations), last pairs %d" % (icp.last["nn_ms"] / 30, icp.last["pairs"]))
icp.match(model, data)
print("ICP loop again (converged from the start): nn_ms per iteration %.4f" % (icp.last["nn_ms"] / 30))
print("after ICP (converged): getPtPairs k_search %.4f ms, pairs %d" % pairs_ms(model, data))
# the same points uploaded anew in the model frame (Morton order computed in the model frame)
cur = data.get_xyz_reduced()
fresh = t.Scan([0, 0, 0], [0, 0, 0], cur); _ = fresh.handle
print("converged points re-uploaded (ordered in the model frame): getPtPairs k_search %.4f ms, pairs %d" % pairs_ms(model, fresh))
