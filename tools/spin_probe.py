"""GPU box: does the runtime's host-wait policy change the ICP iteration time?  usage: python tools/spin_probe.py <flag>
flag: 0 auto, 1 spin, 2 yield, 4 blocking sync (hipSetDeviceFlags before the first GPU work)"""
import ctypes, importlib, os, sys, time
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flag = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hip = ctypes.CDLL("libamdhip64.so")
rc = hip.hipSetDeviceFlags(ctypes.c_uint(flag))
import bench
t = importlib.import_module("3dtk_amd")
m, d, T = bench.make_icp_pair(1000000)
model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
model.getSearchTree(); _ = data.handle
mini = t.icp6D_QUAT(True)
t.icp6D(mini, 25.0, 10, quiet=True, epsilonICP=-1.0).match(model, data)
icp = t.icp6D(mini, 25.0, 100, quiet=True, epsilonICP=-1.0)
t0 = time.perf_counter(); it = icp.match(model, data); dt = time.perf_counter() - t0
L = icp.last
print("hipSetDeviceFlags(%d) rc %d: wall %.4f ms/it  k_search %.4f  sums %.4f  outside %.4f" % (flag, rc, dt * 1e3 / (it + 1), L["nn_ms"] / (it + 1), L["sums_ms"] / (it + 1), dt * 1e3 / (it + 1) - (L["nn_ms"] + L["sums_ms"]) / (it + 1)))
