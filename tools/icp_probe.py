"""GPU box: the bench's ICP loop (configs[1] shape) at a given size; prints wall / k_search / pair-sum time per iteration.
Kernel selection comes from the environment (TDTK_SEARCH_VARIANT, TDTK_REFILL_QPW, TDTK_REFILL_THRESH, TDTK_FUSE_SUMS),
read once per process -- run one process per configuration (tools/r2_sweep.sh).
usage: python tools/icp_probe.py [points] [steps] [warmup]"""
import importlib, os, sys, time
os.environ.setdefault("TDTK_KERNEL_TIMING", "1")   # the probes read the library's per-kernel event times
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 10
m, d, T = bench.make_icp_pair(n)
model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
model.getSearchTree(); _ = data.handle
mini = t.icp6D_QUAT(True)
t.icp6D(mini, 25.0, warm, quiet=True, epsilonICP=-1.0).match(model, data)
icp = t.icp6D(mini, 25.0, steps, quiet=True, epsilonICP=-1.0)
t0 = time.perf_counter(); it = icp.match(model, data); dt = time.perf_counter() - t0
L = icp.last
cfg = " ".join("%s=%s" % (k[5:], os.environ[k]) for k in sorted(os.environ) if k.startswith("TDTK_"))
print("n=%d [%s] wall %.4f ms/it  k_search %.4f  sums %.4f  -> %.3e NN/s  pairs %d rms %.10f poseerr %.2e"
      % (n, cfg, dt * 1e3 / (it + 1), L["nn_ms"] / (it + 1), L["sums_ms"] / (it + 1), n * (it + 1) / dt, L["pairs"], L["rms"],
         float(np.abs(data.get_transMat() - T).max())))
