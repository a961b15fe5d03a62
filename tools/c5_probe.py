"""configs[4] regime, one pair: k_search of a cold whole-scan pass and of ICP iterations, 10M queries against a 10M-point tree,
under whatever TDTK_* knobs the environment holds (TDTK_LIB=lab for the lab switches).  One line of numbers per process:
  python tools/c5_probe.py [points] [reps] [icp iterations]"""
import ctypes as C, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
tdtk = importlib.import_module("3dtk_amd")
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
raw = bench.make_c5_scans(2, npts)
S = [tdtk.Scan(p, th, loc) for (p, th, loc) in raw]
for s in S:
    _ = s.handle
kd = S[0].getSearchTree()
L = tdtk.lib()
tm4 = (C.c_double * 4)()
tdtk.Scan.getPtPairs(S[0], S[1], 0, 0, 100.0)
ks, ss = [], []
L.tdtk_kernel_timing(1)
for _ in range(reps):
    r = tdtk.Scan.getPtPairs(S[0], S[1], 0, 0, 100.0)
    L.tdtk_last_timings(tm4); ks.append(tm4[0]); ss.append(tm4[1])
d = tdtk.Scan(raw[1][0], raw[1][1], raw[1][2]); _ = d.handle
icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 10.0, iters, quiet=True, epsilonICP=-1.0)
t0 = time.perf_counter(); it = icp.match(S[0], d); dt = time.perf_counter() - t0
L.tdtk_kernel_timing(0)
knobs = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("TDTK_"))
print("C5PROBE %-60s pass k_search %.4f ms (min %.4f) sums %.4f pairs %d | icp %d it: k_search %.4f ms/it, %.4f ms/it wall, rms %.9f pairs %d"
      % (knobs, float(np.mean(ks)), min(ks), float(np.mean(ss)), r["n"], it + 1, icp.last["nn_ms"] / (it + 1), dt * 1e3 / (it + 1), icp.last["rms"], icp.last["pairs"]))
