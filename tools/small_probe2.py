"""GPU box: does the per-iteration cost of the small-scan ICP loop depend on what the GPU did just before (clock state)?"""
import importlib, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "dat_scans.npz"))
def clocks(tag):
    try:
        o = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        print(tag, " | ".join(l.strip() for l in o.splitlines() if "sclk" in l or "mclk" in l or "fclk" in l)[:400])
    except Exception as e:
        print("rocm-smi:", e)
def small(tag, reps=3):
    for rep in range(reps):
        S = [t.Scan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(2)]
        S[0].getSearchTree(); _ = S[1].handle
        S[1].mergeCoordinatesWithRoboterPosition(S[0])
        icp = t.icp6D(t.icp6D_QUAT(True), 25.0, 50, quiet=True, epsilonICP=1e-5)
        t0 = time.perf_counter(); it = icp.match(S[0], S[1]); dt = time.perf_counter() - t0
        print("%s dat pair: %d iterations, wall %.3f ms -> %.1f us / iteration" % (tag, it + 1, dt * 1e3, dt * 1e6 / (it + 1)))
clocks("start")
small("cold ")
m, d, T = bench.make_icp_pair(1000000)
model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
model.getSearchTree(); _ = data.handle
t0 = time.perf_counter()
t.icp6D(t.icp6D_QUAT(True), 25.0, 3000, quiet=True, epsilonICP=-1.0).match(model, data)
print("heavy work: %.2f s" % (time.perf_counter() - t0))
small("after heavy work")
clocks("end")
# many repetitions back to back: does the loop itself warm the clocks up?
small("again", 10)
