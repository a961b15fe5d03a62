"""GPU box, lab library: k_search_refill2 (two queries per lane, TDTK_TWO_PER_LANE=2|3) against k_search_refill on the bench's
1M-vs-1M ICP loop: same indices / pairs / rms every iteration, kernel time, visit counters and trip counts."""
import ctypes as C, importlib, os, sys
os.environ["TDTK_LIB"] = "lab"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
L = t.lib()
L.tdtk_lab_trip_counters.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
m, d, T = bench.make_icp_pair(n)
mini = t.icp6D_QUAT(True)


def run(mode, count=False, one=False):
    if mode: os.environ["TDTK_TWO_PER_LANE"] = str(mode)
    else: os.environ.pop("TDTK_TWO_PER_LANE", None)
    if one: os.environ["TDTK_TWO_ONE"] = "1"      # a trip of the node walk serves ONE of the lane's two queries (round 4)
    else: os.environ.pop("TDTK_TWO_ONE", None)
    model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
    t.icp6D(mini, 25.0, 5, quiet=True, epsilonICP=-1.0).match(model, data)
    if count: L.tdtk_visit_counting(0, 1)
    L.tdtk_kernel_timing(1)
    icp = t.icp6D(mini, 25.0, 20, quiet=True, epsilonICP=-1.0)
    icp.match(model, data)
    L.tdtk_kernel_timing(0)
    out = dict(rms=icp.last["rms"], pairs=icp.last["pairs"], nn_ms=icp.last.get("nn_ms"), pose=np.array(data.transMat))
    if count:
        c = (C.c_uint64 * 8)(); L.tdtk_visit_counters(0, c)
        tr = (C.c_uint64 * 2)(); L.tdtk_lab_trip_counters(0, tr)
        L.tdtk_visit_counting(0, 0)
        out["visits"] = (c[0], c[1], c[2], c[3]); out["trips"] = (tr[0], tr[1])
    return out


base = run(0)
for mode, one in ((3, False), (2, False), (3, True), (2, True), (3, True)):
    r = run(mode, one=one)
    print("one visit per trip" if one else "both slots per trip", end=": ")
    same = (r["rms"] == base["rms"] and r["pairs"] == base["pairs"] and np.array_equal(r["pose"], base["pose"]))
    print("two per lane, %d waves/SIMD: nn %.4f ms (one per lane %.4f) | rms/pairs/pose identical: %s; pairs %d vs %d, max|dpose| %.3e" %
          (mode, r["nn_ms"], base["nn_ms"], same, r["pairs"], base["pairs"], float(np.abs(r["pose"] - base["pose"]).max())))
b = run(0, True); r = run(3, True); r1 = run(3, True, one=True)
print("visits one per lane", b["visits"], "trips", b["trips"])
print("visits two per lane", r["visits"], "trips", r["trips"])
print("visits two per lane, one visit per trip", r1["visits"], "trips", r1["trips"])
