"""GPU box, lab library: how full are the trips of the persistent-lane search kernel?  Runs the bench's 1M-vs-1M ICP loop
(W + K iterations at the driver's arguments) with the instrumented instantiation and reads, beside the visit counters, the
number of trips the waves made through the node walk and through the bucket scan:
  fill(phase) = lane-visits of the phase / (64 x trips of the phase)
-- the share of a trip's 64 lane-slots that hold a lane with work in that phase.  What a per-wave pool of query states
(phase-sorted issue: a node step for the lanes at nodes, a bucket step for those at buckets) could at best gain is the empty
share of these trips; NEGATIVES.md has the reading."""
import ctypes as C, importlib, os, sys
os.environ["TDTK_LIB"] = "lab"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
t = importlib.import_module("3dtk_amd")
L = t.lib()
L.tdtk_lab_trip_counters.argtypes = [C.c_int, C.POINTER(C.c_uint64)]
n = 1000000
m, d, T = bench.make_icp_pair(n)
model = t.Scan([0, 0, 0], [0, 0, 0], m); data = t.Scan([0, 0, 0], [0, 0, 0], d)
mini = t.icp6D_QUAT(True)
t.icp6D(mini, 25.0, 5, quiet=True, epsilonICP=-1.0).match(model, data)
L.tdtk_visit_counting(0, 1)
icp = t.icp6D(mini, 25.0, 20, quiet=True, epsilonICP=-1.0)
icp.match(model, data)
c = (C.c_uint64 * 8)(); L.tdtk_visit_counters(0, c)
tr = (C.c_uint64 * 2)(); L.tdtk_lab_trip_counters(0, tr)
L.tdtk_visit_counting(0, 0)
nq = c[3]
print("queries %d | node visits per query %.2f, buckets per query %.2f, points per query %.2f" % (nq, c[0] / nq, c[1] / nq, c[2] / nq))
print("node-walk trips %d (%.1f per wave-slab of 256), lane fill %.3f | bucket trips %d, lane fill %.3f" %
      (tr[0], tr[0] / (nq / 256.0), c[0] / (64.0 * tr[0]), tr[1], c[1] / (64.0 * tr[1])))
print("all trips: %.3f of the lane-slots busy; if every trip were full: %.0f trips instead of %d (x %.2f)" %
      ((c[0] + c[1]) / (64.0 * (tr[0] + tr[1])), (c[0] + c[1]) / 64.0, tr[0] + tr[1], (c[0] + c[1]) / 64.0 / (tr[0] + tr[1])))
