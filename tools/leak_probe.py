import importlib, os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import bench
import torch
t = importlib.import_module("3dtk_amd")
raw = bench.make_graphslam_scans(4, 300000, seed=3)
def free_mb():
    f, tot = torch.cuda.mem_get_info(0); return (tot - f) / 2**20
for rep in range(6):
    scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
    t.icp6D(t.icp6D_QUAT(True), 25.0, 5, quiet=True).doICP(scans, prefetch=True)
    del scans
    print("rep", rep, "device memory in use %.0f MB" % free_mb())
print("done")
