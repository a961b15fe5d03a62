"""GPU box: a saved fuzz case (tools/fuzz_parity.py --trace -> gpurun_out/fuzz_last_case.npz), one step at a time:
device tree against the host twin (tdtk_tree_verify) and the search against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
tdtk = import_module("3dtk_amd")
from oracle import orc
d = np.load(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fuzz_last_case.npz")
p, q, bucket, md2 = d["p"], d["q"], int(d["bucket"]), float(d["md2"])
T = orc.Tree(p, bucket)
kd = tdtk.KDtree(p, bucket)
v = kd.verify()
oi, od = T.find_closest(q, md2)
gi, gd = kd.FindClosestBatch(q, md2)
tag = " ".join("%s=%s" % (k, os.environ[k]) for k in sorted(os.environ) if k.startswith("TDTK_"))
print("[%s] verify %s | search idx equal %s, d2 equal %s, differing queries %d | respeculated %d" % (
    tag, v, np.array_equal(gi, oi), np.array_equal(gd, od), int((gi != oi).sum()), tdtk.lib().tdtk_build_respeculated()), flush=True)
