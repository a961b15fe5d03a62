"""CPU (oracle as the counter -- test infrastructure): what bounds k_ann_normals' lane efficiency?  The kernel gives one
query to a lane, 64 consecutive leaf positions to a wave, and a wave's trip lasts as long as its slowest lane's walk.
Per query the oracle counts splitting nodes and leaf points visited (kd_search.cpp's walk with k = 10, eps = 1); from them:
   mean / max of the walk lengths within a wave  = the share of lane-slots a wave can fill at best
usage: python tools/ann_lane_model.py [points] [waves sampled]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
nw = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rng = np.random.default_rng(21)
for shape in ("uniform", "plane"):
    p = rng.uniform(-1000, 1000, (n, 3))
    if shape == "plane": p[:, 2] = 0.05 * p[:, 0] + rng.normal(0, 1.0, n)
    T = orc.AnnTree(p)
    _, _, leaf = T.structure()                     # the points in leaf order = the kernel's query order
    starts = rng.choice(n // 64 - 1, nw, replace=False) * 64
    eff_steps, eff_split, eff_leaf, ms, ml = [], [], [], [], []
    for s in starts:
        sp, lf = np.empty(64), np.empty(64)
        for j in range(64):
            _, _, v = T.ksearch(p[leaf[s + j]][None, :], 10, 1.0, want_visits=True)
            sp[j], lf[j] = v
        st = sp + lf
        eff_steps.append(st.mean() / st.max()); eff_split.append(sp.mean() / sp.max()); eff_leaf.append(lf.mean() / lf.max())
        ms.append(sp.mean()); ml.append(lf.mean())
    print("%s %d points, %d waves of 64 consecutive leaves: %.1f splitting nodes + %.1f leaf points per query; "
          "mean/max within a wave: walk %.2f (nodes %.2f, leaves %.2f)" % (shape, n, nw, np.mean(ms), np.mean(ml), np.mean(eff_steps), np.mean(eff_split), np.mean(eff_leaf)), flush=True)
