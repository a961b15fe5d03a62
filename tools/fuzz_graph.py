"""GPU box: randomized differential run of the graph-SLAM step -- random scan sets whose sizes mix the three search-kernel
families (lane groups < 96K, one query per lane, persistent lanes >= 256K), random links (scans read by several links of a
launch, scans read by none, links in both directions, links ending at the fixed scan), a few lum6DEuler rounds:
  * scan moves queued and carried out by the link launches (default) against every scan moved every round
    (TDTK_LAZY_MOVES=0): `ret`, poses and the final points bit for bit;
  * the first round against the oracle's lum_iteration (numpy / C restatement): ret and poses to 1e-7;
  * a MetaScan tree over some of the moved scans (device to device) against the oracle's tree over the concatenated points.
usage: python tools/fuzz_graph.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module  # noqa: E402

tdtk = import_module("3dtk_amd")
gs = import_module("3dtk_amd.graphslam")
from oracle import icp_oracle as io  # noqa: E402
from oracle import orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
t0 = time.time()
runs = fails = 0
while time.time() - t0 < a.seconds:
    runs += 1
    ns = int(rng.integers(3, 7))
    nworld = int(rng.choice([20000, 120000, 320000]))
    ext = float(rng.choice([60.0, 400.0]))
    world = rng.uniform(-ext, ext, (nworld, 3)); world[:, 2] *= float(rng.choice([0.2, 1.0]))
    sizes = [int(rng.choice([3000, 40000, 110000, 280000])) for _ in range(ns)]
    sizes = [min(s, nworld) for s in sizes]
    poses, clouds = [], []
    for k in range(ns):
        pos = rng.normal(0, 2.0, 3) * (k > 0); th = rng.normal(0, 0.004, 3) * (k > 0)
        T = tdtk.EulerToMatrix4(pos, th); Ti = tdtk.M4inv(T)
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        sub = world[rng.permutation(nworld)[:sizes[k]]]
        loc = sub @ R.T + Ti[12:15] + rng.normal(0, 0.03, sub.shape)
        est = (pos + rng.normal(0, 0.2, 3) * (k > 0), th + rng.normal(0, 0.002, 3) * (k > 0))
        poses.append(est); clouds.append(np.ascontiguousarray(loc))
    links = [(k, k + 1) for k in range(ns - 1)]
    for _ in range(int(rng.integers(0, 2 * ns))):
        i, j = int(rng.integers(0, ns)), int(rng.integers(0, ns))
        if i != j and (i, j) not in links:
            links.append((i, j))
    md2 = float(rng.choice([25.0, 100.0, 400.0]))
    rounds = int(rng.integers(1, 4))

    def run(lazy):
        if lazy: os.environ.pop("TDTK_LAZY_MOVES", None)
        else: os.environ["TDTK_LAZY_MOVES"] = "0"
        scans = [tdtk.Scan(p[0], p[1], c) for p, c in zip(poses, clouds)]
        tdtk.prepare_scans(scans, trees=True, threads=2)
        gr = tdtk.Graph(ns, links=links)
        rets = [gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, md2, None) for _ in range(rounds)]
        out = (rets, np.stack([s.transMat for s in scans]), [s.get_xyz_reduced() for s in scans])
        for s in scans: s.release()
        return out
    try:
        e, l = run(False), run(True)
        ok = e[0] == l[0] and np.array_equal(e[1], l[1]) and all(np.array_equal(x, y) for x, y in zip(e[2], l[2]))
        O = [io.OScan(p[0], p[1], c) for p, c in zip(poses, clouds)]
        oret = io.lum_iteration(links, O, md2)[0]
        S = [tdtk.Scan(p[0], p[1], c) for p, c in zip(poses, clouds)]
        tdtk.prepare_scans(S, trees=True, threads=2)
        ret1 = gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(ns, links=links), S, md2, None)
        ok_o = abs(ret1 - oret) <= 1e-7 * max(1.0, abs(oret)) and all(
            np.abs(s.get_rPos() - o.rPos).max() < 1e-6 and np.abs(s.get_rPosTheta() - o.rPosTheta).max() < 1e-8 for s, o in zip(S, O))
        # a MetaScan tree over some of the (just moved) scans, built device to device (kdMeta.cc:34-134): neighbours in
        # concatenation order against the oracle's tree over the concatenated points
        members = [S[i] for i in sorted(rng.permutation(ns)[: int(rng.integers(2, ns + 1))])]
        cat = np.concatenate([m.get_xyz_reduced() for m in members])
        mk = tdtk.MetaScan(members).getSearchTree()
        q = np.concatenate([cat[rng.integers(0, len(cat), 2000)] + rng.normal(0, 0.05, (2000, 3)), rng.uniform(cat.min(), cat.max(), (200, 3))])
        gi, gd = mk.FindClosestBatch(q, md2)
        oi, od = orc.Tree(cat, int(members[0].bucketSize)).find_closest(q, md2)
        ok_m = np.array_equal(gi, oi) and np.array_equal(gd, od) and mk.verify() == [0, 0, 0, 0]
        ok_o = ok_o and ok_m
        for s in S: s.release()
    except Exception as ex:   # noqa: BLE001
        ok = ok_o = False
        print("EXCEPTION run %d: %r" % (runs, ex), flush=True)
    if not (ok and ok_o):
        fails += 1
        print("GRAPH MISMATCH run %d: lazy==eager %s, first round == oracle %s | ns %d sizes %s links %s md2 %g rounds %d" % (
            runs, ok, ok_o, ns, sizes, links, md2, rounds), flush=True)
print("fuzz_graph: %d graphs, %d mismatches, %.0f s, seed %d" % (runs, fails, time.time() - t0, a.seed))
sys.exit(1 if fails else 0)
