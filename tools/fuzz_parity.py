"""GPU box: randomized differential run of the HIP path against the CPU oracle -- many small clouds of awkward shapes
(duplicates, lattices, collinear / coplanar points, huge dynamic range, tiny sizes) through calcNormals (k-NN lists
and normals bit for bit) and through the search tree (FindClosest indices / distances bit for bit).
usage: python tools/fuzz_parity.py [--seconds 120] [--seed 0]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module  # noqa: E402

tdtk = import_module("3dtk_amd")
from oracle import orc  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=120.0)
ap.add_argument("--seed", type=int, default=0)
ap.add_argument("--big", action="store_true", help="clouds of 30K..400K points (fewer rounds)")
ap.add_argument("--trace", action="store_true", help="print every run's stages as they start (to find a crash)")
a = ap.parse_args()
rng = np.random.default_rng(a.seed)


def cloud():
    n = int(rng.choice([1, 2, 3, 7, 10, 11, 33, 64, 65, 66, 128, 129, 300, 1000, 4097, 20000]))
    if a.big:
        n = int(rng.choice([30000, 65537, 131072, 200001, 400000]))
    kind = rng.integers(0, 9)
    if kind == 0:
        p = rng.uniform(-100, 100, (n, 3))
    elif kind == 1:
        p = rng.integers(-4, 5, (n, 3)).astype(float)                       # lattice, many duplicates
    elif kind == 2:
        p = rng.uniform(-100, 100, (n, 3)); p[:, rng.integers(0, 3)] = rng.choice([0.0, 1.5, -7.25])   # coplanar
    elif kind == 3:
        p = np.outer(rng.uniform(-50, 50, n), rng.normal(size=3))           # collinear
    elif kind == 4:
        p = rng.normal(0, 1, (n, 3)) * (10.0 ** rng.integers(-6, 7, (n, 1)))   # huge dynamic range
    elif kind == 5:
        b = rng.uniform(-10, 10, (max(1, n // 3), 3)); p = b[rng.integers(0, len(b), n)]   # every point repeated
    elif kind == 6:
        p = np.round(rng.uniform(-20, 20, (n, 3)), 1)                       # one decimal: ties on every axis
    elif kind == 7:
        p = np.concatenate([rng.normal(c, 1e-3, (n // 4 + 1, 3)) for c in rng.uniform(-50, 50, (4, 3))])[:n]
    else:
        p = np.cumsum(rng.exponential(1.0, (n, 3)) ** 3, axis=0)            # skewed: deep sliding splits
    return np.ascontiguousarray(p[rng.permutation(len(p))], dtype=np.float64)


t0 = time.time()
runs = fails = 0
while time.time() - t0 < a.seconds:
    p = cloud()
    n = len(p)
    runs += 1
    tr = (lambda what: print("run %d n=%d: %s" % (runs, n, what), flush=True)) if a.trace else (lambda what: None)
    tr("normals")
    # normals
    k = int(min(n, rng.choice([1, 3, 10, 10, 10, 16, 32])))
    eps = float(rng.choice([0.0, 0.5, 1.0, 1.0, 3.0]))
    rp = rng.uniform(-10, 10, 3)
    want, wk = orc.normals_apx_knn(p, k, rp, eps, want_knn=True)
    got, gk = tdtk.calculateNormalsApxKNN(p, k, rp, eps, want_knn=True)
    if not (np.array_equal(gk, wk) and np.array_equal(got, want, equal_nan=True)):
        fails += 1
        np.save("gpurun_out/fuzz_fail_normals_%d.npy" % runs, p)
        print("NORMALS MISMATCH run %d n=%d k=%d eps=%g knn_equal=%s" % (runs, n, k, eps, np.array_equal(gk, wk)), flush=True)
    tr("search")
    # search tree
    bucket = int(rng.choice([1, 2, 5, 20, 20]))
    q = np.concatenate([p[rng.integers(0, n, min(n, 500))] + rng.normal(0, rng.choice([0.0, 1e-3, 1.0]), (min(n, 500), 3)),
                        rng.uniform(p.min() - 1, p.max() + 1, (50, 3))])
    md2 = float(rng.choice([1e-6, 0.25, 25.0, 1e18]))
    if a.trace:      # the case about to run, kept for whoever has to look at a crash
        np.savez("gpurun_out/fuzz_last_case.npz", p=p, q=q, bucket=bucket, md2=md2, run=runs)
    kd, T = tdtk.KDtree(p, bucket), orc.Tree(p, bucket)
    gi, gd = kd.FindClosestBatch(q, md2)
    oi, od = T.find_closest(q, md2)
    if not (np.array_equal(gi, oi) and np.array_equal(gd, od)) or kd.verify() != [0, 0, 0, 0]:
        fails += 1
        np.save("gpurun_out/fuzz_fail_search_%d.npy" % runs, p)
        print("SEARCH MISMATCH run %d n=%d bucket=%d md2=%g" % (runs, n, bucket, md2), flush=True)
    tr("pairs bucket=%d md2=%g" % (bucket, md2))
    # SearchTree::getPtPairs through the C ABI: random pose, the three pairing modes, a query sub-range
    mode = int(rng.integers(0, 3))
    A = tdtk.EulerToMatrix4(rng.uniform(-1, 1, 3), rng.uniform(-0.05, 0.05, 3))
    nq = len(q)
    nrm = rng.normal(size=(nq, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    lo = int(rng.integers(0, nq // 2 + 1)); hi = int(rng.integers(lo, nq + 1))
    md2p = float(rng.choice([0.25, 25.0, 1e6]))
    r = kd.getPtPairs(A, q, nrm, lo, hi, max_dist_match2=md2p, pairing_mode=mode)
    o = T.get_pt_pairs(A, q, nrm, lo, hi, mode, md2p)
    ok = r["n"] == o["n"] and np.array_equal(r["idx"], o["idx"])
    if ok and o["n"]:
        ok = np.array_equal(r["p1"], o["p1"][:o["n"]]) and np.array_equal(r["p2"], o["p2"][:o["n"]]) and \
            abs(r["sum"] - o["sum"]) <= 1e-9 * max(1e-300, abs(o["sum"]))
    if not ok:
        fails += 1
        np.save("gpurun_out/fuzz_fail_pairs_%d.npy" % runs, p)
        print("PAIRS MISMATCH run %d n=%d mode=%d range=[%d,%d) md2=%g n=%s/%s" % (runs, n, mode, lo, hi, md2p, r["n"], o["n"]), flush=True)
    tr("octree")
    # octree reduction (-r): same cells, same centres, same order
    vox = float(rng.choice([0.3, 2.0, 15.0, 1e5]))
    ext = float(np.abs(p).max()) + 1.0
    if ext / vox < 2.0 ** 20:                      # the 21-level key limit of tdtk_reduce_octree
        gr_, or_ = tdtk.calcReducedPoints(p, vox), orc.octree_center(p, vox)
        if not (gr_.shape == or_.shape and np.array_equal(gr_, or_)):
            fails += 1
            np.save("gpurun_out/fuzz_fail_octree_%d.npy" % runs, p)
            print("OCTREE MISMATCH run %d n=%d voxel=%g" % (runs, n, vox), flush=True)
        # the random modes (-O nrpts): the order of the points inside a leaf, the C library's rand() seeded alike
        if runs % 3 == 0:
            nrp = int(rng.choice([1, 2, 5])); sd = int(rng.integers(1, 1 << 30))
            gq, oq = tdtk.calcReducedPoints(p, vox, nrpts=nrp, seed=sd), orc.octree_random(p, vox, nrp, seed=sd)
            if not (gq.shape == oq.shape and np.array_equal(gq, oq)):
                fails += 1
                np.save("gpurun_out/fuzz_fail_octree_random_%d.npy" % runs, p)
                print("OCTREE RANDOM MISMATCH run %d n=%d voxel=%g nrpts=%d seed=%d" % (runs, n, vox, nrp, sd), flush=True)
    # one in ten: a short ICP (well-conditioned cloud, random minimizer) against the oracle loop
    if runs % 10 == 0:
        tr("icp")
        from oracle import icp_oracle as io
        nm = int(rng.choice([200, 1500, 6000]))
        if a.big:      # the one-query-per-lane and persistent-lane kernels inside the loop, too
            nm = int(rng.choice([6000, 90000, 250000, 600000]))
        m = rng.uniform(-60, 60, (nm, 3)); m[:, 2] *= 0.3
        Tg = io.euler_to_matrix4(rng.uniform(-0.8, 0.8, 3), rng.uniform(-0.02, 0.02, 3))
        inv, _ = orc.m4inv(Tg)
        d = m[rng.permutation(nm)[: max(50, nm // 2)]] + rng.normal(0, 0.02, (max(50, nm // 2), 3)); orc.transform_points(inv, d)
        algo = int(rng.choice([1, 2, 6, 3, 4, 5, 7, 8, 9]))
        cls = {1: tdtk.icp6D_QUAT, 2: tdtk.icp6D_SVD, 6: tdtk.icp6D_APX, 3: tdtk.icp6D_ORTHO, 4: tdtk.icp6D_DUAL,
               5: tdtk.icp6D_HELIX, 7: tdtk.icp6D_LUMEULER, 8: tdtk.icp6D_LUMQUAT, 9: tdtk.icp6D_QUAT_SCALE}[algo]
        S = [tdtk.Scan([0, 0, 0], [0, 0, 0], m), tdtk.Scan([0, 0, 0], [0, 0, 0], d)]
        O = [io.OScan([0, 0, 0], [0, 0, 0], m), io.OScan([0, 0, 0], [0, 0, 0], d)]
        icp = tdtk.icp6D(cls(True), 5.0, 10, quiet=True, epsilonICP=1e-6)
        it = icp.match(S[0], S[1])
        oit, otr = io.match(O[0], O[1], algo, 25.0, 10, 1e-6)
        tol = 1e-8 if algo in (1, 2, 6) else 1e-6
        ok = it == oit and [int(r[0]) for r in icp.last["trace"]] == [t[0] for t in otr] and \
            np.abs(S[1].get_transMat() - O[1].transMat).max() <= tol * max(1.0, np.abs(O[1].transMat).max())
        if not ok:
            fails += 1
            np.save("gpurun_out/fuzz_fail_icp_%d.npy" % runs, m)
            print("ICP MISMATCH run %d algo=%d nm=%d it=%s/%s pairs=%s/%s dT=%g" % (
                runs, algo, nm, it, oit, [int(r[0]) for r in icp.last["trace"]][:4], [t[0] for t in otr][:4],
                np.abs(S[1].get_transMat() - O[1].transMat).max()), flush=True)
    # one in twenty: a random life of two resident scans (moves before / after going resident, tree built late,
    # pose extrapolation, whole-scan pair passes) against the oracle's host-side scans -- points bit for bit
    if runs % 20 == 0 and n >= 33:
        tr("scan life")
        from oracle import icp_oracle as io
        ra, rb = rng.uniform(-5, 5, 3), rng.uniform(-0.2, 0.2, 3)
        qa = p[rng.permutation(n)[: max(10, n // 2)]] + rng.normal(0, 0.1, (max(10, n // 2), 3))
        S = [tdtk.Scan([0, 0, 0], [0, 0, 0], p, bucketSize=int(bucket)), tdtk.Scan(ra, rb, qa)]
        O = [io.OScan([0, 0, 0], [0, 0, 0], p, None, int(bucket)), io.OScan(ra, rb, qa)]
        ok = True
        for step in range(int(rng.integers(3, 9))):
            op = int(rng.integers(0, 6))
            k = int(rng.integers(0, 2))
            if op == 0:
                X = tdtk.EulerToMatrix4(rng.uniform(-1, 1, 3), rng.uniform(-0.05, 0.05, 3))
                S[k].transform(X); O[k].transform(X)
            elif op == 1:
                _ = S[k].handle                       # go resident now
            elif op == 2:
                S[1].mergeCoordinatesWithRoboterPosition(S[0]); O[1].mergeCoordinatesWithRoboterPosition(O[0])
            elif op == 3:
                r = tdtk.Scan.getPtPairs(S[0], S[1], max_dist_match2=md2p)
                o = io.get_pt_pairs(O[0], O[1], md2p)
                ok = ok and r["n"] == o["n"] and abs(r["sum"] - o["sum"]) <= 1e-9 * max(1e-300, abs(o["sum"]))
            elif op == 4:
                rP, rT = rng.uniform(-3, 3, 3), rng.uniform(-0.1, 0.1, 3)
                S[k].transformToEuler(rP, rT); O[k].transformToEuler(rP, rT)
            else:
                ok = ok and np.array_equal(S[k].get_xyz_reduced(), O[k].xyz)
        for k in range(2):
            ok = ok and np.array_equal(S[k].get_xyz_reduced(), O[k].xyz) and np.array_equal(S[k].get_transMat(), O[k].transMat)
        if not ok:
            fails += 1
            np.save("gpurun_out/fuzz_fail_life_%d.npy" % runs, p)
            print("SCAN LIFE MISMATCH run %d n=%d" % (runs, n), flush=True)
    # one in forty: a small pose graph (chain + a closure) through one iteration of each graph-SLAM back-end
    if runs % 40 == 0:
        from oracle import icp_oracle as io
        nsc = int(rng.integers(4, 7))
        world = rng.uniform(-40, 40, (3000, 3)); world[:, 2] *= 0.4
        poses = [(np.array([3.0 * k, 0.4 * k, -0.2 * k]) + rng.normal(0, 0.2, 3), rng.normal(0, 0.01, 3) + [0.0, 0.0, 0.01 * k])
                 for k in range(nsc)]
        raws = []
        for (pp, th) in poses:
            inv, _ = orc.m4inv(io.euler_to_matrix4(pp, th))
            loc = world[rng.permutation(len(world))[:2000]].copy(); orc.transform_points(inv, loc)
            drift = (pp + rng.normal(0, 0.15, 3), th + rng.normal(0, 0.003, 3))
            raws.append((drift[0], drift[1], loc + rng.normal(0, 0.02, loc.shape)))
        links = [(k, k + 1) for k in range(nsc - 1)] + [(0, nsc - 1)] + ([(1, nsc - 1)] if rng.random() < 0.5 else [])
        for name in ("lum6DEuler", "lum6DQuat", "ghelix6DQ2", "gapx6D"):
            S = [tdtk.Scan(p_, t_, l_) for (p_, t_, l_) in raws]
            O = [io.OScan(p_, t_, l_) for (p_, t_, l_) in raws]
            gr = tdtk.Graph(nsc, links=links)
            ret = getattr(tdtk, name)(None, 5.0, 5.0, epsilonLUM=-1.0).doGraphSlam6D(gr, S, 1)
            if name == "lum6DEuler":
                oret = io.lum_iteration(links, O, 25.0)[0]
            elif name == "lum6DQuat":
                oret = io.lumquat_iteration(links, O, 25.0)
                oret = oret[0] if isinstance(oret, tuple) else oret
            elif name == "ghelix6DQ2":
                oret = io.ghelix_iteration(links, O, 25.0)[0]
            else:
                oret = io.gapx_iteration(links, O, 25.0)[0]
            ok = abs(ret - oret) <= 1e-6 * max(1.0, abs(oret))
            for s_, o_ in zip(S, O):
                ok = ok and np.abs(s_.get_transMat() - o_.transMat).max() <= 1e-6 * max(1.0, np.abs(o_.transMat).max())
            if not ok:
                fails += 1
                print("GRAPH MISMATCH run %d %s scans=%d links=%d ret %g / %g" % (runs, name, nsc, len(links), ret, oret), flush=True)
print("fuzz: %d clouds, %d mismatches, %.0f s, seed %d" % (runs, fails, time.time() - t0, a.seed))
sys.exit(1 if fails else 0)
