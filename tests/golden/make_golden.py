"""Generates the golden fixtures in this directory.  Run ONCE in the build container, where
/root/reference exists:   python tests/golden/make_golden.py
It reads reference *data files* (dat/scan00?.3d/.pose) and calls the reference's own compiled
translation units through oracle/_ref (oracle/build_ref.sh); it commits only numbers.

Fixtures written:
  dat_scans.npz     the three bundled uos scans (float64, as strtod parses them) + poses
  k4_random.npz     testing/kdtree/kdtree_indexed_random.cc recipe: 10 000 pts in [-10,10]^3,
                    100 queries x maxdist2 in {0.5,...,5.0}; expected = reference KDtreeIndexed
                    (cross-checked against brute force, first strictly smaller d2 wins)
  k5_hashes.json    1M-vs-1M K5 stream (mt19937_64(42), U(-1000,1000)): found counts + XOR
                    hashes of the reference KDtreeIndexed for maxdist2 = 625 and 1e18, tree stats
  k6_minimizers.json  reference icp6D_{QUAT,SVD,APX,NAPX}::Align on a synthetic rigid motion
  b1_dat_icp.json   sequential ICP on dat/ (-d 25 -i 50, QUAT): per-iteration pairs / RMS /
                    alignxf and final transMat, computed with the oracle loop + the REFERENCE
                    minimizer; cross-checked against SURVEY appendix B1
  b4_dat_lum.json   LUM link systems at the B1 final poses (-D 25)
  k6_serial_minimizers.json  reference icp6D_{ORTHO,DUAL,HELIX,LUMEULER,LUMQUAT,QUAT_SCALE}::Align
                    (-a 3,4,5,7,8,9) on the K6 clouds; `python make_golden.py k6s` regenerates only this
  b1_dat_icp_idx.json  the correspondence indices behind B1 (SURVEY 8(c) "still to generate"): the stored alignxf
                    sequence is replayed and at every iteration the REFERENCE KDtreeIndexed answers the whole data
                    scan; per iteration the XOR hash of the index array, for the first and last iteration of each pair
                    also the first / last 100 indices; `python make_golden.py b1idx` regenerates only this
  k7_ann_normals.npz  Scan::calcNormals = calculateNormalsApxKNN(k = 10, eps = 1.0): the vendored ANN library's
                    k-NN lists and the normals (real annkSearch + real newmat EigenValues through
                    oracle/ref_ann_driver.cc) for the first 6000 points of dat/scan000 and for a seeded noisy
                    plane; `python make_golden.py k7` regenerates only this
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc, icp_oracle as io  # noqa: E402

REF = os.environ.get("TDTK_REF", "/root/reference")


def brute(m, q, maxd2):
    out = np.full(len(q), -1, np.int32)
    for i, p in enumerate(q):
        d = m - p
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        j = int(np.argmin(d2))            # first minimum == first strictly smaller
        if d2[j] < maxd2:
            out[i] = j
    return out


def gen_k6_serial():
    """K6 clouds through the reference's serial-only minimizers.  LUMEULER / LUMQUAT receive the current
    scan pose in alignxf (icp6D.cc:237-241); QUAT_SCALE gets a cloud that really is scaled."""
    d = orc.gen_mt64_uniform(7, 3000, -100, 100).reshape(1000, 3)
    T = io.euler_to_matrix4([1.5, -2.0, 0.7], [0.02, -0.03, 0.05])
    mm = d.copy(); orc.transform_points(T, mm)
    noise = orc.gen_mt64_uniform(9, 3000, -0.5, 0.5).reshape(1000, 3)
    pose = io.euler_to_matrix4([10.0, -5.0, 3.0], [0.01, 0.02, -0.03])
    out = {"seed_points": 7, "seed_noise": 9, "rPos": [1.5, -2.0, 0.7], "rPosTheta": [0.02, -0.03, 0.05],
           "pose": pose.tolist(), "scale_case": 1.02, "cases": {}}
    cd = d.mean(axis=0)
    for tag, pm in (("clean", mm), ("noisy", mm + noise), ("scaled", 1.02 * mm + noise)):
        cm = pm.mean(axis=0)
        out["cases"][tag] = {}
        for algo in (3, 4, 5, 7, 8, 9):
            a, err = orc.ref_align(algo, pm, d, cm, cd, None, pose)
            out["cases"][tag][str(algo)] = {"alignxf": a.tolist(), "rms": err}
            print("K6s", tag, algo, err, a[12:15], a[0])
    json.dump(out, open(os.path.join(HERE, "k6_serial_minimizers.json"), "w"), indent=1)


def k7_clouds():
    """The two K7 clouds: a slice of the bundled scan (needs dat_scans.npz) and a seeded noisy plane."""
    z = np.load(os.path.join(HERE, "dat_scans.npz"))
    u = orc.gen_mt64_uniform(11, 9000, -1.0, 1.0).reshape(3000, 3)
    plane = np.stack([300.0 * u[:, 0], 300.0 * u[:, 1], 40.0 + 0.1 * 300.0 * u[:, 0] + 0.5 * u[:, 2]], axis=1)
    return {"dat": np.ascontiguousarray(z["scan000"][:6000]), "plane": plane}


def gen_k7_ann():
    out = {}
    rPos = np.array([0.0, 0.0, 0.0])
    for tag, pts in k7_clouds().items():
        t = orc.AnnTree(pts, "ref")
        idx, dist = t.ksearch(pts, 10, 1.0)
        nrm = orc.normals_apx_knn(pts, 10, rPos, 1.0, "ref")
        out[tag + "_knn"] = idx
        out[tag + "_normals"] = nrm
        out[tag + "_stats"] = np.array(t.stats())
        print("K7", tag, pts.shape, t.stats(), "knn hash", hex(int(orc.k5_hash(idx.reshape(-1)))))
    np.savez_compressed(os.path.join(HERE, "k7_ann_normals.npz"), **out)


def gen_b1_idx():
    """Needs dat_scans.npz and b1_dat_icp.json (written by the full run)."""
    z = np.load(os.path.join(HERE, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(HERE, "b1_dat_icp.json")))
    S = [io.OScan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(3)]
    out = {"maxdist2": 625.0, "pairs": []}
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        R = orc.RefTree(S[i - 1].xyz_orig, 20)                 # the tree lives in the frame of "xyz reduced original"
        inv, _ = orc.m4inv(S[i - 1].dalignxf)                  # searchTree.cc:110,122
        rows = []
        for it, a in enumerate(pr["alignxf"]):
            q = S[i].xyz.copy(); orc.transform_points(inv, q)
            idx = R.find_closest(q, 625.0, 8)
            row = {"found": int((idx >= 0).sum()), "hash": "0x%x" % orc.k5_hash(idx)}
            if it in (0, len(pr["alignxf"]) - 1):
                row["first100"] = idx[:100].tolist(); row["last100"] = idx[-100:].tolist()
            assert row["found"] == pr["trace"][it][0], (row["found"], pr["trace"][it][0])
            rows.append(row)
            S[i].transform(np.array(a))
        out["pairs"].append({"prev": i - 1, "cur": i, "iterations": rows})
        print("B1idx pair", i, len(rows), rows[0]["hash"], rows[-1]["hash"])
    json.dump(out, open(os.path.join(HERE, "b1_dat_icp_idx.json"), "w"), indent=1)


def main():
    assert orc.have_ref() or os.path.isdir(REF), "needs the reference checkout"
    orc.build()
    if sys.argv[1:] == ["b1idx"]:
        return gen_b1_idx()
    if sys.argv[1:] == ["k6s"]:
        return gen_k6_serial()
    if sys.argv[1:] == ["k7"]:
        return gen_k7_ann()
    gen_k6_serial()

    # ---- dat scans ------------------------------------------------------------------
    scans, poses = {}, {}
    for k in range(3):
        scans["scan%03d" % k] = io.read_uos(os.path.join(REF, "dat", "scan%03d.3d" % k))
        rP, rPT = io.read_pose(os.path.join(REF, "dat", "scan%03d.pose" % k))
        poses["pose%03d" % k] = np.concatenate([rP, rPT])
        print("scan", k, scans["scan%03d" % k].shape, poses["pose%03d" % k])
    np.savez_compressed(os.path.join(HERE, "dat_scans.npz"), **scans, **poses)

    # ---- K4 ---------------------------------------------------------------------------
    rng = np.random.RandomState(42)
    m = rng.uniform(-10, 10, (10000, 3))
    qs, exp, mds = [], [], []
    R = orc.RefTree(m, 20)
    for j in range(1, 11):
        md2 = 0.5 * j
        q = rng.uniform(-10, 10, (100, 3))
        e = R.find_closest(q, md2)
        assert np.array_equal(e, brute(m, q, md2)), "reference tree != brute force"
        qs.append(q); exp.append(e); mds.append(md2)
    np.savez_compressed(os.path.join(HERE, "k4_random.npz"), model=m, queries=np.array(qs),
                        expected=np.array(exp), maxdist2=np.array(mds))

    # ---- K5 ---------------------------------------------------------------------------
    M = 1000000
    stream = orc.gen_mt64_uniform(42, 6 * M, -1000, 1000)
    m5 = stream[:3 * M].reshape(M, 3).copy()
    q5 = stream[3 * M:].reshape(M, 3).copy()
    R5 = orc.RefTree(m5, 20)
    T5 = orc.Tree(m5, 20)
    k5 = {"M": M, "seed": 42, "lo": -1000, "hi": 1000, "bucket": 20, "tree": T5.stats(), "cases": []}
    for md2 in (625.0, 1e18):
        e = R5.find_closest(q5, md2, 8)
        oi, _, cnt = T5.find_closest(q5, md2, 8, True)
        assert np.array_equal(e, oi)
        k5["cases"].append({"maxdist2": md2, "found": int((e >= 0).sum()), "hash": "0x%x" % orc.k5_hash(e),
                            "visits_per_query": [c / M for c in cnt],
                            "first32": e[:32].tolist(), "last32": e[-32:].tolist()})
        print("K5", k5["cases"][-1]["maxdist2"], k5["cases"][-1]["found"], k5["cases"][-1]["hash"])
    json.dump(k5, open(os.path.join(HERE, "k5_hashes.json"), "w"), indent=1)

    # ---- K6 minimizers -----------------------------------------------------------------
    d = orc.gen_mt64_uniform(7, 3000, -100, 100).reshape(1000, 3)
    T = io.euler_to_matrix4([1.5, -2.0, 0.7], [0.02, -0.03, 0.05])
    mm = d.copy(); orc.transform_points(T, mm)
    nr = orc.gen_mt64_uniform(8, 3000, -1, 1).reshape(1000, 3)
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    cm, cd = mm.mean(axis=0), d.mean(axis=0)
    k6 = {"seed_points": 7, "seed_normals": 8, "rPos": [1.5, -2.0, 0.7], "rPosTheta": [0.02, -0.03, 0.05], "algos": {}}
    for algo in (1, 2, 6, 10):
        a, err = orc.ref_align(algo, mm, d, cm, cd, nr if algo == 10 else None)
        k6["algos"][str(algo)] = {"alignxf": a.tolist(), "rms": err}
        print("K6", algo, err, a[12:15], a[0])
    # a noisy, non-exact case as well
    noise = orc.gen_mt64_uniform(9, 3000, -0.5, 0.5).reshape(1000, 3)
    mn = mm + noise
    cmn = mn.mean(axis=0)
    k6["noisy"] = {"seed_noise": 9, "algos": {}}
    for algo in (1, 2, 6, 10):
        a, err = orc.ref_align(algo, mn, d, cmn, cd, nr if algo == 10 else None)
        k6["noisy"]["algos"][str(algo)] = {"alignxf": a.tolist(), "rms": err}
    json.dump(k6, open(os.path.join(HERE, "k6_minimizers.json"), "w"), indent=1)

    # ---- B1: dat sequential ICP (oracle loop + REFERENCE minimizer) --------------------
    def ref_align_fn(algo, p1, p2, cm, cd, pn, pose=None):
        a, err = orc.ref_align(algo, p1, p2, cm, cd, pn, pose)
        return err, a
    S = [io.OScan(poses["pose%03d" % k][:3], poses["pose%03d" % k][3:], scans["scan%03d" % k]) for k in range(3)]
    b1 = {"params": {"algo": 1, "max_dist_match": 25.0, "max_num_iterations": 50, "epsilonICP": 1e-5, "eP": True},
          "pairs": []}
    for i in (1, 2):
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        it, tr = io.match(S[i - 1], S[i], 1, 625.0, 50, 1e-5, 0, ref_align_fn)
        b1["pairs"].append({"prev": i - 1, "cur": i, "iter": it,
                            "trace": [[int(n), float(r)] for n, r, _ in tr],
                            "alignxf": [a.tolist() for _, _, a in tr],
                            "final_transMat": S[i].transMat.tolist()})
        print("B1 pair", i, "ITER", it, tr[0][:2], tr[-1][:2], S[i].transMat[12:15])
    json.dump(b1, open(os.path.join(HERE, "b1_dat_icp.json"), "w"), indent=1)

    # ---- B4: LUM link systems at the B1 final poses -----------------------------------
    b4 = {"max_dist_match_LUM": 25.0, "links": []}
    for (a, b) in ((0, 1), (1, 2)):
        C, CD, m_, ss, D = io.covariance_euler(S[a], S[b], 625.0)
        b4["links"].append({"first": a, "second": b, "m": int(m_), "ss": float(ss), "D": D.tolist(),
                            "C": C.tolist(), "CD": CD.tolist()})
        print("B4 link", a, b, m_, ss, D[:3], CD[:3])
    links = [(0, 1), (1, 2)]
    ret, G, B, X = io.lum_iteration(links, S, 625.0)
    b4["one_iteration"] = {"ret": float(ret), "X": X.tolist(),
                           "poses_after": [np.concatenate([s.rPos, s.rPosTheta]).tolist() for s in S]}
    print("B4 lum iteration ret", ret)
    json.dump(b4, open(os.path.join(HERE, "b4_dat_lum.json"), "w"), indent=1)
    gen_k7_ann()      # after dat_scans.npz has been written
    gen_b1_idx()


if __name__ == "__main__":
    main()
