"""CPU tier: pin the oracle (oracle/oracle.c + oracle/icp_oracle.py) against
  * the reference's own known-answer tests (testing/kdtree/kdtree.cc:20-99),
  * the reference's own translation units built into oracle/_ref (when present),
  * the committed golden fixtures (tests/golden/, generated with oracle/_ref).
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def _need_ref(orc):
    if not orc.have_ref():
        pytest.skip("oracle/_ref not built (no reference checkout on this box)")


# ---- KATs of testing/kdtree/kdtree.cc ------------------------------------------------------
KAT_CLOSEST = [
    ([[2.0, 0.0, 0.0]], None),                       # find_closest1: exactly maxdist2 away -> NULL
    ([[1.99999999999, 0.0, 0.0]], 0),                # find_closest2
    ([[1.5, 0.0, 0.0], [1.0, 0.0, 0.0]], 1),         # find_closest3
]
KAT_DIR = [
    ([[1.0, 2.0, 0.0]], None),                       # find_closest_along_dir1
    ([[1.0, 1.99999999999, 0.0]], 0),
    ([[0.5, 0.1, 0.0], [1.0, 0.0, 0.0]], 1),
    ([[-1.0, 0.0, 0.0]], 0),
]


@pytest.mark.parametrize("pts,want", KAT_CLOSEST)
def test_kat_find_closest(orc, pts, want):
    t = orc.Tree(np.array(pts), 20)
    idx, _ = t.find_closest(np.zeros((1, 3)), 4.0)
    assert idx[0] == (-1 if want is None else want)
    if orc.have_ref():
        assert orc.RefTree(np.array(pts), 20).find_closest(np.zeros((1, 3)), 4.0)[0] == idx[0]


@pytest.mark.parametrize("pts,want", KAT_DIR)
def test_kat_find_closest_along_dir(orc, pts, want):
    t = orc.Tree(np.array(pts), 20)
    idx, _ = t.find_closest_along_dir(np.zeros((1, 3)), np.array([[1.0, 0.0, 0.0]]), 4.0)
    assert idx[0] == (-1 if want is None else want)
    if orc.have_ref():
        r = orc.RefTree(np.array(pts), 20).find_closest_along_dir(np.zeros((1, 3)), np.array([[1.0, 0, 0]]), 4.0)
        assert r[0] == idx[0]


def _clouds():
    rng = np.random.default_rng(3)
    uni = rng.uniform(-100, 100, (30000, 3))
    dup = uni.copy(); dup[1000:1400] = dup[0:400]            # exact duplicates -> ties
    clu = np.concatenate([rng.normal(c, 0.003, (300, 3)) for c in rng.uniform(-50, 50, (40, 3))])  # <0.01 boxes
    plane = rng.uniform(-100, 100, (20000, 3)); plane[:, 2] = 0.0
    tiny = rng.uniform(-1, 1, (7, 3))
    grid = np.stack(np.meshgrid(*[np.arange(12.0)] * 3), -1).reshape(-1, 3)      # many equal distances
    return {"uniform": uni, "duplicates": dup, "clusters": clu, "plane": plane, "tiny": tiny, "grid": grid}


@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "tiny", "grid"])
@pytest.mark.parametrize("bucket", [1, 20])
def test_oracle_tree_equals_reference(orc, name, bucket):
    _need_ref(orc)
    m = _clouds()[name]
    rng = np.random.default_rng(11)
    q = np.concatenate([m[rng.integers(0, len(m), 2000)] + rng.normal(0, 0.5, (2000, 3)),
                        m[:500],                                     # exact hits (d2 == 0, ties on dups)
                        rng.uniform(-120, 120, (1500, 3))])
    if name == "grid":
        q = np.concatenate([q, m[:500] + 0.5])                      # equidistant to 8 corners
    T, R = orc.Tree(m, bucket), orc.RefTree(m, bucket)
    for md2 in (0.25, 25.0, 1e18):
        a, _ = T.find_closest(q, md2)
        b = R.find_closest(q, md2)
        assert np.array_equal(a, b), (name, bucket, md2)
    d = rng.normal(size=(len(q), 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    a, _ = T.find_closest_along_dir(q[:800], d[:800], 4.0)
    assert np.array_equal(a, R.find_closest_along_dir(q[:800], d[:800], 4.0))


def test_k4_fixture(orc):
    z = np.load(os.path.join(G, "k4_random.npz"))
    T = orc.Tree(z["model"], 20)
    for q, e, md2 in zip(z["queries"], z["expected"], z["maxdist2"]):
        assert np.array_equal(T.find_closest(q, md2)[0], e)


def test_k5_hashes(orc):
    k5 = json.load(open(os.path.join(G, "k5_hashes.json")))
    M = k5["M"]
    s = orc.gen_mt64_uniform(k5["seed"], 6 * M, k5["lo"], k5["hi"])
    m, q = s[:3 * M].reshape(M, 3).copy(), s[3 * M:].reshape(M, 3).copy()
    T = orc.Tree(m, k5["bucket"])
    assert T.stats() == k5["tree"]
    for c in k5["cases"]:
        idx, _, cnt = T.find_closest(q, c["maxdist2"], min(8, orc.lib().orc_max_threads()), True)
        assert int((idx >= 0).sum()) == c["found"]
        assert "0x%x" % orc.k5_hash(idx) == c["hash"]
        assert idx[:32].tolist() == c["first32"] and idx[-32:].tolist() == c["last32"]
        np.testing.assert_allclose(np.array(cnt) / M, c["visits_per_query"], rtol=1e-12)


def test_k6_minimizers(orc):
    from oracle import icp_oracle as io
    k6 = json.load(open(os.path.join(G, "k6_minimizers.json")))
    d = orc.gen_mt64_uniform(k6["seed_points"], 3000, -100, 100).reshape(1000, 3)
    T = io.euler_to_matrix4(k6["rPos"], k6["rPosTheta"])
    mm = d.copy(); orc.transform_points(T, mm)
    nr = orc.gen_mt64_uniform(k6["seed_normals"], 3000, -1, 1).reshape(1000, 3)
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    cd = d.mean(axis=0)
    for tag, pm in (("clean", mm), ("noisy", mm + orc.gen_mt64_uniform(k6["noisy"]["seed_noise"], 3000, -0.5, 0.5).reshape(1000, 3))):
        exp = k6["algos"] if tag == "clean" else k6["noisy"]["algos"]
        cm = pm.mean(axis=0)
        for algo in (1, 2, 6, 10):
            rms, a = io.align(algo, pm, d, cm, cd, nr)
            np.testing.assert_allclose(a, exp[str(algo)]["alignxf"], rtol=0, atol=2e-9, err_msg="%s %d" % (tag, algo))
            assert abs(rms - exp[str(algo)]["rms"]) <= 1e-9 * max(1.0, abs(rms))


def _k6s_inputs(orc):
    from oracle import icp_oracle as io
    k = json.load(open(os.path.join(G, "k6_serial_minimizers.json")))
    d = orc.gen_mt64_uniform(k["seed_points"], 3000, -100, 100).reshape(1000, 3)
    T = io.euler_to_matrix4(k["rPos"], k["rPosTheta"])
    mm = d.copy(); orc.transform_points(T, mm)
    noise = orc.gen_mt64_uniform(k["seed_noise"], 3000, -0.5, 0.5).reshape(1000, 3)
    clouds = {"clean": mm, "noisy": mm + noise, "scaled": k["scale_case"] * mm + noise}
    return k, d, clouds


def test_k6_serial_only_minimizers(orc):
    """numpy restatements of ORTHO / DUAL / HELIX / LUMEULER / LUMQUAT / QUAT_SCALE (-a 3,4,5,7,8,9) against
    the fixture generated with the reference's own TUs."""
    from oracle import icp_oracle as io
    k, d, clouds = _k6s_inputs(orc)
    pose = np.array(k["pose"])
    cd = d.mean(axis=0)
    for tag, pm in clouds.items():
        cm = pm.mean(axis=0)
        for algo in (3, 4, 5, 7, 8, 9):
            exp = k["cases"][tag][str(algo)]
            rms, a = io.align(algo, pm, d, cm, cd, None, pose)
            np.testing.assert_allclose(a, exp["alignxf"], rtol=0, atol=5e-9, err_msg="%s %d" % (tag, algo))
            assert abs(rms - exp["rms"]) <= 1e-9 * max(1.0, abs(rms))


def test_align_parallel_quat_vs_reference(orc):
    _need_ref(orc)
    from oracle import icp_oracle as io
    rng = np.random.default_rng(5)
    d = rng.uniform(-100, 100, (8000, 3))
    T = io.euler_to_matrix4([3, -2, 1], [0.03, 0.01, -0.02])
    m = d.copy(); orc.transform_points(T, m); m += rng.normal(0, 0.3, m.shape)
    n = np.full(8, 1000, np.uint32)
    cm = np.array([m[i * 1000:(i + 1) * 1000].mean(0) for i in range(8)])
    cd = np.array([d[i * 1000:(i + 1) * 1000].mean(0) for i in range(8)])
    s = np.array([((m[i * 1000:(i + 1) * 1000] - d[i * 1000:(i + 1) * 1000]) ** 2).sum() for i in range(8)])
    Si = np.array([((m[i * 1000:(i + 1) * 1000] - cm[i]).T @ (d[i * 1000:(i + 1) * 1000] - cd[i])).reshape(9)
                   for i in range(8)])
    ra, re = orc.ref_align_parallel(1, n, s, cm, cd, Si)
    rms, a = io.align_parallel_quat(n, s, cm, cd, Si)
    np.testing.assert_allclose(a, ra, atol=1e-10)
    assert abs(rms - re) < 1e-12 * re


def test_m4inv_mmult_roundtrip(orc):
    from oracle import icp_oracle as io
    A = io.euler_to_matrix4([10, -5, 3], [0.02, -0.03, 0.05])
    inv, ok = orc.m4inv(A)
    assert ok
    np.testing.assert_allclose(orc.mmult(A, inv), np.eye(4).reshape(16), atol=1e-12)
    sing, ok = orc.m4inv(np.zeros(16))
    assert not ok and np.array_equal(sing, np.eye(4).reshape(16))     # globals.icc:765-769


def test_globals_icc_primitives_bit_exact_against_reference(orc, tdtk):
    """A12: M4inv / MMult / transform3 (both forms) / transform3normal / EulerToMatrix4 / Matrix4ToEuler /
    QuatToMatrix4 / Matrix4ToQuat as the reference's own include/slam6d/globals.icc compiles them (oracle/_ref),
    against (a) the oracle's restatements and (b) the product's host algebra (the tdtk_host_* diagnostics and the
    Python mirror): every output bit for bit, on random poses, ill-scaled matrices and the singular case."""
    import ctypes as C
    from oracle import icp_oracle as io
    if not orc.have_ref():
        pytest.skip("reference TUs not built here")
    L = tdtk.lib()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    rng = np.random.default_rng(12)
    for trial in range(300):
        pos = rng.uniform(-1000, 1000, 3) * (10.0 ** rng.integers(-3, 4))
        th = rng.uniform(-np.pi, np.pi, 3)
        if trial % 7 == 0:
            th[1] = (np.pi / 2) * (1 if trial % 2 else -1) + rng.normal(0, 1e-4)     # near the Euler singularity
        A = orc.ref_euler_to_matrix4(pos, th)
        a_prod = np.empty(16); L.tdtk_host_euler_to_matrix4(dp(pos), dp(th), dp(a_prod))
        assert np.array_equal(A, io.euler_to_matrix4(pos, th)) and np.array_equal(A, tdtk.EulerToMatrix4(pos, th))
        assert np.array_equal(A, a_prod)
        B = orc.ref_euler_to_matrix4(rng.uniform(-50, 50, 3), rng.uniform(-0.1, 0.1, 3))
        rigid = orc.ref_mmult(A, B)
        if trial % 5 == 0:                                # a general (non-rigid) matrix
            A = A + rng.normal(0, 0.05, 16); A[15] = 1.0
        inv_r, rc = orc.ref_m4inv_raw(A)
        inv_o, ok = orc.m4inv(A)
        assert rc == 1 and ok and np.array_equal(inv_r, inv_o) and np.array_equal(inv_r, tdtk.M4inv(A))
        mm = orc.ref_mmult(A, B)
        assert np.array_equal(mm, orc.mmult(A, B)) and np.array_equal(mm, tdtk.MMult(A, B))
        if abs(rigid[8]) > 1.0:                           # asin domain: C gives NaN, Python raises; rounding can do it
            rigid[8] = np.sign(rigid[8])
        th_r, pos_r = orc.ref_matrix4_to_euler(rigid)
        th_o, pos_o = io.matrix4_to_euler(rigid)
        th_p, pos_p = tdtk.Matrix4ToEuler(rigid)
        th_c, pos_c = np.empty(3), np.empty(3); L.tdtk_host_matrix4_to_euler(dp(rigid), dp(th_c), dp(pos_c))
        for t_, p_ in ((th_o, pos_o), (th_p, pos_p), (th_c, pos_c)):
            assert np.array_equal(th_r, np.asarray(t_)) and np.array_equal(pos_r, np.asarray(p_))
        q_r, t_r = orc.ref_matrix4_to_quat(B)
        q_o, t_o = io.matrix4_to_quat(B)
        q_c, t_c = np.empty(4), np.empty(3); L.tdtk_host_matrix4_to_quat(dp(B), dp(q_c), dp(t_c))
        assert np.array_equal(q_r, q_o) and np.array_equal(t_r, t_o) and np.array_equal(q_r, q_c) and np.array_equal(t_r, t_c)
        assert np.array_equal(q_r, tdtk.Matrix4ToQuat(B)[0])
        m_r = orc.ref_quat_to_matrix4(q_r, t_r)
        m_c = np.empty(16); L.tdtk_host_quat_to_matrix4(dp(q_r), dp(t_r), dp(m_c))
        assert np.array_equal(m_r, io.quat_to_matrix4(q_r, t_r)) and np.array_equal(m_r, m_c)
        assert np.array_equal(m_r, tdtk.QuatToMatrix4(q_r, t_r))
        pts = rng.uniform(-1000, 1000, (64, 3))
        moved = orc.ref_transform3_inplace(A, pts)
        p2 = pts.copy(); orc.transform_points(A, p2)
        assert np.array_equal(moved, p2)
        nr = rng.normal(size=(64, 3))
        n2 = nr.copy(); orc.transform_normals(A, n2)
        assert np.array_equal(orc.ref_transform3normal(A, nr), n2)
    # singular input: "Error matrix inverting!", identity out, return 0 (globals.icc:765-769)
    inv_r, rc = orc.ref_m4inv_raw(np.zeros(16))
    inv_o, ok = orc.m4inv(np.zeros(16))
    assert rc == 0 and not ok and np.array_equal(inv_r, inv_o) and np.array_equal(inv_r, tdtk.M4inv(np.zeros(16)))


def test_transform3_out_of_place_form_against_reference(orc):
    """transform3(alignxf, in, out) (globals.icc:1477-1490) -- the form SearchTree::getPtPairs uses for the query
    and the hit (searchTree.cc:122,147) -- differs from the in-place form in its association; the oracle's
    getPtPairs must use this one: pair lists through a non-trivial pose are compared with the reference's own
    transform3 applied to the reference tree's hits."""
    from oracle import icp_oracle as io
    if not orc.have_ref():
        pytest.skip("reference TUs not built here")
    rng = np.random.default_rng(5)
    m = rng.uniform(-100, 100, (20000, 3))
    A = io.euler_to_matrix4([3.0, -2.0, 1.0], [0.03, -0.02, 0.05])
    inv, _ = orc.ref_m4inv_raw(A)
    world = orc.ref_transform3(A, m)[rng.permutation(len(m))[:5000]] + rng.normal(0, 0.3, (5000, 3))
    T = orc.Tree(m, 10)
    o = T.get_pt_pairs(A, world, None, 0, None, 0, 4.0)
    q = orc.ref_transform3(inv, world)                        # searchTree.cc:122
    ridx = orc.RefTree(m, 10).find_closest(q, 4.0)
    assert np.array_equal(o["idx"], ridx)
    found = ridx >= 0
    assert np.array_equal(o["p1"], orc.ref_transform3(A, m[ridx[found]]))    # searchTree.cc:147
    assert np.array_equal(o["p2"], world[found])


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_get_pt_pairs_all_modes_against_reference_pieces(orc, mode):
    """A4: SearchTree::getPtPairs in the three pairing modes (searchTree.cc:92-189) -- closest point, closest point
    along the normal (FindClosestAlongDir with the normalised normal rotated into the tree frame), closest point
    projected onto the data point's tangent plane -- with the loop restated in ref_driver.cc and every operation the
    reference's own compiled code: the oracle's restatement gives the same indices, pair lists (p1, p2, the normal the
    pair carries) and accumulators, bit for bit, through a non-trivial pose, with duplicates in the model."""
    from oracle import icp_oracle as io
    if not orc.have_ref():
        pytest.skip("reference TUs not built here")
    rng = np.random.default_rng(40 + mode)
    m = rng.uniform(-100, 100, (12000, 3)); m[:, 2] *= 0.05; m[500:700] = m[0:200]
    A = io.euler_to_matrix4([4.0, -1.0, 2.0], [0.02, -0.05, 0.03])
    world = orc.ref_transform3(A, m)[rng.permutation(len(m))[:3000]] + rng.normal(0, 0.4, (3000, 3))
    nrm = rng.normal(size=(3000, 3)) * rng.uniform(0.1, 5.0, (3000, 1))      # un-normalised on purpose
    md2 = 9.0
    r = orc.RefTree(m, 10).get_pt_pairs(A, world, nrm if mode else None, mode, md2)
    o = orc.Tree(m, 10).get_pt_pairs(A, world, nrm if mode else None, 0, None, mode, md2)
    assert r["n"] == o["n"] > 500 and np.array_equal(r["idx"], o["idx"])
    assert np.array_equal(r["p1"], o["p1"]) and np.array_equal(r["p2"], o["p2"])
    if mode:
        assert np.array_equal(r["pn"], o["pn"])
    assert r["sum"] == o["sum"] and np.array_equal(r["centroid_m"], o["centroid_m"]) and np.array_equal(r["centroid_d"], o["centroid_d"])


def test_spd_solve_against_newmat(orc, tdtk):
    """A16: graphSlam6D::solveSparseCholesky goes through CSparse, which this image does not have; the product's dense
    envelope Cholesky (tdtk_solve_spd, with convertToCS's |v| > 1e-5 entry filter) is checked against the reference's
    vendored matrix library instead: newmat's `G.i() * B` on a C4-shaped block system (63 unknown poses, 6x6 SPD link
    blocks spanning 1e3 .. 1e9 like real LUM blocks, chain + closures)."""
    import ctypes as C
    if not orc.have_ref():
        pytest.skip("reference TUs not built here")
    rng = np.random.default_rng(16)
    n = 63
    G = np.zeros((6 * n, 6 * n)); B = np.zeros(6 * n)
    links = [(i, i + 1) for i in range(n)] + [(0, 40), (5, 52), (10, 63), (20, 61)]
    scale = np.array([1.0, 1.0, 1.0, 300.0, 300.0, 300.0])
    for (fa, fb) in links:
        M = rng.normal(size=(6, 6))
        Cm = (M @ M.T + 6 * np.eye(6)) * 4e3 * np.outer(scale, scale)
        CD = rng.normal(size=6) * 50.0 * scale
        a, b = fa - 1, fb - 1
        if a >= 0:
            B[a * 6:a * 6 + 6] += CD; G[a * 6:a * 6 + 6, a * 6:a * 6 + 6] += Cm
        if b >= 0:
            B[b * 6:b * 6 + 6] -= CD; G[b * 6:b * 6 + 6, b * 6:b * 6 + 6] += Cm
        if a >= 0 and b >= 0:
            G[a * 6:a * 6 + 6, b * 6:b * 6 + 6] -= Cm; G[b * 6:b * 6 + 6, a * 6:a * 6 + 6] -= Cm
    _, Xn = orc.ref_newmat_inverse_solve(G, B)
    X = np.empty(6 * n)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert tdtk.lib().tdtk_solve_spd(dp(np.ascontiguousarray(G)), dp(B), 6 * n, dp(X)) == 0
    np.testing.assert_allclose(X, Xn, rtol=1e-7, atol=1e-9 * np.abs(Xn).max())


def test_lum_link_system_against_newmat(orc, tdtk):
    """A14: covarianceEuler's arithmetic (lum6Deuler.cc:143-232) evaluated with the reference's own newmat objects
    (`MM.i() * MZ`, `MM * ss`) in oracle/_ref on the pinned dat/ pair lists: the numpy restatement, the committed
    B4 fixture and the product's dense inverse (tdtk_invert, host) all agree with it."""
    import ctypes as C
    from oracle import icp_oracle as io
    if not orc.have_ref():
        pytest.skip("reference TUs not built here")
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    b4 = json.load(open(os.path.join(G, "b4_dat_lum.json")))
    S = [io.OScan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(3)]
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        for a in pr["alignxf"]:
            S[i].transform(np.array(a))
    for Lk in b4["links"]:
        r = io.get_pt_pairs(S[Lk["first"]], S[Lk["second"]], 625.0)
        Cr, CDr, ssr, Dr = orc.ref_lum_covariance_euler(r["p1"], r["p2"])
        Co, CDo, m, sso, Do = io.covariance_euler(S[Lk["first"]], S[Lk["second"]], 625.0)
        assert m == Lk["m"] == len(r["p1"])
        np.testing.assert_allclose(sso, ssr, rtol=1e-12)
        np.testing.assert_allclose(Do, Dr, rtol=1e-8, atol=1e-13)
        np.testing.assert_allclose(Co, Cr, rtol=1e-11, atol=1e-9)
        np.testing.assert_allclose(CDo, CDr, rtol=1e-10, atol=1e-10)
        np.testing.assert_allclose(Lk["C"], Cr, rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(Lk["ss"], ssr, rtol=1e-10)
        # the product's inverse (pivoted Gauss-Jordan, linalg.cpp) on the same MM against newmat's .i()
        MM = Cr * ssr
        MZ = CDr * ssr
        Ai, x = orc.ref_newmat_inverse_solve(MM, MZ)
        mine = np.empty((6, 6))
        assert tdtk.lib().tdtk_invert(MM.ctypes.data_as(C.POINTER(C.c_double)), 6, mine.ctypes.data_as(C.POINTER(C.c_double))) == 0
        np.testing.assert_allclose(mine @ MZ, x, rtol=1e-9, atol=1e-14)
        np.testing.assert_allclose(x, Dr, rtol=1e-9, atol=1e-14)


def test_b1_dat_icp_trace(orc):
    """The oracle loop with the numpy minimizer reproduces the committed trace that was
    generated with the REFERENCE minimizer (and SURVEY appendix B1)."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    S = [io.OScan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(3)]
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        it, tr = io.match(S[i - 1], S[i], 1, 625.0, 50, 1e-5)
        assert it == pr["iter"]
        assert [t[0] for t in tr] == [t[0] for t in pr["trace"]]
        np.testing.assert_allclose([t[1] for t in tr], [t[1] for t in pr["trace"]], rtol=1e-10)
        np.testing.assert_allclose(S[i].transMat, pr["final_transMat"], rtol=1e-9, atol=1e-9)
    # SURVEY appendix B1 known answers
    assert b1["pairs"][0]["trace"][0][0] == 73343 and abs(b1["pairs"][0]["trace"][0][1] - 5.9733177799) < 1e-9
    assert b1["pairs"][0]["iter"] == 38 and b1["pairs"][1]["iter"] == 49


def test_b4_lum_links(orc):
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    b4 = json.load(open(os.path.join(G, "b4_dat_lum.json")))
    S = [io.OScan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(3)]
    for pr in b1["pairs"]:       # replay the recorded alignxf sequence (bit-identical point motion)
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        for a in pr["alignxf"]:
            S[i].transform(np.array(a))
    for L in b4["links"]:
        C, CD, m, ss, D = io.covariance_euler(S[L["first"]], S[L["second"]], 625.0)
        assert m == L["m"]
        np.testing.assert_allclose(ss, L["ss"], rtol=1e-10)
        np.testing.assert_allclose(C, L["C"], rtol=1e-9, atol=1e-6)
        np.testing.assert_allclose(CD, L["CD"], rtol=1e-8, atol=1e-8)
    # SURVEY appendix B4
    assert b4["links"][0]["m"] == 73335 and abs(b4["links"][0]["ss"] - 17.34408661) < 1e-7


# ---- normals: ANN kd-tree + approximate k-NN + PCA (oracle_normals.c) ---------------------------
def _ann_clouds():
    rng = np.random.default_rng(0)
    c = {"uniform": rng.uniform(-100, 100, (5000, 3))}
    p = rng.uniform(-50, 50, (4000, 3)); p[:, 2] = 0.01 * p[:, 0] + rng.normal(0, 0.05, 4000)
    c["plane"] = p
    c["lattice"] = np.stack(np.meshgrid(np.arange(12.0), np.arange(12.0), np.arange(12.0)), -1).reshape(-1, 3)
    d = rng.uniform(-1, 1, (300, 3))
    c["duplicates"] = np.concatenate([d, d, d[:100]])
    c["tiny"] = rng.uniform(-1, 1, (11, 3))
    c["line"] = np.outer(np.arange(200.0), [1.0, 0.0, 0.0])
    c["clusters"] = np.concatenate([rng.normal(m, 0.5, (400, 3)) for m in ((0, 0, 0), (50, 0, 0), (0, 80, 5))])
    return c


@pytest.mark.parametrize("name", ["uniform", "plane", "lattice", "duplicates", "tiny", "line", "clusters"])
def test_ann_oracle_equals_vendored_library(orc, name):
    """Tree (pre-order cut dimensions and leaf points), k-NN lists (indices in list order, distances) and
    normals of the restatement == the vendored ANN 1.1.1 + newmat, bit for bit."""
    _need_ref(orc)
    p = _ann_clouds()[name]
    a, b = orc.AnnTree(p, "oracle"), orc.AnnTree(p, "ref")
    assert a.stats() == b.stats()
    sa, sb = a.structure(), b.structure()
    assert np.array_equal(sa[0], sb[0]) and np.array_equal(sa[2], sb[2])
    assert np.allclose(sa[1], sb[1], rtol=1e-13, atol=0)            # dumped with 15 digits
    rng = np.random.default_rng(1)
    q = np.concatenate([p, p[:500] + rng.normal(0, 3, (min(500, len(p)), 3))])
    for k, eps in ((10, 1.0), (10, 0.0), (1, 1.0), (5, 0.3), (len(p) if len(p) < 20 else 16, 2.0)):
        i1, d1 = a.ksearch(q, k, eps)
        i2, d2 = b.ksearch(q, k, eps)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    rp = [1.0, 2.0, 3.0]
    assert np.array_equal(orc.normals_apx_knn(p, 10, rp, 1.0), orc.normals_apx_knn(p, 10, rp, 1.0, "ref"))


def test_eigen3_equals_newmat(orc):
    _need_ref(orc)
    rng = np.random.default_rng(2)
    for trial in range(300):
        X = rng.normal(size=(10, 3)) * rng.uniform(0.01, 100, 3)
        if trial % 5 == 0:
            X[:, 2] = 0.0                                            # exactly planar neighbourhoods
        if trial % 7 == 0:
            X[:, 1] = X[:, 0]
        A = X.T @ X
        d1, u1 = orc.eigen3(A)
        d2, u2 = orc.eigen3(A, "ref")
        assert np.array_equal(d1, d2) and np.array_equal(u1, u2)


def test_ann_oracle_against_golden(orc):
    """K7 fixture (generated with the vendored ANN + newmat): k-NN lists and normals of the restatement."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    z = np.load(os.path.join(G, "k7_ann_normals.npz"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    for tag, pts in mg.k7_clouds().items():
        t = orc.AnnTree(pts)
        assert tuple(z[tag + "_stats"]) == t.stats()
        idx, _ = t.ksearch(pts, 10, 1.0)
        assert np.array_equal(idx, z[tag + "_knn"])
        nrm, knn = orc.normals_apx_knn(pts, 10, [0.0, 0.0, 0.0], 1.0, want_knn=True)
        assert np.array_equal(knn, z[tag + "_knn"]) and np.array_equal(nrm, z[tag + "_normals"])
        assert np.array_equal(orc.normals_from_knn(pts, knn, [0.0, 0.0, 0.0]), nrm)


def test_ann_search_errors(orc):
    t = orc.AnnTree(np.random.default_rng(0).uniform(-1, 1, (5, 3)))
    with pytest.raises(RuntimeError):
        t.ksearch(np.zeros((1, 3)), 6, 1.0)          # "Requesting more near neighbors than data points"


@pytest.mark.parametrize("group", [64, 7, 1])
def test_visiting_order_free_traversal_is_exact(orc, group):
    """The reference's answer is argmin over points with d2 < maxdist2 of (d2, rank in the query's own near-first
    depth-first order).  A walk in ANY other order returns the same index if it prunes with strict '>' and breaks
    exact distance ties by the path key (far-child bits, most significant = root) and the position in the bucket --
    shown here with several queries walking the tree together on clouds where ties are the rule."""
    rng = np.random.default_rng(group)
    g = np.stack(np.meshgrid(np.arange(14.0), np.arange(14.0), np.arange(14.0)), -1).reshape(-1, 3)
    d = rng.uniform(-10, 10, (1500, 3))
    clouds = {"lattice": (g, np.concatenate([g[:1500] + 0.5, g[:1500] + [0.5, 0, 0], g[:800], g[:800] + [0, 0.5, 0.5]]), 4.0),
              "duplicates": (np.concatenate([d, d, d]), np.concatenate([d, d + rng.normal(0, 0.3, d.shape)]), 1.0),
              "uniform": (rng.uniform(-50, 50, (20000, 3)), rng.uniform(-55, 55, (4000, 3)), 25.0)}
    for name, (m, q, md2) in clouds.items():
        for bucket in (1, 20):
            T = orc.Tree(m, bucket)
            oi, od = T.find_closest(q, md2)
            pi, pd, _ = T.packet_find_closest(q, md2, group)
            assert np.array_equal(pi, oi) and np.array_equal(pd, od), (name, bucket)


def test_b1_correspondence_indices_oracle(orc):
    """B1 replayed on the oracle: at every ICP iteration the index array of the whole data scan hashes to what the
    REFERENCE KDtreeIndexed returned (tests/golden/b1_dat_icp_idx.json), first / last 100 indices included."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    bi = json.load(open(os.path.join(G, "b1_dat_icp_idx.json")))
    S = [io.OScan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k]) for k in range(3)]
    for pr, pi in zip(b1["pairs"], bi["pairs"]):
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        inv, _ = orc.m4inv(S[i - 1].dalignxf)
        for it, a in enumerate(pr["alignxf"]):
            q = S[i].xyz.copy(); orc.transform_points(inv, q)
            idx, _ = S[i - 1].tree().find_closest(q, 625.0)
            row = pi["iterations"][it]
            assert int((idx >= 0).sum()) == row["found"] and "0x%x" % orc.k5_hash(idx) == row["hash"], (i, it)
            if "first100" in row:
                assert idx[:100].tolist() == row["first100"] and idx[-100:].tolist() == row["last100"]
            S[i].transform(np.array(a))


def test_deferred_quick_check_gives_the_references_answer_on_the_roundings(orc):
    """Round 5: the ARGUMENT behind the GPU kernel's deferred quick check, tried on the CPU where the reference's
    own walk is at hand.  oracle.c states the rule on the reference's tree with the check skipped at EVERY node: start from
    d2(q, previous hit) + 2 tie, mark a query whose closest_d2 ever improved by `tie` or less, search those again with every
    check.  Index and d2 must be FindClosest's on clouds built to sit on the roundings the bound is about: a lattice with a
    non-dyadic pitch and queries displaced along one axis (mid-way between neighbours: ties to within an ulp; and right on a
    neighbour), twins 1e-12 .. 1e-11 apart, the same far from the origin (1e5: the bound's absolute term), and a plain noisy
    pair; with the true neighbour, a wrong neighbour and a far point as the previous hit.  (These clouds give the same answers
    with tie = 0 as well -- for the reference's own check to hide a point, a box face, a near-tie and an axis-aligned offset have
    to coincide to the last bit; the bound is what the proof needs, and what this test shows is that the rule built on it
    costs no answer while its second searches run by the thousand.)"""
    rng = np.random.default_rng(5)
    total_redo = 0
    for case in range(8):
        off = 0.0 if case < 4 else 1.0e5
        if case % 4 == 0:        # lattice, pitch 0.1 (not a dyadic rational), queries on the axis between / on neighbours
            g = np.arange(24) * 0.1
            m = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + off
            pick = rng.integers(0, len(m), 6000)
            q = m[pick].copy()
            ax = rng.integers(0, 3, len(q))
            q[np.arange(len(q)), ax] += rng.choice([0.05, -0.05, 0.1, 0.025, 1e-13, 0.0], len(q))
        elif case % 4 == 1:      # twins
            m = rng.uniform(-50, 50, (20000, 3)) + off
            tw = rng.choice(len(m), 4000, replace=False)
            m[tw[:2000]] = m[tw[2000:]] + rng.uniform(1e-12, 1e-11, (2000, 3)) * rng.choice([-1.0, 1.0], (2000, 3))
            q = m[rng.integers(0, len(m), 8000)] + rng.normal(0, 0.3, (8000, 3))
        elif case % 4 == 2:      # exact duplicates and coplanar points
            m = rng.uniform(-50, 50, (20000, 3)) + off
            m[:5000, 2] = off + 1.0
            m[rng.integers(0, len(m), 3000)] = m[rng.integers(0, len(m), 3000)]
            q = m[rng.integers(0, len(m), 8000)] + rng.normal(0, 0.5, (8000, 3)) * np.array([1.0, 1.0, 0.0])
        else:                    # the ordinary case: a noisy copy
            m = rng.uniform(-100, 100, (30000, 3)) + off
            q = m[rng.integers(0, len(m), 8000)] + rng.normal(0, 1.0, (8000, 3))
        t = orc.Tree(m, 20 if case % 2 else 5)
        maxd2 = 25.0 if case % 4 else 0.0625
        absmax = float(np.abs(m).max())
        ref_i, ref_d = t.find_closest(q, maxd2)
        assert (ref_i >= 0).sum() > len(q) // 3
        # previous hits: the answer itself; a neighbour of it in the array (a wrong but near point); anything
        for warm in (ref_i, np.where(ref_i >= 0, (ref_i + 1) % len(m), -1), rng.integers(0, len(m), len(q)).astype(np.int32)):
            i2, d2, redo = t.find_closest_deferred(q, maxd2, warm, absmax)
            assert np.array_equal(i2, ref_i), (case, int((i2 != ref_i).sum()))
            assert np.array_equal(d2, ref_d), case
            total_redo += redo
    assert total_redo > 1000        # the thin path ran (lattices and twins make thin acceptances by the thousand)


# ---- the -m / -M range filter against the reference's PointFilter (round 6) -----------------------
@pytest.mark.parametrize("rmax,rmin", [(-1, -1), (500, -1), (-1, 100), (500, 100), (1000000, 1), (123.456789, 10.5), (0, 0)])
def test_range_filter_of_the_uos_reader_equals_the_references_pointfilter(tdtk, orc, tmp_path, rmax, rmin):
    """tdtk_io_read_uos's -m / -M filter (io.cpp) against the reference's own compiled PointFilter (pointfilter.cc in
    oracle/_ref, driven as BasicScan drives it: setRange(max, min) then check() per point) on dat/ scan 0 (81 360 points,
    the committed fixture) and on points placed exactly on, one ulp inside and one ulp outside both radii -- including a
    non-integer range, which the reference filters at six significant digits (it passes the range through a stringstream)."""
    _need_ref(orc)
    z = np.load(os.path.join(G, "dat_scans.npz"))
    pts = [z["scan000"]]
    for r in (rmax, rmin):
        if r > 0:
            rr = float("%g" % r)
            edge = np.array([rr, np.nextafter(rr, 0.0), np.nextafter(rr, np.inf), float(r)])
            e = np.zeros((12, 3))
            for ax in range(3):
                e[4 * ax:4 * ax + 4, ax] = edge
            d = np.array([3.0, 4.0, 12.0]) / 13.0                       # (3, 4, 12) / 13: a direction whose norm is exact
            pts += [e, -e, np.outer(edge, d)]
    pts = np.concatenate(pts)
    f = tmp_path / "scan000.3d"
    with open(f, "w") as fh:
        for p in pts:
            fh.write("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))
    got = tdtk.read_uos(f, rmax, rmin)
    keep = orc.ref_point_filter_range(pts, rmax, rmin)
    assert np.array_equal(got, pts[keep]), (len(got), int(keep.sum()))
    if rmax <= 0 and rmin <= 0:
        assert len(got) == len(pts)
