"""GPU tier (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
inputs, against the committed golden fixtures (generated with the reference's own TUs), and --
at BASELINE.json's full 1M-vs-1M size -- through hashes and size-independent properties.

Bar: correspondence indices and squared distances BIT-EXACT; sums / poses within the tolerance
BASELINE.json states (1e-5 relative on the pose; we assert much tighter where the arithmetic
allows)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")

POSE_RTOL = 1e-5          # BASELINE.json north_star: pose within 1e-5 relative


def _rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return np.abs(a - b).max() / max(1e-300, np.abs(b).max())


# ---- known-answer tests of testing/kdtree/kdtree.cc:20-99 through the C ABI -------------------
@pytest.mark.parametrize("pts,want", [
    ([[2.0, 0.0, 0.0]], None), ([[1.99999999999, 0.0, 0.0]], 0), ([[1.5, 0.0, 0.0], [1.0, 0.0, 0.0]], 1)])
def test_kat_find_closest(tdtk, gpu, pts, want):
    assert tdtk.KDtree(np.array(pts)).FindClosest([0.0, 0.0, 0.0], 4.0) == want


@pytest.mark.parametrize("pts,want", [
    ([[1.0, 2.0, 0.0]], None), ([[1.0, 1.99999999999, 0.0]], 0),
    ([[0.5, 0.1, 0.0], [1.0, 0.0, 0.0]], 1), ([[-1.0, 0.0, 0.0]], 0)])
def test_kat_find_closest_along_dir(tdtk, gpu, pts, want):
    assert tdtk.KDtree(np.array(pts)).FindClosestAlongDir([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], 4.0) == want


def test_errors(tdtk, gpu):
    with pytest.raises(RuntimeError):
        tdtk.KDtree(np.zeros((0, 3)))                     # kdTreeImpl.h:86-88
    kd = tdtk.KDtree(np.random.default_rng(0).uniform(-1, 1, (50, 3)))
    idx, d2 = kd.FindClosestBatch(np.zeros((0, 3)), 1.0)   # empty batch
    assert len(idx) == 0
    with pytest.raises(tdtk.TdtkError) as e:                # resident-scan pass: rnd > 1 unsupported
        tdtk.Scan.getPtPairs(tdtk.Scan([0, 0, 0], [0, 0, 0], np.zeros((4, 3))),
                             tdtk.Scan([0, 0, 0], [0, 0, 0], np.zeros((4, 3))), rnd=5)
    assert e.value.code == -5
    with pytest.raises(tdtk.TdtkError):
        kd.getPtPairs(tdtk.M4identity(), np.zeros((4, 3)), pairing_mode=2)   # needs normals


def _clouds():
    rng = np.random.default_rng(3)
    uni = rng.uniform(-100, 100, (30000, 3))
    dup = uni.copy(); dup[1000:1400] = dup[0:400]
    clu = np.concatenate([rng.normal(c, 0.003, (300, 3)) for c in rng.uniform(-50, 50, (40, 3))])
    plane = rng.uniform(-100, 100, (20000, 3)); plane[:, 2] = 0.0
    tiny = rng.uniform(-1, 1, (7, 3))
    grid = np.stack(np.meshgrid(*[np.arange(12.0)] * 3), -1).reshape(-1, 3)
    line = np.zeros((5000, 3)); line[:, 0] = np.sort(rng.uniform(0, 1e4, 5000)) ** 2 / 1e4   # skewed -> deep tree
    return {"uniform": uni, "duplicates": dup, "clusters": clu, "plane": plane, "tiny": tiny, "grid": grid,
            "line": line}


@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "tiny", "grid", "line"])
@pytest.mark.parametrize("bucket", [1, 20])
def test_find_closest_bit_exact(tdtk, orc, gpu, name, bucket):
    m = _clouds()[name]
    rng = np.random.default_rng(11)
    q = np.concatenate([m[rng.integers(0, len(m), 4000)] + rng.normal(0, 0.5, (4000, 3)), m[:700],
                        rng.uniform(-120, 120, (2000, 3))])
    if name == "grid":
        q = np.concatenate([q, m[:700] + 0.5])
    kd, T = tdtk.KDtree(m, bucket), orc.Tree(m, bucket)
    inf = kd.info(); st = T.stats()
    assert (inf["n_internal"], inf["n_leaves"], inf["max_depth"]) == (st["internal"], st["leaves"], st["depth"])
    for md2 in (0.25, 25.0, 1e18):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2)
        assert np.array_equal(idx, oi), (name, bucket, md2)
        assert np.array_equal(d2, od2)
        assert kd.count_visits(q, md2) == T.find_closest(q, md2, 1, True)[2]    # same traversal, node for node
    d = rng.normal(size=(len(q), 3)); d /= np.linalg.norm(d, axis=1)[:, None]
    idx, d2 = kd.FindClosestAlongDirBatch(q[:1500], d[:1500], 4.0)
    oi, od2 = T.find_closest_along_dir(q[:1500], d[:1500], 4.0)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od2)


@pytest.mark.parametrize("nq", [6000, 300000])
def test_box_shortcut_gives_way_outside_float_range(tdtk, orc, gpu, nq):
    """The fp32 box test may only decide what it can decide rigorously (kernels.hip, BoxF32): a query with a component
    beyond float range, an infinite or a NaN component, or a search radius beyond FLT_MAX must take the exact fp64 test,
    so that not only the hits but the VISITS stay the reference's (round-2 advice: +inf pruned without the exact test,
    fmaxf dropped a NaN component).  Small batch = lane-group kernel, large batch = persistent-lane kernel."""
    rng = np.random.default_rng(77)
    m = rng.uniform(-1000, 1000, (50000, 3))
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    q = m[rng.integers(0, len(m), nq)] + rng.normal(0, 2.0, (nq, 3))
    weird = [1e39, -1e39, 3.3e38, -3.39e38, 1e300, np.inf, -np.inf, np.nan, 5e38, 1e45]
    rows = rng.choice(nq, size=nq // 10, replace=False)
    for k, r in enumerate(rows):
        q[r, k % 3] = weird[k % len(weird)]
        if k % 7 == 0:
            q[r, (k + 1) % 3] = weird[(k + 3) % len(weird)]
    for md2 in (25.0, 1e39, 1e300):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi), md2
        assert np.array_equal(d2, od2, equal_nan=True), md2
        assert kd.count_visits(q, md2) == T.find_closest(q, md2, 8, True)[2], md2


@pytest.mark.parametrize("case", ["one_cell", "outside", "lattice", "flat"])
def test_sixteen_bit_bucket_shadows_only_reject_on_proof(tdtk, orc, gpu, case):
    """Round 5: the persistent-lane kernel filters buckets on a 16-bit shadow (kernels.hip, bucket_scan_q16: ONE grid of
    65536 cells per axis over the root box).  Clouds on which that grid says little or nothing -- a cluster a million times
    smaller than the box (every point of it in one cell), queries far outside the box (quantised onto its faces: only the
    lower bound of a distance survives that), points exactly on cell boundaries with exact ties, a cloud with no extent along
    one axis -- must come out index for index and bit for bit like the oracle's, visit counters included; 300K queries =
    the persistent-lane kernel."""
    rng = np.random.default_rng({"one_cell": 1, "outside": 2, "lattice": 3, "flat": 4}[case])
    nq = 300000
    if case == "one_cell":
        m = np.concatenate([rng.uniform(-5e-4, 5e-4, (120000, 3)) + [3.0, -2.0, 1.0], rng.uniform(-1e3, 1e3, (400, 3))])
        m[1000:1200] = m[0:200]
        q = m[rng.integers(0, 120000, nq)] + rng.normal(0, 2e-5, (nq, 3))
        radii = (1e-8, 1e-4, 1e12)
    elif case == "outside":
        m = rng.uniform(-100, 100, (100000, 3))
        q = rng.uniform(-100, 100, (nq, 3))
        far = rng.integers(0, nq, nq // 2)
        q[far, rng.integers(0, 3, len(far))] += rng.choice([-1.0, 1.0], len(far)) * rng.uniform(150, 1e5, len(far))
        radii = (400.0, 1e8, 1e30)
    elif case == "lattice":
        g = np.arange(40, dtype=np.float64)
        m = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) * (65535.0 / 39.0 / 65535.0 * 39.0)   # 64000 points, unit lattice
        m = np.concatenate([m, m[:5000]])                       # + exact duplicates
        q = m[rng.integers(0, len(m), nq)] + rng.choice([0.0, 0.5, -0.5, 0.25], (nq, 3))      # ties between lattice neighbours
        radii = (0.2, 1.0, 1e6)
    else:
        m = rng.uniform(-50, 50, (90000, 3)); m[:, 1] = 7.0     # no extent along y
        q = m[rng.integers(0, len(m), nq)] + rng.normal(0, 0.3, (nq, 3))
        radii = (0.05, 9.0, 1e20)
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    assert kd.verify() == [0, 0, 0, 0]
    for md2 in radii:
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi), (case, md2, int((idx != oi).sum()))
        assert np.array_equal(d2, od2), (case, md2)
        assert kd.count_visits(q, md2) == T.find_closest(q, md2, 8, True)[2], (case, md2)


@pytest.mark.parametrize("nq", [20000, 300000])
@pytest.mark.parametrize("offset", [1.0e6, 1.0e8, 1.0e9])
def test_fp32_shortcuts_on_clouds_far_from_the_origin(tdtk, orc, gpu, nq, offset):
    """Round 4: a cloud offset by 1e6 .. 1e9 with centimetre structure -- coordinates whose fp32 shadows
    cannot tell neighbours apart (at 1e9 the fp32 spacing is 64 units: every point of a bucket has the same shadow).  The
    box shortcut and the bucket filter may then decide almost nothing and must hand every such decision to the exact fp64
    path: indices, squared distances and visit counters are the oracle's, for the lane-group kernels (20K queries) and
    the persistent-lane kernel (300K)."""
    rng = np.random.default_rng(int(offset) % 1000 + nq)
    base = np.array([offset, -0.5 * offset, 0.25 * offset])
    m = base + rng.uniform(-3.0, 3.0, (60000, 3))            # a 6 m cube, centimetre-to-metre structure
    m[2000:2400] = m[0:400]                                    # exact duplicates
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    assert kd.verify() == [0, 0, 0, 0]
    q = m[rng.integers(0, len(m), nq)] + rng.normal(0, 0.02, (nq, 3))
    q[: nq // 20] += rng.normal(0, 5.0, (nq // 20, 3))        # some queries outside the cube
    for md2 in (0.01, 4.0, 1e30):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi), (offset, md2)
        assert np.array_equal(d2, od2), (offset, md2)
        assert kd.count_visits(q, md2) == T.find_closest(q, md2, 8, True)[2], (offset, md2)


@pytest.mark.parametrize("nq", [20000, 300000])
def test_fp32_shortcuts_with_model_coordinates_beyond_float_range(tdtk, orc, gpu, nq):
    """Round 4: the MODEL holds coordinates fp32 cannot represent -- beyond FLT_MAX (their shadows are
    +-inf), within a rounding step of it (3.4028235e38 rounds up to inf), and huge-but-finite ones whose differences
    overflow fp32 --, so node boxes and shadow groups carry inf / NaN-producing values.  Nothing may be decided from
    those: same indices, distances and visit counters as the oracle, with queries both ordinary and huge."""
    rng = np.random.default_rng(31 + nq)
    m = rng.uniform(-1000, 1000, (50000, 3))
    wild = [1e39, -1e39, 3.4028235e38, -3.4028236e38, 3.3e38, -3.0e38, 1e300, -1e305, 2e38, 1.7e38]
    rows = rng.choice(len(m), 600, replace=False)
    for k, r in enumerate(rows):
        m[r, k % 3] = wild[k % len(wild)]
        if k % 5 == 0:
            m[r, (k + 1) % 3] = wild[(k + 4) % len(wild)]
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    assert kd.verify() == [0, 0, 0, 0]
    q = m[rng.integers(0, len(m), nq)] + rng.normal(0, 2.0, (nq, 3))      # includes queries next to the wild points
    q[: nq // 50, 0] = 2.5e38
    for md2 in (25.0, 1e39, 1e300):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi), md2
        assert np.array_equal(d2, od2, equal_nan=True), md2
        assert kd.count_visits(q, md2) == T.find_closest(q, md2, 8, True)[2], md2


def test_leaf_table_mode(tdtk, orc, gpu):
    """bits(M) + bits(max leaf) > 30 switches child references to the leaf table."""
    rng = np.random.default_rng(5)
    m = rng.uniform(-1000, 1000, (1100000, 3))
    m[5000:5600] = m[4999]                               # a 601-point degenerate bucket
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    assert kd.info()["max_leaf_points"] >= 601
    q = np.concatenate([rng.uniform(-1000, 1000, (20000, 3)), m[4990:5010] + 0.001])
    for md2 in (625.0, 1e18):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od2)


def test_k4_fixture(tdtk, gpu):
    z = np.load(os.path.join(G, "k4_random.npz"))
    kd = tdtk.KDtree(z["model"], 20)
    for q, e, md2 in zip(z["queries"], z["expected"], z["maxdist2"]):
        assert np.array_equal(kd.FindClosestBatch(q, md2)[0], e)


@pytest.fixture(scope="module")
def k5(orc):
    k = json.load(open(os.path.join(G, "k5_hashes.json")))
    M = k["M"]
    s = orc.gen_mt64_uniform(k["seed"], 6 * M, k["lo"], k["hi"])
    return k, s[:3 * M].reshape(M, 3).copy(), s[3 * M:].reshape(M, 3).copy()


def test_k5_full_size_hashes(tdtk, orc, gpu, k5):
    """BASELINE configs[1] size: 1M-vs-1M, hashes of the REFERENCE KDtree's index stream."""
    k, m, q = k5
    kd = tdtk.KDtree(m, k["bucket"])
    inf = kd.info()
    assert (inf["n_internal"], inf["n_leaves"], inf["max_depth"]) == (k["tree"]["internal"], k["tree"]["leaves"], k["tree"]["depth"])
    for c in k["cases"]:
        idx, d2 = kd.FindClosestBatch(q, c["maxdist2"])
        assert int((idx >= 0).sum()) == c["found"]
        assert "0x%x" % orc.k5_hash(idx) == c["hash"]
        assert idx[:32].tolist() == c["first32"] and idx[-32:].tolist() == c["last32"]
        cnt = kd.count_visits(q, c["maxdist2"])
        np.testing.assert_allclose(np.array(cnt) / k["M"], c["visits_per_query"], rtol=1e-12)
        # size-independent properties: reported d2 is the true distance to the reported point,
        # nothing within the radius is closer (brute force on a sample), strict '<' radius
        f = idx >= 0
        dd = m[idx[f]] - q[f]
        assert np.array_equal(dd[:, 0] * dd[:, 0] + dd[:, 1] * dd[:, 1] + dd[:, 2] * dd[:, 2], d2[f])
        assert (d2[f] < c["maxdist2"]).all() and (d2[~f] == c["maxdist2"]).all()
        for i in np.random.default_rng(1).integers(0, k["M"], 300):
            e = m - q[i]
            b2 = (e[:, 0] * e[:, 0] + e[:, 1] * e[:, 1] + e[:, 2] * e[:, 2]).min()
            assert (b2 == d2[i]) if b2 < c["maxdist2"] else (idx[i] == -1)


def test_k5_idempotence_and_presorted_device_path(tdtk, gpu, k5):
    """Querying the model with itself returns every point (d2 == 0); the device-pointer entry
    (inputs resident in HBM) gives the same answers with and without its own binning pass."""
    import ctypes as C
    import torch
    k, m, q = k5
    kd = tdtk.KDtree(m, 20)
    idx, d2 = kd.FindClosestBatch(m, 1e-9)
    assert np.array_equal(idx, np.arange(len(m), dtype=np.int32)) and (d2 == 0).all()
    tq = torch.from_numpy(q).cuda()
    out_i = torch.empty(len(q), dtype=torch.int32, device="cuda")
    out_d = torch.empty(len(q), dtype=torch.float64, device="cuda")
    ref_idx, ref_d2 = kd.FindClosestBatch(q, 625.0)
    for presorted in (0, 1):
        out_i.fill_(-7)
        rc = tdtk.lib().tdtk_find_closest_dev(kd._h, C.c_void_p(tq.data_ptr()), len(q), 625.0,
                                              C.c_void_p(out_i.data_ptr()), C.c_void_p(out_d.data_ptr()),
                                              presorted, None)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(out_i.cpu().numpy(), ref_idx) and np.array_equal(out_d.cpu().numpy(), ref_d2)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_get_pt_pairs_vs_oracle(tdtk, orc, gpu, mode):
    """SearchTree::getPtPairs with a non-trivial Source->dalignxf: indices and the PtPair list
    bit-exact, the fused sums to rounding."""
    rng = np.random.default_rng(8)
    m = rng.uniform(-200, 200, (60000, 3)); m[100:160] = m[0:60]
    A = tdtk.EulerToMatrix4([12.0, -7.0, 3.0], [0.03, -0.02, 0.04])
    d = m[rng.permutation(len(m))[:40000]].copy(); orc.transform_points(A, d)
    d += rng.normal(0, 0.4, d.shape)
    nr = rng.normal(size=d.shape)
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    md2 = 4.0 if mode != 1 else 0.5
    got = kd.getPtPairs(A, d, nr, 100, 39000, max_dist_match2=md2, pairing_mode=mode,
                        want=tdtk.WANT_APX | tdtk.WANT_NAPX | tdtk.WANT_LUM)
    ref = T.get_pt_pairs(A, d, nr, 100, 39000, mode, md2)
    assert got["n"] == ref["n"] and got["n"] > 1000
    assert np.array_equal(got["idx"], ref["idx"])
    assert np.array_equal(got["p1"], ref["p1"]) and np.array_equal(got["p2"], ref["p2"])
    if mode == 0:
        # the reference leaves PtPair's normal uninitialised in CLOSEST_POINT mode
        # (searchTree.cc:126-131); we hand the NAPX block the normalised data normal instead
        nn = nr[100:39000][ref["idx"] >= 0]
        ref["pn"] = nn / np.sqrt(nn[:, 0] * nn[:, 0] + nn[:, 1] * nn[:, 1] + nn[:, 2] * nn[:, 2])[:, None]
    assert np.array_equal(got["pn"], ref["pn"])
    n = ref["n"]
    assert _rel(got["sum"], ref["sum"]) < 1e-12
    assert _rel(got["centroid_m"], ref["centroid_m"] / n) < 1e-13
    assert _rel(got["centroid_d"], ref["centroid_d"] / n) < 1e-13
    cm, cd = ref["p1"].mean(0), ref["p2"].mean(0)
    Si = (ref["p1"] - cm).T @ (ref["p2"] - cd)
    assert _rel(got["Si"], Si.reshape(9)) < 1e-11
    # APX / NAPX / LUM blocks against direct evaluation on the oracle's pair list
    p12, p2c, pn = ref["p1"] - ref["p2"], ref["p2"] - cd, ref["pn"]
    A6 = [(p2c[:, 1] ** 2 + p2c[:, 2] ** 2).sum(), -(p2c[:, 0] * p2c[:, 1]).sum(), -(p2c[:, 0] * p2c[:, 2]).sum(),
          (p2c[:, 0] ** 2 + p2c[:, 2] ** 2).sum(), -(p2c[:, 1] * p2c[:, 2]).sum(), (p2c[:, 0] ** 2 + p2c[:, 1] ** 2).sum()]
    B3 = [(p12[:, 2] * p2c[:, 1] - p12[:, 1] * p2c[:, 2]).sum(), (p12[:, 0] * p2c[:, 2] - p12[:, 2] * p2c[:, 0]).sum(),
          (p12[:, 1] * p2c[:, 0] - p12[:, 0] * p2c[:, 1]).sum()]
    assert _rel(got["apx_A"], A6) < 1e-11 and np.abs(np.array(got["apx_B"]) - B3).max() < 1e-7 * np.abs(A6).max() ** 0.5
    v = np.hstack([np.cross(p2c, pn), pn])
    AA = v.T @ v
    assert _rel(got["napx_A"], AA[np.triu_indices(6)]) < 1e-10
    assert np.abs(np.array(got["napx_B"]) - v.sum(0)).max() < 1e-8 * np.abs(AA).max() ** 0.5
    assert _rel(got["napx_sum"], ((p12 * pn).sum(1) ** 2).sum()) < 1e-11
    u = (ref["p1"] + ref["p2"]) / 2.0
    x, y, z = u.T; dx, dy, dz = p12.T
    lum = [x.sum(), y.sum(), z.sum(), (x * x + y * y).sum(), (x * x + z * z).sum(), (y * y + z * z).sum(),
           (x * y).sum(), (x * z).sum(), (y * z).sum(), dx.sum(), dy.sum(), dz.sum(),
           (-z * dy + y * dz).sum(), (-y * dx + x * dy).sum(), (z * dx - x * dz).sum()]
    assert np.abs((np.array(got["lum"]) - lum) / (np.abs(lum) + np.abs(lum).max() * 1e-6)).max() < 1e-9


def test_get_pt_pairs_rnd_subsampling(tdtk, orc, gpu):
    """-R 5 (README config 1): keep-mask = (int)(rnd*rand()/(RAND_MAX+1.0)) == 0 per candidate in
    index order (searchTree.cc:118, globals.icc:607-610), serial-build semantics."""
    import ctypes as C
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    rng = np.random.default_rng(12)
    m = rng.uniform(-50, 50, (20000, 3))
    q = m[rng.permutation(len(m))[:9000]] + rng.normal(0, 0.2, (9000, 3))
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    libc.srand(1234)
    got = kd.getPtPairs(tdtk.M4identity(), q, rnd=5, max_dist_match2=4.0)
    libc.srand(1234)
    keep = np.array([int(5.0 * libc.rand() / (2147483647 + 1.0)) == 0 for _ in range(len(q))])
    assert 0.15 < keep.mean() < 0.25
    ref = T.get_pt_pairs(np.eye(4).reshape(16), q[keep], maxdist2=4.0)
    assert got["n"] == ref["n"] and got["n_queries"] == int(keep.sum())
    assert np.array_equal(got["idx"][keep], ref["idx"]) and (got["idx"][~keep] == -1).all()
    assert np.array_equal(got["p1"], ref["p1"]) and np.array_equal(got["p2"], ref["p2"])
    assert abs(got["sum"] - ref["sum"]) <= 1e-12 * ref["sum"]


@pytest.mark.parametrize("n", [20000, 150000, 300000])
def test_rnd_in_the_resident_loop_equals_the_stepped_loop(tdtk, orc, gpu, n):
    """Round 5: `-R 5` inside the device-resident loop (tdtk_icp_match_rnd: per iteration the keep-mask is
    drawn on the host -- one std::rand() per point in index order, searchTree.cc:116-118 -- and sent as bits; the search
    kernels skip what was not drawn) against the loop driven from the host, one tdtk_get_pt_pairs(rnd) per iteration, with
    libc seeded identically: the same pairs per iteration, the same RMS and alignxf to 1e-9, the same final pose -- for the
    three search-kernel families (four lanes per query, one query per lane, persistent lanes) -- and the same number of
    std::rand() draws consumed (the next draw of the process is the same)."""
    import ctypes as C
    import bench
    libc = C.CDLL(None)
    libc.rand.restype = C.c_int
    m, d, T = bench.make_icp_pair(n, seed=5)
    out = []
    for stepped in (False, True):
        model = tdtk.Scan([0, 0, 0], [0, 0, 0], m)
        data = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 7, quiet=True, rnd=5, epsilonICP=1e-9)
        icp.stepped_rnd = stepped
        libc.srand(4711)
        it = icp.match(model, data)
        out.append((it, icp.last["trace"].copy(), data.get_transMat().copy(), data.get_xyz_reduced(), libc.rand()))
    (it0, tr0, tm0, x0, r0), (it1, tr1, tm1, x1, r1) = out
    assert it0 == it1 and len(tr0) == len(tr1) == it0 + 1
    assert np.array_equal(tr0[:, 0], tr1[:, 0])                               # pairs per iteration
    assert 3 < tr0[0, 0] < 0.23 * n                                            # at most about a fifth of the points are candidates
    np.testing.assert_allclose(tr0[:, 1:], tr1[:, 1:], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(tm0, tm1, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(x0, x1, rtol=0, atol=1e-7)
    assert r0 == r1


def test_scan_transform_bit_exact(tdtk, orc, gpu):
    rng = np.random.default_rng(2)
    p = rng.uniform(-500, 500, (50000, 3)); nr = rng.normal(size=p.shape)
    s = tdtk.Scan([1.0, 2.0, 3.0], [0.1, -0.2, 0.3], p, nr)
    from oracle import icp_oracle as io
    o = io.OScan([1.0, 2.0, 3.0], [0.1, -0.2, 0.3], p, nr)
    assert np.array_equal(s.get_xyz_reduced(), o.xyz)
    for k in range(3):          # incremental, in place, k-fold (SURVEY N-a)
        A = tdtk.EulerToMatrix4([0.3 * k, -0.1, 0.2], [0.01, 0.02 * k, -0.01])
        s.transform(A); o.transform(A)
    assert np.array_equal(s.get_xyz_reduced(), o.xyz)
    assert np.array_equal(s.transMat, o.transMat) and np.array_equal(s.dalignxf, o.dalignxf)
    out = np.empty((50000, 3)); on = np.empty((50000, 3))
    import ctypes as C
    dp = C.POINTER(C.c_double)
    assert tdtk.lib().tdtk_scan_download(s.handle, out.ctypes.data_as(dp), on.ctypes.data_as(dp)) == 0
    assert np.array_equal(on, o.normals)


def _dat_scans(cls, z, **kw):
    return [cls(z["pose%03d" % k][:3], z["pose%03d" % k][3:], z["scan%03d" % k], **kw) for k in range(3)]


def test_dat_sequential_icp_matches_reference_trace(tdtk, gpu):
    """`bin/slam6D -d 25 -i 50 dat` (QUAT, --epsICP 1e-5): per-iteration pair counts exact, RMS
    and the final 6-DoF pose within 1e-5 relative of the trace generated with the reference's
    own minimizer TU (tests/golden/b1_dat_icp.json)."""
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    S = _dat_scans(tdtk.Scan, z)
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 50, quiet=True, epsilonICP=1e-5)
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        it = icp.match(S[i - 1], S[i])
        tr = icp.last["trace"]
        assert it == pr["iter"]
        assert [int(r[0]) for r in tr] == [t[0] for t in pr["trace"]]
        np.testing.assert_allclose(tr[:, 1], [t[1] for t in pr["trace"]], rtol=1e-9)
        assert _rel(S[i].get_transMat(), pr["final_transMat"]) < POSE_RTOL
        assert _rel(S[i].get_transMat(), pr["final_transMat"]) < 1e-9     # what we actually reach


@pytest.mark.parametrize("algo", [1, 2, 6, 3, 4, 5, 7, 8, 9])
def test_icp_vs_oracle_loop(tdtk, orc, gpu, algo):
    """icp6D::match for every point-to-point minimizer (-a 1..9) against the oracle loop on the bundled
    scans: QUAT / SVD / APX (the OpenMP-capable ones) and ORTHO / DUAL / HELIX / LUMEULER / LUMQUAT /
    QUAT_SCALE, which the device loop computes from the second-moment block."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    S, O = _dat_scans(tdtk.Scan, z), _dat_scans(io.OScan, z)
    cls = {1: tdtk.icp6D_QUAT, 2: tdtk.icp6D_SVD, 6: tdtk.icp6D_APX, 3: tdtk.icp6D_ORTHO, 4: tdtk.icp6D_DUAL,
           5: tdtk.icp6D_HELIX, 7: tdtk.icp6D_LUMEULER, 8: tdtk.icp6D_LUMQUAT, 9: tdtk.icp6D_QUAT_SCALE}[algo]
    icp = tdtk.icp6D(cls(True), 25.0, 12, quiet=True, epsilonICP=1e-5)
    S[1].mergeCoordinatesWithRoboterPosition(S[0]); O[1].mergeCoordinatesWithRoboterPosition(O[0])
    it = icp.match(S[0], S[1])
    oit, otr = io.match(O[0], O[1], algo, 625.0, 12, 1e-5)
    assert it == oit
    assert [int(r[0]) for r in icp.last["trace"]] == [t[0] for t in otr]
    tol = 1e-9 if algo in (1, 2, 6) else 1e-7
    np.testing.assert_allclose(icp.last["trace"][:, 1], [t[1] for t in otr], rtol=tol)
    assert _rel(S[1].get_transMat(), O[1].transMat) < tol
    assert np.abs(S[1].get_xyz_reduced() - O[1].xyz).max() < (1e-8 if algo in (1, 2, 6) else 1e-5)


def _base_sums(p1, p2, shift):
    """the 17 base sums of a pass about `shift` (kernels.h ACC_N .. ACC_P) and the PairSums tdtk_align reads (finish_sums)"""
    import sys
    capi = sys.modules["3dtk_amd._capi"]
    m, d = p1 - shift, p2 - shift
    n = float(len(p1))
    acc = np.zeros(17)
    acc[0] = n
    acc[1] = ((p1 - p2) ** 2).sum()
    acc[2:5] = m.sum(0); acc[5:8] = d.sum(0)
    acc[8:17] = (m[:, :, None] * d[:, None, :]).sum(0).reshape(9)
    s = capi.PairSums()
    s.n_queries = len(p1); s.n = len(p1); s.sum = acc[1]
    for a in range(3):
        s.centroid_m[a] = shift[a] + acc[2 + a] / n if n else 0.0
        s.centroid_d[a] = shift[a] + acc[5 + a] / n if n else 0.0
    for a in range(3):
        for b in range(3):
            s.Si[a * 3 + b] = acc[8 + a * 3 + b] - acc[2 + a] * acc[5 + b] / n if n else 0.0
    return acc, s


def test_device_solver_equals_the_host_solver(tdtk, gpu, lab):
    """Round 6, lab (NEGATIVES.md, "the small-scan loop without the host"): the solve every workgroup makes in the prologue of a
    launch of the host-free ICP loop (loop_dev.h: Newton on the characteristic polynomial + inverse iteration) run once on the
    sums of a pass (tdtk_lab_icp_device_solve, lab library) against tdtk_align (linalg.cpp: Jacobi) on the same sums:
    the K6 pose (rPos (1.5, -2, 0.7), rPosTheta (0.02, -0.03, 0.05), 1000 points), poses of every size from
    1e-9 rad to a half turn about tilted axes, far-off clouds (coordinates ~1e5), noise, exactly identical clouds (the zero
    matrix), fewer than four pairs (status 4) and a NaN among the sums (status 5: the host takes over)."""
    import sys
    capi = sys.modules["3dtk_amd._capi"]
    L = tdtk.lib()
    import ctypes as C
    rng = np.random.default_rng(7)

    L.tdtk_lab_icp_device_solve.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]

    def dev(acc, shift):
        xf = np.empty(16); rms = C.c_double(0.0); st = C.c_int(0)
        capi.check(L.tdtk_lab_icp_device_solve(gpu, capi.dptr(np.ascontiguousarray(acc)), capi.dptr(np.ascontiguousarray(shift, dtype=np.float64)),
                                           capi.dptr(xf), C.byref(rms), C.byref(st)))
        return xf, rms.value, st.value

    def host(s):
        xf = np.eye(4).reshape(16).copy(); rms = C.c_double(0.0)
        capi.check(L.tdtk_align(1, C.byref(s), capi.dptr(xf), C.byref(rms)))
        return xf, rms.value

    cases = [([1.5, -2.0, 0.7], [0.02, -0.03, 0.05], 100.0, 0.0, 1000)]
    for ang in (1e-9, 1e-6, 1e-3, 0.3, 1.5, 3.0, np.pi):
        cases.append((rng.normal(0, 50, 3), ang * np.array([0.6, -0.3, 0.74]), 1000.0, 0.5, 20000))
    cases.append(([3e4, -2e4, 1e4], [0.001, 0.002, -0.001], 1e5, 1.0, 50000))
    worst = 0.0
    for rPos, th, extent, noise, n in cases:
        T = tdtk.EulerToMatrix4(rPos, th)
        R = np.array([[T[0], T[4], T[8]], [T[1], T[5], T[9]], [T[2], T[6], T[10]]])
        d = rng.uniform(-extent, extent, (n, 3))
        m = d @ R.T + T[12:15] + rng.normal(0, noise, (n, 3)) if noise else d @ R.T + T[12:15]
        shift = m.mean(0) + rng.normal(0, 10, 3)
        acc, s = _base_sums(m, d, shift)
        xd, rd, st = dev(acc, shift)
        xh, rh = host(s)
        assert st == 1
        err = np.abs(xd - xh).max() / max(1.0, np.abs(xh).max())
        worst = max(worst, err)
        assert err < 1e-9, (rPos, th, err)
        assert abs(rd - rh) <= 1e-12 * max(1.0, rh)
        if not noise:
            assert np.abs(xd - T).max() < 1e-9 * max(1.0, np.abs(T).max())
    # identical clouds: Q = 0 up to rounding -- identity rotation either way
    d = rng.uniform(-10, 10, (500, 3))
    acc, s = _base_sums(d, d, np.zeros(3))
    xd, rd, st = dev(acc, np.zeros(3))
    xh, rh = host(s)
    assert st == 1 and np.abs(xd - xh).max() < 1e-9 and rd == 0.0
    # three pairs: icp6D.cc:235-243 ends the loop; a NaN among the sums: the host's turn
    acc, _ = _base_sums(d[:3], d[:3] + 0.1, np.zeros(3))
    assert dev(acc, np.zeros(3))[2] == 4
    acc, _ = _base_sums(d, d + 0.1, np.zeros(3))
    acc[9] = np.nan
    assert dev(acc, np.zeros(3))[2] == 5


def test_small_scan_loop_without_the_host_equals_the_stepped_loop(tdtk, gpu, lab):
    """Round 6, lab: tdtk_icp_match on a small scan with -a 1 and TDTK_ICP_DEVICE_LOOP=1 runs without the host
    in its iteration (every launch's workgroups make the previous iteration's solve themselves, the host follows a record in
    pinned memory; built, measured, a tie with the stepped loop: NEGATIVES.md).  Against the same call with the switch off
    (every iteration solved on the host, linalg.cpp): same iteration count, same pair count in every iteration, RMS / alignxf /
    final pose / moved points within 1e-11 relative -- on the dat/ pairs thinned to every fourth point (20K points: the loop
    takes scans of up to ~32K), for max_num_iterations 1, 2, 7 (the cap reached mid-flight), without a stopping rule over
    more iterations than the record's ring holds (150 > 64), and on a pair with no partner within reach (fewer than four
    pairs: the loop ends at iteration 0 and nothing moves)."""
    z0 = np.load(os.path.join(G, "dat_scans.npz"))
    z = {k: (z0[k][::4] if k.startswith("scan") else z0[k]) for k in z0.files}

    def run(loop_on, i, max_it, eps, dist=25.0):
        os.environ["TDTK_ICP_DEVICE_LOOP"] = "1" if loop_on else "0"
        try:
            S = _dat_scans(tdtk.Scan, z)
            S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
            icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), dist, max_it, quiet=True, epsilonICP=eps)
            it = icp.match(S[i - 1], S[i])
            out = (it, icp.last["trace"].copy(), S[i].get_transMat().copy(), S[i].get_xyz_reduced().copy(), icp.last["pairs"],
                   icp.last["converged"])
            for s in S:
                s.release()
            return out
        finally:
            os.environ.pop("TDTK_ICP_DEVICE_LOOP", None)

    ran_long = False
    for (i, max_it, eps) in ((1, 50, 1e-5), (2, 50, 1e-5), (1, 1, 1e-5), (1, 2, 1e-5), (1, 7, 1e-5), (1, 150, -1.0)):
        a = run(True, i, max_it, eps)
        b = run(False, i, max_it, eps)
        assert a[0] == b[0] and a[4] == b[4] and a[5] == b[5], (i, max_it, a[0], b[0])
        assert np.array_equal(a[1][:, 0], b[1][:, 0]), (i, max_it)                       # pairs per iteration
        np.testing.assert_allclose(a[1][:, 1], b[1][:, 1], rtol=1e-11)
        np.testing.assert_allclose(a[1][:, 2:], b[1][:, 2:], rtol=0, atol=1e-11 * 200.0)
        assert _rel(a[2], b[2]) < 1e-11
        assert np.abs(a[3] - b[3]).max() < 1e-8
        ran_long = ran_long or a[0] > 64
        # (the two runs are not bit-identical -- the solvers differ in the last places -- which also shows the loop ran)
        assert max_it < 3 or not np.array_equal(a[1][:, 2:], b[1][:, 2:])
    assert ran_long
    # nothing within reach: fewer than four pairs
    a = run(True, 1, 20, 1e-5, dist=1e-6)
    b = run(False, 1, 20, 1e-5, dist=1e-6)
    assert a[0] == b[0] == 0 and a[4] == b[4] and len(a[1]) == len(b[1]) and np.array_equal(a[3], b[3]) and np.array_equal(a[2], b[2])


def test_kernel_timing_is_opt_in_and_changes_nothing(tdtk, gpu):
    """The HIP events around the search / pair-sum launches (tdtk_kernel_timing) are off by default -- nn_ms / sums_ms
    read 0 -- and switching them on changes no result: same iterations, same trace, same pose, bit for bit."""
    z = np.load(os.path.join(G, "dat_scans.npz"))
    L = tdtk.lib()
    assert L.tdtk_kernel_timing(0) in (0, 1)
    runs = []
    for on in (0, 1, 0):
        assert L.tdtk_kernel_timing(on) in (0, 1)
        S = _dat_scans(tdtk.Scan, z)
        S[1].mergeCoordinatesWithRoboterPosition(S[0])
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 15, quiet=True, epsilonICP=1e-5)
        it = icp.match(S[0], S[1])
        runs.append((it, icp.last["trace"].copy(), S[1].get_transMat().copy(), icp.last["nn_ms"], icp.last["sums_ms"]))
    assert L.tdtk_kernel_timing(0) == 0
    assert runs[0][3] == 0.0 and runs[0][4] == 0.0 and runs[2][3] == 0.0
    assert runs[1][3] > 0.0 and runs[1][4] > 0.0
    for r in runs[1:]:
        assert r[0] == runs[0][0] and np.array_equal(r[1], runs[0][1]) and np.array_equal(r[2], runs[0][2])


def test_pair_sums_inside_the_search_launch_agree_with_k_accum(tdtk, gpu, lab, monkeypatch):
    """From 256K queries up to one generation of resident waves the persistent-lane kernel's waves add up the base pair
    sums of their own slabs after their last query (FUSE 3, the default since round 3); TDTK_FUSE_SUMS=0 restores the
    separate k_accum pass.  Same hits, so the pair counts are equal exactly and the sums to rounding (the order of the
    additions differs); the fused path itself is deterministic run to run."""
    rng = np.random.default_rng(5)
    m = rng.uniform(-500, 500, (300000, 3))
    d = m[rng.permutation(len(m))] + rng.normal(0, 0.5, m.shape) + np.array([4.0, -3.0, 2.0])
    runs = []
    for mode in (None, None, "0"):
        if mode is not None:
            monkeypatch.setenv("TDTK_FUSE_SUMS", mode)
        S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], m); S1 = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 8, quiet=True, epsilonICP=-1.0)
        it = icp.match(S0, S1)
        runs.append((it, icp.last["trace"].copy(), S1.get_transMat().copy()))
    monkeypatch.delenv("TDTK_FUSE_SUMS")
    assert runs[0][0] == runs[1][0] == runs[2][0] == 7
    assert np.array_equal(runs[0][1], runs[1][1]) and np.array_equal(runs[0][2], runs[1][2])
    assert np.array_equal(runs[0][1][:, 0], runs[2][1][:, 0]) and runs[0][1][-1, 0] > 0.99 * len(m)     # pairs per iteration
    np.testing.assert_allclose(runs[0][1][:, 1], runs[2][1][:, 1], rtol=1e-12)                            # RMS per iteration
    np.testing.assert_allclose(runs[0][2], runs[2][2], rtol=0, atol=1e-10)


@pytest.mark.parametrize("partial", [False, True])
def test_expensive_queries_first_changes_nothing(tdtk, gpu, lab, monkeypatch, partial):
    """From the second ICP iteration on the persistent-lane kernel hands a wave's slab out with the queries first that
    visited most buckets in the previous pass (TDTK_COST_ORDER=0: in slab order).  Only the order in which a wave works
    through its own queries changes: every iteration's pair count, RMS and pose are the same bit for bit."""
    rng = np.random.default_rng(77)
    m = rng.uniform(-600, 600, (300000, 3))
    d = m[rng.permutation(len(m))] + rng.normal(0, 1.0, m.shape)
    d = d + np.array([6.0, -4.0, 3.0])
    if partial:
        # half of the data scan lies outside the model (no partner within reach: walks of very different length in one
        # slab), a tenth of it is a dense clump, and the model has repeated points
        d[: len(d) // 2, 0] += 900.0
        d[-len(d) // 10:] = d[-1] + rng.normal(0, 0.05, (len(d) // 10, 3))
        m[1000:3000] = m[0:2000]
    runs = []
    for on in ("0", "1", "1"):
        monkeypatch.setenv("TDTK_COST_ORDER", on)
        S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], m); S1 = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 12, quiet=True, epsilonICP=-1.0)
        it = icp.match(S0, S1)
        runs.append((it, icp.last["trace"].copy(), S1.get_transMat().copy()))
    monkeypatch.delenv("TDTK_COST_ORDER")
    assert runs[0][0] == 11 and runs[0][1][-1, 0] > (0.3 if partial else 0.9) * len(m)
    for r in runs[1:]:
        assert r[0] == runs[0][0] and np.array_equal(r[1], runs[0][1]) and np.array_equal(r[2], runs[0][2])


def test_icp_point_to_plane_napx(tdtk, orc, gpu):
    """-a 10 (icp6D_NAPX, the 6x6 point-to-plane system) with -z style plane projection."""
    from oracle import icp_oracle as io
    rng = np.random.default_rng(4)
    g = np.stack(np.meshgrid(np.linspace(-50, 50, 120), np.linspace(-50, 50, 120)), -1).reshape(-1, 2)
    m = np.concatenate([np.c_[g, 0.02 * g[:, 0] * np.sin(g[:, 1] / 9)], np.c_[g[:, 0], np.full(len(g), 50.0), g[:, 1] + 50],
                        np.c_[np.full(len(g), -50.0), g[:, 0], g[:, 1] + 50]])
    nrm = np.concatenate([np.tile([0, 0, 1.0], (len(g), 1)), np.tile([0, 1.0, 0], (len(g), 1)), np.tile([1.0, 0, 0], (len(g), 1))])
    T = io.euler_to_matrix4([0.4, -0.3, 0.2], [0.004, -0.003, 0.005])
    inv, _ = orc.m4inv(T)
    d = m + rng.normal(0, 0.01, m.shape); orc.transform_points(inv, d)
    dn = nrm.copy(); orc.transform_normals(inv, dn)
    S = [tdtk.Scan([0, 0, 0], [0, 0, 0], m, nrm), tdtk.Scan([0, 0, 0], [0, 0, 0], d, dn)]
    O = [io.OScan([0, 0, 0], [0, 0, 0], m, nrm), io.OScan([0, 0, 0], [0, 0, 0], d, dn)]
    icp = tdtk.icp6D(tdtk.icp6D_NAPX(True), 3.0, 8, quiet=True, epsilonICP=1e-9)
    it = icp.match(S[0], S[1], pairing_mode=2)
    oit, otr = io.match(O[0], O[1], 10, 9.0, 8, 1e-9, 2)
    assert it == oit and [int(r[0]) for r in icp.last["trace"]] == [t[0] for t in otr]
    np.testing.assert_allclose(icp.last["trace"][:, 1], [t[1] for t in otr], rtol=1e-7, atol=1e-12)
    assert _rel(S[1].get_transMat(), O[1].transMat) < 1e-8


def test_lum_links_and_iteration_vs_fixture(tdtk, gpu):
    """lum6DEuler::covarianceEuler per link and one doGraphSlam6D iteration on dat/ at the B1
    final poses (tests/golden/b4_dat_lum.json)."""
    from importlib import import_module
    sl = import_module("3dtk_amd.slam6d")
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    b4 = json.load(open(os.path.join(G, "b4_dat_lum.json")))
    S = _dat_scans(tdtk.Scan, z)
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        for a in pr["alignxf"]:
            S[i].transform(np.array(a))
    for L in b4["links"]:
        Cm, CD, m, ss = sl.covarianceEuler(S[L["first"]], S[L["second"]], 625.0)
        assert m == L["m"]
        assert abs(ss - L["ss"]) < 1e-10 * L["ss"]
        np.testing.assert_allclose(Cm, L["C"], rtol=1e-9, atol=1e-5)
        np.testing.assert_allclose(CD, L["CD"], rtol=1e-7, atol=1e-7)
    g = tdtk.Graph(3)
    assert list(zip(g.frm, g.to)) == [(0, 1), (1, 2)]
    # two more copies of the scans at the same poses: python-orchestrated path vs batched native path
    S2 = _dat_scans(tdtk.Scan, z)
    for pr in b1["pairs"]:
        i = pr["cur"]
        S2[i].mergeCoordinatesWithRoboterPosition(S2[i - 1])
        for a in pr["alignxf"]:
            S2[i].transform(np.array(a))
    ret = tdtk.lum6DEuler(None, 25.0, 25.0).doGraphSlam6D(g, S, 1, native=False)
    ret2 = tdtk.lum6DEuler(None, 25.0, 25.0).doGraphSlam6D(g, S2, 1, native=True)
    assert abs(ret - b4["one_iteration"]["ret"]) < 1e-6 * max(1.0, b4["one_iteration"]["ret"])
    assert abs(ret2 - ret) < 1e-9
    for s, s2, want in zip(S, S2, b4["one_iteration"]["poses_after"]):
        got = np.concatenate([s.get_rPos(), s.get_rPosTheta()])
        assert np.abs(got - want).max() <= POSE_RTOL * np.abs(want).max() + 1e-12
        assert np.abs(np.concatenate([s2.get_rPos(), s2.get_rPosTheta()]) - got).max() < 1e-9
        assert np.abs(s2.get_xyz_reduced() - s.get_xyz_reduced()).max() < 1e-7


def test_full_size_icp_recovers_pose(tdtk, orc, gpu, k5):
    """BASELINE configs[1]: synthetic 1M-vs-1M point-to-point ICP.  Pose
    recovered to the noise floor; first-iteration pair count equals the oracle's."""
    from oracle import icp_oracle as io
    k, m, _ = k5
    rng = np.random.default_rng(9)
    T = io.euler_to_matrix4([10.0, -5.0, 3.0], [0.02, -0.03, 0.05])
    inv, _ = orc.m4inv(T)
    d = (m + rng.normal(0, 1.0, m.shape))[rng.permutation(len(m))]
    orc.transform_points(inv, d)
    ms, ds = tdtk.Scan([0, 0, 0], [0, 0, 0], m), tdtk.Scan([0, 0, 0], [0, 0, 0], d)
    first = tdtk.Scan.getPtPairs(ms, ds, max_dist_match2=625.0, want_idx=True)
    oi, _ = orc.Tree(m, 20).find_closest(d, 625.0, 8)
    assert np.array_equal(first["idx"], oi) and first["n"] == int((oi >= 0).sum())
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 100, quiet=True, epsilonICP=1e-5)
    it = icp.match(ms, ds)
    assert it < 99
    assert np.abs(ds.get_transMat() - T).max() < 5e-3           # noise floor of sigma=1 on 1M points
    assert icp.last["rms"] < 2.0


def _range_filter(p, rmax):
    return np.ascontiguousarray(p[p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1] + p[:, 2] * p[:, 2] < rmax * rmax])


@pytest.mark.parametrize("n,dups", [(1000000, 0), (300000, 30000)])
def test_resident_loop_indices_every_iteration(tdtk, orc, gpu, n, dups):
    """Round 5: the regime bench.py times -- tdtk_icp_match on >= 262144 queries: persistent-lane kernel,
    warm start from the previous hit, cost-ordered hand-out, pair sums inside the launch -- pinned at the level of INDICES.
    With tdtk_icp_index_hashes on, every pass of the loop reduces its correspondences to the K5 hash on the
    device; for the first iterations that hash must equal the hash of the multi-threaded oracle's FindClosest over the
    points where the loop has moved them (the trace's alignxf applied with the reference's transform3 arithmetic), and pair
    count + RMS must equal the reference's own OpenMP-branch iterations (oracle/_ref: ref_icp_iterations) where that library
    exists.  Second case: 30 000 exact duplicates in the model, i.e. ties that the warm radius (one ulp above the previous
    hit's distance) must not resolve differently from a cold search."""
    import bench
    rng = np.random.default_rng(n + dups)
    m, d, T = bench.make_icp_pair(n, seed=42 if not dups else 43)
    if dups:
        m[rng.integers(0, n, dups)] = m[rng.integers(0, n, dups)]
    model = tdtk.Scan([0, 0, 0], [0, 0, 0], m)
    data = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
    iters = 6
    was = tdtk.lib().tdtk_icp_index_hashes(1)
    try:
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, iters, quiet=True, epsilonICP=-1.0)
        assert icp.match(model, data) == iters - 1
    finally:
        tdtk.lib().tdtk_icp_index_hashes(was)
    hashes = icp.last["index_hashes"]
    trace = icp.last["trace"]
    assert len(hashes) == iters and len(trace) == iters
    ot = orc.Tree(m, 20)
    cur = d.copy()
    nt = max(8, min(96, os.cpu_count() or 8))
    for it in range(iters):
        idx, _ = ot.find_closest(cur, 625.0, nt)                      # model at the identity pose: tree frame = world
        assert int((idx >= 0).sum()) == int(trace[it, 0]), it
        assert orc.k5_hash(idx) == hashes[it], (it, "0x%x" % orc.k5_hash(idx), "0x%x" % hashes[it])
        orc.transform_points(trace[it, 2:], cur)                      # Scan::transform, same arithmetic (scan.cc:851-875)
    assert np.array_equal(cur, data.get_xyz_reduced())                # the resident points are where the trace says
    if orc.have_ref():
        rt = orc.RefTree(m, 20)
        _, rtr = rt.icp_iterations(np.eye(4).reshape(16), d, 625.0, 16, iters)
        assert np.array_equal(rtr[:, 0], trace[:, 0])                 # pairs per iteration
        np.testing.assert_allclose(rtr[:, 1], trace[:, 1], rtol=1e-9)  # RMS per iteration
        np.testing.assert_allclose(rtr[:, 2:], trace[:, 2:], rtol=0, atol=1e-9)
    # switched off again: the next match keeps no hashes
    icp2 = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 2, quiet=True, epsilonICP=-1.0)
    icp2.match(model, data)
    assert icp2.last["index_hashes"] == []


def test_reference_walk_counters_of_the_resident_loop_equal_the_oracles(tdtk, orc, gpu):
    """Round 6: SURVEY 8(d)'s n_int / n_pts under `roofline.frac` are the REFERENCE's walk, not the kernel's
    own.  tdtk_visit_counting(device, 2) makes every search of the loop start cold (no warm start, no deferred quick check)
    while it counts; over the bench pair (1M-vs-1M, seed 42: the pair whose numbers the bench line quotes) the counters of a
    4-iteration loop must equal, exactly, the oracle's _FindClosest counters (kdTreeImpl.h:345-383 restated, oracle.c) summed
    over the same four query sets -- and the loop's results must not depend on the counting mode (same trace as mode 1 and as
    no counting at all), while mode 1 (the kernel's own walk: warm start, deferred check) visits something else."""
    import ctypes as C
    import bench
    n, iters = 1000000, 4
    m, d, T = bench.make_icp_pair(n, seed=42)
    model = tdtk.Scan([0, 0, 0], [0, 0, 0], m)
    L = tdtk.lib()

    def loop(mode):
        data = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
        c = (C.c_uint64 * 8)()
        if mode:
            L.tdtk_visit_counting(gpu, mode)
        try:
            icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, iters, quiet=True, epsilonICP=-1.0)
            assert icp.match(model, data) == iters - 1
            if mode:
                L.tdtk_visit_counters(gpu, c)
        finally:
            L.tdtk_visit_counting(gpu, 0)
        data.release()
        return icp.last["trace"].copy(), (int(c[0]), int(c[1]), int(c[2]), int(c[3]))

    tr0, _ = loop(0)
    tr2, ref_walk = loop(2)
    tr1, own_walk = loop(1)
    assert np.array_equal(tr0, tr2) and np.array_equal(tr0, tr1)
    assert ref_walk[3] == own_walk[3] == n * iters
    ot = orc.Tree(m, 20)
    cur = d.copy()
    nt = max(8, min(96, os.cpu_count() or 8))
    tot = np.zeros(3, np.int64)
    for it in range(iters):
        tot += np.array(ot.find_closest(cur, 625.0, nt, True)[2], np.int64)
        orc.transform_points(tr0[it, 2:], cur)
    assert tuple(int(x) for x in tot) == ref_walk[:3], (tot, ref_walk)
    assert own_walk[:3] != ref_walk[:3]
    # the figure the bench line quotes for this pair's first iterations: SURVEY 8(d)'s ~2.4 KB per query
    bq = bench.algorithmic_bytes_per_query(ref_walk[0] / ref_walk[3], ref_walk[2] / ref_walk[3])
    assert 2300.0 < bq < 2600.0, bq


def test_thin_acceptances_are_searched_again(tdtk, orc, gpu):
    """Round 5, "the quick check deferred" (kernels.hip): from its second pass on a query of the resident loop walks without
    the quick check of its divergent visits, and is searched AGAIN -- cold, every check made: the reference's own walk -- if it
    ever accepted a point that improved closest_d2 by no more than SearchArgs::tie (a rounding's worth: what a skipped check
    could have hidden).  30 000 model points get a twin 1e-12 .. 1e-11 away, so that thousands of queries see two candidates
    whose d2 differ by less than that.  Every iteration's index hash must equal the oracle's FindClosest over the moved
    points, and the instrumented run must report such second searches (tdtk_visit_counters out[7]) with the same hashes."""
    import ctypes as C
    import bench
    n, twins = 300000, 30000
    rng = np.random.default_rng(77)
    m, d, T = bench.make_icp_pair(n, seed=44)
    src = rng.choice(n, twins, replace=False)
    dst = np.setdiff1d(np.arange(n), src)[:twins]
    off = np.zeros((twins, 3))
    off[np.arange(twins), rng.integers(0, 3, twins)] = rng.uniform(1e-12, 1e-11, twins) * rng.choice([-1.0, 1.0], twins)
    m[dst] = m[src] + off
    assert not np.array_equal(m[dst], m[src])
    model = tdtk.Scan([0, 0, 0], [0, 0, 0], m)
    iters = 5
    L = tdtk.lib()

    def loop(counting):
        data = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
        was = L.tdtk_icp_index_hashes(1)
        if counting:
            L.tdtk_visit_counting(gpu, 1)
        try:
            icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, iters, quiet=True, epsilonICP=-1.0)
            assert icp.match(model, data) == iters - 1
            c = (C.c_uint64 * 8)()
            if counting:
                L.tdtk_visit_counters(gpu, c)
        finally:
            L.tdtk_icp_index_hashes(was)
            if counting:
                L.tdtk_visit_counting(gpu, 0)
        data.release()
        return icp.last["index_hashes"], icp.last["trace"].copy(), int(c[7])

    hashes, trace, _ = loop(False)
    ot = orc.Tree(m, 20)
    cur = d.copy()
    nt = max(8, min(96, os.cpu_count() or 8))
    for it in range(iters):
        idx, _ = ot.find_closest(cur, 625.0, nt)
        assert int((idx >= 0).sum()) == int(trace[it, 0]), it
        assert orc.k5_hash(idx) == hashes[it], (it, "0x%x" % orc.k5_hash(idx), "0x%x" % hashes[it])
        orc.transform_points(trace[it, 2:], cur)
    h2, t2, again = loop(True)
    assert h2 == hashes and np.array_equal(t2, trace)
    if os.environ.get("TDTK_DEFER_CHECK", "1") != "0":
        assert again > 100, again                                     # the path ran: twins do make thin acceptances


@pytest.mark.parametrize("rnd", [1, 5])
def test_config1_metascan_dat(tdtk, orc, gpu, rnd, tmp_path):
    """BASELINE configs[0]: `slam6D -m 500 -R 5 -d 25.0 --metascan dat` (plumbing).  -R 5 draws
    std::rand() per candidate, so the comparison seeds libc identically for both runs (serial-build
    semantics); rnd=1 runs the device-resident loop against the MetaScan tree."""
    import ctypes as C
    from oracle import icp_oracle as io
    libc = C.CDLL(None)
    z = np.load(os.path.join(G, "dat_scans.npz"))
    pts = [_range_filter(z["scan%03d" % k], 500.0) for k in range(3)]
    assert [len(p) for p in pts] != [81360] * 3                      # the filter bites
    S = [tdtk.Scan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], pts[k]) for k in range(3)]
    O = [io.OScan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], pts[k]) for k in range(3)]
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 50, quiet=True, meta=True, rnd=rnd, epsilonICP=1e-5)
    libc.srand(42)
    traces = []
    orig_match = icp.match

    def rec(a, b, pm=0):
        r = orig_match(a, b, pm)
        traces.append((r, icp.last["trace"].copy()))
        return r
    icp.match = rec
    icp.doICP(S)
    libc.srand(42)
    want = io.do_icp(O, 1, 625.0, 50, 1e-5, True, rnd, True)
    assert len(traces) == len(want) == 2
    for (it, tr), (oit, otr) in zip(traces, want):
        assert it == oit
        assert [int(r[0]) for r in tr] == [t[0] for t in otr]
        np.testing.assert_allclose(tr[:, 1], [t[1] for t in otr], rtol=1e-9)
    for s, o in zip(S, O):
        assert _rel(s.get_transMat(), o.transMat) < 1e-9
    # second match ran against a MetaScan of two scans (concatenated, current poses)
    assert traces[1][1][0][0] > traces[0][1][0][0] * 0.5
    # .frames output (N2): 16 entries at 6 significant digits + AlgoType per line
    S[1].identifier, S[1].path = "001", str(tmp_path)
    fn = tdtk.saveFrames(S[1])
    lines = open(fn).read().strip().split("\n")
    assert len(lines) == len(S[1].frames)
    last = [float(t) for t in lines[-1].split()]
    assert len(last) == 17 and int(last[16]) == 1
    np.testing.assert_allclose(last[:16], S[1].get_transMat(), rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("meta,maxmeta,rnd", [(1, -1, 1), (1, 1, 1), (0, -1, 1), (1, -1, 5)])
def test_config1_metascan_dat_through_the_cpp_glue(tdtk, gpu, tmp_path, meta, maxmeta, rnd):
    """BASELINE configs[0] (`slam6D -m 500 -d 25.0 --metascan dat`) through the C++ binding's body: icp6D_hip::doICP =
    hip_do_icp (adapters/icp_glue.h), whose meta branch (icp6D.cc:396-434) matches every scan against a MetaScan tree
    built on the device over the scans before it (tdtk_tree_create_from_scans = KDtreeMetaManaged).  Executed by
    adapters/harness/slam_glue_harness.cc on the bundled scans; poses, frame counts and iteration counts equal the Python
    mirror's doICP -- which test_config1_metascan_dat pins to the oracle -- bit for bit.  Also max_num_metascans = 1 and
    the plain sequential branch."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "adapters", "harness", "_bin", "slam_glue_harness")
    if not os.path.exists(exe):
        r = subprocess.run([os.path.join(os.path.dirname(HERE), "adapters", "harness", "build_glue.sh")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(os.path.join(G, "dat_scans.npz"))
    pts = [_range_filter(z["scan%03d" % k], 500.0) for k in range(3)]
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(np.int32(3).tobytes())
        for k in range(3):
            f.write(np.int32(len(pts[k])).tobytes())
            f.write(np.ascontiguousarray(z["pose%03d" % k], dtype=np.float64).tobytes())
            f.write(np.ascontiguousarray(pts[k], dtype=np.float64).tobytes())
    # (rnd = 5: BASELINE configs[0] to the letter, `-R 5` -- the keep-masks drawn per iteration inside tdtk_icp_match_rnd, libc
    # seeded with 42 in the harness and, below, for the Python mirror: the same draws, the same poses bit for bit)
    r = subprocess.run([exe, "doicp", fin, fout, str(meta), str(maxmeta), "25.0", "50", "1e-5", str(rnd), "42"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SLAM GLUE HARNESS OK" in r.stdout, r.stdout + r.stderr
    out = np.fromfile(fout, dtype=np.uint8)
    tm_cpp = np.frombuffer(out[:3 * 128].tobytes(), dtype=np.float64).reshape(3, 16)
    tail = np.frombuffer(out[3 * 128:].tobytes(), dtype=np.int32)
    S = [tdtk.Scan(z["pose%03d" % k][:3], z["pose%03d" % k][3:], pts[k]) for k in range(3)]
    tdtk.Scan.allScans = S
    try:
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 50, quiet=True, meta=bool(meta), rnd=rnd, epsilonICP=1e-5, max_num_metascans=maxmeta)
        import ctypes as _C
        _C.CDLL(None).srand(42)
        its = []
        orig_match = icp.match

        def rec(a, b, pm=0):
            its.append(orig_match(a, b, pm))
            return its[-1]
        icp.match = rec
        icp.doICP(S)
        tm_py = np.stack([s.transMat for s in S])
        assert tm_py.tobytes() == tm_cpp.tobytes(), float(np.abs(tm_py - tm_cpp).max())
        assert [len(s.frames) for s in S] == tail[:3].tolist()
        assert its == tail[3:5].tolist()
        assert np.abs(tm_cpp[2] - tdtk.EulerToMatrix4(z["pose002"][:3], z["pose002"][3:])).max() > 1e-3      # it matched something
    finally:
        tdtk.Scan.allScans = []


def test_gapx6d_links_and_iterations(tdtk, orc, gpu):
    """-G 4 (gapx6D): per-link genBArotForLinkedPair blocks (literal formulas incl. the
    `p1x*p2x + p1y + p2y` terms) and two doGraphSlam6D iterations vs the oracle restatement."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    S, O = _dat_scans(tdtk.Scan, z), _dat_scans(io.OScan, z)
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1]); O[i].mergeCoordinatesWithRoboterPosition(O[i - 1])
        for a in pr["alignxf"][:8]:                       # not fully converged: leaves something to do
            S[i].transform(np.array(a)); O[i].transform(np.array(a))
    r = tdtk.Scan.getPtPairs(S[0], S[1], max_dist_match2=625.0, want=tdtk.WANT_GAPX)
    o = io.get_pt_pairs(O[0], O[1], 625.0)
    blocks = io.gapx_link_blocks(o["p1"], o["p2"], o["cm"])
    for name, want in zip(("gapx_MkMkt", "gapx_DkDkt", "gapx_MkDkt", "gapx_DkMkt", "gapx_Ak1", "gapx_Ak2"), blocks):
        scale = max(np.abs(blocks[0]).max(), 1.0)
        assert np.abs(r[name] - want).max() < 1e-9 * scale, name
    links = [(0, 1), (1, 2), (0, 2)]
    g = tdtk.Graph(3, links=links)
    g.nrScans = 3
    gx = tdtk.gapx6D(None, 25.0, 25.0, epsilonLUM=-1.0)
    ret = gx.doGraphSlam6D(g, S, 2)
    T = None
    for _ in range(2):
        oret, T, X = io.gapx_iteration(links, O, 625.0, T)
    assert abs(ret - oret) < 1e-7 * max(1.0, abs(oret))
    for s, o in zip(S, O):
        assert _rel(s.get_transMat(), o.transMat) < POSE_RTOL
        assert _rel(s.get_transMat(), o.transMat) < 1e-8


def test_concurrent_host_threads(tdtk, orc, gpu):
    """The reference drives a tree from several OpenMP threads at once (disjoint query ranges in
    icp6D::match, different links in FillGB3D).  The C ABI keeps a stream + workspace per calling
    thread, so concurrent calls on shared handles must give the single-threaded answers."""
    import threading
    rng = np.random.default_rng(21)
    m = rng.uniform(-300, 300, (200000, 3))
    q = rng.uniform(-300, 300, (120000, 3))
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    want_idx, want_d2 = T.find_closest(q, 400.0, 8)
    chunks = np.array_split(np.arange(len(q)), 6)
    out = [None] * len(chunks)
    errs = []

    def work(k):
        try:
            for _ in range(3):
                out[k] = kd.FindClosestBatch(q[chunks[k]], 400.0)
                r = kd.getPtPairs(tdtk.M4identity(), q, None, int(chunks[k][0]), int(chunks[k][-1]) + 1,
                                  max_dist_match2=400.0, want_pairs=False)
                assert np.array_equal(r["idx"], want_idx[chunks[k]])
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(len(chunks))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for k, c in enumerate(chunks):
        assert np.array_equal(out[k][0], want_idx[c]) and np.array_equal(out[k][1], want_d2[c])


def test_large_scan_4m_points(tdtk, orc, gpu):
    """Towards configs[4] (~10M points per scan): a 4M-point model, 24-bit point indices, deeper
    tree; indices bit-exact on a query sample, whole-scan pass consistent with it."""
    rng = np.random.default_rng(31)
    M = 4000000
    m = rng.uniform(-3000, 3000, (M, 3))
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    inf, st = kd.info(), T.stats()
    assert (inf["n_internal"], inf["n_leaves"], inf["max_depth"]) == (st["internal"], st["leaves"], st["depth"])
    q = m[rng.integers(0, M, 300000)] + rng.normal(0, 5.0, (300000, 3))
    for md2 in (100.0, 1e18):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od2)


@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "tiny", "grid", "line"])
@pytest.mark.parametrize("bucket", [1, 7, 20])
def test_device_tree_build_equals_host_build(tdtk, gpu, name, bucket):
    """The tree built level by level on the GPU (build.hip) is the host builder's tree
    (kd_build.cpp == KDTreeImpl::create), record for record."""
    kd = tdtk.KDtree(_clouds()[name], bucket)
    assert kd.verify() == [0, 0, 0, 0]


@pytest.mark.parametrize("case", ["fixture", "generated"])
def test_device_tree_of_a_lopsided_cloud_is_handed_over_late(tdtk, orc, gpu, case):
    """A cloud whose densest cluster stays above the finisher's 3584-point limit for sixteen levels (coordinates over
    twelve orders of magnitude: tools/fuzz_parity.py kind 4, seed 4401 run 2446 -- tests/golden/fuzz_case_dynamic_range_20000.npz,
    written by `tools/fuzz_parity.py --trace` -- and a second one generated here).  The level is handed to the subtree
    finisher sixteen levels later than a balanced tree's would be; round 4 launched 2^level workgroups and copied that many
    root records there -- a million, over the end of the build's arena: five wrong split values, 125 of 550 wrong neighbours
    and, inside a long process, a GPU memory fault.  Now: as many as the level has nodes.  Record for record the host
    builder's tree, the oracle's neighbours, and a tree built right after it in the same context is sound too."""
    if case == "fixture":
        d = np.load(os.path.join(HERE, "golden", "fuzz_case_dynamic_range_20000.npz"))
        p, q, bucket, md2 = d["p"], d["q"], int(d["bucket"]), float(d["md2"])
    else:
        rng = np.random.default_rng(77)
        p = rng.normal(0, 1, (30000, 3)) * (10.0 ** rng.integers(-6, 7, (30000, 1)))
        q = np.concatenate([p[rng.integers(0, len(p), 500)] + rng.normal(0, 1e-3, (500, 3)), rng.uniform(p.min() - 1, p.max() + 1, (50, 3))])
        bucket, md2 = 20, 1e18
    kd, T = tdtk.KDtree(p, bucket), orc.Tree(p, bucket)
    assert kd.verify() == [0, 0, 0, 0]
    gi, gd = kd.FindClosestBatch(q, md2)
    oi, od = T.find_closest(q, md2)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    m = np.random.default_rng(5).uniform(-50, 50, (40000, 3))
    kd2 = tdtk.KDtree(m, 20)
    assert kd2.verify() == [0, 0, 0, 0]


def test_lab_library_guards_the_build_arena(tdtk, gpu, lab, monkeypatch):
    """The lab library puts 256 guard bytes behind every region of the tree build's arena and looks at them behind the
    build (round 4, after a table of 8192 root records had been written a million deep): an ordinary build passes, and with
    TDTK_GUARD_SELFTEST=1 -- one guard word overwritten on purpose -- the build is refused.  (Every test that loads the
    lab library, and tools/fuzz_parity.py under TDTK_LIB=lab, builds its trees under these guards.)"""
    p = np.random.default_rng(3).uniform(-100, 100, (50000, 3))
    kd = tdtk.KDtree(p, 20)
    assert kd.verify() == [0, 0, 0, 0]
    monkeypatch.setenv("TDTK_GUARD_SELFTEST", "1")
    with pytest.raises(tdtk.TdtkError):
        tdtk.KDtree(p, 20)
    monkeypatch.delenv("TDTK_GUARD_SELFTEST")
    assert tdtk.KDtree(p, 20).verify() == [0, 0, 0, 0]


def test_device_tree_build_full_size(tdtk, gpu, k5):
    k, m, _ = k5
    kd = tdtk.KDtree(m, 20)
    inf = kd.info()
    assert (inf["n_internal"], inf["n_leaves"], inf["max_depth"]) == (k["tree"]["internal"], k["tree"]["leaves"], k["tree"]["depth"])
    assert kd.verify() == [0, 0, 0, 0]
    rng = np.random.default_rng(5)
    big = rng.uniform(-1000, 1000, (1100000, 3)); big[5000:5600] = big[4999]       # leaf-table mode
    assert tdtk.KDtree(big, 20).verify() == [0, 0, 0, 0]


@pytest.mark.parametrize("shape", ["uniform", "lopsided", "repeated"])
def test_device_tree_build_of_millions_by_subtree_size(tdtk, gpu, shape):
    """Round 6: from two million points on the levels run until a balanced node holds <= 384 points (the partition of a
    level in two passes: k_part_scan's segmented look-back scan + k_part_swap) and the subtrees are finished by size --
    k_fin_wave (one wave per subtree of <= 512 points), k_fin_subtrees_half (<= 1792), k_fin_subtrees (<= 3584).  A uniform
    cloud is all waves; a lopsided one (a dense cluster inside a sparse field: the hand-over level's largest node several
    times its average) needs all three launches; coordinates that repeat put points ON splitting planes.  The tree is the
    host builder's, record for record, and a second tree built in the same context (the arena reused) is too."""
    rng = np.random.default_rng(606)
    n = 2200000
    if shape == "uniform":
        p = rng.uniform(-1000.0, 1000.0, (n, 3))
    elif shape == "lopsided":
        p = np.concatenate([rng.uniform(-1000.0, 1000.0, (n - 600000, 3)), rng.normal(0.0, 6.0, (600000, 3)) + [300.0, -200.0, 50.0]])
        p = p[rng.permutation(len(p))]
    else:
        p = np.round(rng.uniform(-1000.0, 1000.0, (n, 3)), 1)
    kd = tdtk.KDtree(np.ascontiguousarray(p), 20)
    assert kd.verify() == [0, 0, 0, 0]
    assert tdtk.KDtree(np.ascontiguousarray(p[: n // 2]), 7).verify() == [0, 0, 0, 0]


def test_lab_switches_of_the_round6_tree_build_give_the_same_tree(tdtk, gpu, lab, monkeypatch):
    """The lab library's switches back to round 5's build -- five partition passes (TDTK_BUILD_PART=0), the workgroup
    finisher alone (TDTK_BUILD_FINWAVE=0, TDTK_BUILD_FINHALF=0), the hand-over at 2048-point nodes, the piecewise path for
    every speculated level (TDTK_BUILD_CHAINFROM=99), two side streams -- each give the host builder's tree as well."""
    p = np.random.default_rng(607).uniform(-500.0, 500.0, (2100000, 3))
    for kv in ("TDTK_BUILD_PART=0", "TDTK_BUILD_FINWAVE=0", "TDTK_BUILD_FINHALF=0", "TDTK_BUILD_HANDOFF=2048", "TDTK_BUILD_CHAINFROM=99",
               "TDTK_BUILD_STREAMS=3", "TDTK_BUILD_BIGGROW=0"):
        k, v = kv.split("=")
        monkeypatch.setenv(k, v)
        assert tdtk.KDtree(p, 20).verify() == [0, 0, 0, 0], kv
        monkeypatch.delenv(k)


def _stress_clouds(n, seed=12):
    """clouds chosen to break the piecewise centroid sum of the device tree build (build.hip, k_big_*): exact rounding
    ties, sums that wander through zero, huge dynamic range, values at both ends of the exponent range"""
    rng = np.random.default_rng(seed)
    return {
        "zero-mean": rng.uniform(-1000, 1000, (n, 3)),
        "one-sided": rng.uniform(0, 2000, (n, 3)),
        "integers": rng.integers(-500, 501, (n, 3)).astype(float),
        "half-integers": rng.integers(-2000, 2001, (n, 3)) * 0.5,
        "dyadic": rng.integers(-1 << 20, 1 << 20, (n, 3)) * (2.0 ** -7),
        "wide-range": rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.uniform(-6, 6, (n, 1)),
        "tiny": rng.uniform(-1, 1, (n, 3)) * 1e-300,
        "huge": rng.uniform(-1, 1, (n, 3)) * 1e300,
        "repeated": np.repeat(rng.uniform(-100, 100, (n // 50, 3)), 50, axis=0),
        "back-to-zero": np.concatenate([rng.uniform(0, 1000, (n // 2, 3)), -rng.uniform(0, 1000, (n // 2, 3))])[rng.permutation(n // 2 * 2)],
        "alternating": np.where((np.arange(n) % 2 == 0)[:, None], 1e8, -1e8) + rng.uniform(-1, 1, (n, 3)),
    }


@pytest.mark.parametrize("mode", ["", "1", "2"])
def test_device_tree_build_piecewise_sum(tdtk, gpu, lab, monkeypatch, mode):
    """Nodes of 8192 points and more get their left-to-right centroid sum piecewise (integer mantissa offsets per
    64-point piece, runs folded by a scan, exact walk where the sum changes binade).  The tree must stay the host
    builder's, record for record -- also when no folded run is trusted (mode 1) and when every piece is walked (2)."""
    if mode:
        monkeypatch.setenv("TDTK_BIG_DEBUG", mode)
    for name, pts in _stress_clouds(70000).items():
        for bucket in (20, 3):
            kd = tdtk.KDtree(np.ascontiguousarray(pts), bucket)
            assert kd.verify() == [0, 0, 0, 0], (name, bucket, mode)
    # Morton-ordered input (what a resident scan hands to the builder): partial sums return to zero at every scale
    s = tdtk.Scan([0, 0, 0], [0, 0, 0], _stress_clouds(300000, seed=5)["zero-mean"])
    assert s.getSearchTree().verify() == [0, 0, 0, 0]


def test_device_tree_build_from_eight_threads(tdtk, gpu):
    """Eight host threads building trees of big scans at once (prepare_scans): every tree still the host builder's.
    (The first version of the piecewise sum lost an addend here: inline-asm LDS reads whose destination registers the
    compiler had moved before the data arrived.)"""
    rng = np.random.default_rng(8)
    scans = [tdtk.Scan([0, 0, 0], [0, 0, 0], rng.uniform(-1500, 1500, (400000, 3))) for _ in range(16)]
    tdtk.prepare_scans(scans, trees=True, threads=8)
    assert [s.getSearchTree().verify() for s in scans] == [[0, 0, 0, 0]] * 16


def test_batched_scan_moves_are_visible_to_every_thread(tdtk, orc, gpu):
    """tdtk_scans_transform2 returns before the device has moved the scans; the next library call of ANY host thread
    on the device must wait for the move first."""
    import threading
    import ctypes as C
    rng = np.random.default_rng(9)
    pts = [rng.uniform(-100, 100, (200000, 3)) for _ in range(6)]
    scans = [tdtk.Scan([0, 0, 0], [0, 0, 0], p) for p in pts]
    for s in scans:
        _ = s.handle
    A = [tdtk.EulerToMatrix4(rng.uniform(-5, 5, 3), rng.uniform(-0.2, 0.2, 3)) for _ in scans]
    hs = (C.c_void_p * len(scans))(*[s._h for s in scans])
    A1 = np.ascontiguousarray(np.stack(A))
    from importlib import import_module
    capi = import_module("3dtk_amd._capi")
    capi.check(capi.lib().tdtk_scans_transform2(len(scans), hs, capi.dptr(A1), None))
    got = [None] * len(scans)

    def fetch(k):
        out = np.empty((len(pts[k]), 3))
        capi.check(capi.lib().tdtk_scan_download(scans[k]._h, capi.dptr(out), None))
        got[k] = out
    th = [threading.Thread(target=fetch, args=(k,)) for k in range(len(scans))]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(len(scans)):
        want = pts[k].copy()
        orc.transform_points(A[k], want)
        assert np.array_equal(got[k], want), k


def test_lum_links_fused_into_the_search_agree(tdtk, gpu, lab, monkeypatch):
    """The ways a batch of lum6DEuler links is evaluated on big scans agree: all links in one launch
    (k_search_refill_multi) with k_accum_multi behind it (TDTK_LINK_FUSE=0) and one search + k_accum per link on three
    streams (TDTK_LINK_BATCH=0) -- bit for bit; the default since round 3, the sums added up inside that one launch by
    each wave over its own slab, and TDTK_FUSE_LUM=1, the 17 sums accumulated when a query retires inside the search kernel (a measured
    negative kept selectable): same blocks to rounding (the order of the additions differs), same pair counts exactly."""
    import ctypes as C
    from importlib import import_module
    capi = import_module("3dtk_amd._capi")
    rng = np.random.default_rng(17)
    world = rng.uniform(-400, 400, (300000, 3))
    scans = []
    for k in range(4):
        T = tdtk.EulerToMatrix4([3.0 * k, -1.0 * k, 2.0 * k], [0.002 * k, -0.003 * k, 0.004 * k])
        Ti = tdtk.M4inv(T)
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        loc = world @ R.T + Ti[12:15] + rng.normal(0, 0.05, world.shape)
        scans.append(tdtk.Scan([3.0 * k + 0.3, -1.0 * k, 2.0 * k - 0.2], [0.002 * k, -0.003 * k + 0.001, 0.004 * k], loc))
    tdtk.prepare_scans(scans, trees=True, threads=2)
    links = [(0, 1), (1, 2), (2, 3), (0, 3), (1, 3)]
    nl = len(links)
    first = (C.c_void_p * nl)(*[scans[a].getSearchTree()._h for a, b in links])
    second = (C.c_void_p * nl)(*[scans[b].handle for a, b in links])
    dal = np.ascontiguousarray(np.stack([scans[a].dalignxf for a, b in links]))

    def blocks():
        Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
        capi.check(capi.lib().tdtk_lum_links(nl, first, capi.dptr(dal), second, 100.0, capi.dptr(Cm), capi.dptr(CD), m, capi.dptr(ss)))
        return Cm, CD, list(m), ss
    one_launch = blocks()                       # default for big scans: all links in one launch, the sums added up by the
                                                # search waves over their own slabs (round 3)
    monkeypatch.setenv("TDTK_LINK_FUSE", "0")
    one_launch_accum = blocks()                 # ... and with k_accum_multi behind the search launch
    monkeypatch.delenv("TDTK_LINK_FUSE")
    # (round 4, lab: the same launch in workgroups of ONE wave and with the hand-out that does not wait -- one row of
    # partial sums per wave: same pairs, same blocks to rounding)
    for env in ({"TDTK_MULTI_BLOCK": "64"}, {"TDTK_PIPE": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        alt = blocks()
        assert alt[2] == one_launch[2], env
        for a, b in zip((one_launch[0], one_launch[1], one_launch[3]), (alt[0], alt[1], alt[3])):
            np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-9 * np.abs(a).max())
        for k in env:
            monkeypatch.delenv(k)
    monkeypatch.setenv("TDTK_LINK_BATCH", "0")
    base = blocks()                             # three streams, k_accum behind every search
    monkeypatch.setenv("TDTK_FUSE_LUM", "1")
    fused = blocks()
    assert base[2] == fused[2] == one_launch[2] == one_launch_accum[2] and min(base[2]) > 100000
    for a, b, c in zip((base[0], base[1], base[3]), (fused[0], fused[1], fused[3]), (one_launch[0], one_launch[1], one_launch[3])):
        np.testing.assert_allclose(b, a, rtol=1e-9, atol=1e-9 * np.abs(a).max())
        np.testing.assert_allclose(c, a, rtol=1e-9, atol=1e-9 * np.abs(a).max())   # (the order of the additions differs)
    for a, b in zip((base[0], base[1], base[3]), (one_launch_accum[0], one_launch_accum[1], one_launch_accum[3])):
        assert np.array_equal(a, b)             # same kernels per link, only launched together
    # how many links share a launch changes nothing either: groups of two, and 70 links (the same five pairs over and
    # over: more than the 64 a launch took until round 3) in one
    monkeypatch.delenv("TDTK_FUSE_LUM")
    monkeypatch.setenv("TDTK_LINK_BATCH", "2")
    pairs = blocks()
    for a, b in zip((one_launch[0], one_launch[1], one_launch[3]), (pairs[0], pairs[1], pairs[3])):
        assert np.array_equal(a, b)
    monkeypatch.delenv("TDTK_LINK_BATCH")
    links = [links[i % 5] for i in range(70)]
    nl = len(links)
    first = (C.c_void_p * nl)(*[scans[a].getSearchTree()._h for a, b in links])
    second = (C.c_void_p * nl)(*[scans[b].handle for a, b in links])
    dal = np.ascontiguousarray(np.stack([scans[a].dalignxf for a, b in links]))
    many = blocks()
    again = blocks()            # the second pass over the same links hands its slabs out by the first pass's costs ...
    monkeypatch.setenv("TDTK_LINK_ORDERED", "0")
    plain = blocks()            # ... and this one in slab order: the order never shows in a result
    monkeypatch.delenv("TDTK_LINK_ORDERED")
    assert many[2] == again[2] == plain[2]
    for a, b, c in zip((many[0], many[1], many[3]), (again[0], again[1], again[3]), (plain[0], plain[1], plain[3])):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    for i in range(nl):
        assert many[2][i] == one_launch[2][i % 5]
        for a, b in zip((one_launch[0], one_launch[1], one_launch[3]), (many[0], many[1], many[3])):
            assert np.array_equal(a[i % 5], b[i])


def test_a_links_sums_do_not_depend_on_its_company(tdtk, gpu):
    """Since round 3 the sums of a big scan's link are added up inside the search launch, so the path a link takes decides
    the last bits of its sums -- and that path depends on the link alone: a graph whose scans are of several size classes
    (300K points: persistent lanes, sums inside the launch; 100K: lane groups, k_accum) gives every link the same blocks
    whether it is evaluated with all the others, with some of them (a rank's share) or alone."""
    import ctypes as C
    from importlib import import_module
    capi = import_module("3dtk_amd._capi")
    rng = np.random.default_rng(23)
    world = rng.uniform(-400, 400, (300000, 3))
    scans = []
    for k in range(4):
        T = tdtk.EulerToMatrix4([3.0 * k, -1.0 * k, 2.0 * k], [0.002 * k, -0.003 * k, 0.004 * k])
        Ti = tdtk.M4inv(T)
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        loc = world @ R.T + Ti[12:15] + rng.normal(0, 0.05, world.shape)
        if k == 3:
            loc = loc[:100000]
        scans.append(tdtk.Scan([3.0 * k + 0.3, -1.0 * k, 2.0 * k - 0.2], [0.002 * k, -0.003 * k + 0.001, 0.004 * k], loc))
    tdtk.prepare_scans(scans, trees=True, threads=2)
    links = [(0, 1), (1, 2), (2, 3), (3, 0), (0, 2), (1, 3)]

    def blocks(sel):
        nl = len(sel)
        first = (C.c_void_p * nl)(*[scans[links[i][0]].getSearchTree()._h for i in sel])
        second = (C.c_void_p * nl)(*[scans[links[i][1]].handle for i in sel])
        dal = np.ascontiguousarray(np.stack([scans[links[i][0]].dalignxf for i in sel]))
        Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
        capi.check(capi.lib().tdtk_lum_links(nl, first, capi.dptr(dal), second, 100.0, capi.dptr(Cm), capi.dptr(CD), m, capi.dptr(ss)))
        return Cm, CD, np.array(list(m)), ss
    whole = blocks(list(range(len(links))))
    assert whole[2].min() > 50000
    for share in ([0, 2, 4], [1, 3, 5], [2, 5], [0, 1, 4], [3], [0], [5]):
        part = blocks(share)
        for a, b in zip(whole, part):
            assert np.array_equal(a[share], b), share


def test_lazy_scan_moves_equal_moving_every_round(tdtk, gpu, monkeypatch):
    """Round 4: the pose update of a graph-SLAM round queues its transforms on the resident scans
    (Scan::transformToEuler, scan.cc:1061-1083: two in-place transforms per round) and the link passes of the next round
    apply them in registers -- the link that owns a scan's update stores the moved points, other links of the same launch
    that read the scan move it again on the fly, a scan no link reads keeps its chain until somebody asks for its points.
    Same arithmetic in the same order as moving every scan every round (TDTK_LAZY_MOVES=0): poses, `ret`, every link's
    sums and the final points are bit-identical."""
    from importlib import import_module
    gs = import_module("3dtk_amd.graphslam")
    rng = np.random.default_rng(29)
    world = rng.uniform(-400, 400, (300000, 3))

    def make():
        scans = []
        for k in range(5):
            T = tdtk.EulerToMatrix4([3.0 * k, -1.0 * k, 2.0 * k], [0.002 * k, -0.003 * k, 0.004 * k])
            Ti = tdtk.M4inv(T)
            R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
            loc = world @ R.T + Ti[12:15] + np.random.default_rng(100 + k).normal(0, 0.05, world.shape)
            scans.append(tdtk.Scan([3.0 * k + 0.3, -1.0 * k, 2.0 * k - 0.2], [0.002 * k, -0.003 * k + 0.001, 0.004 * k], loc))
        tdtk.prepare_scans(scans, trees=True, threads=2)
        return scans
    # scan 1 is read by one link, scan 2 by two, scan 3 by three links of the launch; scan 4 by none (it only lends its tree)
    links = [(0, 1), (1, 2), (2, 3), (0, 3), (0, 2), (1, 3), (4, 1)]

    def run(lazy):
        if lazy:
            monkeypatch.delenv("TDTK_LAZY_MOVES", raising=False)
        else:
            monkeypatch.setenv("TDTK_LAZY_MOVES", "0")
        scans = make()
        gr = tdtk.Graph(5, links=links)
        rets = [gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 100.0, None) for _ in range(3)]
        # a pass of another kind in between reads what is queued as well (scan 3 has just been moved by round 3)
        pairs = tdtk.Scan.getPtPairs(scans[2], scans[3], max_dist_match2=100.0)
        rets.append(gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 100.0, None))
        poses = np.stack([s.transMat for s in scans])
        pts = [s.get_xyz_reduced() for s in scans]
        for s in scans:
            s.release()
        return rets, poses, pts, (pairs["n"], pairs["sum"])
    eager = run(False)
    lazy = run(True)
    assert eager[0] == lazy[0] and np.array_equal(eager[1], lazy[1]) and eager[3] == lazy[3]
    assert eager[3][0] > 100000
    for a, b in zip(eager[2], lazy[2]):
        assert np.array_equal(a, b)
    assert not np.array_equal(eager[2][4], make()[4].get_xyz_reduced())     # (scan 4 did move)


def test_links_that_repeat_start_warm_and_change_nothing(tdtk, gpu, lab, monkeypatch):
    """Round 5: a link that repeats -- same tree, same scan, same position of the launch, told by the handles' numbers --
    starts every search from the previous round's hit (k_search's warm radius: a point of the link's static tree bounds the
    nearest neighbour's distance whatever the scans have done since).  Same index, same d2: four rounds of lum6DEuler over
    seven links of 300K-point scans, the scans moving between the rounds, give the same `ret`, poses and points bit for bit
    with the warm start switched off (TDTK_LINK_WARM=0, lab library).  A second graph over OTHER scans through the same
    context's slots in between must not inherit anybody's hits."""
    from importlib import import_module
    gs = import_module("3dtk_amd.graphslam")
    rng = np.random.default_rng(31)
    world = rng.uniform(-400, 400, (300000, 3))

    def make(seed0, n=None):
        scans = []
        w = world if n is None else world[:n]
        for k in range(5):
            T = tdtk.EulerToMatrix4([3.0 * k, -1.0 * k, 2.0 * k], [0.002 * k, -0.003 * k, 0.004 * k])
            Ti = tdtk.M4inv(T)
            R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
            loc = w @ R.T + Ti[12:15] + np.random.default_rng(seed0 + k).normal(0, 0.05, w.shape)
            scans.append(tdtk.Scan([3.0 * k + 0.3, -1.0 * k, 2.0 * k - 0.2], [0.002 * k, -0.003 * k + 0.001, 0.004 * k], loc))
        tdtk.prepare_scans(scans, trees=True, threads=2)
        return scans
    links = [(0, 1), (1, 2), (2, 3), (0, 3), (0, 2), (1, 3), (4, 1)]

    def run(warm):
        if warm:
            monkeypatch.delenv("TDTK_LINK_WARM", raising=False)
        else:
            monkeypatch.setenv("TDTK_LINK_WARM", "0")
        scans = make(100)
        other = make(200, 280000)                   # other handles, other sizes, through the same slots
        gr = tdtk.Graph(5, links=links)
        rets = [gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 100.0, None) for _ in range(2)]
        r_other = gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(5, links=links), other, 100.0, None)
        rets += [gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 100.0, None) for _ in range(2)]
        poses = np.stack([s.transMat for s in scans])
        pts = [s.get_xyz_reduced() for s in scans]
        for s in scans + other:
            s.release()
        return rets, r_other, poses, pts
    cold = run(False)
    warm = run(True)
    assert cold[0] == warm[0] and cold[1] == warm[1] and np.array_equal(cold[2], warm[2])
    for a, b in zip(cold[3], warm[3]):
        assert np.array_equal(a, b)


def test_tree_edge_cases(tdtk, orc, gpu):
    """One point, two points, all-identical points (one degenerate bucket larger than the bucket size),
    non-finite coordinates (the reference would recurse on an empty side; we return an error)."""
    for pts in ([[1.0, 2.0, 3.0]], [[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]], np.tile([[5.0, 5.0, 5.0]], (100, 1)),
                np.arange(90, dtype=float).reshape(30, 3)):
        pts = np.asarray(pts, float)
        for bucket in (1, 20):
            kd, T = tdtk.KDtree(pts, bucket), orc.Tree(pts, bucket)
            assert kd.verify() == [0, 0, 0, 0]
            q = np.concatenate([pts[:5] + 0.25, pts[:3]])
            for md2 in (0.01, 1e18):
                idx, d2 = kd.FindClosestBatch(q, md2)
                oi, od2 = T.find_closest(q, md2)
                assert np.array_equal(idx, oi) and np.array_equal(d2, od2)
    bad = np.random.default_rng(0).uniform(-1, 1, (200, 3)); bad[17, 1] = np.nan
    with pytest.raises(tdtk.TdtkError):
        tdtk.KDtree(bad, 5)
    # an empty resident scan is legal and pairs with nothing
    m = tdtk.Scan([0, 0, 0], [0, 0, 0], np.random.default_rng(1).uniform(-1, 1, (50, 3)))
    e = tdtk.Scan([0, 0, 0], [0, 0, 0], np.zeros((0, 3)))
    r = tdtk.Scan.getPtPairs(m, e, max_dist_match2=1.0)
    assert r["n"] == 0 and r["n_queries"] == 0


# ---- octree reduction (-r), centre mode ------------------------------------------------------
@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "tiny", "grid", "line"])
@pytest.mark.parametrize("voxel", [0.5, 10.0, 1e6])
def test_octree_reduction_equals_oracle(tdtk, orc, gpu, name, voxel):
    """tdtk_reduce_octree (key sort on the device) == the recursive restatement of
    BOctTree + GetOctTreeCenter: same cells, same centres bit for bit, same depth-first order."""
    pts = _clouds()[name]
    got = tdtk.calcReducedPoints(pts, voxel)
    want = orc.octree_center(pts, voxel)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "tiny", "grid", "line"])
@pytest.mark.parametrize("voxel", [0.5, 10.0, 1e6])
@pytest.mark.parametrize("nrpts", [1, 3])
def test_octree_reduction_random_modes_equal_oracle(tdtk, orc, gpu, name, voxel, nrpts):
    """`-r <voxel> -O <nrpts>` (Scan::calcReducedPoints with reduction_nrpts >= 1, BOctTree::GetOctTreeRandom): the
    leaves, their depth-first order and the order of the points INSIDE a leaf (which rand() indexes: the reference's
    in-place z / y / x partitions level after level) come from the device; with the C library's rand() seeded alike the
    kept points equal the recursive restatement's, bit for bit and in order."""
    pts = _clouds()[name]
    got = tdtk.calcReducedPoints(pts, voxel, nrpts=nrpts, seed=1234 + nrpts)
    want = orc.octree_random(pts, voxel, nrpts, seed=1234 + nrpts)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_octree_reduction_random_modes_full_size_and_refusals(tdtk, orc, gpu, k5):
    """1M points: one random point per leaf == the oracle; one point per centre-mode cell, each inside its cell; the two
    modes the reference cannot run reproducibly are refused (TDTK_EUNSUP), not approximated."""
    _, m, _ = k5
    got = tdtk.calcReducedPoints(m, 25.0, nrpts=1, seed=7)
    want = orc.octree_random(m, 25.0, 1, seed=7)
    assert np.array_equal(got, want)
    centres = tdtk.calcReducedPoints(m, 25.0)
    assert len(got) == len(centres)
    size = (0.5 * (m.max(0) - m.min(0))).max() + 1.0
    while size > 25.0:
        size /= 2.0
    assert np.all(np.abs(got - centres).max(1) <= size)          # same leaf, in the same depth-first position
    with pytest.raises(tdtk.TdtkError):
        tdtk.calcReducedPoints(m[:1000], 25.0, nrpts=-1)
    with pytest.raises(tdtk.TdtkError):
        tdtk.calcReducedPoints(m[:1000], 25.0, nrpts=2, rm_scatter=True)


def test_octree_reduction_dat_and_icp(tdtk, orc, gpu):
    """-r 10 on the bundled scans, then the pairwise ICP of scan001 onto scan000 on the reduced clouds
    against the numpy restatement run on the oracle's reduced clouds."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    S, O = [], []
    for k in range(2):
        pose, pts = z["pose%03d" % k], z["scan%03d" % k]
        got = tdtk.calcReducedPoints(pts, 10.0)
        want = orc.octree_center(pts, 10.0)
        assert np.array_equal(got, want) and 0 < len(got) < len(pts)
        S.append(tdtk.Scan(pose[:3], pose[3:], got))
        O.append(io.OScan(pose[:3], pose[3:], want))
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 30, quiet=True, epsilonICP=1e-5)
    it = icp.match(S[0], S[1])
    oit, otr = io.match(O[0], O[1], 1, 625.0, 30, 1e-5)
    assert it == oit
    assert np.allclose(S[1].transMat, O[1].transMat, rtol=1e-9, atol=1e-9)


def test_octree_reduction_full_size_properties(tdtk, orc, gpu, k5):
    """1M points, voxel 25: equals the oracle; every input point lies in exactly the emitted cell its
    own descent names (checked through the cell containing it), cells are unique and sorted DFS."""
    _, m, _ = k5
    got = tdtk.calcReducedPoints(m, 25.0)
    want = orc.octree_center(m, 25.0)
    assert np.array_equal(got, want)
    assert len(np.unique(got, axis=0)) == len(got)
    # leaf half size: root = 1000-ish + 1, halved until <= 25
    lo, hi = m.min(0), m.max(0)
    size = (0.5 * (hi - lo)).max() + 1.0
    while size > 25.0:
        size /= 2.0
    kd = tdtk.KDtree(got, 20)
    idx, d2 = kd.FindClosestBatch(m[:200000], 1e18)
    assert np.all(np.abs(m[:200000] - got[idx]).max(1) <= size)
    # idempotent at the same voxel only in cell count terms: reducing the centres keeps one per cell
    again = tdtk.calcReducedPoints(got, 25.0)
    assert len(again) <= len(got)


def test_octree_reduction_edge_cases(tdtk, orc, gpu):
    one = np.array([[3.0, -2.0, 7.5]])
    assert np.array_equal(tdtk.calcReducedPoints(one, 10.0), orc.octree_center(one, 10.0))
    same = np.tile([[5.0, 5.0, 5.0]], (1000, 1))
    g = tdtk.calcReducedPoints(same, 0.1)
    assert len(g) == 1 and np.array_equal(g, orc.octree_center(same, 0.1))
    assert len(tdtk.calcReducedPoints(np.zeros((0, 3)), 10.0)) == 0
    assert len(tdtk.calcReducedPoints(same, -1.0)) == 1000            # no reduction requested
    far = np.array([[0.0, 0.0, 0.0], [1e9, 1e9, 1e9]])
    with pytest.raises(tdtk.TdtkError):
        tdtk.calcReducedPoints(far, 1e-3)                             # > 21 levels


def test_clpairs_graph_and_lum_round(tdtk, orc, gpu):
    """graphSlam6D::matchGraph6Dautomatic(allScans, nrIt, clpairs, loopsize): the all-ordered-pairs graph
    (pair counts exact) and one LUM iteration over it -- links in both directions and links that end at
    the fixed scan 0 -- against the numpy restatement."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    S, O = _dat_scans(tdtk.Scan, z), _dat_scans(io.OScan, z)
    for pr in b1["pairs"]:
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1]); O[i].mergeCoordinatesWithRoboterPosition(O[i - 1])
        for a in pr["alignxf"]:
            S[i].transform(np.array(a)); O[i].transform(np.array(a))
    links, counts = io.graph_links_clpairs(O, 30000, 625.0)
    gr = tdtk.computeGraph6Dautomatic(S, 30000, 625.0)
    assert gr.pair_counts == counts
    assert list(zip(gr.frm, gr.to)) == links and 0 < len(links) < 6
    gr_all = tdtk.computeGraph6Dautomatic(S, 100, 625.0)
    assert gr_all.getNrLinks() == 6
    ret = tdtk.lum6DEuler(None, 25.0, 25.0).doGraphSlam6D(gr_all, S, 1)
    oret = io.lum_iteration(list(zip(gr_all.frm, gr_all.to)), O, 625.0)[0]
    assert abs(ret - oret) < 1e-7 * max(1.0, oret)
    for s, o in zip(S, O):
        assert np.abs(s.get_rPos() - o.rPos).max() < 1e-7 and np.abs(s.get_rPosTheta() - o.rPosTheta).max() < 1e-9


def _dat_after_icp(tdtk, cls_list):
    """the bundled scans at the B1 final poses, one copy per class in cls_list"""
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    out = []
    for cls in cls_list:
        S = _dat_scans(cls, z)
        for pr in b1["pairs"]:
            i = pr["cur"]
            S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
            for a in pr["alignxf"]:
                S[i].transform(np.array(a))
        out.append(S)
    return out


def test_lum6DQuat_iterations_vs_oracle(tdtk, orc, gpu):
    """-G 2: two lum6DQuat::doGraphSlam6D iterations on dat/ (chain + one closure) against the numpy
    restatement: ret, poses and moved points."""
    from oracle import icp_oracle as io
    S, O = _dat_after_icp(tdtk, [tdtk.Scan, io.OScan])
    links = [(0, 1), (1, 2), (0, 2)]
    gr = tdtk.Graph(3, links=links); gr.nrScans = 3
    slam = tdtk.lum6DQuat(None, 25.0, 25.0, epsilonLUM=-1.0)
    for _ in range(2):
        ret = slam.doGraphSlam6D(gr, S, 1)
        oret = io.lumquat_iteration(links, O, 625.0)[0]
        assert abs(ret - oret) < 1e-7 * max(1.0, oret)
    for s, o in zip(S, O):
        assert _rel(s.get_transMat(), o.transMat) < 1e-8
        assert np.abs(s.get_xyz_reduced() - o.xyz).max() < 1e-6
        assert np.abs(s.get_rPosQuat() - o.get_rPosQuat()).max() < 1e-9


def test_ghelix6DQ2_iterations_vs_oracle(tdtk, orc, gpu):
    """-G 3: ghelix6DQ2::doGraphSlam6D with nrIt = 2 (B and bd carried from the first iteration into the
    second, as the reference does) against the numpy restatement."""
    from oracle import icp_oracle as io
    S, O = _dat_after_icp(tdtk, [tdtk.Scan, io.OScan])
    links = [(0, 1), (1, 2), (0, 2)]
    gr = tdtk.Graph(3, links=links); gr.nrScans = 3
    ret = tdtk.ghelix6DQ2(None, 25.0, 25.0, epsilonLUM=-1.0).doGraphSlam6D(gr, S, 2)
    Bm = bd = None
    for _ in range(2):
        oret, Bm, bd, _ccs = io.ghelix_iteration(links, O, 625.0, Bm, bd)
    assert abs(ret - oret) < 1e-6 * max(1.0, oret)
    for s, o in zip(S, O):
        assert _rel(s.get_transMat(), o.transMat) < 1e-7
        assert np.abs(s.get_xyz_reduced() - o.xyz).max() < 1e-5


def test_scans_transform2_equals_two_transforms(tdtk, gpu):
    """tdtk_scans_transform2 (one launch, both matrices) == Scan::transform twice, bit for bit."""
    rng = np.random.default_rng(3)
    pts = rng.uniform(-500, 500, (5000, 3)); nrm = rng.normal(size=(5000, 3))
    A = [tdtk.EulerToMatrix4(rng.uniform(-50, 50, 3), rng.uniform(-1, 1, 3)) for _ in range(4)]
    a = [tdtk.Scan([1, 2, 3], [0.1, 0.2, 0.3], pts, nrm) for _ in range(2)]
    b = [tdtk.Scan([1, 2, 3], [0.1, 0.2, 0.3], pts, nrm) for _ in range(2)]
    for s in a + b:
        _ = s.handle
    from importlib import import_module
    sl = import_module("3dtk_amd.slam6d")
    sl.transform_many(a, [A[0], A[1]], [A[2], A[3]], "LUM")
    b[0].transform(A[0], "INVALID"); b[0].transform(A[2], "LUM")
    b[1].transform(A[1], "INVALID"); b[1].transform(A[3], "LUM")
    for x, y in zip(a, b):
        assert np.array_equal(x.get_xyz_reduced(), y.get_xyz_reduced())
        assert np.array_equal(x.transMat, y.transMat) and np.array_equal(x.dalignxf, y.dalignxf)
        assert len(x.frames) == len(y.frames)


def test_ten_million_point_model_with_normals(tdtk, orc, gpu):
    """configs[4] scale: a 10M-point model (leaf-table / packed reference limits, 24-bit indices and beyond),
    the device-built tree against the host builder, indices bit-exact on a query sample; a 2M-point scan
    with normals: whole-scan pass and the point-to-plane (NAPX) step equal to the oracle's, and a
    point-to-point ICP that recovers the known motion."""
    from oracle import icp_oracle as io
    rng = np.random.default_rng(77)
    M = 10_000_000
    m = rng.uniform(-2000, 2000, (M, 3))
    kd = tdtk.KDtree(m, 20)
    assert kd.verify() == [0, 0, 0, 0]
    T = orc.Tree(m, 20)
    inf, st = kd.info(), T.stats()
    assert (inf["n_internal"], inf["n_leaves"], inf["max_depth"]) == (st["internal"], st["leaves"], st["depth"])
    q = m[rng.integers(0, M, 300000)] + rng.normal(0, 4.0, (300000, 3))
    idx, d2 = kd.FindClosestBatch(q, 400.0)
    oi, od2 = T.find_closest(q, 400.0, 8)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od2)
    del kd
    # a 2M-point scan of the same surface sample, moved by a small rigid motion
    sel = rng.choice(M, 2_000_000, replace=False)
    Tgt = io.euler_to_matrix4([3.0, -2.0, 1.5], [0.004, -0.003, 0.005])
    inv, _ = orc.m4inv(Tgt)
    d = m[sel] + rng.normal(0, 0.3, (len(sel), 3))
    orc.transform_points(inv, d)
    nrm = rng.normal(size=(len(sel), 3)); nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], m)
    S1 = tdtk.Scan([0, 0, 0], [0, 0, 0], d, nrm)
    r = tdtk.Scan.getPtPairs(S0, S1, 0, 0, 400.0, 2, tdtk.WANT_NAPX)    # CLOSEST_PLANE_SIMPLE, as -a 10 is meant to run
    o = T.get_pt_pairs(np.eye(4).reshape(16), d, nrm, 0, None, 2, 400.0)
    assert r["n"] == o["n"] and abs(r["sum"] - o["sum"]) < 1e-9 * o["sum"]
    rms, a = tdtk.icp6D_NAPX(True).Align_Parallel(r)
    orms, oa = io.align(10, o["p1"], o["p2"], o["centroid_m"] / o["n"], o["centroid_d"] / o["n"], o["pn"])
    assert abs(rms - orms) < 1e-9 * orms and np.abs(a - oa).max() < 1e-8
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 20.0, 15, quiet=True, epsilonICP=1e-6)
    icp.match(S0, S1)
    assert np.abs(S1.get_transMat() - Tgt).max() < 5e-3
    assert np.abs(S1.get_transMat()[:12] - Tgt[:12]).max() < 1e-5


def test_eight_million_queries_cold_start_deep_stacks(tdtk, orc, gpu):
    """More queries than the largest grid has lanes (the persistent-lane kernel then gives every wave a longer
    slab instead of adding waves), cold start with an unbounded radius on a 19-deep tree: every first descent
    pushes a far child per level, so the per-lane stacks spill 14 levels deep into the overflow area, which is
    sized for the lanes of the grid that is actually launched.  All 8M indices and distances against the oracle."""
    rng = np.random.default_rng(2026)
    M, K = 2_000_000, 8_000_000
    m = rng.uniform(-2000, 2000, (M, 3))
    m[:200000] = rng.normal(0, 3.0, (200000, 3)) + rng.uniform(-1500, 1500, (200, 3)).repeat(1000, axis=0)   # deepens the tree
    kd, T = tdtk.KDtree(m, 20), orc.Tree(m, 20)
    assert kd.info()["max_depth"] >= 18
    q = rng.uniform(-2100, 2100, (K, 3))
    nt = max(8, min(64, os.cpu_count() or 8))
    for md2 in (1e18, 400.0):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, nt)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od2)
    # the same through a resident scan (Morton-sorted, fused pair sums): pair count and sums against the index list
    S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], m); S0.kd = kd
    S1 = tdtk.Scan([0, 0, 0], [0, 0, 0], q)
    r = tdtk.Scan.getPtPairs(S0, S1, 0, 0, 400.0, 0, 0, want_idx=True)
    assert np.array_equal(r["idx"], oi)
    found = oi >= 0
    assert r["n"] == int(found.sum())
    assert abs(r["sum"] - od2[found].sum()) <= 1e-10 * od2[found].sum()
    cm = m[oi[found]].mean(axis=0)
    assert np.abs(np.asarray(r["centroid_m"]) - cm).max() < 1e-7


def test_config3_shape_graphslam_sharded(tdtk, orc, gpu):
    """configs[3] at reduced size (16 scans x 40K points on a closed circle, chain + loop closures): one
    lum6DEuler iteration of the native path against the numpy restatement, and the sharded exchange
    emulated for 2, 4 and 8 ranks on this one GPU -- every rank's links through tdtk_lum_links, blocks
    summed as the all-reduce would, C-side scatter + solve -- gives bit-identical X for every rank count."""
    import sys
    from importlib import import_module
    from oracle import icp_oracle as io
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    gs = import_module("3dtk_amd.graphslam")
    capi = import_module("3dtk_amd._capi")
    import ctypes as C
    raw = bench.make_graphslam_scans(16, 40000, seed=11)
    S = [tdtk.Scan(p, th, loc) for (p, th, loc) in raw]
    O = [io.OScan(p, th, loc) for (p, th, loc) in raw]
    gr = tdtk.Graph(16, 700.0 ** 2, 5, S)
    links = list(zip(gr.frm, gr.to))
    assert links == io.graph_links(O, 700.0 ** 2, 5) and len(links) > 15
    # emulated ranks
    def blocks_for(world):
        out = np.zeros((len(links), 42))
        for r in range(world):
            mine = gs.shard_links(gr, r, world)
            if not mine:
                continue
            nl = len(mine)
            first = (C.c_void_p * nl)(*[S[links[i][0]].getSearchTree()._h for i in mine])
            second = (C.c_void_p * nl)(*[S[links[i][1]].handle for i in mine])
            dal = np.ascontiguousarray(np.stack([S[links[i][0]].dalignxf for i in mine]))
            Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
            capi.check(capi.lib().tdtk_lum_links(nl, first, capi.dptr(dal), second, 625.0, capi.dptr(Cm), capi.dptr(CD),
                                                 m, capi.dptr(ss)))
            part = np.zeros((len(links), 42)); part[mine, :36] = Cm; part[mine, 36:] = CD
            out = out + part                     # what the all-reduce does: every link has one owner
        return out
    Xs = []
    for world in (1, 2, 4, 8):
        b = blocks_for(world)
        Xs.append(gs.lum_reduce_solve(gr, list(range(len(links))), b[:, :36], b[:, 36:], 1))
    for X in Xs[1:]:
        assert np.array_equal(X, Xs[0])
    ret = gs.lum_iteration_native(gr, S, 625.0)
    oret, _, _, Xo = io.lum_iteration(links, O, 625.0)
    np.testing.assert_allclose(Xs[0], Xo, rtol=1e-6, atol=1e-9)
    assert abs(ret - oret) < 1e-7 * max(1.0, oret)
    for s, o in zip(S, O):
        assert np.abs(s.get_rPos() - o.rPos).max() < 1e-6 and np.abs(s.get_rPosTheta() - o.rPosTheta).max() < 1e-9


def test_find_closest_randomized_shapes(tdtk, orc, gpu):
    """Seeded random clouds of random size, bucket size, quantisation (forcing exact duplicates and
    equidistant candidates), anisotropy and search radius; queries on, near and far from the cloud.
    Indices and squared distances bit-exact against the oracle in every case; the device-built tree equals the
    host builder's."""
    rng = np.random.default_rng(2024)
    for case in range(40):
        n = int(rng.choice([1, 2, 3, 17, 64, 65, 300, 2000, 9000]))
        scale = rng.choice([1.0, 100.0, 1e4]) * np.array([1.0, rng.choice([1.0, 0.1, 1e-3]), rng.choice([1.0, 0.5])])
        pts = rng.normal(0, 1, (n, 3)) * scale + rng.uniform(-1e3, 1e3, 3)
        q_step = rng.choice([0.0, 0.0, 0.5, 8.0])
        if q_step > 0:
            pts = np.round(pts / q_step) * q_step                # duplicates and lattice ties
        bucket = int(rng.choice([1, 2, 5, 20, 64]))
        kd, T = tdtk.KDtree(pts, bucket), orc.Tree(pts, bucket)
        assert kd.verify() == [0, 0, 0, 0], case
        k = 400
        q = np.concatenate([pts[rng.integers(0, n, k)],                                   # on the points
                            pts[rng.integers(0, n, k)] + rng.normal(0, 0.3, (k, 3)) * scale.max() * 0.01,
                            pts[rng.integers(0, n, k)] + (np.round(rng.normal(0, 2, (k, 3))) * (q_step if q_step else 1.0)),
                            rng.uniform(-3e4, 3e4, (k, 3))])                               # far away
        for md2 in (float(rng.choice([1e-6, 1.0, 100.0])), 1e18):
            idx, d2 = kd.FindClosestBatch(q, md2)
            oi, od2 = T.find_closest(q, md2)
            assert np.array_equal(idx, oi), (case, n, bucket, q_step, md2)
            assert np.array_equal(d2[idx >= 0], od2[oi >= 0])


def test_big_batch_kernel_on_lattice_ties(tdtk, orc, gpu):
    """The persistent-lane kernel (batches of 256K queries and more) on a lattice cloud where most queries
    have several equidistant candidates: the first-visited one must win, as in the reference."""
    rng = np.random.default_rng(77)
    pts = np.round(rng.uniform(-40, 40, (60000, 3)))             # many exact duplicates, unit lattice
    kd, T = tdtk.KDtree(pts, 10), orc.Tree(pts, 10)
    q = np.round(rng.uniform(-45, 45, (300000, 3)) * 2) / 2      # on lattice points and exactly between them
    for md2 in (0.75, 4.0, 1e18):
        idx, d2 = kd.FindClosestBatch(q, md2)
        oi, od2 = T.find_closest(q, md2, 8)
        assert np.array_equal(idx, oi) and np.array_equal(d2[idx >= 0], od2[oi >= 0])
    S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], pts, bucketSize=10)
    S1 = tdtk.Scan([0.5, 0, 0], [0, 0, 0], q)
    r = tdtk.Scan.getPtPairs(S0, S1, 0, 0, 4.0, 0, 0, None, True)
    o = orc.Tree(pts, 10).get_pt_pairs(np.eye(4).reshape(16), q + np.array([0.5, 0, 0]), None, 0, None, 0, 4.0)
    assert r["n"] == o["n"] and np.array_equal(r["idx"], o["idx"])


def test_doicp_prefetch_is_transparent(tdtk, gpu):
    """icp6D::doICP with the next scan uploaded and its tree built on a second host thread while the current
    pair is matched: poses and moved points bit-identical to the serial order."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    raw = bench.make_graphslam_scans(6, 30000, seed=3)
    out = []
    for pf in (False, True):
        scans = [tdtk.Scan(p, th, loc) for (p, th, loc) in raw]
        tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 20, quiet=True, epsilonICP=1e-6).doICP(scans, prefetch=pf)
        out.append((np.stack([s.transMat for s in scans]), scans[-1].get_xyz_reduced(), [len(s.frames) for s in scans]))
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]


def test_point_point_error(tdtk, orc, gpu):
    """icp6D::Point_Point_Error on the bundled pair, before and after matching."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    S, O = _dat_scans(tdtk.Scan, z), _dat_scans(io.OScan, z)
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 10, quiet=True, epsilonICP=1e-5)
    for scale_max in (0.000001, 0.01):
        e, n = icp.Point_Point_Error(S[0], S[1], 25.0, scale_max)
        oe, on = io.point_point_error(O[0], O[1], 25.0, scale_max)
        assert n == on and abs(e - oe) <= 1e-11 * abs(oe)
    S[1].mergeCoordinatesWithRoboterPosition(S[0]); O[1].mergeCoordinatesWithRoboterPosition(O[0])
    icp.match(S[0], S[1]); io.match(O[0], O[1], 1, 625.0, 10, 1e-5)
    e, n = icp.Point_Point_Error(S[0], S[1], 20.0, 0.001)
    oe, on = io.point_point_error(O[0], O[1], 20.0, 0.001)
    assert n == on and abs(e - oe) <= 1e-9 * abs(oe)


@pytest.mark.parametrize("backend", [1, 2, 3, 4])
def test_graph_backends_blocks_are_rank_count_independent(tdtk, gpu, backend):
    """tdtk_graph_link_blocks per link does not depend on which other links share the batch, so any dealing of
    the links over ranks reproduces the single-rank blocks bit for bit (what makes X independent of N)."""
    import ctypes as C, sys
    from importlib import import_module
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    gs = import_module("3dtk_amd.graphslam"); capi = import_module("3dtk_amd._capi")
    raw = bench.make_graphslam_scans(8, 20000, seed=5)
    S = [tdtk.Scan(p, th, loc) for (p, th, loc) in raw]
    gr = tdtk.Graph(8, 900.0 ** 2, 3, S)
    nl = gr.getNrLinks()
    Bn = capi.lib().tdtk_graph_block_doubles(backend)
    def blocks(idx):
        n = len(idx)
        first = (C.c_void_p * n)(*[S[gr.getLink(i, 0)].getSearchTree()._h for i in idx])
        second = (C.c_void_p * n)(*[S[gr.getLink(i, 1)].handle for i in idx])
        dal = np.ascontiguousarray(np.stack([S[gr.getLink(i, 0)].dalignxf for i in idx]))
        out = np.empty((n, Bn))
        capi.check(capi.lib().tdtk_graph_link_blocks(backend, n, first, capi.dptr(dal), second, 625.0, capi.dptr(out)))
        return out
    full = blocks(list(range(nl)))
    for world in (2, 3):
        part = np.zeros_like(full)
        for r in range(world):
            mine = gs.shard_links(gr, r, world)
            if mine:
                part[mine] = blocks(mine)
        assert np.array_equal(part, full)
    assert np.abs(full).sum() > 0


def test_match_graph6d_automatic(tdtk, orc, gpu):
    """matchGraph6Dautomatic (slam6D.cc:387-548, no ELCH): sequential ICP along a closed circle of scans, loop
    detection, global LUM rounds -- with the next scan prefetched on a second thread -- against the numpy
    restatement: same number of rounds, poses within tolerance."""
    import sys
    from oracle import icp_oracle as io
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    raw = bench.make_graphslam_scans(10, 15000, seed=21)
    S = [tdtk.Scan(p, th, loc) for (p, th, loc) in raw]
    O = [io.OScan(p, th, loc) for (p, th, loc) in raw]
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 30, quiet=True, epsilonICP=1e-5)
    slam = tdtk.lum6DEuler(icp, 25.0, 25.0, epsilonLUM=0.5)
    rounds = tdtk.matchGraph6Dautomatic(900.0, 4, S, icp, False, slam, 5, 0.05, 25.0, eP=False)
    orounds = io.match_graph6d_automatic(900.0, 4, O, 1, 625.0, 30, 1e-5, 5, 0.05, 625.0, eP=False)
    assert rounds == orounds and rounds >= 2
    for s, o in zip(S, O):
        assert np.abs(s.get_rPos() - o.rPos).max() < 1e-5 and np.abs(s.get_rPosTheta() - o.rPosTheta).max() < 1e-7
        assert np.abs(s.get_xyz_reduced() - o.xyz).max() < 1e-4


# ---- normals: Scan::calcNormals = calculateNormalsApxKNN(k = 10, eps = 1.0) -----------------------------
def _normal_clouds():
    rng = np.random.default_rng(5)
    c = dict(_clouds())
    p = rng.uniform(-300, 300, (40000, 3)); p[:, 2] = 40.0 + 0.1 * p[:, 0] + rng.normal(0, 0.5, len(p))
    c["noisy_plane"] = p
    # a room seen from inside: four walls and a floor, duplicates where they meet
    u, v = rng.uniform(-500, 500, 12000), rng.uniform(0, 250, 12000)
    walls = [np.stack([u[:3000], np.full(3000, 500.0), v[:3000]], 1), np.stack([u[3000:6000], np.full(3000, -500.0), v[3000:6000]], 1),
             np.stack([np.full(3000, 500.0), u[6000:9000], v[6000:9000]], 1), np.stack([np.full(3000, -500.0), u[9000:], v[9000:]], 1),
             np.stack([u[:4000], u[4000:8000], np.zeros(4000)], 1)]
    c["room"] = np.concatenate(walls) + rng.normal(0, 0.3, (16000, 3))
    c["eleven"] = rng.uniform(-1, 1, (11, 3))
    c["exp_line"] = np.outer(2.0 ** np.arange(60), [1.0, 0.5, 0.25])     # every split slides: a 59-deep comb
    return c


@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "grid", "line", "noisy_plane", "room",
                                  "eleven", "exp_line"])
def test_normals_equal_oracle(tdtk, orc, gpu, name):
    """tdtk_normals_apx_knn == the restatement of calculateNormalsApxKNN (itself bit-identical to the vendored
    ANN + newmat): neighbour lists in list order and normals, bit for bit."""
    pts = _normal_clouds()[name]
    rPos = np.array([3.0, -2.0, 10.0])
    want, wknn = orc.normals_apx_knn(pts, 10, rPos, 1.0, want_knn=True)
    got, gknn = tdtk.calculateNormalsApxKNN(pts, 10, rPos, 1.0, want_knn=True)
    assert np.array_equal(gknn, wknn)
    assert np.array_equal(got, want, equal_nan=True)


@pytest.mark.parametrize("k,eps", [(1, 1.0), (5, 0.3), (10, 0.0), (16, 2.0), (32, 1.0)])
def test_normals_other_k_eps(tdtk, orc, gpu, k, eps):
    pts = _normal_clouds()["room"][:6000]
    want, wknn = orc.normals_apx_knn(pts, k, [0.0, 0.0, 100.0], eps, want_knn=True)
    got, gknn = tdtk.calculateNormalsApxKNN(pts, k, [0.0, 0.0, 100.0], eps, want_knn=True)
    assert np.array_equal(gknn, wknn) and np.array_equal(got, want, equal_nan=True)


def test_lab_switch_of_the_round6_ann_build_gives_the_same_lists(tdtk, gpu, lab, monkeypatch):
    """The ANN build's Hoare passes as scan + swap with the cells' counts inside the first pass's scan (round 6) against round 3's
    count / misplaced / scan / list / swap launches (TDTK_ANN_PART=0, lab): the same 10-NN lists and normals bit for bit, on a
    uniform cloud, on one whose rounded coordinates put points ON cutting planes (the second pass has work) and on a plane."""
    rng = np.random.default_rng(608)
    clouds = [rng.uniform(-300.0, 300.0, (300000, 3)), np.round(rng.uniform(-50.0, 50.0, (250000, 3)), 1)]
    pl = rng.uniform(-300.0, 300.0, (200000, 3)); pl[:, 2] = 0.05 * pl[:, 0] + rng.normal(0, 0.5, len(pl)); clouds.append(pl)
    for p in clouds:
        a, ak = tdtk.calculateNormalsApxKNN(p, 10, [0.0, 0.0, 400.0], 1.0, want_knn=True)
        monkeypatch.setenv("TDTK_ANN_PART", "0")
        b, bk = tdtk.calculateNormalsApxKNN(p, 10, [0.0, 0.0, 400.0], 1.0, want_knn=True)
        monkeypatch.delenv("TDTK_ANN_PART")
        assert np.array_equal(ak, bk) and np.array_equal(a, b, equal_nan=True)


def test_normals_golden_and_reference_library(tdtk, orc, gpu):
    """K7 fixture (vendored ANN + newmat, generated in the build container) and, where oracle/_ref travelled,
    the library itself."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(G, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    z = np.load(os.path.join(G, "k7_ann_normals.npz"))
    for tag, pts in mg.k7_clouds().items():
        got, knn = tdtk.calculateNormalsApxKNN(pts, 10, [0.0, 0.0, 0.0], 1.0, want_knn=True)
        assert np.array_equal(knn, z[tag + "_knn"]) and np.array_equal(got, z[tag + "_normals"])
        if orc.have_ref():
            assert np.array_equal(got, orc.normals_apx_knn(pts, 10, [0.0, 0.0, 0.0], 1.0, "ref"))


def test_normals_dat_scan_and_resident(tdtk, orc, gpu):
    """A whole bundled scan (81 360 points); Scan.calcNormals before going resident == the oracle's normals
    moved with the points; on a resident scan == the oracle on the downloaded points."""
    z = np.load(os.path.join(G, "dat_scans.npz"))
    pts = z["scan001"]
    pose = z["pose001"]
    want = orc.normals_apx_knn(pts, 10, pose[:3], 1.0)
    sc = tdtk.Scan(pose[:3], pose[3:], pts).calcNormals()
    assert np.array_equal(sc._local_n, want)
    # resident: points have moved to the global frame; recompute there
    h = sc.handle
    xyz = np.empty_like(pts); nrm = np.empty_like(pts)
    from importlib import import_module
    cap = import_module("3dtk_amd._capi")
    cap.check(cap.lib().tdtk_scan_download(h, cap.dptr(xyz), cap.dptr(nrm)))
    moved = want.copy(); orc.transform_normals(sc.transMatOrg, moved)
    assert np.allclose(nrm, moved, rtol=0, atol=1e-15)
    sc.calcNormals()
    cap.check(cap.lib().tdtk_scan_download(h, cap.dptr(xyz), cap.dptr(nrm)))
    assert np.array_equal(nrm, orc.normals_apx_knn(xyz, 10, sc.rPos, 1.0))


def test_normals_feed_point_to_plane_pairs(tdtk, orc, gpu):
    """the computed normals drive pairing mode 2 and the NAPX minimizer (-a 10 -z) like supplied ones"""
    rng = np.random.default_rng(8)
    m = _normal_clouds()["room"]
    T = tdtk.EulerToMatrix4([2.0, -1.0, 0.5], [0.01, -0.02, 0.015])
    d = m[rng.permutation(len(m))[:8000]] + rng.normal(0, 0.05, (8000, 3))
    nrm = tdtk.calculateNormalsApxKNN(d, 10, [0.0, 0.0, 100.0], 1.0)
    assert np.array_equal(nrm, orc.normals_apx_knn(d, 10, [0.0, 0.0, 100.0], 1.0))
    kd, Tm = tdtk.KDtree(m), orc.Tree(m)
    r = kd.getPtPairs(T, d, normal_r=nrm, pairing_mode=2, max_dist_match2=400.0)
    o = Tm.get_pt_pairs(T, d, nrm, 0, len(d), 2, 400.0)
    assert r["n"] == o["n"] and np.array_equal(r["idx"], o["idx"])


def test_normals_errors(tdtk, gpu):
    with pytest.raises(tdtk.TdtkError):
        tdtk.calculateNormalsApxKNN(np.zeros((0, 3)), 10, [0, 0, 0], 1.0)           # scan.cc:408
    with pytest.raises(tdtk.TdtkError):
        tdtk.calculateNormalsApxKNN(np.zeros((5, 3)), 10, [0, 0, 0], 1.0)           # k > n: the library aborts
    bad = np.random.default_rng(0).uniform(-1, 1, (200, 3)); bad[17, 1] = np.nan
    with pytest.raises(tdtk.TdtkError):
        tdtk.calculateNormalsApxKNN(bad, 10, [0, 0, 0], 1.0)
    with pytest.raises(tdtk.TdtkError):
        tdtk.calculateNormalsApxKNN(np.zeros((50, 3)), 33, [0, 0, 0], 1.0)


def test_normals_full_size(tdtk, orc, gpu):
    """1M points (BASELINE configs' scan size): list hash and normals against the oracle on a 1M cloud would take
    the CPU minutes, so: the oracle on a 200k sub-cloud bit for bit, and at 1M the size-independent properties --
    unit length, orientation towards the sensor, self as first neighbour, normals close to the plane's."""
    rng = np.random.default_rng(21)
    p = rng.uniform(-1000, 1000, (1000000, 3)); p[:, 2] = 0.05 * p[:, 0] + rng.normal(0, 1.0, len(p))
    sub = p[:200000]
    want, wk = orc.normals_apx_knn(sub, 10, [0.0, 0.0, 500.0], 1.0, want_knn=True)
    got, gk = tdtk.calculateNormalsApxKNN(sub, 10, [0.0, 0.0, 500.0], 1.0, want_knn=True)
    assert np.array_equal(gk, wk) and np.array_equal(got, want)
    n, knn = tdtk.calculateNormalsApxKNN(p, 10, [0.0, 0.0, 500.0], 1.0, want_knn=True)
    assert np.abs(np.linalg.norm(n, axis=1) - 1.0).max() < 1e-14
    assert (np.einsum("ij,ij->i", n, p - np.array([0.0, 0.0, 500.0])) >= 0).all()
    assert np.array_equal(knn[:, 0], np.arange(len(p)))
    assert (np.abs(n[:, 2]) > 0.5).mean() > 0.95        # a slightly tilted plane, noise ~ half the point spacing


def test_normals_10m_properties(tdtk, gpu):
    """A 10M-point scan (configs[4]'s scan size): the size-independent properties of the ANN answer -- every list
    sorted, the query itself first, and the library's guarantee that the i-th reported neighbour is at most
    (1 + eps) times farther than the true i-th nearest neighbour (checked against brute force on a sample) -- plus
    unit normals oriented towards the sensor."""
    rng = np.random.default_rng(33)
    n = 10000000
    p = rng.uniform(-3000, 3000, (n, 3)); p[:, 2] = 0.02 * p[:, 1] + rng.normal(0, 2.0, n)
    rp = np.array([0.0, 0.0, 800.0])
    nrm, knn = tdtk.calculateNormalsApxKNN(p, 10, rp, 1.0, want_knn=True)
    assert np.array_equal(knn[:, 0], np.arange(n, dtype=np.int32))
    assert np.abs(np.linalg.norm(nrm, axis=1) - 1.0).max() < 1e-14
    assert (np.einsum("ij,ij->i", nrm, p - rp) >= 0).all()
    for i in rng.integers(0, n, 200):
        d2 = ((p - p[i]) ** 2).sum(axis=1)
        true = np.sort(np.partition(d2, 10)[:10])
        got = d2[knn[i]]
        assert (np.diff(got) >= 0).all() and (got <= 4.0 * true * (1 + 1e-12)).all() and (got >= true).all()


@pytest.mark.parametrize("algo,mode", [(1, 2), (10, 2), (1, 1)])
def test_dat_icp_with_computed_normals(tdtk, orc, gpu, algo, mode):
    """slam6D -z (point-to-plane pairs, mode 2) / --normal_shoot-simple (mode 1) on the bundled scans: the normals
    come from Scan::calcNormals on the device (oracle side: the restated calculateNormalsApxKNN), then the pairwise
    ICP of scan001 onto scan000 -- pair counts per iteration exact, pose within tolerance."""
    from oracle import icp_oracle as io
    z = np.load(os.path.join(G, "dat_scans.npz"))
    S, O = [], []
    for k in range(2):
        pts, pose = z["scan%03d" % k], z["pose%03d" % k]
        s = tdtk.Scan(pose[:3], pose[3:], pts).calcNormals()
        on = orc.normals_apx_knn(pts, 10, pose[:3], 1.0)
        assert np.array_equal(s._local_n, on)
        S.append(s); O.append(io.OScan(pose[:3], pose[3:], pts, on))
    mini = tdtk.icp6D_QUAT(True) if algo == 1 else tdtk.icp6D_NAPX(True)
    icp = tdtk.icp6D(mini, 25.0, 15, quiet=True, epsilonICP=1e-5)
    it = icp.match(S[0], S[1], pairing_mode=mode)
    oit, otr = io.match(O[0], O[1], algo, 625.0, 15, 1e-5, mode)
    assert it == oit and [int(r[0]) for r in icp.last["trace"]] == [t[0] for t in otr]
    np.testing.assert_allclose(icp.last["trace"][:, 1], [t[1] for t in otr], rtol=1e-7)
    assert _rel(S[1].get_transMat(), O[1].transMat) < 1e-7


def test_bench_graphslam_two_ranks_on_this_box(gpu):
    """bench.py's N > 1 leg end to end (torch.distributed.run, links dealt over the ranks, per-link exchange, every
    rank solving and moving its resident scans), with both ranks sharing this box's GPU and gloo carrying the exchange
    (RCCL refuses two ranks on one device): the iteration's result equals the one-process run bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    common = ["--workload", "graphslam", "--scans", "12", "--points", "60000", "--steps", "3", "--warmup", "1"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, cwd=root,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    import socket
    with socket.socket() as sk:                      # a free rendezvous port on this box
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TDTK_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2"] + common, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and a["config"]["links"] == b["config"]["links"]
    assert a["last_ret"] == b["last_ret"]
    # the RCCL branch itself: one rank under torch.distributed.run with the nccl backend (device mapping, device_id,
    # banner kept off stdout) and the library's own communicator forced on (TDTK_FORCE_ALLREDUCE): one ncclAllReduce
    # per LUM iteration inside tdtk_graph_iteration, same result bit for bit, stdout = the one JSON line
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, TDTK_FORCE_ALLREDUCE="1")
    env.pop("TDTK_BENCH_BACKEND", None)
    rc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                         "--gpus", "1"] + common, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert rc.returncode == 0, rc.stderr[-2000:]
    lines = [ln for ln in rc.stdout.strip().splitlines() if ln.strip()]
    assert len(lines) == 1, rc.stdout[-2000:]
    c = json.loads(lines[0])
    assert c["last_ret"] == a["last_ret"] and "RCCL ncclAllReduce" in c["exchange"]
    assert int(c["exchange"].split(",")[-1].split()[0]) >= 4          # 1 warm-up + 3 timed + the counting step


def test_two_ranks_with_lazy_scan_moves_equal_one_rank_moving_eagerly(gpu):
    """Round-4 advice: the multi-rank behaviour of the queued scan moves -- ranks queue moves for scans none of their links
    reads, the link launch of the rank that does read a scan carries its chain out, chains grow over rounds on the others --
    had only run with one rank.  Two ranks (sharing this box's GPU, gloo carrying the exchange) over scans big enough for the
    persistent-lane launch (300K points: the launch that applies the moves itself), lazy moves on, five rounds, against ONE
    process with TDTK_LAZY_MOVES=0 (every scan moved at once): the last round's result bit for bit."""
    import socket
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    common = ["--workload", "graphslam", "--scans", "8", "--points", "300000", "--steps", "4", "--warmup", "1", "--no-rehearsal"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"] + common, cwd=root,
                         env=dict(os.environ, TDTK_LAZY_MOVES="0"), capture_output=True, text=True, timeout=900)
    assert one.returncode == 0, one.stderr[-2000:]
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"),
                          "--gpus", "2"] + common, cwd=root, env=dict(os.environ, TDTK_BENCH_BACKEND="gloo", TDTK_LAZY_MOVES="1"),
                         capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads(one.stdout.strip().splitlines()[-1])
    b = json.loads(two.stdout.strip().splitlines()[-1])
    assert b["n_gpus"] == 2 and a["config"]["links"] == b["config"]["links"] and min(b["links_per_rank"]) >= 1
    assert a["last_ret"] == b["last_ret"], (a["last_ret"], b["last_ret"])


@pytest.mark.parametrize("name", ["one", "two", "identical70", "identical64", "two_values", "axis_ties"])
def test_normals_degenerate_clouds(tdtk, orc, gpu, name):
    """Clouds on which every split of the ANN tree is a tie-break: a single point, two points, all points identical
    (one global level + wave-built cells / wave-built only), two distinct positions repeated, and integer
    coordinates with many equal values per axis (every cutting plane passes through points)."""
    rng = np.random.default_rng(12)
    pts = {"one": np.array([[1.0, 2.0, 3.0]]), "two": np.array([[0.0, 0.0, 0.0], [1.0, 0.0, 0.0]]),
           "identical70": np.tile([[5.0, -5.0, 2.5]], (70, 1)), "identical64": np.tile([[5.0, -5.0, 2.5]], (64, 1)),
           "two_values": np.repeat(np.array([[0.0, 0.0, 0.0], [3.0, 1.0, 2.0]]), 150, axis=0)[rng.permutation(300)],
           "axis_ties": rng.integers(0, 6, (5000, 3)).astype(float)}[name]
    k = min(10, len(pts))
    want, wknn = orc.normals_apx_knn(pts, k, [0.5, 0.25, -1.0], 1.0, want_knn=True)
    got, gknn = tdtk.calculateNormalsApxKNN(pts, k, [0.5, 0.25, -1.0], 1.0, want_knn=True)
    assert np.array_equal(gknn, wknn)
    assert np.array_equal(got, want, equal_nan=True)
    if orc.have_ref():
        assert np.array_equal(want, orc.normals_apx_knn(pts, k, [0.5, 0.25, -1.0], 1.0, "ref"), equal_nan=True)


def test_get_pt_pairs_empty_range_with_normals(tdtk, gpu):
    """startindex == endindex in the plane / normal-shooting modes: no query, no pair, no error (searchTree.cc:112)"""
    rng = np.random.default_rng(0)
    m = rng.uniform(-1, 1, (100, 3)); q = rng.uniform(-1, 1, (10, 3)); nr = np.tile([0.0, 0.0, 1.0], (10, 1))
    kd = tdtk.KDtree(m)
    for mode in (0, 1, 2):
        r = kd.getPtPairs(tdtk.M4identity(), q, nr, 4, 4, pairing_mode=mode)
        assert r["n"] == 0 and len(r["idx"]) == 0


@pytest.mark.parametrize("n", [1000, 1001, 2])
def test_find_closest_dev_odd_and_tiny_batches(tdtk, orc, gpu, n):
    """regression: batches whose size is not a multiple of the grouped kernel's lanes-per-query, and a 2-query batch
    (a KAT-sized call), through every host-buffer entry point that bins queries"""
    rng = np.random.default_rng(n)
    m = rng.uniform(-10, 10, (5000, 3))
    kd, T = tdtk.KDtree(m, 5), orc.Tree(m, 5)
    q = m[rng.integers(0, len(m), n)] + rng.normal(0, 0.3, (n, 3))
    gi, gd = kd.FindClosestBatch(q, 4.0)
    oi, od = T.find_closest(q, 4.0)
    assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_b1_correspondence_indices_every_iteration(tdtk, orc, gpu):
    """B1 replayed on resident scans: at every one of the 39 + 50 ICP iterations the whole-scan pass returns exactly
    the index array the REFERENCE KDtreeIndexed returned (hash per iteration, first / last 100 indices at the ends;
    tests/golden/b1_dat_icp_idx.json)."""
    z = np.load(os.path.join(G, "dat_scans.npz"))
    b1 = json.load(open(os.path.join(G, "b1_dat_icp.json")))
    bi = json.load(open(os.path.join(G, "b1_dat_icp_idx.json")))
    S = _dat_scans(tdtk.Scan, z)
    for pr, pi in zip(b1["pairs"], bi["pairs"]):
        i = pr["cur"]
        S[i].mergeCoordinatesWithRoboterPosition(S[i - 1])
        for it, a in enumerate(pr["alignxf"]):
            r = tdtk.Scan.getPtPairs(S[i - 1], S[i], max_dist_match2=625.0, want_idx=True)
            row = pi["iterations"][it]
            assert r["n"] == row["found"] and "0x%x" % orc.k5_hash(r["idx"]) == row["hash"], (i, it)
            if "first100" in row:
                assert r["idx"][:100].tolist() == row["first100"] and r["idx"][-100:].tolist() == row["last100"]
            S[i].transform(np.array(a))


def test_open_directory_with_normals(tdtk, orc, gpu, tmp_path):
    """openDirectory(..., use_normals=True): uos files on disk -> range filter -> Scan::calcNormals per scan, equal to
    the oracle on the same filtered points; refused together with -r (the reference cannot carry them either)."""
    z = np.load(os.path.join(G, "dat_scans.npz"))
    for k in range(2):
        pts, pose = z["scan%03d" % k][:5000], z["pose%03d" % k]
        with open(tmp_path / ("scan%03d.3d" % k), "w") as f:
            for p in pts:
                f.write("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))
        with open(tmp_path / ("scan%03d.pose" % k), "w") as f:
            f.write("%r %r %r\n%r %r %r\n" % (*[float(v) for v in pose[:3]], *[float(np.degrees(v)) for v in pose[3:]]))
    scans = tdtk.openDirectory(str(tmp_path), 0, 1, range_max=500.0, use_normals=True)
    assert len(scans) == 2
    for k, s in enumerate(scans):
        pts = _range_filter(z["scan%03d" % k][:5000], 500.0)
        assert np.array_equal(s._local, pts)
        assert np.array_equal(s._local_n, orc.normals_apx_knn(pts, 10, s.rPos, 1.0))
    with pytest.raises(ValueError):
        tdtk.openDirectory(str(tmp_path), 0, 1, red=10.0, use_normals=True)


def test_prepare_scans_with_normals_concurrently(tdtk, orc, gpu):
    """prepare_scans(normals=True) on 4 host threads: normals, upload and tree of several scans side by side -- the
    same normals as the oracle, the same trees as a serial preparation."""
    rng = np.random.default_rng(9)
    clouds = [rng.uniform(-50, 50, (6000, 3)) * [1.0, 1.0, 0.1] + rng.normal(0, 0.05, (6000, 3)) for _ in range(6)]
    S = [tdtk.Scan([0.5 * k, 0, 2.0], [0, 0, 0.01 * k], c) for k, c in enumerate(clouds)]
    tdtk.prepare_scans(S, trees=True, threads=4, normals=True)
    for k, s in enumerate(S):
        assert np.array_equal(s._local_n, orc.normals_apx_knn(clouds[k], 10, s.rPos, 1.0))
        assert s.getSearchTree().verify() == [0, 0, 0, 0]


def test_randomized_differential_run(gpu):
    """20 s of tools/fuzz_parity.py (random awkward clouds through normals, search + tree verification, getPtPairs,
    octree reduction, short ICP loops, scan lives, pose graphs -- each against the oracle)."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--seconds", "20", "--seed", "101"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert " 0 mismatches" in r.stdout


def test_randomized_differential_run_big_clouds(gpu):
    """20 s of tools/fuzz_parity.py --big (round 4): the same differential run on clouds of 30K .. 400K
    points -- the one-query-per-lane and persistent-lane kernels, the piecewise centroid sums and the subtree finisher of
    the tree build; the small-cloud run above only ever reaches the lane-group kernels."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--seconds", "20", "--seed", "202", "--big"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert " 0 mismatches" in r.stdout


def test_randomized_differential_run_graphs(gpu):
    """15 s of tools/fuzz_graph.py (round 4): random scan sets whose sizes mix the three search-kernel families, random
    links, a few lum6DEuler rounds -- queued scan moves against moving every scan every round bit for bit, the first
    round against the oracle's lum_iteration."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_graph.py"), "--seconds", "15", "--seed", "303"],
                       cwd=root, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert " 0 mismatches" in r.stdout


def test_reference_side_binding_executes(gpu):
    """adapters/hip_search_tree.cc compiled against the reference's own headers and linked with lib3dtk_hip.so
    (adapters/harness/build.sh, built where the checkout exists, travels in oracle/_ref/): HipSearchTree::getPtPairs
    through the SearchTree base pointer with a real DataXYZ, pairing modes 0 and 2, appending semantics, the legacy
    pointer overload, FindClosest returning the caller's pointer -- against the C ABI called directly and brute
    force (the checks are in adapters/harness/hip_search_tree_harness.cc)."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "hip_search_tree_harness")
    if not os.path.exists(exe):
        pytest.skip("harness not built (no reference checkout where this snapshot was made)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "HARNESS OK" in r.stdout, r.stdout + r.stderr


def test_icp_glue_executes(gpu):
    """adapters/icp_glue.h executed: the body of icp6D_hip::match and the C++ doICP with scans prepared ahead, instantiated
    with a minimal scan type (adapters/harness/icp_glue_harness.cc) -- equal to the same matches issued one by one
    through the C ABI (matrices bit for bit), independent of the prefetch depth (poses, frames, resident points), frame
    bookkeeping of icp6D.cc:246-268, and the no-pairs ending."""
    import subprocess
    exe = os.path.join(os.path.dirname(HERE), "adapters", "harness", "_bin", "icp_glue_harness")
    if not os.path.exists(exe):
        r = subprocess.run([os.path.join(os.path.dirname(HERE), "adapters", "harness", "build_glue.sh")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ICP GLUE HARNESS OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_slam_glue_executes(tdtk, gpu, tmp_path, variant):
    """(Round 4 -- variant 1: meta_icp with max_num_metascans = 3, slam6D.cc:436-448, and the closing -DlastSLAM pass,
    slam6D.cc:535-547; variants 2 / 3 / 4: the loop closed by hip_elch_close_loop_quat / _unitquat / _slerp, -L 2 / 3 / 4.)
    adapters/slam6d_glue.h executed (adapters/harness/slam_glue_harness.cc): matchGraph6Dautomatic in C++ on the C ABI
    -- sequential ICP with scans prepared ahead, loop detection, ELCH loop closing (batched covariance passes, balancer,
    MetaScan-against-MetaScan match), global lum6DEuler rounds through graph_slam_glue.h -- ends, on the same scans, in
    the poses of the Python mirror (matchGraph6Dautomatic + elch6Deuler + Graph + the library's graph iteration), bit
    for bit, with the same number of frames per scan and of global rounds; prefetch depth 3 == none (checked by the
    harness itself)."""
    import subprocess
    from importlib import import_module
    gs = import_module("3dtk_amd.graphslam")
    exe = os.path.join(os.path.dirname(HERE), "adapters", "harness", "_bin", "slam_glue_harness")
    if not os.path.exists(exe):
        r = subprocess.run([os.path.join(os.path.dirname(HERE), "adapters", "harness", "build_glue.sh")], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    r = subprocess.run([exe, fin, fout, "15", "30000", "3", str(variant)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "SLAM GLUE HARNESS OK" in r.stdout, r.stdout + r.stderr
    raw = np.fromfile(fin, dtype=np.uint8)
    nscans, npts = np.frombuffer(raw[:8].tobytes(), dtype=np.int32)
    body = np.frombuffer(raw[8:].tobytes(), dtype=np.float64).reshape(nscans, 6 + 3 * npts)
    out = np.fromfile(fout, dtype=np.uint8)
    tm_cpp = np.frombuffer(out[:nscans * 128].tobytes(), dtype=np.float64).reshape(nscans, 16)
    tail = np.frombuffer(out[nscans * 128:].tobytes(), dtype=np.int32)
    frames_cpp, rounds_cpp = tail[:nscans], int(tail[nscans])
    assert rounds_cpp >= 2          # the loop was detected and relaxed (else the scenario tests nothing)

    S = [tdtk.Scan(body[k, :3], body[k, 3:6], body[k, 6:].reshape(-1, 3)) for k in range(nscans)]
    tdtk.Scan.allScans = S
    try:
        class Relax:                # the -G 1 plug point served by the library's own iteration, like graph_slam_glue.h
            mdm2 = 25.0 ** 2

            def set_mdmll(self, mdmll):
                self.mdm2 = mdmll * mdmll

            def doGraphSlam6D(self, gr, allScans, nrIt):
                ret, it = float("inf"), 0
                while it < nrIt and ret > 0.5:
                    ret = gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, allScans, self.mdm2, None)
                    it += 1
                return ret
        mini = tdtk.icp6D_QUAT(True)
        icp = tdtk.icp6D(mini, 25.0, 30, quiet=True, epsilonICP=1e-5)
        closer = {2: tdtk.elch6Dquat, 3: tdtk.elch6DunitQuat, 4: tdtk.elch6Dslerp}.get(variant, tdtk.elch6Deuler)
        loop = closer(True, mini, 25.0, 30, epsilonICP=1e-5)
        extra = dict(max_num_metascans=3, mdmll=15.0, graphDist=140.0) if variant == 1 else {}
        rounds = tdtk.matchGraph6Dautomatic(90.0, 6, S, icp, variant == 1, Relax(), 3, 0.05, 25.0, eP=True, prefetch=True,
                                            my_loopSlam6D=loop, **extra)
        assert rounds == rounds_cpp
        tm_py = np.stack([s.transMat for s in S])
        assert tm_py.tobytes() == tm_cpp.tobytes(), float(np.abs(tm_py - tm_cpp).max())
        assert [len(s.frames) for s in S] == frames_cpp.tolist()
        # and the loop closing did something: the last scan ends nearer to the truth than its odometry said
        assert np.abs(tm_cpp[-1, 12:15] - body[-1, :3]).max() > 0.5
    finally:
        tdtk.Scan.allScans = []


@pytest.mark.parametrize("variant", ["elch6Deuler", "elch6Dquat", "elch6DunitQuat", "elch6Dslerp"])
def test_elch_variants_distribute_a_loop_error(tdtk, gpu, variant):
    """-L 1 .. 4 (elch6Deuler / elch6Dquat / elch6DunitQuat / elch6Dslerp::close_loop; parity UNPINNED -- their TUs need
    Boost.Graph): thirteen scans of one cloud around a closed path whose odometry drifts linearly; closing the loop
    (0, 12) over the chain graph must take the drift out of the scans in between in proportion -- the error of every
    scan from the third on drops below a third of what it was (the three matched scans at the end move as one rigid
    MetaScan and keep their relative drift), scan 0 stays, and the four
    interpolation rules agree with each other to a fraction of the drift."""
    rng = np.random.default_rng(5)
    world = rng.uniform(-400, 400, (20000, 3))
    n = 13
    truth, S = [], []
    for k in range(n):
        ang = 2 * np.pi * k / 12.0
        tP = np.array([120 * np.sin(ang), 3 * np.sin(2 * ang), 120 * (1 - np.cos(ang))])
        tT = np.array([0.01 * np.sin(ang), 0.15 * np.sin(ang), 0.008 * np.cos(ang) - 0.008])
        Ti = tdtk.M4inv(tdtk.EulerToMatrix4(tP, tT)).reshape(4, 4)
        w = world + rng.uniform(-0.2, 0.2, world.shape)
        local = w @ Ti[:3, :3] + Ti[3, :3]
        drift = 0.35 * k
        S.append(tdtk.Scan(tP + drift * np.array([1.0, -0.3, 0.6]), tT + np.array([0.0, 0.0006 * k, 0.0]), local))
        truth.append(tP)
    truth = np.array(truth)
    before = np.linalg.norm(np.array([s.get_rPos() for s in S]) - truth, axis=1)
    tdtk.Scan.allScans = S
    try:
        loop = getattr(tdtk, variant)(True, tdtk.icp6D_QUAT(True), 25.0, 40, epsilonICP=1e-6)
        loop.close_loop(S, 0, n - 1, [(i - 1, i) for i in range(1, n)])
        pos = np.array([s.get_rPos() for s in S])
    finally:
        tdtk.Scan.allScans = []
    after = np.linalg.norm(pos - truth, axis=1)
    assert after[0] == before[0] == 0.0 or after[0] < 1e-9
    assert (after[3:] < before[3:] / 3.0).all(), (before, after)
    ref = getattr(test_elch_variants_distribute_a_loop_error, "_first", None)
    if ref is None:
        test_elch_variants_distribute_a_loop_error._first = pos
    else:
        assert np.abs(pos - ref).max() < 1.0, np.abs(pos - ref).max()


def test_speculative_tree_build_and_its_fallback(gpu):
    """build.hip, speculative splits: the levels are cut at the plain parallel sums of the big nodes while the exact
    serial sums run beside them, and every big node is checked against its exact sum at the end.  On ordinary clouds (a
    bundled scan with its duplicate points, uniform, a dense clump) no build needs redoing and the tree is the host
    builder's; with the cut forced elsewhere (TDTK_BUILD_SPEC_FAULT=1) the check fails on every build, the in-order path
    takes over, and the tree is still the host builder's."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    for fault, want in (("0", "0"), ("1", "+")):
        env = dict(os.environ, TDTK_BUILD_SPEC_FAULT=fault, TDTK_LIB="lab")      # (the fault switch is a lab switch)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "fault_probe.py")], cwd=root, env=env, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        rows = [ln.split() for ln in r.stdout.splitlines() if "respeculated" in ln]
        assert len(rows) == 4, r.stdout
        for k, row in enumerate(rows):
            assert "".join(row[:4]) == "[0,0,0,0]", r.stdout
            n_redone = int(row[5])
            assert n_redone == (0 if want == "0" else k + 1), r.stdout


def test_handle_pool_reuses_and_releases(tdtk, orc, gpu):
    """pool.cpp: the device arrays of a destroyed tree / scan are kept and handed to the next handle; a tree built in
    reused blocks is the same tree (device == host build, same answers), tdtk_pool_trim gives the kept bytes back and a
    second trim has nothing left."""
    capi = __import__("importlib").import_module("3dtk_amd._capi")
    rng = np.random.default_rng(17)
    m = rng.uniform(-300, 300, (150000, 3))
    q = m[:5000] + rng.normal(0, 1.0, (5000, 3))
    capi.pool_trim()
    kd = tdtk.KDtree(m, 20)
    idx0, d0 = kd.FindClosestBatch(q, 100.0)
    del kd
    released = capi.pool_trim()
    assert released > 150000 * 32          # at least the point array came back
    assert capi.pool_trim() == 0
    kd = tdtk.KDtree(m, 20); del kd         # fills the shelves again ...
    kd = tdtk.KDtree(m[::-1].copy(), 20)    # ... and this one is built inside the kept blocks
    assert kd.verify() == [0, 0, 0, 0]
    idx1, d1 = kd.FindClosestBatch(q, 100.0)
    assert np.array_equal(np.where(idx0 >= 0, len(m) - 1 - idx1, -1), idx0) and np.array_equal(d0, d1)


def test_lab_library_default_path_is_the_products(tdtk, gpu):
    """The lab library (tests of measured-and-lost variants run inside it) takes, without any of its switches set, the
    product's path: indices, squared distances, pair sums and an ICP loop's poses are bit-identical between
    lib3dtk_hip.so and lib3dtk_hip_lab.so -- on a batch for each kernel family (four lanes per query, one query per
    lane, persistent lanes with the sums inside the launch)."""
    from importlib import import_module
    capi = import_module("3dtk_amd._capi")
    rng = np.random.default_rng(123)
    for npts in (30000, 150000, 300000):
        m = rng.uniform(-300, 300, (npts, 3)); m[100:300] = m[0:200]
        d = m[rng.permutation(npts)] + rng.normal(0, 0.5, m.shape) + np.array([2.0, -1.5, 1.0])
        out = []
        for name in ("product", "lab"):
            with capi.library(name):
                S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], m); S1 = tdtk.Scan([0, 0, 0], [0, 0, 0], d)
                r = tdtk.Scan.getPtPairs(S0, S1, 0, 0, 100.0, 0, 0, want_idx=True)
                icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 25.0, 6, quiet=True, epsilonICP=-1.0)
                it = icp.match(S0, S1)
                out.append((r["idx"].copy(), r["n"], r["sum"], np.asarray(r["Si"]).copy(), it, icp.last["trace"].copy(), S1.get_transMat().copy()))
                S0.release(); S1.release()
        a, b = out
        assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3]), npts
        assert a[4] == b[4] and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6]), npts


def test_alternative_search_kernels_agree(tdtk, orc, gpu, lab, monkeypatch):
    """The kernels kept beside the default as measured alternatives (fused retire-time sums, the work-queue kernel,
    other slab lengths / refill thresholds, static slab + per-XCD pool, 256-thread persistent lanes, one query per lane,
    two queries per lane, upper levels in LDS, slabs shared by a workgroup's waves) walk the same tree the
    same way: identical indices, pair sums equal to rounding, and the instrumented instantiations count the same
    visits as the oracle -- on a batch large enough for the persistent-lane kernels, with duplicates in the model."""
    rng = np.random.default_rng(99)
    m = rng.uniform(-500, 500, (400000, 3)); m[1000:3000] = m[0:2000]
    q = m[rng.integers(0, len(m), 600000)] + rng.normal(0, 2.0, (600000, 3))
    S0 = tdtk.Scan([0, 0, 0], [0, 0, 0], m); S1 = tdtk.Scan([0, 0, 0], [0, 0, 0], q)
    T = orc.Tree(m, 20)
    oi, od2, cnt = T.find_closest(q, 100.0, 8, want_counters=True)
    base = tdtk.Scan.getPtPairs(S0, S1, 0, 0, 100.0, 0, 0, want_idx=True)
    assert np.array_equal(base["idx"], oi)
    import ctypes as C
    L = tdtk.lib()
    for env in ({"TDTK_FUSE_SUMS": "1"}, {"TDTK_FUSE_SUMS": "3"}, {"TDTK_FUSE_SUMS": "3", "TDTK_REFILL_PHASES": "1"}, {"TDTK_SEARCH_VARIANT": "30"}, {"TDTK_SEARCH_VARIANT": "30", "TDTK_STREAM_SLAB": "64", "TDTK_STREAM_WPS": "7"},
                {"TDTK_SEARCH_VARIANT": "8"}, {"TDTK_SEARCH_VARIANT": "4"}, {"TDTK_SEARCH_VARIANT": "0"},
                {"TDTK_SEARCH_VARIANT": "4", "TDTK_FUSE_SUMS": "0"}, {"TDTK_SEARCH_VARIANT": "10"}, {"TDTK_SEARCH_VARIANT": "10", "TDTK_FUSE_SUMS": "0"},
                {"TDTK_SEARCH_VARIANT": "40"}, {"TDTK_SEARCH_VARIANT": "41"},
                {"TDTK_REFILL_QPW": "128", "TDTK_REFILL_THRESH": "8"}, {"TDTK_REFILL_QPW": "512", "TDTK_REFILL_THRESH": "32"},
                {"TDTK_REFILL_POOL": "25"}, {"TDTK_REFILL_POOL": "60", "TDTK_REFILL_POOL_SLAB": "48"}, {"TDTK_REFILL_PHASES": "1"},
                {"TDTK_TWO_PER_LANE": "3"}, {"TDTK_TWO_PER_LANE": "2", "TDTK_FUSE_SUMS": "3"},
                # round 4: the tree's upper levels staged in LDS by workgroups of 512 / 1024 threads; the slabs of a workgroup's
                # waves handed out together from one cursor in LDS
                {"TDTK_TOP_BLOCK": "1024"}, {"TDTK_TOP_BLOCK": "512"},
                {"TDTK_SHARE_BLOCK": "256"}, {"TDTK_SHARE_BLOCK": "512"}, {"TDTK_SHARE_BLOCK": "1024"},
                # ... a hand-out that does not wait for its loads; two queries per lane with one visit per trip
                {"TDTK_PIPE": "1"}, {"TDTK_TWO_PER_LANE": "3", "TDTK_TWO_ONE": "1"}, {"TDTK_SINGLE_BLOCK": "64"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for counting in (0, 1):
            L.tdtk_visit_counting(0, counting)
            r = tdtk.Scan.getPtPairs(S0, S1, 0, 0, 100.0, 0, 0, want_idx=True)
            assert np.array_equal(r["idx"], oi), env
            assert r["n"] == base["n"] and abs(r["sum"] - base["sum"]) <= 1e-11 * base["sum"], env
            assert np.abs(np.asarray(r["Si"]) - np.asarray(base["Si"])).max() <= 1e-9 * np.abs(base["Si"]).max(), env
            if counting:
                c = (C.c_uint64 * 8)()
                L.tdtk_visit_counters(0, c)
                assert (c[0], c[1], c[2], c[3]) == (cnt[0], cnt[1], cnt[2], len(q)), (env, list(c), cnt)
        L.tdtk_visit_counting(0, 0)
        for k in env:
            monkeypatch.delenv(k)


def test_rccl_exchange_inside_the_library_one_rank(tdtk, orc, gpu, monkeypatch):
    """The library's own communicator (tdtk_comm_*: ncclGetUniqueId / ncclCommInitRank / ncclAllReduce bound at run
    time) with a 1-rank world: the all-reduce of a block list returns it unchanged, and a graph-SLAM iteration of
    each back-end that goes through tdtk_graph_iteration WITH the forced exchange gives bit for bit the poses of the
    one without a communicator -- the path `bench.py --gpus N` takes, with no Python between link blocks and solve."""
    import ctypes as C
    gs = __import__("importlib").import_module("3dtk_amd.graphslam")
    comm = gs.NativeComm(0, 1, 0)
    rng = np.random.default_rng(8)
    blocks = rng.normal(size=84 * 42)
    got = blocks.copy()
    assert tdtk.lib().tdtk_graph_exchange(comm._h, got.ctypes.data_as(C.POINTER(C.c_double)), got.size) == 0
    assert np.array_equal(got, blocks) and comm.n_allreduce() == 1

    def scans():
        r = np.random.default_rng(5)
        world = r.uniform(-300, 300, (120000, 3))
        out = []
        for k in range(6):
            ang = 2 * np.pi * k / 6
            pos = np.array([60 * np.cos(ang), 0.0, 60 * np.sin(ang)]) + r.normal(0, 0.3, 3)
            th = np.array([0.0, -ang + r.normal(0, 0.002), 0.0])
            T = tdtk.EulerToMatrix4([60 * np.cos(ang), 0.0, 60 * np.sin(ang)], [0.0, -ang, 0.0])
            Ti = tdtk.M4inv(T)
            R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
            sel = r.choice(len(world), 30000, replace=False)
            out.append(tdtk.Scan(pos, th, world[sel] @ R.T + Ti[12:15] + r.normal(0, 0.05, (30000, 3))))
        return out
    for backend in (1, 2, 3, 4):
        res = []
        for forced in (False, True):
            S = scans()
            gr = tdtk.Graph(6, links=[(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]) if backend != 4 else \
                tdtk.Graph(6, links=[(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)])
            state = gs.graph_state(backend, 6)
            if forced:
                monkeypatch.setenv("TDTK_FORCE_ALLREDUCE", "1")
            before = comm.n_allreduce()
            rets = [gs.graph_iteration_comm(backend, gr, S, 400.0, comm if forced else None, state) for _ in range(2)]
            if forced:
                assert comm.n_allreduce() == before + 2
                monkeypatch.delenv("TDTK_FORCE_ALLREDUCE")
            res.append((rets, [s.transMat.copy() for s in S], [s.get_xyz_reduced()[:100].copy() for s in S]))
        assert res[0][0] == res[1][0]
        for a, b in zip(res[0][1], res[1][1]):
            assert np.array_equal(a, b)
        for a, b in zip(res[0][2], res[1][2]):
            assert np.array_equal(a, b)
    comm.close()


def test_failure_paths_of_the_exchange(tdtk, gpu, lab, monkeypatch):
    """Round 6: the two ways a rank's failure reaches the others (comm.cpp) had never executed anywhere.
    On the one GPU of this box (RCCL refuses two ranks on one device, so the peers' side -- ncclAllReduce returning an
    error after ncclCommAbort, TDTK_EPEER from the status slot -- stays unexecuted) a 1-rank communicator with the exchange
    forced runs this rank's side of both, the failures injected by a lab switch:
      * TDTK_COMM_FAIL=links: the rank's link passes fail; it still enters the one collective of the iteration with zeros
        for its blocks and the status slot raised, and returns ITS error (not a hang, not a solve over zeros): the poses
        and the resident points are untouched, the communicator stays usable and the next iteration (switch off) equals
        the iteration of a run that never failed;
      * TDTK_COMM_FAIL=reserve: the rank cannot get staging memory -- nothing can be reported through the collective --:
        ncclCommAbort, TDTK_ENOMEM, and every later call on that communicator returns TDTK_EDEVICE at once."""
    import bench
    gs = __import__("importlib").import_module("3dtk_amd.graphslam")
    capi = __import__("importlib").import_module("3dtk_amd._capi")
    raw = bench.make_graphslam_scans(6, 20000, seed=9)
    links = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5)]
    monkeypatch.setenv("TDTK_FORCE_ALLREDUCE", "1")

    def fresh():
        return [tdtk.Scan(p, th, loc) for (p, th, loc) in raw]
    # the run that never fails
    comm = gs.NativeComm(0, 1, 0)
    S = fresh()
    want = gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(6, links=links), S, 625.0, comm, None)
    want_T = [s.transMat.copy() for s in S]
    for s in S: s.release()
    # link passes fail: reported through the exchange, nothing moves, the communicator lives
    S = fresh()
    before_T = [s.transMat.copy() for s in S]
    before_x = [s.get_xyz_reduced()[:50].copy() for s in S]
    n0 = comm.n_allreduce()
    monkeypatch.setenv("TDTK_COMM_FAIL", "links")
    with pytest.raises(capi.TdtkError) as e:
        gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(6, links=links), S, 625.0, comm, None)
    assert "reported to the other ranks through the exchange" in str(e.value) and "TDTK_COMM_FAIL=links" in str(e.value)
    assert comm.n_allreduce() == n0 + 1                                  # it DID take part in the collective
    assert all(np.array_equal(a, s.transMat) for a, s in zip(before_T, S))
    assert all(np.array_equal(a, s.get_xyz_reduced()[:50]) for a, s in zip(before_x, S))
    monkeypatch.delenv("TDTK_COMM_FAIL")
    got = gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(6, links=links), S, 625.0, comm, None)
    assert got == want and all(np.array_equal(a, s.transMat) for a, s in zip(want_T, S))
    # no staging memory: abort, and the communicator is dead for good
    monkeypatch.setenv("TDTK_COMM_FAIL", "reserve")
    with pytest.raises(capi.TdtkError) as e:
        gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(6, links=links), S, 625.0, comm, None)
    assert "communicator aborted after a local failure" in str(e.value)
    monkeypatch.delenv("TDTK_COMM_FAIL")
    with pytest.raises(capi.TdtkError) as e:
        gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(6, links=links), S, 625.0, comm, None)
    assert "aborted after an earlier failure" in str(e.value)
    blocks = np.zeros(42)
    assert tdtk.lib().tdtk_graph_exchange(comm._h, capi.dptr(blocks), blocks.size) == -2      # TDTK_EDEVICE
    comm.close()
    for s in S: s.release()
    # a new communicator works again
    comm = gs.NativeComm(0, 1, 0)
    S = fresh()
    assert gs.graph_iteration_comm(gs.GRAPH_LUMEULER, tdtk.Graph(6, links=links), S, 625.0, comm, None) == want
    comm.close()
    for s in S: s.release()


def _loop_scans(tdtk, io, nscans=12, npts=25000, seed=17, drift=0.25):
    rng = np.random.default_rng(seed)
    world = np.concatenate([rng.uniform(-260, 260, (150000, 3)) * np.array([1.0, 0.15, 1.0]),
                            rng.normal(0, 1.0, (60000, 3)) + rng.uniform(-200, 200, (60, 3)).repeat(1000, axis=0)])
    S, O = [], []
    dp = np.zeros(3)
    for k in range(nscans):
        ang = 2 * np.pi * k / nscans
        pos = np.array([90 * np.cos(ang), 0.0, 90 * np.sin(ang)])
        th = np.array([0.0, -ang, 0.0])
        T = tdtk.EulerToMatrix4(pos, th)
        Ti = tdtk.M4inv(T)
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        sel = rng.choice(len(world), npts, replace=False)
        loc = world[sel] @ R.T + Ti[12:15] + rng.normal(0, 0.05, (npts, 3))
        if k > 0:
            dp = dp + rng.normal(0.0, drift, 3) + np.array([0.15, 0.0, 0.1])      # a systematic odometry error: the loop does not close
        p0 = pos + dp
        S.append(tdtk.Scan(p0, th, loc)); O.append(io.OScan(p0, th, loc))
    return S, O


def test_elch_close_loop_vs_oracle(tdtk, orc, gpu):
    """-L 1 (elch6Deuler::close_loop, elch6Deuler.cc:44-138): per-edge covariances -> 6 weighted graphs -> graph
    balancer -> MetaScan(first..first+2) matched against MetaScan(last-2..last) -> the error distributed over the loop.
    Against the oracle restatement (parity unpinned: the reference TUs need Boost.Graph): edge weights, balancer output,
    the meta-vs-meta match trace, delta and every pose."""
    from oracle import icp_oracle as io
    S, O = _loop_scans(tdtk, io)
    n = len(S)
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 5.0, 40, quiet=True, epsilonICP=1e-6)
    icp.doICP(S, prefetch=False)
    io.do_icp(O, 1, 25.0, 40, 1e-6)
    for s, o in zip(S, O):
        assert np.abs(s.transMat - o.transMat).max() < 1e-8
    g = [(i - 1, i) for i in range(1, n)]
    gap_before = np.linalg.norm(S[n - 1].get_rPos() - O[n - 1].rPos)
    assert gap_before < 1e-7
    loop = tdtk.elch6Deuler(True, tdtk.icp6D_QUAT(True), 5.0, 40, epsilonICP=1e-6)
    loop.close_loop(S, 0, n - 1, g)
    delta, weights = io.elch_close_loop(O, 0, n - 1, g, 1, 25.0, 40, 1e-6)
    np.testing.assert_allclose(loop.last_delta, delta, rtol=1e-6, atol=1e-9)
    assert np.abs(delta[:3]).max() > 0.05                       # there was a loop error to distribute
    for s, o in zip(S, O):
        assert np.abs(s.transMat - o.transMat).max() < 1e-6 * max(1.0, np.abs(o.transMat).max())
        assert np.abs(s.get_xyz_reduced() - o.xyz).max() < 1e-6
    # the loop is closed better than before: scan n-1 against scan 0
    err, npairs = icp.Point_Point_Error(S[0], S[n - 1], 5.0)
    assert npairs > 1000


@pytest.mark.parametrize("variant", ["elch6Dquat", "elch6DunitQuat", "elch6Dslerp"])
def test_elch_quaternion_variants_vs_oracle(tdtk, orc, gpu, variant):
    """-L 2 / 3 / 4 (elch6Dquat.cc:44-148, elch6DunitQuat.cc:45-199, elch6Dslerp.cc:44-184) against the oracle's
    restatement (round 4; parity UNPINNED -- the TUs need Boost.Graph): the per-edge covarianceQuat passes run in one
    batched device call, the balancer in the library, the interpolation rules on the host; every pose after the loop
    closing, the loop error found by the MetaScan match and the balancer's weights agree with the oracle."""
    from oracle import icp_oracle as io
    S, O = _loop_scans(tdtk, io)
    n = len(S)
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 5.0, 40, quiet=True, epsilonICP=1e-6)
    icp.doICP(S, prefetch=False)
    io.do_icp(O, 1, 25.0, 40, 1e-6)
    g = [(i - 1, i) for i in range(1, n)]
    first, last = (2, n - 1) if variant == "elch6Dslerp" else (0, n - 1)      # (-L 4 reaches two scans in front of `first`)
    loop = getattr(tdtk, variant)(True, tdtk.icp6D_QUAT(True), 5.0, 40, epsilonICP=1e-6)
    before = np.stack([s.transMat.copy() for s in S])
    loop.close_loop(S, first, last, g)
    out = getattr(io, {"elch6Dquat": "elch_close_loop_quat", "elch6DunitQuat": "elch_close_loop_unitquat",
                       "elch6Dslerp": "elch_close_loop_slerp"}[variant])(O, first, last, g, 1, 25.0, 40, 1e-6)
    want_delta = np.concatenate([np.ravel(v) for v in out[:-1]])
    assert np.abs(loop.last_delta - want_delta).max() < 1e-7 * max(1.0, np.abs(want_delta).max())
    moved = 0.0
    for k, (s, o) in enumerate(zip(S, O)):
        assert np.abs(s.transMat - o.transMat).max() < 1e-7 * max(1.0, np.abs(o.transMat).max()), k
        moved = max(moved, np.abs(s.transMat - before[k]).max())
    assert moved > 0.05                                        # the loop error was there and has been distributed
    # the resident points followed their poses
    for k in (1, n // 2, n - 1):
        assert np.abs(S[k].get_xyz_reduced() - O[k].xyz).max() < 1e-6


def test_match_graph6d_automatic_with_elch(tdtk, orc, gpu):
    """matchGraph6Dautomatic with -L 1 (slam6D.cc:387-548): sequential ICP, loop detection by pose distance with the
    closest (first, last) pair, ELCH loop closing, then the global lum6DEuler rounds -- against the oracle's loop."""
    from oracle import icp_oracle as io
    S, O = _loop_scans(tdtk, io, nscans=10, npts=20000, seed=23)
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 5.0, 30, quiet=True, epsilonICP=1e-6)
    loop = tdtk.elch6Deuler(True, tdtk.icp6D_QUAT(True), 5.0, 30, epsilonICP=1e-6)
    lum = tdtk.lum6DEuler(icp, 5.0, 5.0, epsilonLUM=0.5)
    rounds = tdtk.matchGraph6Dautomatic(60.0, 5, S, icp, False, lum, 3, 0.05, 5.0, eP=True, prefetch=False, my_loopSlam6D=loop)
    orounds, closed = io.match_graph6d_automatic(60.0, 5, O, 1, 25.0, 30, 1e-6, 3, 0.05, 25.0, True, elch=True)
    assert rounds == orounds and len(closed) >= 1
    for s, o in zip(S, O):
        assert np.abs(s.transMat - o.transMat).max() < 1e-5 * max(1.0, np.abs(o.transMat).max())
