"""CPU tier: the C-ABI library loads and exports every symbol include/tdtk_hip.h declares, the
host-side logic (tree construction order, 4x4 helpers, minimizer solves, SPD solve, graph
construction, link sharding + all-reduce) is right -- and the compute entry points fail loudly
when there is no GPU (no CPU fallback)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
G = os.path.join(HERE, "golden")


def test_library_exports_every_declared_symbol(tdtk):
    hdr = open(os.path.join(ROOT, "include", "tdtk_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tdtk_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    L = tdtk.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    capi = sys.modules["3dtk_amd._capi"]
    assert declared == set(capi.EXPORTS)
    assert "gfx950" in tdtk.version()


def test_struct_layout_matches_header(tdtk):
    capi = sys.modules["3dtk_amd._capi"]
    # 2 u64 + (1+3+3+9+6+3+21+6+1+15+1) + (4*9+2*3) doubles
    assert C.sizeof(capi.PairSums) == 16 + 8 * (69 + 42 + 12 + 1)
    assert C.sizeof(capi.IcpParams) == 40 and C.sizeof(capi.IcpResult) == 48


def _clouds():
    rng = np.random.default_rng(3)
    uni = rng.uniform(-100, 100, (30000, 3))
    dup = uni.copy(); dup[1000:1400] = dup[0:400]
    clu = np.concatenate([rng.normal(c, 0.003, (300, 3)) for c in rng.uniform(-50, 50, (40, 3))])
    plane = rng.uniform(-100, 100, (20000, 3)); plane[:, 2] = 0.0
    tiny = rng.uniform(-1, 1, (7, 3))
    one = np.array([[1.0, 2.0, 3.0]])
    return {"uniform": uni, "duplicates": dup, "clusters": clu, "plane": plane, "tiny": tiny, "one": one}


@pytest.mark.parametrize("name", ["uniform", "duplicates", "clusters", "plane", "tiny", "one"])
@pytest.mark.parametrize("bucket", [1, 5, 20])
def test_host_tree_layout_equals_oracle(tdtk, orc, name, bucket):
    """Same tree as KDTreeImpl::create: identical leaf-order permutation and node counts."""
    m = _clouds()[name]
    perm, st = tdtk.host_tree_layout(m, bucket)
    T = orc.Tree(m, bucket)
    assert np.array_equal(perm, T.perm())
    o = T.stats()
    assert (st["internal"], st["leaves"], st["depth"]) == (o["internal"], o["leaves"], o["depth"])


def test_host_tree_errors(tdtk):
    with pytest.raises(tdtk.TdtkError):
        tdtk.host_tree_layout(np.zeros((0, 3)), 20)          # "cannot create kdtree with zero points"
    with pytest.raises(tdtk.TdtkError):
        tdtk.host_tree_layout(np.zeros((5, 3)), 0)
    # non-finite coordinates: an error, not an endless partition scan
    for v in (np.nan, np.inf, -np.inf):
        bad = np.random.default_rng(0).uniform(-1, 1, (200, 3)); bad[17, 1] = v
        with pytest.raises(tdtk.TdtkError):
            tdtk.host_tree_layout(bad, 5)


def test_m4inv_mmult_bit_exact(tdtk, orc):
    rng = np.random.default_rng(0)
    for _ in range(200):
        A = tdtk.EulerToMatrix4(rng.uniform(-500, 500, 3), rng.uniform(-3, 3, 3))
        B = tdtk.EulerToMatrix4(rng.uniform(-500, 500, 3), rng.uniform(-3, 3, 3))
        assert np.array_equal(tdtk.M4inv(A), orc.m4inv(A)[0])
        assert np.array_equal(tdtk.MMult(A, B), orc.mmult(A, B))
    assert np.array_equal(tdtk.M4inv(np.zeros(16)), np.eye(4).reshape(16))   # singular -> identity


def _sums_from_pairs(capi, p1, p2, pn=None):
    n = len(p1)
    s = capi.PairSums()
    s.n = n; s.n_queries = n
    s.sum = float(((p1 - p2) ** 2).sum())
    cm, cd = p1.mean(0), p2.mean(0)
    for k in range(3):
        s.centroid_m[k] = cm[k]; s.centroid_d[k] = cd[k]
    Si = (p1 - cm).T @ (p2 - cd)
    for k in range(9):
        s.Si[k] = Si.reshape(9)[k]
    MMc, DDc = (p1 - cm).T @ (p1 - cm), (p2 - cd).T @ (p2 - cd)
    q = 0
    for a in range(3):
        for b in range(a, 3):
            s.mom_mm[q] = MMc[a, b]; s.mom_dd[q] = DDc[a, b]; q += 1
    p12, p2c = p1 - p2, p2 - cd
    A = [(p2c[:, 1] ** 2 + p2c[:, 2] ** 2).sum(), -(p2c[:, 0] * p2c[:, 1]).sum(), -(p2c[:, 0] * p2c[:, 2]).sum(),
         (p2c[:, 0] ** 2 + p2c[:, 2] ** 2).sum(), -(p2c[:, 1] * p2c[:, 2]).sum(), (p2c[:, 0] ** 2 + p2c[:, 1] ** 2).sum()]
    B = [(p12[:, 2] * p2c[:, 1] - p12[:, 1] * p2c[:, 2]).sum(), (p12[:, 0] * p2c[:, 2] - p12[:, 2] * p2c[:, 0]).sum(),
         (p12[:, 1] * p2c[:, 0] - p12[:, 0] * p2c[:, 1]).sum()]
    for k in range(6):
        s.apx_A[k] = A[k]
    for k in range(3):
        s.apx_B[k] = B[k]
    if pn is not None:
        v = np.hstack([np.cross(p2c, pn), pn])
        AA = v.T @ v
        q = 0
        for r in range(6):
            for c in range(r, 6):
                s.napx_A[q] = AA[r, c]; q += 1
        for r in range(6):
            s.napx_B[r] = v[:, r].sum()
        s.napx_sum = float((((p1 - p2) * pn).sum(1) ** 2).sum())
    return s


def test_align_against_reference_fixture(tdtk, orc):
    """tdtk_align (Jacobi / Cholesky solves of the product) vs the reference's Align outputs."""
    from oracle import icp_oracle as io
    capi = sys.modules["3dtk_amd._capi"]
    k6 = json.load(open(os.path.join(G, "k6_minimizers.json")))
    d = orc.gen_mt64_uniform(k6["seed_points"], 3000, -100, 100).reshape(1000, 3)
    T = io.euler_to_matrix4(k6["rPos"], k6["rPosTheta"])
    mm = d.copy(); orc.transform_points(T, mm)
    nr = orc.gen_mt64_uniform(k6["seed_normals"], 3000, -1, 1).reshape(1000, 3)
    nr /= np.linalg.norm(nr, axis=1)[:, None]
    noisy = mm + orc.gen_mt64_uniform(k6["noisy"]["seed_noise"], 3000, -0.5, 0.5).reshape(1000, 3)
    for pm, exp in ((mm, k6["algos"]), (noisy, k6["noisy"]["algos"])):
        s = _sums_from_pairs(capi, pm, d, nr)
        for algo, cls in ((1, tdtk.icp6D_QUAT), (2, tdtk.icp6D_SVD), (6, tdtk.icp6D_APX), (10, tdtk.icp6D_NAPX)):
            rms, a = cls(True).Align_Parallel(s)
            np.testing.assert_allclose(a, exp[str(algo)]["alignxf"], rtol=0, atol=5e-9, err_msg=str(algo))
            assert abs(rms - exp[str(algo)]["rms"]) <= 1e-9 * max(1.0, abs(rms))


def test_serial_only_minimizers_against_reference_fixture(tdtk, orc):
    """ORTHO / DUAL / HELIX / LUMEULER / LUMQUAT / QUAT_SCALE computed from the second-moment block
    (tdtk_align, TDTK_WANT_MOM2) vs the reference's serial Align on explicit pair lists."""
    from oracle import icp_oracle as io
    capi = sys.modules["3dtk_amd._capi"]
    k = json.load(open(os.path.join(G, "k6_serial_minimizers.json")))
    d = orc.gen_mt64_uniform(k["seed_points"], 3000, -100, 100).reshape(1000, 3)
    T = io.euler_to_matrix4(k["rPos"], k["rPosTheta"])
    mm = d.copy(); orc.transform_points(T, mm)
    noise = orc.gen_mt64_uniform(k["seed_noise"], 3000, -0.5, 0.5).reshape(1000, 3)
    clouds = {"clean": mm, "noisy": mm + noise, "scaled": k["scale_case"] * mm + noise}
    pose = np.array(k["pose"])
    classes = {3: tdtk.icp6D_ORTHO, 4: tdtk.icp6D_DUAL, 5: tdtk.icp6D_HELIX, 7: tdtk.icp6D_LUMEULER,
               8: tdtk.icp6D_LUMQUAT, 9: tdtk.icp6D_QUAT_SCALE}
    for tag, pm in clouds.items():
        s = _sums_from_pairs(capi, pm, d)
        for algo, cls in classes.items():
            exp = k["cases"][tag][str(algo)]
            rms, a = cls(True).Align_Parallel(s, pose)
            np.testing.assert_allclose(a, exp["alignxf"], rtol=0, atol=2e-8, err_msg="%s %d" % (tag, algo))
            assert abs(rms - exp["rms"]) <= 1e-9 * max(1.0, abs(rms))
    # far from the origin (coordinates ~1e5): the moment algebra keeps its digits
    off = np.array([1.2e5, -0.7e5, 3.0e4])
    s = _sums_from_pairs(capi, clouds["noisy"] + off, d + off)
    for algo, cls in classes.items():
        want_rms, want = io.align(algo, clouds["noisy"] + off, d + off, (clouds["noisy"] + off).mean(0),
                                  (d + off).mean(0), None, pose)
        rms, a = cls(True).Align_Parallel(s, pose)
        assert np.abs(a - want).max() <= 1e-6 * max(1.0, np.abs(want).max()), algo


def test_align_degenerate_cases(tdtk):
    capi = sys.modules["3dtk_amd._capi"]
    rng = np.random.default_rng(1)
    # coplanar + reflection-prone input: SVD must still return a proper rotation (icp6Dsvd.cc:103-116)
    p2 = rng.uniform(-10, 10, (50, 3)); p2[:, 2] = 0
    p1 = p2 * np.array([1, 1, -1]) + rng.normal(0, 1e-3, p2.shape)
    s = _sums_from_pairs(capi, p1, p2)
    _, a = tdtk.icp6D_SVD(True).Align_Parallel(s)
    R = np.array([[a[0], a[4], a[8]], [a[1], a[5], a[9]], [a[2], a[6], a[10]]])
    assert abs(np.linalg.det(R) - 1) < 1e-9
    # APX with all data points identical -> Cholesky fails -> -1.0 ("Couldn't find transform.")
    p2 = np.tile([[1.0, 2.0, 3.0]], (10, 1)); p1 = p2 + 0.1
    rms, _ = tdtk.icp6D_APX(True).Align_Parallel(_sums_from_pairs(capi, p1, p2))
    assert rms == -1.0


def test_solve_spd(tdtk):
    from importlib import import_module
    sl = import_module("3dtk_amd.slam6d")
    rng = np.random.default_rng(2)
    A = rng.normal(size=(60, 60)); Gm = A @ A.T + 60 * np.eye(60)
    Gm[np.abs(Gm) < 0.5] = 1e-6                                  # below the 1e-5 filter -> dropped
    Gm = (Gm + Gm.T) / 2
    B = rng.normal(size=60)
    x = sl.solveSparseCholesky(Gm, B)
    Gf = np.where(np.abs(Gm) > 1e-5, Gm, 0.0)
    np.testing.assert_allclose(x, np.linalg.solve(Gf, B), rtol=1e-9, atol=1e-12)
    with pytest.raises(tdtk.TdtkError):
        sl.solveSparseCholesky(-np.eye(4), np.ones(4))


def test_no_cpu_fallback(tdtk):
    """Without a GPU every compute entry point must fail with TDTK_EDEVICE, not compute."""
    if tdtk.device_count() > 0:
        pytest.skip("a GPU is present")
    pts = np.random.default_rng(0).uniform(-1, 1, (100, 3))
    with pytest.raises(tdtk.TdtkError) as e:
        tdtk.KDtree(pts)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(tdtk.TdtkError) as e:
        _ = tdtk.Scan([0, 0, 0], [0, 0, 0], pts).handle      # residency is lazy: first use uploads
    assert e.value.code == -2
    # the housekeeping entry points need no device: nothing is kept, nothing was rebuilt
    capi = __import__("importlib").import_module("3dtk_amd._capi")
    assert capi.pool_trim() == 0 and capi.build_respeculated() == 0


def test_euler_roundtrip_and_graph(tdtk):
    from oracle import icp_oracle as io
    rng = np.random.default_rng(4)
    for _ in range(50):
        p, th = rng.uniform(-100, 100, 3), rng.uniform(-1.5, 1.5, 3)
        A = tdtk.EulerToMatrix4(p, th)
        assert np.array_equal(A, io.euler_to_matrix4(p, th))
        th2, p2 = tdtk.Matrix4ToEuler(A)
        np.testing.assert_allclose(th2, th, atol=1e-12); np.testing.assert_allclose(p2, p)

    class P:
        def __init__(self, p): self.p = np.array(p, float)
        def get_rPos(self): return self.p
    scans = [P([800 * np.cos(a), 0, 800 * np.sin(a)]) for a in np.linspace(0, 2 * np.pi, 16, endpoint=False)]
    g = tdtk.Graph(16, 500.0 ** 2, 5, scans)
    links = list(zip(g.frm, g.to))
    assert links == io.graph_links(scans, 500.0 ** 2, 5)
    assert links[:15] == [(i, i + 1) for i in range(15)] and (0, 15) in links


_GLOO_WORKER = textwrap.dedent("""
    import importlib, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    gs = importlib.import_module("3dtk_amd.graphslam")
    from oracle import icp_oracle as io
    rng = np.random.default_rng(0)
    world = rng.uniform(-60, 60, (6000, 3))
    poses = [([4.0 * k, 0.5 * k, -0.3 * k], [0.01 * k, -0.02 * k, 0.015 * k]) for k in range(5)]
    scans = []
    for k, (p, th) in enumerate(poses):
        T = io.euler_to_matrix4(p, th)
        inv, _ = io.orc.m4inv(T)
        loc = world.copy(); io.orc.transform_points(inv, loc)
        drift = ([p[0] + 0.3 * k, p[1] - 0.2 * k, p[2] + 0.1 * k], [th[0], th[1] + 0.002 * k, th[2]])
        scans.append(io.OScan(drift[0], drift[1], loc + rng.normal(0, 0.02, loc.shape)))
    links = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (0, 3), (1, 4)]
    class Gr:
        def getNrScans(self): return 5
        def getNrLinks(self): return len(links)
        def getLink(self, i, ft): return links[i][ft]
    link_fn = lambda a, b, md2: io.covariance_euler(a, b, md2)[:2]
    # the exchange step of the native path: per-link blocks, one all-reduce, C-side scatter + solve
    rank, wsz = dist.get_rank(), dist.get_world_size()
    mine = gs.shard_links(Gr(), rank, wsz)
    blk = [link_fn(scans[links[i][0]], scans[links[i][1]], 9.0) for i in mine]
    Cm = np.array([b[0].reshape(36) for b in blk]).reshape(len(mine), 36)
    CD = np.array([b[1] for b in blk]).reshape(len(mine), 6)
    X = gs.lum_reduce_solve(Gr(), mine, Cm, CD, wsz)
    np.save(os.path.join(%(tmp)r, "X%%d.npy" %% rank), X)
    ret = gs.lum_iteration(Gr(), scans, 9.0, None, link_fn, None, io.solve_sparse_cholesky)
    out = np.concatenate([[ret]] + [np.concatenate([s.rPos, s.rPosTheta]) for s in scans])
    np.save(os.path.join(%(tmp)r, "rank%%d.npy" %% dist.get_rank()), out)
    dist.barrier()
    dist.destroy_process_group()
""")


def test_lum_links_sharded_over_two_ranks_gloo(tmp_path, orc):
    """world_size-2 gloo run of the sharded FillGB3D + all-reduce: both ranks end with the same
    poses, equal to the single-process oracle LUM iteration."""
    from oracle import icp_oracle as io
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT, "tmp": str(tmp_path)})
    import socket
    with socket.socket() as sk:                      # a free rendezvous port on this box
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r))) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    np.testing.assert_allclose(r0, r1, rtol=0, atol=1e-12)
    # single-process oracle on the same data
    rng = np.random.default_rng(0)
    world = rng.uniform(-60, 60, (6000, 3))
    poses = [([4.0 * k, 0.5 * k, -0.3 * k], [0.01 * k, -0.02 * k, 0.015 * k]) for k in range(5)]
    scans = []
    for k, (p, th) in enumerate(poses):
        T = io.euler_to_matrix4(p, th)
        inv, _ = orc.m4inv(T)
        loc = world.copy(); orc.transform_points(inv, loc)
        drift = ([p[0] + 0.3 * k, p[1] - 0.2 * k, p[2] + 0.1 * k], [th[0], th[1] + 0.002 * k, th[2]])
        scans.append(io.OScan(drift[0], drift[1], loc + rng.normal(0, 0.02, loc.shape)))
    links = [(0, 1), (1, 2), (2, 3), (3, 4), (0, 4), (0, 3), (1, 4)]
    ret, _, _, Xo = io.lum_iteration(links, scans, 9.0)
    # per-link all-reduce + tdtk_lum_assemble_solve: bit-identical on both ranks, equal to the oracle's X
    X0, X1 = np.load(tmp_path / "X0.npy"), np.load(tmp_path / "X1.npy")
    assert np.array_equal(X0, X1)
    np.testing.assert_allclose(X0, Xo, rtol=1e-8, atol=1e-11)
    single = np.concatenate([[ret]] + [np.concatenate([s.rPos, s.rPosTheta]) for s in scans])
    np.testing.assert_allclose(r0, single, rtol=1e-9, atol=1e-10)
    # the iteration must actually pull the drifted poses towards the truth
    assert abs(scans[4].rPos[0] - 16.0) < abs(16.0 + 1.2 - 16.0)


def test_shard_links_partition(tdtk):
    from importlib import import_module
    gs = import_module("3dtk_amd.graphslam")
    n = 64
    chain = [(i, i + 1) for i in range(n - 1)]
    closures = [(j, k) for j in range(n) for k in range(j + 1, n) if k - j >= 58]
    g = tdtk.Graph(n, links=chain + closures)
    g.nrScans = n
    for world in (1, 2, 4, 8):
        shards = [gs.shard_links(g, r, world) for r in range(world)]
        assert sorted(sum(shards, [])) == list(range(g.getNrLinks()))       # a partition
        sizes = [len(x) for x in shards]
        assert max(sizes) - min(sizes) <= 4
    # a closure that appears between rounds must not move any other link to another rank
    g2 = tdtk.Graph(n, links=chain + closures[:5] + [(2, 57)] + closures[5:])
    g2.nrScans = n
    for world in (2, 8):
        owner = {}
        for r in range(world):
            for i in gs.shard_links(g, r, world):
                owner[(g.getLink(i, 0), g.getLink(i, 1))] = r
        for r in range(world):
            for i in gs.shard_links(g2, r, world):
                key = (g2.getLink(i, 0), g2.getLink(i, 1))
                assert owner.get(key, r) == r


def test_adapter_compiles_against_reference_headers(tmp_path):
    """adapters/hip_search_tree.{h,cc} (the SearchTree binding INTEGRATION.md describes) must
    compile against the reference's own headers.  Only possible where the checkout exists."""
    ref = os.environ.get("TDTK_REF", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "include", "slam6d")):
        pytest.skip("no reference checkout on this box")
    inc = tmp_path / "slam6d"
    inc.mkdir()
    (inc / "hip_search_tree.h").write_text(open(os.path.join(ROOT, "adapters", "hip_search_tree.h")).read())
    obj = tmp_path / "hst.o"
    cmd = ["g++", "-std=c++17", "-c", "-fopenmp", "-DMAX_OPENMP_NUM_THREADS=8", "-DOPENMP_NUM_THREADS=8",
           "-I" + str(tmp_path), "-I" + os.path.join(ref, "include"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "adapters", "hip_search_tree.cc"), "-o", str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    syms = subprocess.run(["nm", "-C", str(obj)], capture_output=True, text=True).stdout
    for want in ("HipSearchTree::getPtPairs", "HipSearchTree::FindClosest", "tdtk_get_pt_pairs", "tdtk_tree_create"):
        assert want in syms
    # adapters/normals_hip.h: the calculateNormalsApxKNN-shaped binding (needs slam6d/point.h only)
    src = tmp_path / "nrm.cc"
    src.write_text('#include "normals_hip.h"\nvoid use(std::vector<Point>& n, const std::vector<Point>& p, const double* r)'
                   ' { calculateNormalsApxKNN_hip(n, p, 10, r, 1.0); }\n')
    obj2 = tmp_path / "nrm.o"
    r = subprocess.run(["g++", "-std=c++17", "-c", "-I" + os.path.join(ref, "include"), "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "adapters"), str(src), "-o", str(obj2)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "tdtk_normals_apx_knn" in subprocess.run(["nm", "-C", str(obj2)], capture_output=True, text=True).stdout


def test_class_shells_pass_a_syntax_check(tmp_path):
    """The four class shells that adapt the reference's classes to the executed glue -- icp6D_hip.h, graphSlam6D_hip.h,
    slam6D_hip.h, normals_hip.h -- cannot be compiled against the reference's real headers here (scan.h -> Boost,
    graphSlam6D.h -> SuiteSparse).  BOUNDARY TEST INFRASTRUCTURE, not an oracle: adapters/harness/stubs/slam6d/ declares
    the reference classes they touch (names, signatures, member names and types as in the reference's headers; no bodies),
    and `g++ -fsyntax-only` runs over a TU that instantiates every shell.  Catches what a maintainer's first compile
    would: wrong argument counts / types, members that do not exist, overrides that do not override."""
    inc = tmp_path / "slam6d"
    inc.mkdir()
    stubs = os.path.join(ROOT, "adapters", "harness", "stubs", "slam6d")
    for fn in os.listdir(stubs):
        if fn.endswith(".h"):
            (inc / fn).write_text(open(os.path.join(stubs, fn)).read())
    for fn in ("hip_search_tree.h", "icp_glue.h", "graph_slam_glue.h", "slam6d_glue.h", "icp6D_hip.h", "graphSlam6D_hip.h",
               "slam6D_hip.h", "normals_hip.h"):
        (inc / fn).write_text(open(os.path.join(ROOT, "adapters", fn)).read())
    src = tmp_path / "shells.cc"
    src.write_text(
        '#include "slam6d/slam6D_hip.h"\n#include "slam6d/normals_hip.h"\n'
        "struct S : Scan { DataPointer get(const std::string&) override; void addFrame(AlgoType) override; };\n"
        "int use(icp6Dminimizer* m, std::vector<Scan*>& scans, Graph& g, tdtk_comm* comm, std::vector<Point>& n, const double* r) {\n"
        "  icp6D_hip icp(m, 25.0, 50, true, true, 1, true, -1, 1e-7, HipKD, false, false, 2);\n"
        "  icp6D* base = &icp;\n"
        "  base->doICP(scans);\n"
        "  int it = base->match(scans[0], scans[1]);\n"
        "  graphSlam6D_hip gs(TDTK_GRAPH_LUMEULER, m, 25.0, 25.0, 50, true, false, 1, true, -1, 1e-7, HipKD, 0.5, comm);\n"
        "  graphSlam6D* gb = &gs;\n"
        "  double ret = gb->doGraphSlam6D(g, scans, 1);\n"
        "  it += matchGraph6Dautomatic_hip(500.0, 20, scans, &icp, &icp, TDTK_GRAPH_LUMEULER, 50, 0.5, 25.0, 0.5, 3, comm, true, 3, 15.0, 140.0);\n"
        "  calculateNormalsApxKNN_hip(n, n, 10, r, 1.0);\n"
        "  return it + (int)ret;\n}\n")
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-I" + str(tmp_path), "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]


def test_reference_patch_applies(tmp_path):
    """adapters/reference.patch -- the edits the reference itself needs (enum value, `case HipKD:`, the two Scan
    additions, the four icp6D construction sites, the CMake option) -- applies cleanly to the checkout's files, and
    the patched tree holds what INTEGRATION.md promises.  Only where the checkout exists; works on a copy."""
    ref = os.environ.get("TDTK_REF", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "include", "slam6d")):
        pytest.skip("no reference checkout on this box")
    import shutil
    files = ["include/slam6d/scan.h", "include/slam6d/icp6D.h", "src/slam6d/scan.cc", "src/slam6d/basicScan.cc", "src/slam6d/slam6D.cc",
             "src/slam6d/CMakeLists.txt", "CMakeLists.txt"]
    for f in files:
        (tmp_path / f).parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(ref, f), tmp_path / f)
    script = os.path.join(ROOT, "adapters", "apply_to_reference.sh")
    r = subprocess.run([script, str(tmp_path), "--check"], capture_output=True, text=True)
    assert r.returncode == 0 and "FAILED" not in r.stdout and "fuzz" not in r.stdout, r.stdout + r.stderr
    r = subprocess.run([script, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    scan_h = (tmp_path / "include/slam6d/scan.h").read_text()
    assert "BruteForce, HipKD" in scan_h and "transformMatrixAndFrames" in scan_h and "hipResident()" in scan_h
    assert "case HipKD:" in (tmp_path / "src/slam6d/basicScan.cc").read_text()
    assert (tmp_path / "src/slam6d/slam6D.cc").read_text().count("NEW_ICP6D(my_icp6Dminimizer") == 4
    scan_cc = (tmp_path / "src/slam6d/scan.cc").read_text()
    assert "void Scan::transformMatrixAndFrames" in scan_cc
    assert "tdtk_scan_transform(hip_resident, alignxf)" in scan_cc          # a resident copy moves with Scan::transformReduced
    assert "virtual void doICP" in (tmp_path / "include/slam6d/icp6D.h").read_text()   # icp6D_hip::doICP (prefetching) is reachable
    assert "WITH_HIP_ICP" in (tmp_path / "CMakeLists.txt").read_text()
    for f in ("include/slam6d/hip_search_tree.h", "include/slam6d/icp6D_hip.h", "include/slam6d/icp_glue.h", "include/tdtk_hip.h",
              "src/slam6d/hip_search_tree.cc"):
        assert (tmp_path / f).exists()
    # the binding compiles in the patched layout against the reference's remaining headers
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-DWITH_HIP_ICP", "-I" + str(tmp_path / "include"),
                        "-I" + os.path.join(ref, "include"), str(tmp_path / "src/slam6d/hip_search_tree.cc")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_graph_slam_glue_compiles_and_links(tdtk, tmp_path):
    """adapters/graph_slam_glue.h -- everything of the graph-SLAM binding that is not the reference's class
    declarations: link dealing, the one tdtk_graph_iteration call (RCCL exchange inside the library), poses back,
    frame replay -- compiled and LINKED against lib3dtk_hip.so with two minimal types of the same shape as the
    reference's Scan / Graph.  Without a GPU the call chain ends in the library's "no HIP device" error, which the
    glue turns into the reference's exception type: that is what the program checks."""
    src = tmp_path / "glue.cc"
    src.write_text(textwrap.dedent('''
        #include <cstdio>
        #include "graph_slam_glue.h"
        struct MiniScan {
          double tm[16], da[16], rp[3], rt[3]; size_t n; int frames;
          const double* get_transMat() const { return tm; } const double* getDAlign() const { return da; }
          const double* get_rPos() const { return rp; } const double* get_rPosTheta() const { return rt; }
          size_t hipPoints() { return n; } tdtk_tree* hipTree() { return 0; }
          tdtk_scan* hipResident() { return 0; } tdtk_scan* hipResidentOrNull() { return 0; }
          void transformMatrixAndFrames(const double*, int, int) { frames++; }
        };
        struct MiniGraph { int ns; std::vector<int> f, t; int getNrScans() { return ns; } int getNrLinks() { return (int)f.size(); }
                           int getLink(int i, int w) { return w ? t[i] : f[i]; } };
        int main() {
          std::vector<MiniScan> sc(3); std::vector<MiniScan*> p;
          for (auto& s : sc) { for (int k = 0; k < 16; k++) s.tm[k] = s.da[k] = (k % 5 == 0); s.n = 100; s.frames = 0; p.push_back(&s); }
          MiniGraph g{3, {0, 1}, {1, 2}};
          try { hip_graph_slam(TDTK_GRAPH_LUMEULER, g, p, 1, 0.5, 625.0, (tdtk_comm*)0, 0, 3); }
          catch (const std::runtime_error& e) { std::printf("caught: %s\\n", e.what()); return 0; }
          std::printf("ran\\n"); return 0;
        }
    '''))
    exe = tmp_path / "glue"
    r = subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "adapters"), "-I" + os.path.join(ROOT, "include"), str(src),
                        "-L" + os.path.join(ROOT, "3dtk_amd"), "-l3dtk_hip", "-Wl,-rpath," + os.path.join(ROOT, "3dtk_amd"),
                        "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and ("caught:" in out.stdout or "ran" in out.stdout), out.stdout + out.stderr


def test_icp_glue_harness_compiles_and_links(tdtk):
    """adapters/icp_glue.h -- the body of icp6D_hip::match and the prefetching doICP, templated on the scan type --
    instantiated with a minimal scan type by adapters/harness/icp_glue_harness.cc: compiles and links against
    lib3dtk_hip.so here; the binary travels to the GPU box, where tests/test_gpu_parity.py::test_icp_glue_executes runs it.
    Without a GPU it must end in the library's "no HIP device" error turned into the reference's exception type."""
    r = subprocess.run([os.path.join(ROOT, "adapters", "harness", "build_glue.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    exe = os.path.join(ROOT, "adapters", "harness", "_bin", "icp_glue_harness")
    assert os.path.exists(exe)
    if tdtk.device_count() == 0:
        out = subprocess.run([exe, "3", "2000"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 1 and "no HIP device" in out.stdout, out.stdout + out.stderr
    # adapters/slam6d_glue.h (matchGraph6Dautomatic, ELCH close_loop, MetaScan match, Graph) + graph_slam_glue.h likewise
    exe2 = os.path.join(ROOT, "adapters", "harness", "_bin", "slam_glue_harness")
    assert os.path.exists(exe2)
    if tdtk.device_count() == 0:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            out = subprocess.run([exe2, os.path.join(d, "in.bin"), os.path.join(d, "out.bin"), "4", "500"], capture_output=True,
                                 text=True, timeout=120)
        assert out.returncode == 1 and "no HIP device" in out.stdout, out.stdout + out.stderr


def test_link_dealing_round_robin_and_lpt(tdtk):
    """tdtk_graph_deal_links: equal scans -> chain round-robin, closures by (from + to) % world (stable under graph
    growth); unequal scans -> longest-processing-time-first by the point count of the link's second scan: every link
    has one owner and no rank carries more than the lightest rank plus one link."""
    L = tdtk.lib()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    n = 40
    frm = np.array(list(range(n - 1)) + [0, 3, 7, 11], np.int32)
    to = np.array(list(range(1, n)) + [25, 30, 33, 39], np.int32)
    for world in (1, 2, 3, 4, 8):
        own = np.full(len(frm), -1, np.int32)
        assert L.tdtk_graph_deal_links(len(frm), ip(frm), ip(to), None, n, world, ip(own)) == 0
        want = [(i % world) if i < n - 1 else int((frm[i] + to[i]) % world) for i in range(len(frm))]
        assert own.tolist() == want
        rng = np.random.default_rng(world)
        pts = rng.integers(100000, 12000000, n).astype(np.uint64)
        own2 = np.full(len(frm), -1, np.int32)
        assert L.tdtk_graph_deal_links(len(frm), ip(frm), ip(to), pts.ctypes.data_as(C.POINTER(C.c_uint64)), n, world, ip(own2)) == 0
        assert own2.min() >= 0 and own2.max() < world
        cost = pts[to].astype(np.float64)
        load = np.array([cost[own2 == r].sum() for r in range(world)])
        assert load.max() - load.min() <= cost.max() + 1e-9
        # against the round-robin dealing on the same sizes: never worse
        load_rr = np.array([cost[np.array(want) == r].sum() for r in range(world)])
        assert load.max() <= load_rr.max() + 1e-9
        own3 = np.full(len(frm), -1, np.int32)
        L.tdtk_graph_deal_links(len(frm), ip(frm), ip(to), pts.ctypes.data_as(C.POINTER(C.c_uint64)), n, world, ip(own3))
        assert np.array_equal(own2, own3)


def test_elch_graph_balancer_against_restatement(tdtk):
    """tdtk_elch_graph_balancer (C++, no Boost) against the Python restatement of elch6D::graph_balancer
    (elch6D.cc:186-279) on chains with loop closures, branches and parallel edges; plus the known answers of the
    simplest case: on a plain chain f .. l the weights are the distance fractions along it."""
    from oracle import icp_oracle as io
    w = tdtk.graph_balancer(5, [(0, 1), (1, 2), (2, 3), (3, 4)], [1.0, 1.0, 2.0, 4.0], 0, 4)
    np.testing.assert_allclose(w, [0.0, 0.125, 0.25, 0.5, 1.0], rtol=0, atol=1e-15)
    rng = np.random.default_rng(4)
    for case in range(60):
        n = int(rng.integers(6, 40))
        edges = [(i, i + 1) for i in range(n - 1)]
        for _ in range(int(rng.integers(0, 4))):                 # earlier loop closures
            a = int(rng.integers(0, n - 3)); b = int(rng.integers(a + 2, n))
            edges.append((a, b))
        if case % 7 == 0:
            edges.append(edges[1])                               # a parallel edge
        ew = rng.uniform(0.1, 10.0, len(edges))
        f = int(rng.integers(0, n - 4)); l = int(rng.integers(f + 3, n))
        got = tdtk.graph_balancer(n, edges, ew, f, l)
        want = io.graph_balancer(n, edges, list(ew), f, l)
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-15, err_msg=str((case, n, f, l, edges)))
        assert got[f] == 0.0 and got[l] == 1.0 and got.min() >= -1e-12 and got.max() <= 1.0 + 1e-12


def test_pair_sums_merge_equals_sums_of_the_union(tdtk):
    """tdtk_pair_sums_merge: the base block of several passes merged like Align_Parallel merges per-thread partial
    sums (icp6Dquat.cc:533-578) equals the block computed from the union of the pair lists."""
    capi = sys.modules["3dtk_amd._capi"]
    rng = np.random.default_rng(6)
    parts, P1, P2 = (capi.PairSums * 3)(), [], []
    for k in range(3):
        n = int(rng.integers(50, 400))
        p2 = rng.uniform(-50, 50, (n, 3)) + 100 * k
        p1 = p2 + rng.normal(0, 0.3, (n, 3)) + np.array([0.5, -0.2, 0.1])
        parts[k] = _sums_from_pairs(capi, p1, p2)
        P1.append(p1); P2.append(p2)
    merged = capi.PairSums()
    assert tdtk.lib().tdtk_pair_sums_merge(3, parts, C.byref(merged)) == 0
    want = _sums_from_pairs(capi, np.concatenate(P1), np.concatenate(P2))
    assert merged.n == want.n
    np.testing.assert_allclose(merged.sum, want.sum, rtol=1e-13)
    np.testing.assert_allclose(list(merged.centroid_m), list(want.centroid_m), rtol=1e-13)
    np.testing.assert_allclose(list(merged.centroid_d), list(want.centroid_d), rtol=1e-13)
    np.testing.assert_allclose(list(merged.Si), list(want.Si), rtol=1e-10, atol=1e-8)
    r1, a1 = tdtk.icp6D_QUAT(True).Align_Parallel(merged)
    r2, a2 = tdtk.icp6D_QUAT(True).Align_Parallel(want)
    np.testing.assert_allclose(a1, a2, atol=1e-12)


def test_reference_full_iteration_loop_equals_oracle_loop(orc):
    """A6: full iterations of the OpenMP branch of icp6D::match assembled from the reference's own compiled pieces
    (oracle/ref_driver.cc: ref_icp_iterations -- KDtreeIndexed::FindClosest, transform3, M4inv, PtPair,
    icp6D_QUAT::Align_Parallel) against the oracle's restated loop: pair counts exact every iteration, RMS and the
    moved points equal to the difference between the serial and the parallel merge (< 1e-6)."""
    from oracle import icp_oracle as io
    if not orc.have_ref():
        pytest.skip("reference TUs not built here")
    rng = np.random.default_rng(1)
    m = rng.uniform(-100, 100, (60000, 3))
    T = io.euler_to_matrix4([1.0, -0.5, 0.3], [0.01, -0.02, 0.015])
    inv, _ = orc.m4inv(T)
    d = m[rng.permutation(len(m))[:50000]] + rng.normal(0, 0.05, (50000, 3))
    orc.transform_points(inv, d)
    for nthreads in (1, 8):
        p, tr = orc.RefTree(m, 20).icp_iterations(np.eye(4).reshape(16), d, 25.0, nthreads, 6)
        S0, S1 = io.OScan([0, 0, 0], [0, 0, 0], m), io.OScan([0, 0, 0], [0, 0, 0], d)
        it, tro = io.match(S0, S1, 1, 25.0, 6, -1.0)
        assert [int(t[0]) for t in tr] == [int(t[0]) for t in tro]
        np.testing.assert_allclose([t[1] for t in tr], [t[1] for t in tro], rtol=1e-6)
        np.testing.assert_allclose(p, S1.xyz, atol=1e-6)


def test_scan_io_uos_pose_frames(tdtk, tmp_path):
    """uos reader (header line, comments, blank lines, CRLF, -m/-M range filter), .pose reader
    (deg -> rad with rad()), .frames writer (default-ostream formatting + AlgoType)."""
    f = tmp_path / "scan000.3d"
    f.write_text("3\n# a comment\n1.5 2 -3\r\n\n10.1 0 0 # trailing comment\n  600 1 1\n0.1 0.1 0.1\n")
    p = tdtk.read_uos(f)
    assert np.array_equal(p, [[1.5, 2, -3], [10.1, 0, 0], [600, 1, 1], [0.1, 0.1, 0.1]])
    assert np.array_equal(tdtk.read_uos(f, 500.0), [[1.5, 2, -3], [10.1, 0, 0], [0.1, 0.1, 0.1]])
    assert np.array_equal(tdtk.read_uos(f, 500.0, 1.0), [[1.5, 2, -3], [10.1, 0, 0]])
    bad = tmp_path / "bad.3d"
    bad.write_text("1 2 3\n4 5\n")
    with pytest.raises(tdtk.TdtkError):
        tdtk.read_uos(bad)
    with pytest.raises(tdtk.TdtkError):
        tdtk.read_uos(tmp_path / "missing.3d")
    (tmp_path / "scan000.pose").write_text("-3.10605 -7.50803 156.917\n1.35694 -0.852409 -0.56224\n")
    rP, rT = tdtk.read_pose(tmp_path / "scan000.pose")
    assert np.array_equal(rP, [-3.10605, -7.50803, 156.917])
    assert np.array_equal(rT, [(2 * np.pi * a) / 360 for a in (1.35694, -0.852409, -0.56224)])

    class S:
        pass
    s = S()
    s.identifier, s.path = "000", str(tmp_path)
    M = np.eye(4).reshape(16).copy(); M[12:15] = [1234.56789, -0.000123456789, 1e-7]
    s.frames = [(np.eye(4).reshape(16), "ICP"), (M, "LUM"), (M, "INVALID")]
    fn = tdtk.saveFrames(s)
    lines = open(fn).read().split("\n")
    assert lines[0] == "1 0 0 0 0 1 0 0 0 0 1 0 0 0 0 1 1"
    assert lines[1] == "1 0 0 0 0 1 0 0 0 0 1 0 1234.57 -0.000123457 1e-07 1 3"
    assert lines[2].endswith(" 0") and lines[3] == ""


def test_oracle_octree_random_modes_against_cell_membership(orc):
    """The restatement of the random octree modes (oracle.c, parity unpinned) checked against what does not depend on how
    the partitions are written: the order the partitions leave behind groups the points by leaf in the depth-first order
    of the centre mode, every leaf contributes exactly one point (nrpts = 1) resp. min(nrpts, length) points (the
    reference draws rand(length - 1) there, so a leaf's LAST point is only ever kept when the whole leaf is), and each
    kept point lies inside the cell whose centre sits at the same depth-first position."""
    rng = np.random.default_rng(3)
    clouds = [rng.uniform(-100, 100, (30000, 3)), np.round(rng.uniform(-50, 50, (20000, 3)), 1),
              np.stack(np.meshgrid(*[np.arange(16.0)] * 3), -1).reshape(-1, 3)]
    for pts in clouds:
        for voxel in (1.0, 7.5):
            centres = orc.octree_center(pts, voxel)
            size = (0.5 * (pts.max(0) - pts.min(0))).max() + 1.0
            while size > voxel:
                size /= 2.0
            one, perm = orc.octree_random(pts, voxel, 1, seed=11, want_perm=True)
            assert sorted(perm.tolist()) == list(range(len(pts)))
            assert len(one) == len(centres) and np.all(np.abs(one - centres).max(1) <= size)
            # leaf of every position of the leaf order: nearest centre in max-norm (cells are disjoint cubes)
            ordered = pts[perm]
            leaf = np.empty(len(pts), np.int64)
            pos = 0
            for k, c in enumerate(centres):
                n_in = 0
                while pos + n_in < len(pts) and np.all(np.abs(ordered[pos + n_in] - c) <= size):
                    n_in += 1
                assert n_in > 0
                leaf[pos:pos + n_in] = k
                pos += n_in
            assert pos == len(pts)
            lens = np.bincount(leaf, minlength=len(centres))
            three = orc.octree_random(pts, voxel, 3, seed=12)
            assert len(three) == int(np.minimum(lens, 3).sum())
            # which positions were kept: a leaf longer than 3 never keeps its last point
            keep_pos = 0
            for k in range(len(centres)):
                start = int(lens[:k].sum())
                kept = three[keep_pos:keep_pos + min(3, lens[k])]
                keep_pos += min(3, lens[k])
                block = ordered[start:start + lens[k]]
                for q in kept:
                    hits = np.flatnonzero((block == q).all(1))
                    assert len(hits) > 0
                if lens[k] > 3 and not (block[:-1] == block[-1]).all(1).any():
                    assert not (kept == block[-1]).all(1).any()


def test_oracle_octree_center_against_grid_formulation(orc):
    """The recursive octree restatement (oracle.c, parity unpinned: Boctree.h is not buildable here)
    checked against an independent closed-form formulation: occupied cells of the regular 2^D grid over
    the root cube, ordered by their (x lowest) Morton code = depth-first child order."""
    rng = np.random.default_rng(11)
    # a point ON the split planes goes to the UPPER child: Scan::calcReducedPoints reaches the array constructor, whose
    # fullsort cuts with `< centre` | `>= centre` (Boctree.h:1784-1816) -- not childIndex's strict `>` (round 3 correction)
    one = np.array([[4.0, -2.0, 8.0]])
    assert np.array_equal(orc.octree_center(one, 5.0), one + 0.5)
    z = np.load(os.path.join(G, "dat_scans.npz"))
    clouds = [(rng.uniform(-300, 500, (n, 3)) * np.array([1.0, 0.5, 0.1]), voxel) for n, voxel in ((2000, 3.0), (50000, 10.0), (3000, 1e6))]
    clouds += [(z["scan%03d" % k], voxel) for k in range(3) for voxel in (10.0, 2.5)]      # the bundled scans, -r 10 / -r 2.5
    for pts, voxel in clouds:
        n = len(pts)
        got = orc.octree_center(pts, voxel)
        lo, hi = pts.min(0), pts.max(0)
        c = 0.5 * (lo + hi)
        size = (0.5 * (hi - lo)).max() + 1.0
        D, s = 1, size / 2.0
        while s > voxel:
            s /= 2.0; D += 1
        cell = np.floor((pts - (c - size)) / (2.0 * size / 2 ** D)).astype(np.int64)
        assert cell.min() >= 0 and cell.max() < 2 ** D
        code = np.zeros(n, np.int64)
        for b in range(D):
            for a in range(3):
                code |= ((cell[:, a] >> b) & 1) << (3 * b + a)
        ucode, first = np.unique(code, return_index=True)
        ucell = cell[first]
        want = (c - size) + (2 * ucell + 1) * (size / 2 ** D)
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=0, atol=1e-9 * size)
    assert len(orc.octree_center(np.zeros((0, 3)), 1.0)) == 0


def test_lum_assemble_solve_matches_dense_fill(tdtk):
    """tdtk_lum_assemble_solve == FillGB3D's scatter (numpy restatement) + dense solve, including links
    that start or end at the fixed scan 0 and links in both directions."""
    capi = sys.modules["3dtk_amd._capi"]
    rng = np.random.default_rng(4)
    nscans = 6
    links = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (0, 5), (3, 0), (5, 2), (1, 4)]
    Cs, CDs = [], []
    for _ in links:
        A = rng.normal(size=(6, 6))
        Cs.append(A @ A.T + 6 * np.eye(6)); CDs.append(rng.normal(size=6))
    n = nscans - 1
    G = np.zeros((6 * n, 6 * n)); B = np.zeros(6 * n)
    for (fa, fb), Cab, CDab in zip(links, Cs, CDs):
        a, b = fa - 1, fb - 1
        if a >= 0:
            B[a * 6:a * 6 + 6] += CDab; G[a * 6:a * 6 + 6, a * 6:a * 6 + 6] += Cab
        if b >= 0:
            B[b * 6:b * 6 + 6] -= CDab; G[b * 6:b * 6 + 6, b * 6:b * 6 + 6] += Cab
        if a >= 0 and b >= 0:
            G[a * 6:a * 6 + 6, b * 6:b * 6 + 6] -= Cab; G[b * 6:b * 6 + 6, a * 6:a * 6 + 6] -= Cab
    frm = np.array([l[0] for l in links], np.int32); to = np.array([l[1] for l in links], np.int32)
    Call = np.ascontiguousarray(np.array(Cs).reshape(len(links), 36)); CDall = np.ascontiguousarray(np.array(CDs))
    X = np.empty(6 * n); Go = np.empty((6 * n, 6 * n)); Bo = np.empty(6 * n)
    capi.check(capi.lib().tdtk_lum_assemble_solve(len(links), capi.iptr(frm), capi.iptr(to), capi.dptr(Call),
                                                  capi.dptr(CDall), nscans, capi.dptr(X), capi.dptr(Go), capi.dptr(Bo)))
    assert np.array_equal(Go, G) and np.array_equal(Bo, B)
    np.testing.assert_allclose(X, np.linalg.solve(np.where(np.abs(G) > 1e-5, G, 0.0), B), rtol=1e-10, atol=1e-12)
    with pytest.raises(tdtk.TdtkError):
        bad = np.array([7] + [0] * (len(links) - 1), np.int32)
        capi.check(capi.lib().tdtk_lum_assemble_solve(len(links), capi.iptr(bad), capi.iptr(to), capi.dptr(Call),
                                                      capi.dptr(CDall), nscans, capi.dptr(X), None, None))


def test_lum_assemble_straight_into_the_skyline_equals_the_dense_fill(tdtk):
    """Round 5: without G_out / B_out tdtk_lum_assemble_solve puts the link blocks straight into the skyline the solve
    factors (a block row reaches left to the smallest scan it shares a link with) instead of clearing and re-reading a
    dense G: X must be the dense path's BIT FOR BIT -- chain graphs, closures across the whole loop, a link given
    high -> low, a self link (from == to), links at the fixed scan, entries under the 1e-5 filter inside the blocks."""
    capi = sys.modules["3dtk_amd._capi"]
    L = capi.lib()
    rng = np.random.default_rng(3)
    for nscans, extra in ((64, 21), (13, 6), (3, 1), (40, 0), (2, 0)):
        links = [(i, i + 1) for i in range(nscans - 1)]
        while len(links) < nscans - 1 + extra:
            a, b = sorted(rng.integers(0, nscans, 2).tolist())
            if b - a > 1 and (a, b) not in links:
                links.append((a, b))
        if nscans == 13:
            links.append((7, 2))
            links.append((5, 5))       # a self link (net files and addLink can produce one): nets to zero in both fills
        nl = len(links)
        Cm = np.empty((nl, 36)); CD = rng.normal(0, 50, (nl, 6))
        for l in range(nl):
            A = rng.normal(0, 1, (6, 6)); M = A @ A.T * 1e3 + np.eye(6)
            M[np.abs(M) < 30] *= 1e-8
            Cm[l] = ((M + M.T) / 2).reshape(36)
        frm = np.ascontiguousarray([l[0] for l in links], np.int32); to = np.ascontiguousarray([l[1] for l in links], np.int32)
        N = 6 * (nscans - 1)
        X1 = np.empty(N); X2 = np.empty(N); G = np.empty((N, N)); B = np.empty(N)
        capi.check(L.tdtk_lum_assemble_solve(nl, capi.iptr(frm), capi.iptr(to), capi.dptr(Cm), capi.dptr(CD), nscans, capi.dptr(X1), None, None))
        capi.check(L.tdtk_lum_assemble_solve(nl, capi.iptr(frm), capi.iptr(to), capi.dptr(Cm), capi.dptr(CD), nscans, capi.dptr(X2), capi.dptr(G), capi.dptr(B)))
        assert np.array_equal(X1, X2), nscans
        Gf = np.where(np.abs(G) > 1e-5, G, 0.0)
        assert np.abs(Gf @ X1 - B).max() <= 1e-9 * np.abs(B).max()


def test_spd_solve_skyline_storage_shapes(tdtk):
    """solveSparseCholesky's stand-in factors in skyline storage (every row from its first entry above the 1e-5 filter
    to the diagonal).  Shapes that stress the bookkeeping: a chain of 6x6 blocks with far loop closures (envelope
    rows of very different length), a dense matrix, a diagonal one, 1x1, entries at and below the filter threshold in
    front of a row, two calls of different size from one thread (buffers are kept per thread), several threads."""
    import threading
    capi = sys.modules["3dtk_amd._capi"]
    rng = np.random.default_rng(41)

    def solve(G, B):
        x = np.empty(len(B))
        capi.check(capi.lib().tdtk_solve_spd(capi.dptr(np.ascontiguousarray(G)), capi.dptr(np.ascontiguousarray(B)), len(B), capi.dptr(x)))
        return x

    def chain(nb, closures):
        G = np.zeros((6 * nb, 6 * nb))
        for a, b in [(i, i + 1) for i in range(nb - 1)] + closures:
            A = rng.normal(size=(6, 6)); Cab = A @ A.T + 6 * np.eye(6)
            G[a * 6:a * 6 + 6, a * 6:a * 6 + 6] += Cab; G[b * 6:b * 6 + 6, b * 6:b * 6 + 6] += Cab
            G[a * 6:a * 6 + 6, b * 6:b * 6 + 6] -= Cab; G[b * 6:b * 6 + 6, a * 6:a * 6 + 6] -= Cab
        G[:6, :6] += 6 * np.eye(6)     # scan 0 is fixed in the real system: anchor the chain
        return G
    cases = [chain(63, [(i, min(62, i + 20 + i % 7)) for i in range(0, 42, 2)]), chain(5, []), chain(12, [(0, 11), (3, 9)])]
    A = rng.normal(size=(40, 40)); cases.append(A @ A.T + 40 * np.eye(40))           # dense
    cases.append(np.diag(rng.uniform(1, 2, 17)))                                     # diagonal
    cases.append(np.array([[3.0]]))
    G = chain(8, [(1, 6)]); G[30, 2] = G[2, 30] = 1e-5; G[31, 0] = G[0, 31] = 9e-6; G[40, 5] = G[5, 40] = 1.0000001e-5
    cases.append(G)                                                                  # entries at / below / just above the filter
    for G in cases:
        B = rng.normal(size=len(G))
        np.testing.assert_allclose(solve(G, B), np.linalg.solve(np.where(np.abs(G) > 1e-5, G, 0.0), B), rtol=1e-9, atol=1e-12)
    with pytest.raises(tdtk.TdtkError):
        solve(-np.eye(4), np.ones(4))
    errs = []

    def work(k):
        try:
            r = np.random.default_rng(100 + k)
            for G in (cases[0], cases[3], cases[2]):
                B = r.normal(size=len(G))
                np.testing.assert_allclose(solve(G, B), np.linalg.solve(np.where(np.abs(G) > 1e-5, G, 0.0), B), rtol=1e-9, atol=1e-12)
        except Exception as e:      # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(k,)) for k in range(6)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


def test_bench_asks_for_more_hardware_queues_and_the_package_leaves_the_environment_alone():
    """the link passes run on three streams; a host with streams of its own (PyTorch, RCCL) needs more than the
    runtime's four hardware queues.  bench.py asks for them before the runtime starts; the
    package itself must not touch the environment of the process that imports it (round-2 advice)."""
    assert 'setdefault("GPU_MAX_HW_QUEUES"' in open(os.path.join(ROOT, "bench.py")).read()
    import subprocess
    code = ("import os, sys, importlib; sys.path.insert(0, %r); before = dict(os.environ); "
            "importlib.import_module('3dtk_amd'); assert dict(os.environ) == before, "
            "sorted(set(os.environ.items()) ^ set(before.items()))" % ROOT)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)


def test_graph_netfile_chain_addlink(tdtk, tmp_path):
    """Graph(netfile) (slam6D -n, graph.cc:52-74), Graph(n, loop) (:84-105), addLink's scan counting (:157-174)"""
    f = tmp_path / "net"
    f.write_text("4\n4\n0 1\n1 2\n2 3\n3 0\nignored 9 9\n")
    g = tdtk.Graph.from_netfile(str(f))
    assert g.getNrScans() == 4 and g.getNrLinks() == 4
    assert [(g.getLink(i, 0), g.getLink(i, 1)) for i in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 0)]
    bad = tmp_path / "bad"
    bad.write_text("3 5\n0 1\n")
    with pytest.raises(RuntimeError):
        tdtk.Graph.from_netfile(str(bad))
    c = tdtk.Graph.chain(5)
    assert c.getNrScans() == 5 and [(c.getLink(i, 0), c.getLink(i, 1)) for i in range(c.getNrLinks())] == [(0, 1), (1, 2), (2, 3), (3, 4)]
    l = tdtk.Graph.chain(4, loop=True)
    assert l.getNrScans() == 4 and [(l.getLink(i, 0), l.getLink(i, 1)) for i in range(l.getNrLinks())] == [(0, 1), (1, 2), (2, 3), (3, 0)]
    a = tdtk.Graph(0, links=[])
    a.addLink(7, 7)                      # counted twice, as written
    assert a.getNrScans() == 2
    a.addLink(7, 2)
    assert a.getNrScans() == 3 and a.getNrLinks() == 2
    for cls in (tdtk.lum6DEuler, tdtk.lum6DQuat, tdtk.ghelix6DQ2, tdtk.gapx6D):
        b = cls(None, 25.0, 25.0)
        b.set_mdmll(10.0)
        assert b.max_dist_match2_LUM == 100.0


def test_bench_line_is_compact_strict_json_with_roofline_and_cpu_baseline():
    """Round 6: round 5's 21.7 KB bench line did not parse in the driver.  bench.build_line makes the ONE
    stdout line from a full result dict -- here the record of a real run (tests/golden/bench_result_canned.json =
    bench_legs.json of `python bench.py --gpus 1 --steps 20 --warmup 5` on an MI355X): under 6 KB, strict JSON (no NaN /
    Infinity), the contract's keys, `roofline` with frac / achieved / peak / traffic / bytes_per_query / the reference's
    visit counts, `cpu_baseline` with value / cores / kind -- and a line that would not fit raises instead of printing."""
    import bench
    res = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_result_canned.json")))
    line = bench.build_line(res)
    assert "\n" not in line and len(line) < 6144 and len(line) < bench.LINE_MAX

    def no_constants(c):
        raise ValueError("not strict JSON: " + c)
    d = json.loads(line, parse_constant=no_constants)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 * r["frac"]
    # achieved = SURVEY 8(d)'s bytes per query (the reference's walk) x queries per launch / the kernel's duration
    bq = bench.algorithmic_bytes_per_query(r["visits_per_query"]["internal"], r["visits_per_query"]["points"])
    assert abs(bq - r["bytes_per_query"]) < 1e-3 * bq
    assert abs(r["achieved"] - bq * d["config"]["points"] / (r["kernel_ms"] * 1e-3) / 1e9) < 2e-3 * r["achieved"]
    assert r["kernel_ms"] <= d["ms_per_step"]
    assert r["visits_walked"]["bytes_ratio"] > 1.0          # the kernel's own (deferred-check) walk visits more than the reference's
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["unit"] == d["unit"]
    assert d["value"] == res["value"] and d["ms_per_step"] == res["ms_per_step"]       # the contract's numbers are not rounded
    # a NaN among the contract's numbers raises; among the secondary fields it becomes null (still strict JSON); a line that
    # outgrows the limit raises
    with pytest.raises(ValueError):
        bench.build_line(dict(res, value=float("nan")))
    assert json.loads(bench.build_line(dict(res, rms_last=float("nan"))), parse_constant=no_constants)["rms_last"] is None
    fat = dict(res, config=dict(res["config"], workload="x" * 7000))
    with pytest.raises(RuntimeError):
        bench.build_line(fat)


def test_product_and_bench_keep_clear_of_the_oracle():
    """oracle/ is test infrastructure: nothing under 3dtk_amd/, include/ or adapters/ may mention it, the shared
    library must not link it, and bench.py may import it only inside its cpu_baseline legs."""
    import re
    for top in ("3dtk_amd", "include", "adapters"):
        for dp, _dn, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".cc")) or f == "Makefile":
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    assert not re.search(r"\boracle\b|liboracle|libref3dtk", txt), os.path.join(dp, f)
    so = os.path.join(ROOT, "3dtk_amd", "lib3dtk_hip.so")
    if os.path.exists(so):
        needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
        assert "liboracle" not in needed and "libref3dtk" not in needed
    src = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)from oracle import", src, re.M):
        head = src[:m.start()]
        fn = re.findall(r"^def (\w+)\(", head, re.M)[-1]
        ctx = head[head.rfind("\ndef "):]
        assert fn == "cpu_baseline_nn" or "no_cpu" in ctx[ctx.rfind("if "):], (fn, m.group(0))


def _hoare(vals, pred):
    """the in-place loop of ANN's annPlaneSplit (kd_util.cpp:291-319) on a list of values, statement for statement:
    `pred(v)` = belongs left.  Returns the permutation (indices into the input) and the split position."""
    idx = list(range(len(vals)))
    l, r = 0, len(vals) - 1
    while True:
        while l < len(vals) and pred(vals[idx[l]]): l += 1
        while r >= 0 and not pred(vals[idx[r]]): r -= 1
        if l > r: break
        idx[l], idx[r] = idx[r], idx[l]
        l += 1; r -= 1
    return idx, l


def _hoare_kd(vals, split):
    """KDTreeImpl::create's form of the loop (kdTreeImpl.h:172-182): no step after the swap, no bounds (a run with both sides
    occupied never leaves it)"""
    idx = list(range(len(vals)))
    l, r = 0, len(vals) - 1
    while True:
        while vals[idx[l]] < split: l += 1
        while vals[idx[r]] >= split: r -= 1
        if r < l: break
        idx[l], idx[r] = idx[r], idx[l]
    return idx, l


def _two_pass(vals, pred):
    """what k_part_scan + k_part_swap (build.hip) / k_ann_part_scan + k_ann_part_swap (ann.hip) compute: the index list --
    elements that stay right from the front in order, the others from the back in order --, then slot j of the front part
    swaps with slot j + nge when it lies left of the split position"""
    n = len(vals)
    ge = [not pred(v) for v in vals]
    nge = sum(ge)
    nleft = n - nge
    lst = [None] * n
    g = 0
    for p in range(n):
        if ge[p]: lst[g] = p
        else: lst[n - 1 - (p - g)] = p
        g += ge[p]
    idx = list(range(n))
    for j in range(nge):
        a = lst[j]
        if a < nleft:
            b = lst[j + nge]
            idx[a], idx[b] = idx[b], idx[a]
    return idx, nleft


def test_two_pass_partition_rule_is_the_hoare_loop():
    """Round 6's partition of a level rests on one claim: the reference's Hoare loop leaves the k-th element that
    is not below the split value -- if it lies in the left region -- swapped with the k-th element below it counted from the
    right end.  Checked here against the loop itself, statement for statement, on random runs with repeated values, runs that
    are all on one side, single elements -- for KDTreeImpl::create's predicate (v < split) and for both passes of ANN's
    annPlaneSplit (v < cv over the cell, then v <= cv over what lies right of br1)."""
    rng = np.random.default_rng(62)
    for trial in range(3000):
        n = int(rng.integers(1, 70))
        vals = list(rng.integers(-6, 7, n).astype(float)) if trial % 2 else list(rng.normal(0, 1, n))
        split = float(rng.choice(vals)) if trial % 3 else float(rng.normal(0, 1))
        ref, lpos = _hoare(vals, lambda v: v < split)
        got, nleft = _two_pass(vals, lambda v: v < split)
        assert (ref, lpos) == (got, nleft), (vals, split)
        if 0 < nleft < n:
            assert _hoare_kd(vals, split) == (got, nleft), (vals, split)
        # annPlaneSplit: the second pass works on the first pass's result, from br1 on
        v1 = [vals[i] for i in ref]
        tail = v1[lpos:]
        ref2, l2 = _hoare(tail, lambda v: v <= split)
        got2, n2 = _two_pass(tail, lambda v: v <= split)
        assert (ref2, l2) == (got2, n2), (vals, split)


def test_every_product_kernel_spills_nothing_and_search_kernels_keep_four_waves():
    """The resource remarks of the PRODUCT build (csrc/Makefile keeps them for kernels.hip: every search, pair-sum,
    transform and layout kernel lib3dtk_hip.so can launch; the lab library's extra kernels are not in this file).
    Every kernel: no VGPR spills; no SGPR spills outside three named kernel families.  Every search kernel: no scratch beyond
    the 32 bytes per lane of the stack-overflow helpers' call frame and, for the persistent-lane kernels -- which hold a
    whole bucket's fp32 shadow groups in registers --, at most 128 vector registers = four waves per SIMD, which is what
    their launches are sized for.  Also: none of the lab kernels is in the product (k_search_step, k_search_coop, the
    work-queue / FAT / PROBE instantiations, k_slab_bounds)."""
    import re
    path = os.path.join(ROOT, "3dtk_amd", "csrc", "kernels.resource.txt")
    if not os.path.exists(path):
        pytest.skip("no build in this tree (kernels.resource.txt is written by the Makefile)")
    text = open(path).read()
    for other in ("build", "sort", "reduce", "ann"):      # the tree build, the orderings, the octree reduction, the normals
        po = os.path.join(ROOT, "3dtk_amd", "csrc", other + ".resource.txt")
        assert os.path.exists(po), po
        text += open(po).read()
    blocks = text.split("remark: Function Name: ")[1:]
    assert len(blocks) > 100
    seen_refill = 0
    for b in blocks:
        name = b.split()[0]
        def num(key):
            m = re.search(key + r": (\d+)", b)
            assert m, (name, key)
            return int(m.group(1))
        # (round 5: the single-pass persistent-lane kernels the loops run are compiled for SIX waves per SIMD -- 80 registers --
        # and paid for it with the stack-overflow area's per-lane pointers in scratch; round 6: the overflow area is a
        # wave-uniform base + the lane's column (LaneStackQ::gcol) and nothing is spilled to memory anywhere any more)
        six = re.search(r"k_search_refillILi128ELi[46]ELi(16|32)ELi6ELb0ELi[03]E", name) is not None
        dfr = "k_search_refillI" in name and name.endswith("ELb1EEEvNS_10SearchArgsE")
        assert num("VGPRs Spill") == 0, name
        # scalar spills (into lanes of a vector register, not to memory) are left in two kernel families, by name and with
        # their present counts as caps: k_big_stitch -- one wave per (node, axis) walks the exact centroid chain and keeps
        # its whole walk state wave-uniform --, and k_ann_normals<K> -- the ANN priority search keeps its K-best bookkeeping
        # scalar (K = 10 is what Scan::calcNormals uses; 16 / 32 exist for tdtk_normals_apx_knn callers)
        cap = 0
        if dfr: cap = 2     # (the instantiations that defer the quick check: two scalars ride in lanes of a vector register)
        if "k_big_stitch" in name: cap = 40
        if "k_fin_wave" in name: cap = 8      # (round 6: a wave per subtree keeps the eight rows' predicates and its level state scalar; 4 today)
        m_ann = re.search(r"k_ann_normalsILi(\d+)E", name)
        if m_ann: cap = {10: 10, 16: 16, 32: 142}.get(int(m_ann.group(1)), 0)
        assert num("SGPRs Spill") <= cap, (name, num("SGPRs Spill"))
        if "k_search" in name:
            # 32 bytes per lane: the call frame of the two out-of-line stack-overflow helpers, nothing else.  Round 4 built the
            # two inlined forms (per-lane branch; wave-uniform branch around it) -- scratch 0, and k_search 0.2008-0.2028 ms
            # against 0.1937-0.1946 at the driver's arguments (gpurun_out/r4h, r4i; NEGATIVES.md) -- and kept the call.
            assert num(r"ScratchSize \[bytes/lane\]") <= 32, name
        if "k_search_refill" in name:
            seen_refill += 1
            assert num("VGPRs") <= 128 and num(r"Occupancy \[waves/SIMD\]") >= 4, name
        for lab_only in ("k_search_step", "k_search_coop", "k_slab_bounds", "k_make_fat"):
            assert lab_only not in name, name
        if "k_search_refillI" in name and "ELb0ELi" in name.split("k_search_refillI")[1][:40]:
            # round 5: the single-pass kernel as it is timed (not the instrumented instantiation: COUNT = false) filters buckets on
            # the 16-bit shadow (94 registers: five waves per SIMD) and is compiled for six -- its launch is sized for what the
            # runtime reports
            a = re.search(r"k_search_refillILi128ELi[46]ELi(16|32)ELi[146]ELb([01])", name)
            if a and a.group(2) == "0":
                assert num("VGPRs") <= 80 and num(r"Occupancy \[waves/SIMD\]") >= 6, (name, num("VGPRs"))
        if "k_search_refillI" in name:      # <BLOCK, SD, THRESH, WPS, COUNT, FUSE, DYN, PTS, PROBE, FAT, TOP, SHARE, PIPE>: product = 128 threads,
            # FUSE 0 / 3, static slabs, plain walk, no upper levels in LDS, every wave its own slab, hand-outs that wait; with or
            # without the deferred quick check (DEFER)
            # (round 6: six LDS levels of the traversal stack where the sums are added up behind the launch -- FUSE 0 --, four
            #  where they are added up inside it -- FUSE 3: kernels.hip, REFILL_SD)
            assert re.search(r"ILi128ELi(4ELi(16|32)ELi[146]ELb[01]ELi3|6ELi(16|32)ELi[146]ELb[01]ELi0)ELb0ELi4ELi0ELb0ELi0ELb0ELb0ELb[01]EEE", name), name
    assert seen_refill >= 12
