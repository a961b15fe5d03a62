"""TEST INFRASTRUCTURE ONLY: ctypes faces of the CPU oracle (oracle/liboracle.so,
built from oracle/oracle.c) and of the reference's own translation units
(oracle/_ref/libref3dtk.so, built by oracle/build_ref.sh where /root/reference
exists).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
import this module; the product package never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_long)
_up = C.POINTER(C.c_uint)


def _d(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _i(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def build(force=False):
    """Compile liboracle.so (always possible) and oracle/_ref (only where the
    reference checkout is present).  Building the checker is not using it."""
    so = os.path.join(_HERE, "liboracle.so")
    src = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("oracle.c", "oracle_normals.c"))
    if force or not os.path.exists(so) or os.path.getmtime(so) < src:
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    ref = os.environ.get("TDTK_REF", "/root/reference")
    refso = os.path.join(_HERE, "_ref", "libref3dtk.so")
    if os.path.isdir(os.path.join(ref, "src", "slam6d")):
        drv = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("ref_driver.cc", "ref_ann_driver.cc", "build_ref.sh"))
        if force or not os.path.exists(refso) or os.path.getmtime(refso) < drv:
            subprocess.check_call([os.path.join(_HERE, "build_ref.sh"), ref], stdout=subprocess.DEVNULL)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.orc_tree_create.restype = C.c_void_p
        L.orc_tree_create.argtypes = [_dp, C.c_size_t, C.c_int]
        L.orc_tree_destroy.argtypes = [C.c_void_p]
        L.orc_tree_stats.argtypes = [C.c_void_p, _lp]
        L.orc_tree_perm.argtypes = [C.c_void_p, _ip]
        L.orc_find_closest.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _ip, _dp, _lp, C.c_int]
        L.orc_find_closest_along_dir.argtypes = [C.c_void_p, _dp, _dp, C.c_size_t, C.c_double, _ip, _dp]
        L.orc_find_closest_deferred.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _ip, C.c_double, _ip, _dp, _lp]
        L.orc_search_tie.argtypes = [C.c_double, C.c_double]
        L.orc_search_tie.restype = C.c_double
        L.orc_get_pt_pairs.restype = C.c_size_t
        L.orc_get_pt_pairs.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_size_t, C.c_size_t, C.c_int,
                                       C.c_double, _ip, _dp, _dp, _dp, _dp, _dp, _dp]
        L.orc_m4inv.restype = C.c_int
        L.orc_m4inv.argtypes = [_dp, _dp]
        L.orc_mmult.argtypes = [_dp, _dp, _dp]
        L.orc_transform_points.argtypes = [_dp, _dp, C.c_size_t]
        L.orc_transform_normals.argtypes = [_dp, _dp, C.c_size_t]
        L.orc_max_threads.restype = C.c_int
        L.orc_gen_mt64_uniform.argtypes = [C.c_uint64, C.c_size_t, C.c_double, C.c_double, _dp]
        L.orc_octree_center.restype = C.c_size_t
        L.orc_octree_center.argtypes = [_dp, C.c_size_t, C.c_double, _dp]
        L.orc_octree_random.restype = C.c_size_t
        L.orc_octree_random.argtypes = [_dp, C.c_size_t, C.c_double, C.c_int, _dp, C.POINTER(C.c_uint32)]
        L.orc_k5_hash.restype = C.c_uint64
        L.orc_k5_hash.argtypes = [_ip, C.c_size_t]
        L.orc_packet_find_closest.restype = C.c_int
        L.orc_packet_find_closest.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_int, C.c_double, _ip, _dp, _lp]
        L.orc_ann_create.restype = C.c_void_p
        L.orc_ann_create.argtypes = [_dp, C.c_int]
        L.orc_ann_destroy.argtypes = [C.c_void_p]
        L.orc_ann_stats.argtypes = [C.c_void_p, _lp]
        L.orc_ann_structure.restype = C.c_long
        L.orc_ann_structure.argtypes = [C.c_void_p, _ip, _dp, _ip, _lp]
        L.orc_ann_nodes.argtypes = [C.c_void_p, _ip, _ip, _dp, _ip]
        L.orc_ann_ksearch.restype = C.c_int
        L.orc_ann_ksearch.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_double, _ip, _dp, _lp]
        L.orc_eigen3.restype = C.c_int
        L.orc_eigen3.argtypes = [_dp, _dp, _dp]
        L.orc_normals_apx_knn.restype = C.c_int
        L.orc_normals_apx_knn.argtypes = [_dp, C.c_int, C.c_int, _dp, C.c_double, _dp, _ip]
        L.orc_normals_from_knn.argtypes = [_dp, C.c_int, C.c_int, _ip, _dp, _dp]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(os.path.join(_HERE, "_ref", "libref3dtk.so"))


def ref():
    global _ref
    if _ref is None:
        build()
        R = C.CDLL(os.path.join(_HERE, "_ref", "libref3dtk.so"))
        R.ref_kdi_create.restype = C.c_void_p
        R.ref_kdi_create.argtypes = [_dp, C.c_size_t, C.c_int]
        R.ref_kdi_destroy.argtypes = [C.c_void_p]
        R.ref_kdi_find_closest.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _ip, C.c_int]
        R.ref_kdi_find_closest_along_dir.argtypes = [C.c_void_p, _dp, _dp, C.c_size_t, C.c_double, _ip]
        R.ref_align.restype = C.c_double
        R.ref_align.argtypes = [C.c_int, C.c_size_t, _dp, _dp, _dp, _dp, _dp, _dp]
        R.ref_align_inout.restype = C.c_double
        R.ref_align_inout.argtypes = [C.c_int, C.c_size_t, _dp, _dp, _dp, _dp, _dp]
        R.ref_align_parallel.restype = C.c_double
        R.ref_align_parallel.argtypes = [C.c_int, _up, _dp, _dp, _dp, _dp, _dp]
        R.ref_apx_align_parallel.restype = C.c_double
        R.ref_apx_align_parallel.argtypes = [_up, _dp, _dp, _dp, _dp, _dp, _dp]
        R.ref_host_threads.restype = C.c_int
        R.ref_ann_create.restype = C.c_void_p
        R.ref_ann_create.argtypes = [_dp, C.c_int]
        R.ref_ann_destroy.argtypes = [C.c_void_p]
        R.ref_ann_ksearch.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_double, _ip, _dp]
        R.ref_ann_stats.argtypes = [C.c_void_p, _lp]
        R.ref_ann_structure.restype = C.c_long
        R.ref_ann_structure.argtypes = [C.c_void_p, _ip, _dp, _ip, _lp]
        R.ref_eigen3.argtypes = [_dp, _dp, _dp]
        R.ref_normals_apx_knn.argtypes = [_dp, C.c_int, C.c_int, _dp, C.c_double, _dp]
        R.ref_M4inv.restype = C.c_int
        R.ref_M4inv.argtypes = [_dp, _dp]
        R.ref_MMult.argtypes = [_dp, _dp, _dp]
        R.ref_transform3_inplace.argtypes = [_dp, _dp, C.c_size_t]
        R.ref_transform3.argtypes = [_dp, _dp, _dp, C.c_size_t]
        R.ref_transform3normal.argtypes = [_dp, _dp, C.c_size_t]
        R.ref_EulerToMatrix4.argtypes = [_dp, _dp, _dp]
        R.ref_Matrix4ToEuler.argtypes = [_dp, _dp, _dp]
        R.ref_QuatToMatrix4.argtypes = [_dp, _dp, _dp]
        R.ref_Matrix4ToQuat.argtypes = [_dp, _dp, _dp]
        R.ref_Dist2.restype = C.c_double
        R.ref_Dist2.argtypes = [_dp, _dp]
        R.ref_newmat_inverse_solve.restype = C.c_int
        R.ref_newmat_inverse_solve.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
        R.ref_get_pt_pairs.restype = C.c_size_t
        R.ref_get_pt_pairs.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_size_t, C.c_int, C.c_double, _ip, _dp, _dp, _dp, _dp]
        R.ref_icp_iterations.restype = C.c_int
        R.ref_icp_iterations.argtypes = [C.c_void_p, _dp, _dp, C.c_size_t, C.c_double, C.c_int, C.c_int, _dp]
        R.ref_point_filter_range.restype = C.c_size_t
        R.ref_point_filter_range.argtypes = [_dp, C.c_size_t, C.c_double, C.c_double, C.POINTER(C.c_ubyte)]
        R.ref_lum_covariance_euler.restype = C.c_int
        R.ref_lum_covariance_euler.argtypes = [C.c_size_t, _dp, _dp, _dp, _dp, _dp, _dp]
        _ref = R
    return _ref


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ref_point_filter_range(xyz, max_dist, min_dist):
    """the reference's own PointFilter (pointfilter.cc, compiled into oracle/_ref) with setRange(max, min): boolean keep-mask"""
    xyz = _c(xyz).reshape(-1, 3)
    keep = np.zeros(len(xyz), np.uint8)
    ref().ref_point_filter_range(_d(xyz), len(xyz), float(max_dist), float(min_dist), keep.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return keep.astype(bool)


class Tree:
    """Oracle pointer kd-tree (KDTreeImpl::create restated, kdTreeImpl.h:82-201)."""

    def __init__(self, xyz, bucket=20):
        self.xyz = _c(xyz).reshape(-1, 3)
        self.M = self.xyz.shape[0]
        self.h = lib().orc_tree_create(_d(self.xyz), self.M, int(bucket))
        if not self.h:
            raise RuntimeError("cannot create kdtree with zero points")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tree_destroy(self.h)
            self.h = None

    def stats(self):
        s = (C.c_long * 3)()
        lib().orc_tree_stats(self.h, s)
        return {"internal": s[0], "leaves": s[1], "depth": s[2]}

    def perm(self):
        out = np.empty(self.M, np.int32)
        lib().orc_tree_perm(self.h, _i(out))
        return out

    def find_closest(self, q, maxdist2, nthreads=1, want_counters=False):
        q = _c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        cnt = (C.c_long * 3)(0, 0, 0)
        lib().orc_find_closest(self.h, _d(q), len(q), float(maxdist2), _i(idx), _d(d2), cnt, int(nthreads))
        if want_counters:
            return idx, d2, (cnt[0], cnt[1], cnt[2])
        return idx, d2

    def find_closest_deferred(self, q, maxdist2, warm, absmax):
        """The GPU kernel's "quick check deferred" stated on this tree (oracle.c): every query with a warm point walks without
        the quick check and is searched again if it accepted thinly.  Returns (idx, d2, second searches)."""
        q = _c(q).reshape(-1, 3)
        warm = np.ascontiguousarray(warm, np.int32)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        redo = (C.c_long * 1)(0)
        lib().orc_find_closest_deferred(self.h, _d(q), len(q), float(maxdist2), _i(warm), float(absmax), _i(idx), _d(d2), redo)
        return idx, d2, int(redo[0])

    def packet_find_closest(self, q, maxdist2, group=64):
        """Analysis only (oracle.c, "Packet traversal study"): `group` consecutive queries walk the tree together;
        returns (idx, d2, (nodes, buckets, bucket points, point tests summed over lanes) touched by the packets)."""
        q = _c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        cnt = (C.c_long * 4)(0, 0, 0, 0)
        if lib().orc_packet_find_closest(self.h, _d(q), len(q), int(group), float(maxdist2), _i(idx), _d(d2), cnt):
            raise RuntimeError("packet study: group must be 1..64 and the tree at most 63 levels deep")
        return idx, d2, (cnt[0], cnt[1], cnt[2], cnt[3])

    def find_closest_along_dir(self, q, dirs, maxdist2):
        q = _c(q).reshape(-1, 3)
        dirs = _c(dirs).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        lib().orc_find_closest_along_dir(self.h, _d(q), _d(dirs), len(q), float(maxdist2), _i(idx), _d(d2))
        return idx, d2

    def get_pt_pairs(self, source_alignxf, xyz_r, normal_r=None, start=0, end=None, mode=0,
                     maxdist2=625.0):
        """SearchTree::getPtPairs (searchTree.cc:92-189).  Returns dict with idx, p1, p2,
        pn, n, sum, centroid_m / centroid_d (un-normalised sums, as the reference leaves them)."""
        xyz_r = _c(xyz_r).reshape(-1, 3)
        end = len(xyz_r) if end is None else end
        n = end - start
        A = _c(source_alignxf).reshape(16)
        nr = _c(normal_r).reshape(-1, 3) if normal_r is not None else None
        idx = np.empty(n, np.int32)
        p1 = np.empty((n, 3)); p2 = np.empty((n, 3)); pn = np.zeros((n, 3))
        s = C.c_double(0.0)
        cm = np.zeros(3); cd = np.zeros(3)
        k = lib().orc_get_pt_pairs(self.h, _d(A), _d(xyz_r), _d(nr), start, end, int(mode), float(maxdist2),
                                   _i(idx), _d(p1), _d(p2), _d(pn), C.byref(s), _d(cm), _d(cd))
        return dict(idx=idx, p1=p1[:k].copy(), p2=p2[:k].copy(), pn=pn[:k].copy(), n=int(k),
                    sum=s.value, centroid_m=cm, centroid_d=cd)


def m4inv(A):
    A = _c(A).reshape(16)
    out = np.empty(16)
    ok = lib().orc_m4inv(_d(A), _d(out))
    return out, bool(ok)


def mmult(A, B):
    A = _c(A).reshape(16); B = _c(B).reshape(16)
    out = np.empty(16)
    lib().orc_mmult(_d(A), _d(B), _d(out))
    return out


def transform_points(alignxf, xyz):
    """in place, Scan::transformReduced (scan.cc:851-875)"""
    A = _c(alignxf).reshape(16)
    assert xyz.dtype == np.float64 and xyz.flags.c_contiguous
    lib().orc_transform_points(_d(A), _d(xyz), xyz.size // 3)


def transform_normals(alignxf, nrm):
    A = _c(alignxf).reshape(16)
    assert nrm.dtype == np.float64 and nrm.flags.c_contiguous
    lib().orc_transform_normals(_d(A), _d(nrm), nrm.size // 3)


class RefTree:
    """The reference's own KDtreeIndexed (compiled from /root/reference)."""

    def __init__(self, xyz, bucket=20):
        self.xyz = _c(xyz).reshape(-1, 3)
        self.h = ref().ref_kdi_create(_d(self.xyz), len(self.xyz), int(bucket))

    def __del__(self):
        if getattr(self, "h", None):
            ref().ref_kdi_destroy(self.h)
            self.h = None

    def find_closest(self, q, maxdist2, nthreads=1):
        q = _c(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        ref().ref_kdi_find_closest(self.h, _d(q), len(q), float(maxdist2), _i(idx), int(nthreads))
        return idx

    def get_pt_pairs(self, source_alignxf, xyz_r, normal_r=None, mode=0, maxdist2=625.0):
        """SearchTree::getPtPairs assembled from the reference's compiled pieces (ref_driver.cc: ref_get_pt_pairs)"""
        q = _c(xyz_r).reshape(-1, 3)
        nr = _c(normal_r).reshape(-1, 3) if normal_r is not None else None
        n = len(q)
        idx = np.empty(n, np.int32)
        p1, p2, pn = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 3))
        sums = np.zeros(7)
        k = ref().ref_get_pt_pairs(self.h, _d(_c(source_alignxf).reshape(16)), _d(q), _d(nr), n, int(mode), float(maxdist2),
                                   _i(idx), _d(p1), _d(p2), _d(pn), _d(sums))
        return dict(n=int(k), idx=idx, p1=p1[:k], p2=p2[:k], pn=pn[:k], sum=sums[0], centroid_m=sums[1:4].copy(),
                    centroid_d=sums[4:7].copy())

    def icp_iterations(self, model_dalignxf, xyz, maxdist2, nthreads, iters):
        """Full OpenMP-branch ICP iterations (ref_driver.cc: ref_icp_iterations) -> (moved xyz, trace [iters][18])"""
        p = _c(xyz).reshape(-1, 3).copy()
        trace = np.zeros((iters, 18))
        ref().ref_icp_iterations(self.h, _d(_c(model_dalignxf).reshape(16)), _d(p), len(p), float(maxdist2), int(nthreads),
                                 int(iters), _d(trace))
        return p, trace

    def find_closest_along_dir(self, q, dirs, maxdist2):
        q = _c(q).reshape(-1, 3); dirs = _c(dirs).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        ref().ref_kdi_find_closest_along_dir(self.h, _d(q), _d(dirs), len(q), float(maxdist2), _i(idx))
        return idx


def ref_align(algo, p1, p2, cm, cd, nrm=None, pose=None):
    """reference Align by -a id; pose = alignxf on entry (used by 7 LUMEULER / 8 LUMQUAT)"""
    p1 = _c(p1); p2 = _c(p2)
    if int(algo) in (3, 4, 5, 7, 8, 9):
        out = np.eye(4).reshape(16).copy() if pose is None else _c(pose).copy()
        err = ref().ref_align_inout(int(algo), len(p1), _d(p1), _d(p2), _d(_c(cm)), _d(_c(cd)), _d(out))
        return out, err
    out = np.empty(16)
    err = ref().ref_align(int(algo), len(p1), _d(p1), _d(p2), _d(_c(nrm)) if nrm is not None else None,
                          _d(_c(cm)), _d(_c(cd)), _d(out))
    return out, err


def ref_align_parallel(algo, n, s, cm, cd, Si):
    T = 8
    n = np.ascontiguousarray(n, np.uint32); assert len(n) == T
    out = np.empty(16)
    err = ref().ref_align_parallel(int(algo), n.ctypes.data_as(_up), _d(_c(s)), _d(_c(cm)), _d(_c(cd)),
                                   _d(_c(Si)), _d(out))
    return out, err


def gen_mt64_uniform(seed, n, lo, hi):
    """std::mt19937_64(seed) + std::uniform_real_distribution<double>(lo, hi), n draws."""
    out = np.empty(n)
    lib().orc_gen_mt64_uniform(int(seed), n, float(lo), float(hi), _d(out))
    return out


def k5_hash(idx):
    idx = np.ascontiguousarray(idx, np.int32)
    return int(lib().orc_k5_hash(_i(idx), len(idx)))


def octree_center(xyz, voxel):
    """Octree-centre reduction (parity unpinned, see oracle.c); returns [cells, 3] in DFS order."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    out = np.empty_like(xyz)
    m = lib().orc_octree_center(_d(xyz), len(xyz), float(voxel), _d(out))
    return out[:m].copy()


def octree_random(xyz, voxel, nrpts, seed=None, want_perm=False):
    """Octree reduction with `-O nrpts` (nrpts >= 1; parity unpinned, see oracle.c): the kept points in DFS order.
    seed: srand(seed) of the C library first (the reference draws std::rand())."""
    xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
    out = np.empty_like(xyz)
    perm = np.empty(len(xyz), np.uint32)
    if seed is not None:
        C.CDLL(None).srand(int(seed))
    m = lib().orc_octree_random(_d(xyz), len(xyz), float(voxel), int(nrpts), _d(out), perm.ctypes.data_as(C.POINTER(C.c_uint32)))
    return (out[:m].copy(), perm) if want_perm else out[:m].copy()


# ---- normals: ANN kd-tree, approximate k-NN, PCA (oracle_normals.c / ref_ann_driver.cc) -------------
class AnnTree:
    """The ANN kd-tree calculateNormalsApxKNN builds (bucket size 1, sliding midpoint).
    which="oracle": restatement in oracle_normals.c; which="ref": the vendored library itself."""

    def __init__(self, xyz, which="oracle"):
        self.xyz = _c(xyz).reshape(-1, 3)
        self.n = len(self.xyz)
        self.L = lib() if which == "oracle" else ref()
        self.p = "orc_ann_" if which == "oracle" else "ref_ann_"
        self.which = which
        self.h = getattr(self.L, self.p + "create")(_d(self.xyz), self.n)
        if not self.h:
            raise RuntimeError("cannot create an ANN tree with zero points")

    def __del__(self):
        if getattr(self, "h", None):
            getattr(self.L, self.p + "destroy")(self.h)
            self.h = None

    def stats(self):
        """depth, leaves, splitting nodes"""
        s = (C.c_long * 6)()
        getattr(self.L, self.p + "stats")(self.h, s)
        return (s[0], s[1], s[2]) if self.which == "oracle" else (s[0], s[1], s[3])

    def structure(self):
        """pre-order (cut_dim per splitting node, cut_val per splitting node, point per leaf)"""
        cd = np.empty(max(self.n, 1), np.int32)
        cv = np.empty(max(self.n, 1), np.float64)
        lp = np.empty(self.n, np.int32)
        nl = C.c_long(0)
        ns = getattr(self.L, self.p + "structure")(self.h, _i(cd), _d(cv), _i(lp), C.byref(nl))
        if ns < 0:
            raise RuntimeError("leaf with more than one point")
        return cd[:ns].copy(), cv[:ns].copy(), lp[:nl.value].copy()

    def nodes(self):
        """oracle only: flat splitting nodes (cut_dim, child[2], (cut_val, lo, hi)), root reference"""
        ns = self.stats()[2]
        cd = np.empty(max(ns, 1), np.int32)
        ch = np.empty((max(ns, 1), 2), np.int32)
        cv = np.empty((max(ns, 1), 3), np.float64)
        root = np.zeros(1, np.int32)
        lib().orc_ann_nodes(self.h, _i(cd), _i(ch), _d(cv), _i(root))
        return cd[:ns], ch[:ns], cv[:ns], int(root[0])

    def ksearch(self, q, k, eps, want_visits=False):
        q = _c(q).reshape(-1, 3)
        idx = np.empty((len(q), k), np.int32)
        dist = np.empty((len(q), k), np.float64)
        if self.which == "oracle":
            vis = (C.c_long * 2)(0, 0)
            if lib().orc_ann_ksearch(self.h, _d(q), len(q), int(k), float(eps), _i(idx), _d(dist), vis):
                raise RuntimeError("Requesting more near neighbors than data points")
            if want_visits:
                return idx, dist, (vis[0], vis[1])
        else:
            if k > self.n:
                raise RuntimeError("Requesting more near neighbors than data points")   # the library abort()s
            ref().ref_ann_ksearch(self.h, _d(q), len(q), int(k), float(eps), _i(idx), _d(dist))
        return idx, dist


def eigen3(A, which="oracle"):
    """newmat EigenValues(SymmetricMatrix, D, U) of a 3x3: (ascending eigenvalues, eigenvectors in columns)."""
    a = _c(A).reshape(9)
    d = np.empty(3)
    u = np.empty(9)
    (lib().orc_eigen3 if which == "oracle" else ref().ref_eigen3)(_d(a), _d(d), _d(u))
    return d, u.reshape(3, 3)


def normals_apx_knn(xyz, k, rPos, eps, which="oracle", want_knn=False):
    """calculateNormalsApxKNN (normals.cc:35-111)."""
    xyz = _c(xyz).reshape(-1, 3)
    rp = _c(rPos)
    out = np.empty_like(xyz)
    if which == "oracle":
        knn = np.empty((len(xyz), k), np.int32) if want_knn else None
        if lib().orc_normals_apx_knn(_d(xyz), len(xyz), int(k), _d(rp), float(eps), _d(out), _i(knn)):
            raise RuntimeError("Requesting more near neighbors than data points")
        return (out, knn) if want_knn else out
    ref().ref_normals_apx_knn(_d(xyz), len(xyz), int(k), _d(rp), float(eps), _d(out))
    return out


def normals_from_knn(xyz, knn, rPos):
    xyz = _c(xyz).reshape(-1, 3)
    knn = np.ascontiguousarray(knn, np.int32)
    rp = _c(rPos)
    out = np.empty_like(xyz)
    lib().orc_normals_from_knn(_d(xyz), len(xyz), knn.shape[1], _i(knn), _d(rp), _d(out))
    return out


# ---- the reference's own globals.icc primitives and newmat inverse (oracle/_ref) ---------------------------
def ref_m4inv_raw(A):
    """-> (Mout, return value of M4inv)"""
    out = np.empty(16)
    rc = ref().ref_M4inv(_d(_c(A).reshape(16)), _d(out))
    return out, rc


def ref_mmult(A, B):
    out = np.empty(16)
    ref().ref_MMult(_d(_c(A).reshape(16)), _d(_c(B).reshape(16)), _d(out))
    return out


def ref_transform3_inplace(A, pts):
    p = _c(pts).reshape(-1, 3).copy()
    ref().ref_transform3_inplace(_d(_c(A).reshape(16)), _d(p), len(p))
    return p


def ref_transform3(A, pts):
    p = _c(pts).reshape(-1, 3)
    out = np.empty_like(p)
    ref().ref_transform3(_d(_c(A).reshape(16)), _d(p), _d(out), len(p))
    return out


def ref_transform3normal(A, nrm):
    p = _c(nrm).reshape(-1, 3).copy()
    ref().ref_transform3normal(_d(_c(A).reshape(16)), _d(p), len(p))
    return p


def ref_euler_to_matrix4(rPos, rPosTheta):
    out = np.empty(16)
    ref().ref_EulerToMatrix4(_d(_c(rPos)), _d(_c(rPosTheta)), _d(out))
    return out


def ref_matrix4_to_euler(A):
    th, pos = np.empty(3), np.empty(3)
    ref().ref_Matrix4ToEuler(_d(_c(A).reshape(16)), _d(th), _d(pos))
    return th, pos


def ref_quat_to_matrix4(quat, t):
    out = np.empty(16)
    ref().ref_QuatToMatrix4(_d(_c(quat)), _d(_c(t)), _d(out))
    return out


def ref_matrix4_to_quat(A):
    q, t = np.empty(4), np.empty(3)
    ref().ref_Matrix4ToQuat(_d(_c(A).reshape(16)), _d(q), _d(t))
    return q, t


def ref_newmat_inverse_solve(A, b=None):
    """newmat's A.i() and A.i() * b -> (Ainv, x)"""
    A = _c(A); n = A.shape[0]
    Ai = np.empty((n, n)); x = np.empty(n) if b is not None else None
    rc = ref().ref_newmat_inverse_solve(n, _d(A), _d(_c(b)) if b is not None else None, _d(Ai), _d(x))
    if rc:
        raise RuntimeError("newmat: singular matrix")
    return Ai, x


def ref_lum_covariance_euler(p1, p2):
    """lum6DEuler::covarianceEuler's arithmetic on an explicit pair list with the reference's newmat -> (C, CD, ss, D)"""
    p1, p2 = _c(p1).reshape(-1, 3), _c(p2).reshape(-1, 3)
    Cm, CD, D = np.empty(36), np.empty(6), np.empty(6)
    ss = C.c_double(0.0)
    ref().ref_lum_covariance_euler(len(p1), _d(p1), _d(p2), _d(Cm), _d(CD), C.byref(ss), _d(D))
    return Cm.reshape(6, 6), CD, ss.value, D
