"""Host-side mirror of the reference's slam6d interface for the ICP hot path.

Same class / method names and argument meaning as the reference (file:line cited per
item); the bodies only marshal numpy arrays into the C ABI of lib3dtk_hip.so.  Nothing
here computes correspondences or sums on the CPU.
"""
import ctypes as C
import math

import numpy as np

from . import _capi as capi
_dp_type = lambda: C.POINTER(C.c_double)()
from ._capi import (lib, check, dptr, iptr, f64, PairSums, IcpParams, IcpResult, TreeInfo,
                    ALGO_QUAT, ALGO_SVD, ALGO_APX, ALGO_NAPX, WANT_APX, WANT_NAPX, WANT_LUM)


# ---------------------------------------------------------------------------------------
# include/slam6d/globals.icc helpers
# ---------------------------------------------------------------------------------------
def M4identity():
    return np.eye(4).reshape(16).copy()


def M4inv(M):
    """globals.icc:762-785 (bit-exact host restatement inside the library)."""
    M = f64(M, 16)
    out = np.empty(16)
    lib().tdtk_host_m4inv(dptr(M), dptr(out))
    return out


def MMult(M1, M2):
    """globals.icc:298-328, column-major M1*M2."""
    M1 = f64(M1, 16); M2 = f64(M2, 16)
    out = np.empty(16)
    lib().tdtk_host_mmult(dptr(M1), dptr(M2), dptr(out))
    return out


def EulerToMatrix4(rPos, rPosTheta):
    """globals.icc:501-531."""
    sx, cx = math.sin(rPosTheta[0]), math.cos(rPosTheta[0])
    sy, cy = math.sin(rPosTheta[1]), math.cos(rPosTheta[1])
    sz, cz = math.sin(rPosTheta[2]), math.cos(rPosTheta[2])
    a = np.zeros(16)
    a[0] = cy * cz
    a[1] = sx * sy * cz + cx * sz
    a[2] = -cx * sy * cz + sx * sz
    a[4] = -cy * sz
    a[5] = -sx * sy * sz + cx * cz
    a[6] = cx * sy * sz + sx * cz
    a[8] = sy
    a[9] = -sx * cy
    a[10] = cx * cy
    a[12], a[13], a[14] = rPos[0], rPos[1], rPos[2]
    a[15] = 1
    return a


def Matrix4ToEuler(alignxf):
    """globals.icc:541-576.  Returns (rPosTheta, rPos)."""
    th = [0.0, 0.0, 0.0]
    if alignxf[0] > 0.0:
        th[1] = math.asin(alignxf[8])
    else:
        th[1] = math.pi - math.asin(alignxf[8])
    Cc = math.cos(th[1])
    if abs(Cc) > 0.005:
        th[0] = math.atan2(-alignxf[9] / Cc, alignxf[10] / Cc)
        th[2] = math.atan2(-alignxf[4] / Cc, alignxf[0] / Cc)
    else:
        th[0] = 0.0
        th[2] = math.atan2(alignxf[1], alignxf[5])
    return np.array(th), np.array([alignxf[12], alignxf[13], alignxf[14]])


def QuatToMatrix4(quat, t=None):
    """globals.icc:988-1022 (the library's host restatement, like M4inv / MMult: the C++ glue calls the same code)"""
    q = f64(quat, 4)
    tt = f64(t, 3) if t is not None else np.zeros(3)
    out = np.empty(16)
    lib().tdtk_host_quat_to_matrix4(dptr(q), dptr(tt), dptr(out))
    return out


def Matrix4ToQuat(mat):
    """globals.icc:1032-1075 -> (quat[4] normalised, t[3]) (the library's host restatement)"""
    m = f64(mat, 16)
    q, t = np.empty(4), np.empty(3)
    lib().tdtk_host_matrix4_to_quat(dptr(m), dptr(q), dptr(t))
    return q, t


def transform_many(scans, A1, A2=None, type="LUM"):
    """Scan::transform for many scans at once: scan i is moved by A1[i], then A2[i] (the two steps of
    transformToEuler / transformToQuat).  Matrices and frames are updated per scan exactly as
    Scan::transform does (the first step is the INVALID-type one when A2 is given); the resident point
    sets move in ONE launch (tdtk_scans_transform2); non-resident scans queue the matrices."""
    res = [i for i, s in enumerate(scans) if s._h is not None]
    if res:
        hs = (C.c_void_p * len(res))(*[scans[i]._h for i in res])
        a1 = np.ascontiguousarray(np.stack([f64(A1[i], 16) for i in res]))
        a2 = np.ascontiguousarray(np.stack([f64(A2[i], 16) for i in res])) if A2 is not None else None
        check(lib().tdtk_scans_transform2(len(res), hs, dptr(a1), dptr(a2) if a2 is not None else None))
    for i, s in enumerate(scans):
        a = f64(A1[i], 16).copy()
        if s._h is None:
            s._queue.append(a)
        s._transformMatrix(a)
        if A2 is not None:
            b = f64(A2[i], 16).copy()
            if s._h is None:
                s._queue.append(b)
            s._transformMatrix(b)
        s.frames.append((s.transMat.copy(), type))


def prepare_scans(scans, trees=True, threads=4, normals=False):
    """Upload the scans and (trees=True) build their search trees on `threads` host threads, each with its
    own HIP stream.  A tree build is mostly a serial fp64 chain that occupies three wavefronts, so several
    of them run side by side on one GPU at almost no cost to each other.  normals=True first runs
    Scan::calcNormals on every scan that has none (its ANN-tree build is a few hundred short launches: the same
    argument holds)."""
    from concurrent.futures import ThreadPoolExecutor

    def prep(s):
        if normals and s._local_n is None and s._h is None:
            s.calcNormals()
        _ = s.handle
        if trees:
            s.getSearchTree()
    scans = list(scans)
    if threads <= 1 or len(scans) <= 1:
        for s in scans:
            prep(s)
        return
    with ThreadPoolExecutor(threads) as pool:
        for f in [pool.submit(prep, s) for s in scans]:
            f.result()


def host_tree_layout(xyz, bucketSize=20):
    """Host tree builder only (no device): leaf-order permutation + stats."""
    xyz = f64(xyz).reshape(-1, 3)
    perm = np.empty(len(xyz), np.int32)
    st = (C.c_uint64 * 4)()
    check(lib().tdtk_host_tree_layout(dptr(xyz), len(xyz), int(bucketSize), iptr(perm), st))
    return perm, dict(internal=st[0], leaves=st[1], depth=st[2], max_leaf=st[3])


def _sums_dict(s):
    return dict(n_queries=int(s.n_queries), n=int(s.n), sum=s.sum,
                centroid_m=np.array(s.centroid_m), centroid_d=np.array(s.centroid_d),
                Si=np.array(s.Si), apx_A=np.array(s.apx_A), apx_B=np.array(s.apx_B),
                napx_A=np.array(s.napx_A), napx_B=np.array(s.napx_B), napx_sum=s.napx_sum,
                lum=np.array(s.lum), lum_sumd2=s.lum_sumd2,
                gapx_MkMkt=np.array(s.gapx_MkMkt).reshape(3, 3), gapx_DkDkt=np.array(s.gapx_DkDkt).reshape(3, 3),
                gapx_MkDkt=np.array(s.gapx_MkDkt).reshape(3, 3), gapx_DkMkt=np.array(s.gapx_DkMkt).reshape(3, 3),
                gapx_Ak1=np.array(s.gapx_Ak1), gapx_Ak2=np.array(s.gapx_Ak2))


# ---------------------------------------------------------------------------------------
# KDtree : SearchTree   (include/slam6d/kd.h, src/slam6d/kd.cc, searchTree.cc)
# ---------------------------------------------------------------------------------------
class KDtree:
    """KDtree(pts, n, bucketSize) (kd.cc:46-49).  pts is copied and uploaded."""

    def __init__(self, pts, bucketSize=20, device=0):
        pts = f64(pts).reshape(-1, 3)
        if len(pts) == 0:
            raise RuntimeError("cannot create kdtree with zero points")  # kdTreeImpl.h:86-88
        self.n = len(pts)
        self.device = device
        h = C.c_void_p()
        check(lib().tdtk_tree_create(dptr(pts), len(pts), int(bucketSize), int(device), C.byref(h)))
        self._h = h

    @classmethod
    def from_scan(cls, scan_handle, n, bucketSize=20, device=0):
        """the tree over a resident scan's original points, built device to device (tdtk_tree_create_from_scan)"""
        self = cls.__new__(cls)
        self.n = n
        self.device = device
        h = C.c_void_p()
        check(lib().tdtk_tree_create_from_scan(scan_handle, int(bucketSize), C.byref(h)))
        self._h = h
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().tdtk_tree_destroy(h)
            except Exception:      # interpreter shutdown: module globals are already gone
                pass
            self._h = None

    def info(self):
        ti = TreeInfo()
        check(lib().tdtk_tree_get_info(self._h, C.byref(ti)))
        return {k: getattr(ti, k) for k, _ in TreeInfo._fields_}

    def verify(self):
        """Compare the resident (device-built) tree with the host builder's, bit for bit."""
        mm = (C.c_uint64 * 4)()
        check(lib().tdtk_tree_verify(self._h, mm))
        return [int(v) for v in mm]

    def FindClosest(self, p, maxdist2, threadNum=0):
        """kd.cc:78-87.  Returns the index of the closest point or None (reference: NULL)."""
        idx, _ = self.FindClosestBatch(np.asarray(p, dtype=np.float64).reshape(1, 3), maxdist2)
        return None if idx[0] < 0 else int(idx[0])

    def FindClosestBatch(self, q, maxdist2):
        q = f64(q).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        check(lib().tdtk_find_closest(self._h, dptr(q), len(q), float(maxdist2), iptr(idx), dptr(d2)))
        return idx, d2

    def FindClosestAlongDir(self, p, direction, maxdist2, threadNum=0):
        """kd.cc:89-100."""
        idx, _ = self.FindClosestAlongDirBatch(np.asarray(p, float).reshape(1, 3),
                                               np.asarray(direction, float).reshape(1, 3), maxdist2)
        return None if idx[0] < 0 else int(idx[0])

    def FindClosestAlongDirBatch(self, q, dirs, maxdist2):
        q = f64(q).reshape(-1, 3); dirs = f64(dirs).reshape(-1, 3)
        idx = np.empty(len(q), np.int32)
        d2 = np.empty(len(q), np.float64)
        check(lib().tdtk_find_closest_along_dir(self._h, dptr(q), dptr(dirs), len(q), float(maxdist2),
                                                iptr(idx), dptr(d2)))
        return idx, d2

    def count_visits(self, q, maxdist2):
        q = f64(q).reshape(-1, 3)
        cnt = (C.c_uint64 * 3)()
        check(lib().tdtk_count_visits(self._h, dptr(q), len(q), float(maxdist2), cnt))
        return int(cnt[0]), int(cnt[1]), int(cnt[2])

    def getPtPairs(self, source_alignxf, xyz_r, normal_r=None, startindex=0, endindex=None,
                   thread_num=0, rnd=0, max_dist_match2=625.0, pairing_mode=0, want=0, lum_D=None,
                   want_pairs=True):
        """SearchTree::getPtPairs, DataXYZ overload (searchTree.cc:92-189).  Returns a dict with
        idx, the compact pair list p1/p2/pn and the merged sums (see tdtk_pair_sums)."""
        xyz_r = f64(xyz_r).reshape(-1, 3)
        endindex = len(xyz_r) if endindex is None else endindex
        n = endindex - startindex
        A = f64(source_alignxf, 16)
        nr = f64(normal_r).reshape(-1, 3) if normal_r is not None else None
        idx = np.empty(n, np.int32)
        p1 = np.empty((n, 3)) if want_pairs else None
        p2 = np.empty((n, 3)) if want_pairs else None
        pn = np.empty((n, 3)) if want_pairs else None
        s = PairSums()
        D = f64(lum_D, 6) if lum_D is not None else None
        check(lib().tdtk_get_pt_pairs(self._h, dptr(A), dptr(xyz_r), dptr(nr), startindex, endindex,
                                      int(rnd), int(pairing_mode), float(max_dist_match2), int(want),
                                      dptr(D), iptr(idx), dptr(p1), dptr(p2), dptr(pn), C.byref(s)))
        out = _sums_dict(s)
        out["idx"] = idx
        out["_raw"] = s
        if want_pairs:
            k = out["n"]
            out["p1"], out["p2"], out["pn"] = p1[:k], p2[:k], pn[:k]
        return out


# ---------------------------------------------------------------------------------------
# Scan  (include/slam6d/scan.h, src/slam6d/scan.cc, basicScan.cc) -- the 6 functions of the path
# ---------------------------------------------------------------------------------------
class Scan:
    """Resident scan: "xyz reduced original" (host, tree input), "xyz reduced" (device),
    transMatOrg / transMat / dalignxf (basicScan.cc:175-197, scan.cc:878-898).

    The device copy of the points and the search tree are materialised on first use, so a rank
    that only needs a scan's pose (multi-GPU graph-SLAM) never uploads it; transforms issued
    before materialisation are queued and replayed in order (the motion is incremental and in
    place in the reference, SURVEY N-a)."""

    allScans = []

    def __init__(self, rPos, rPosTheta, points, normals=None, bucketSize=20, device=0):
        self.device = device
        self.bucketSize = bucketSize
        self.rPos = np.array(rPos, dtype=np.float64)
        self.rPosTheta = np.array(rPosTheta, dtype=np.float64)
        self.transMatOrg = EulerToMatrix4(self.rPos, self.rPosTheta)
        self.transMat = M4identity()
        self.dalignxf = M4identity()
        self._transformMatrix(self.transMatOrg)     # basicScan.cc:188
        self.dalignxf = M4identity()                # basicScan.cc:192
        self._local = f64(points).reshape(-1, 3)
        self._local_n = f64(normals).reshape(-1, 3) if normals is not None else None
        self.n = len(self._local)
        self._h = None
        self._queue = []
        self._orig = None
        self.kd = None
        self.frames = []  # (transMat copy, type)

    def release(self):
        """Give the device arrays (and the search tree) back now; the scan can go resident again later."""
        h = getattr(self, "_h", None)
        if h:
            try:
                lib().tdtk_scan_destroy(h)
            except Exception:      # interpreter shutdown
                pass
            self._h = None
        self.kd = None

    def __del__(self):
        self.release()

    K_NEIGHBOURS = 10     # scan.cc:418

    def calcNormals(self):
        """Scan::calcNormals (scan.cc:398-427): normals of the scan-local points from their K_NEIGHBOURS
        approximate neighbours (eps = 1.0), oriented with get_rPos().  Before the scan goes resident they become its
        "normal" array (transformed with the points afterwards, basicScan.cc:730-737); on a resident scan the
        normals are recomputed from the points where they currently are."""
        if self._h is None:
            self._local_n = calculateNormalsApxKNN(self._local, self.K_NEIGHBOURS, self.rPos, 1.0, self.device)
        else:
            check(lib().tdtk_scan_calc_normals(self._h, self.K_NEIGHBOURS, dptr(self.rPos), 1.0))
            self._has_device_normals = True
        return self

    # accessors named as in scan.h
    def get_transMat(self): return self.transMat
    def get_transMatOrg(self): return self.transMatOrg
    def getDAlign(self): return self.dalignxf
    def get_rPos(self): return self.rPos
    def get_rPosTheta(self): return self.rPosTheta

    def _upload(self):
        h = C.c_void_p()
        check(lib().tdtk_scan_create(dptr(self._local), dptr(self._local_n), self.n, int(self.device),
                                     C.byref(h)))
        # calcReducedOnDemandPrivate (basicScan.cc:730-737): transformReduced(transMatOrg), then
        # copyReducedToOriginal: from here on the library keeps the original through any later move
        check(lib().tdtk_scan_transform(h, dptr(self.transMatOrg)))
        check(lib().tdtk_scan_mark_original(h))
        return h

    @property
    def handle(self):
        """device-resident "xyz reduced" (materialised on first use)"""
        if self._h is None:
            self._h = self._upload()
            for A in self._queue:
                check(lib().tdtk_scan_transform(self._h, dptr(A)))
            self._queue = []
        return self._h

    def _download(self, h):
        out = np.empty((self.n, 3))
        check(lib().tdtk_scan_download(h, dptr(out), None))
        return out

    @property
    def xyz_reduced_original(self):
        if self._orig is None:
            if self._h is not None or not self._queue:
                out = np.empty((self.n, 3))
                check(lib().tdtk_scan_download_original(self.handle, dptr(out)))
                self._orig = out
            else:   # only the tree is needed here: do not keep a moved copy resident
                h = self._upload()
                self._orig = self._download(h)
                lib().tdtk_scan_destroy(h)
        return self._orig

    def get_xyz_reduced(self):
        return self._download(self.handle)

    def get_normal_reduced(self):
        """"normal reduced" of the resident scan (None when the scan has no normals)"""
        h = self.handle
        if self._local_n is None and not getattr(self, "_has_device_normals", False):
            return None
        xyz, nrm = np.empty((self.n, 3)), np.empty((self.n, 3))
        check(lib().tdtk_scan_download(h, dptr(xyz), dptr(nrm)))
        return nrm

    def getSearchTree(self):
        """scan.cc:268-306 -> basicScan.cc:702-728: lazily built over "xyz reduced original"."""
        if self.kd is None:
            if self._h is None and self._queue:     # never resident, already moved: go through a temporary upload
                self.kd = KDtree(self.xyz_reduced_original, self.bucketSize, self.device)
            else:                                   # device to device, from the (saved) original points
                self.kd = KDtree.from_scan(self.handle, self.n, self.bucketSize, self.device)
        return self.kd

    def _transformMatrix(self, alignxf):
        """Scan::transformMatrix (scan.cc:878-898)."""
        self.transMat = MMult(alignxf, self.transMat)
        self.rPosTheta, self.rPos = Matrix4ToEuler(self.transMat)
        self.dalignxf = MMult(alignxf, self.dalignxf)

    def transform(self, alignxf, type="ICP", islum=0):
        """Scan::transform (scan.cc:918-1009): transformReduced on the device + matrices."""
        alignxf = f64(alignxf, 16).copy()
        if self._h is not None:
            check(lib().tdtk_scan_transform(self._h, dptr(alignxf)))
        else:
            self._queue.append(alignxf)
        self._transformMatrix(alignxf)
        self._addFrames(type, islum)

    def addFrame(self, type):
        """Scan::addFrame: the scan's CURRENT transMat with the given type"""
        self.frames.append((self.transMat.copy(), type))

    def _addFrames(self, type, islum):
        """frame bookkeeping of Scan::transform (scan.cc:945-1008).  islum 0 writes a frame to EVERY scan of
        Scan.allScans (this scan: `type`; the scans before it: ICPINACTIVE; the scans after it: INVALID -- or
        ICPINACTIVE throughout when this scan is the first), so that all .frames files keep the same length, which
        `show` animations rely on; 1: this scan only; 2: this scan and scan 0, INVALID for the scans after it.
        A scan that is not registered in Scan.allScans (stand-alone use) only ever writes its own frame."""
        if type == "INVALID" or islum == -1:
            return
        scans = Scan.allScans
        if islum == 1 or not any(sc is self for sc in scans):
            self.addFrame(type)
            return
        found = 0
        if islum == 0:
            for i, sc in enumerate(scans):
                if sc is self:
                    found = i
                    sc.addFrame(type)
                else:
                    sc.addFrame("ICPINACTIVE" if found == 0 else "INVALID")
        elif islum == 2:
            for i, sc in enumerate(scans):
                if sc is self:
                    found = i
                    self.addFrame(type)
                    scans[0].addFrame(type)
                    continue
                if found != 0:
                    sc.addFrame("INVALID")
        else:
            raise ValueError("invalid point transformation mode")

    def transformToEuler(self, rP, rPT, type="LUM", islum=1):
        """scan.cc:1061-1083."""
        tinv = M4inv(self.transMat)
        self.transform(tinv, "INVALID")
        self.transform(EulerToMatrix4(rP, rPT), type, islum)

    def get_rPosQuat(self):
        """rQuat, kept equal to Matrix4ToQuat(transMat) by Scan::transformMatrix (scan.cc:886)."""
        return Matrix4ToQuat(self.transMat)[0]

    def transformToQuat(self, rP, rPQ, type="LUM", islum=1):
        """scan.cc:1093-1104."""
        tinv = M4inv(self.transMat)
        self.transform(tinv, "INVALID")
        self.transform(QuatToMatrix4(rPQ, rP), type, islum)

    def mergeCoordinatesWithRoboterPosition(self, prevScan):
        """scan.cc:826-833 (pose extrapolation)."""
        tempMat = M4inv(prevScan.get_transMatOrg())
        deltaMat = MMult(prevScan.get_transMat(), tempMat)
        self.transform(deltaMat, "INVALID")

    @staticmethod
    def getPtPairs(Source, Target, thread_num=0, rnd=0, max_dist_match2=625.0, pairing_mode=0,
                   want=0, lum_D=None, want_idx=False):
        """Scan::getPtPairs (scan.cc:1220-1260): whole-scan pass, centroids normalised."""
        if rnd > 1:
            raise capi.TdtkError(-5, "rnd > 1 is not supported on the device path")
        s = PairSums()
        idx = np.empty(Target.n, np.int32) if want_idx else None
        D = f64(lum_D, 6) if lum_D is not None else None
        tree = Source.getSearchTree()
        check(lib().tdtk_scan_pairs(tree._h, dptr(Source.dalignxf), Target.handle, int(pairing_mode),
                                    float(max_dist_match2), int(want), dptr(D), iptr(idx), C.byref(s)))
        out = _sums_dict(s)
        out["_raw"] = s
        if want_idx:
            out["idx"] = idx
        return out


class MetaScan:
    """MetaScan (src/slam6d/metaScan.cc:27-107) + KDtreeMetaManaged (src/slam6d/kdMeta.cc:34-134):
    one search tree over the CURRENT "xyz reduced" points of several scans, addressed in
    concatenation order (prepareTempIndices, kdMeta.cc:60-79), bucket size of the first scan
    (kdMeta.cc:45-46).  Its dalignxf stays identity (Scan base ctor, scan.cc:193)."""

    def __init__(self, scans, nns_method=0):
        self.m_scans = list(scans)
        self.dalignxf = M4identity()
        self.transMat = M4identity()
        self.kd = None
        self.device = self.m_scans[0].device if self.m_scans else 0

    def size(self): return len(self.m_scans)
    def getScan(self, i): return self.m_scans[i]
    def getDAlign(self): return self.dalignxf

    def transform(self, alignxf, type="ICP", islum=0):
        """Scan::transform on a MetaScan (scan.cc:920-926): every member scan is moved (without frames of its own:
        islum -1), then the MetaScan's own matrices; the frames written for `islum` go to the member scans
        (in_meta, scan.cc:962-975)."""
        alignxf = f64(alignxf, 16).copy()
        members = [m for m in self.m_scans if m._h is not None]
        if members:                                           # one launch for all resident members
            hs = (C.c_void_p * len(members))(*[m._h for m in members])
            A = np.ascontiguousarray(np.tile(alignxf, (len(members), 1)))
            check(lib().tdtk_scans_transform2(len(members), hs, dptr(A), None))
        for m in self.m_scans:
            if m._h is None:
                m._queue.append(alignxf)
            m._transformMatrix(alignxf)
        self.transMat = MMult(alignxf, self.transMat)
        self.dalignxf = MMult(alignxf, self.dalignxf)
        if type != "INVALID" and islum == 0:
            scans = Scan.allScans
            if not any(any(sc is m for m in self.m_scans) for sc in scans):
                for m in self.m_scans:
                    m.addFrame(type)
                return
            found = 0
            for i, sc in enumerate(scans):
                if any(sc is m for m in self.m_scans):
                    found = i
                    sc.addFrame(type)
                else:
                    sc.addFrame("ICPINACTIVE" if found == 0 else "INVALID")

    def getSearchTree(self):
        if self.kd is None:   # device to device: the scans' current points never visit the host
            hs = (C.c_void_p * len(self.m_scans))(*[s.handle for s in self.m_scans])
            kd = KDtree.__new__(KDtree)
            kd.n = sum(s.n for s in self.m_scans)
            kd.device = self.device
            h = C.c_void_p()
            check(lib().tdtk_tree_create_from_scans(hs, len(self.m_scans), int(self.m_scans[0].bucketSize), C.byref(h)))
            kd._h = h
            self.kd = kd
        return self.kd


def read_uos(path, range_max=0.0, range_min=0.0):
    """uos ASCII reader + PointFilter range (-m / -M); see tdtk_io_read_uos."""
    p = _dp_type()
    n = C.c_size_t(0)
    check(lib().tdtk_io_read_uos(str(path).encode(), float(range_max), float(range_min), C.byref(p), C.byref(n)))
    try:
        out = np.ctypeslib.as_array(p, shape=(max(1, n.value) * 3,))[:n.value * 3].copy().reshape(-1, 3)
    finally:
        lib().tdtk_io_free(p)
    return out


def calcReducedPoints(xyz, voxelSize, device=0, nrpts=0, rm_scatter=False, seed=None):
    """Scan::calcReducedPoints (scan.cc:560-603): octree reduction `-r voxelSize [-O nrpts]` on the device.  nrpts == 0:
    the centres of the occupied leaves (tdtk_reduce_octree); nrpts == 1 / N > 1: one / up to N random points per leaf
    (BOctTree::GetOctTreeRandom; the draws are the C library's rand() in the reference's order -- `seed` calls srand
    first).  voxelSize <= 0 keeps every point, as scan.cc:497-558 does."""
    xyz = f64(xyz).reshape(-1, 3)
    if voxelSize <= 0 or len(xyz) == 0:
        return xyz.copy()
    out = np.empty_like(xyz)
    m = C.c_size_t(0)
    if nrpts == 0 and not rm_scatter:
        check(lib().tdtk_reduce_octree(dptr(xyz), len(xyz), float(voxelSize), int(device), dptr(out), C.byref(m)))
    else:
        if seed is not None:
            C.CDLL(None).srand(int(seed))
        check(lib().tdtk_reduce_octree_nrpts(dptr(xyz), len(xyz), float(voxelSize), int(nrpts), int(bool(rm_scatter)), int(device),
                                             dptr(out), C.byref(m)))
    return out[:m.value].copy()


def calculateNormalsApxKNN(points, k, rPos, eps, device=0, want_knn=False):
    """normals.cc:35-111 (calculateNormalsApxKNN(normals, points, k, rPos, eps)) on the device: returns the
    [n][3] normals (and, with want_knn, the [n][k] neighbour lists of the ANN search, nearest first)."""
    xyz = f64(points).reshape(-1, 3)
    out = np.empty_like(xyz)
    knn = np.empty((len(xyz), int(k)), np.int32) if want_knn else None
    check(lib().tdtk_normals_apx_knn(dptr(xyz), len(xyz), int(k), dptr(f64(rPos)), float(eps), int(device),
                                     dptr(out), iptr(knn)))
    return (out, knn) if want_knn else out


def read_pose(path):
    rP = np.empty(3); rT = np.empty(3)
    check(lib().tdtk_io_read_pose(str(path).encode(), dptr(rP), dptr(rT)))
    return rP, rT


ALGO_TYPE = {"INVALID": 0, "ICP": 1, "ICPINACTIVE": 2, "LUM": 3, "ELCH": 4}   # scan.h:126


def openDirectory(path, start=0, end=-1, range_max=0.0, range_min=0.0, bucketSize=20, device=0, red=-1.0,
                  use_normals=False):
    """Scan::openDirectory for the uos format (src/slam6d/scan.cc / basicScan.cc:39-122):
    scanNNN.3d + scanNNN.pose, NNN = start..end; red = -r voxel size (setReductionParameter).
    use_normals = what slam6D.cc:688-692 asks for with -z / --normal_shoot-simple (PointType::USE_NORMAL): every
    scan gets Scan::calcNormals.  Not together with red > 0: the reference's centre-mode octree reduction does not
    carry normals (it copies POINTDIM values out of a 3-element centre, Boctree.h:928-941)."""
    if use_normals and red > 0:
        raise ValueError("normals are not carried through the octree reduction (see DESIGN.md section 9)")
    import os
    scans = []
    i = start
    while end < 0 or i <= end:
        f3d = os.path.join(path, "scan%03d.3d" % i)
        fpose = os.path.join(path, "scan%03d.pose" % i)
        if not (os.path.exists(f3d) and os.path.exists(fpose)):
            break
        rP, rT = read_pose(fpose)
        pts = read_uos(f3d, range_max, range_min)
        if red > 0:
            pts = calcReducedPoints(pts, red, device)
        s = Scan(rP, rT, pts, bucketSize=bucketSize, device=device)
        if use_normals:
            s.calcNormals()
        s.identifier = "%03d" % i
        s.path = path
        scans.append(s)
        i += 1
    Scan.allScans = scans          # the static Scan::allScans the frame bookkeeping of Scan::transform walks
    return scans


def closeDirectory():
    """Scan::closeDirectory (scan.cc:159-165, BasicScan::closeDirectory basicScan.cc:124-137): release every scan of the
    registry that openDirectory filled (device arrays, trees) and clear it.  Like the reference's static
    Scan::allScans the registry keeps the scans alive until this is called or the next openDirectory replaces it."""
    for s in Scan.allScans:
        try:
            s.release()
        except Exception:
            pass
    Scan.allScans = []


def saveFrames(scan, path=None, append=False):
    """BasicScan::saveFrames (basicScan.cc:902-917): scanNNN.frames"""
    import os
    fn = path or os.path.join(scan.path, "scan%s.frames" % scan.identifier)
    mats = np.ascontiguousarray(np.array([m for m, _ in scan.frames], dtype=np.float64).reshape(-1, 16))
    types = (C.c_int * len(scan.frames))(*[ALGO_TYPE.get(t, 0) for _, t in scan.frames])
    check(lib().tdtk_io_write_frames(str(fn).encode(), dptr(mats) if len(scan.frames) else None, types,
                                     len(scan.frames), 1 if append else 0))
    return fn


# ---------------------------------------------------------------------------------------
# icp6Dminimizer family (include/slam6d/icp6Dminimizer.h:31-88)
# ---------------------------------------------------------------------------------------
class icp6Dminimizer:
    algo = 0

    def __init__(self, quiet=False):
        self.quiet = quiet

    def getAlgorithmID(self):
        return self.algo

    want = 0    # accumulator block the minimizer needs on top of the base sums

    def Align_Parallel(self, sums, pose=None):
        """Align_Parallel with the merged sums (slot 0 semantics).  Returns (rms, alignxf).
        pose = the current scan's transMat, read by LUMEULER / LUMQUAT (icp6D.cc:237-241)."""
        raw = sums["_raw"] if isinstance(sums, dict) else sums
        out = np.eye(4).reshape(16).copy() if pose is None else f64(pose).copy()
        rms = C.c_double(0.0)
        rc = lib().tdtk_align(int(self.algo), C.byref(raw), dptr(out), C.byref(rms))
        if rc == -4:       # "Couldn't find transform." -> the reference returns -1.0
            return -1.0, out
        check(rc)
        return rms.value, out


class icp6D_QUAT(icp6Dminimizer):   # src/slam6d/icp6Dquat.cc, -a 1
    algo = ALGO_QUAT


class icp6D_SVD(icp6Dminimizer):    # src/slam6d/icp6Dsvd.cc, -a 2
    algo = ALGO_SVD


class icp6D_APX(icp6Dminimizer):    # src/slam6d/icp6Dapx.cc, -a 6
    algo = ALGO_APX
    want = WANT_APX


class icp6D_NAPX(icp6Dminimizer):   # src/slam6d/icp6Dnapx.cc, -a 10
    algo = ALGO_NAPX
    want = WANT_NAPX


# The minimizers the reference only has as serial Align.  getAlgorithmID() here is the -a value
# (the reference's icp6D_LUMEULER answers 3, like ORTHO: include/slam6d/icp6Dlumeuler.h:33).
class icp6D_ORTHO(icp6Dminimizer):       # src/slam6d/icp6Dortho.cc, -a 3
    algo = capi.ALGO_ORTHO
    want = capi.WANT_MOM2


class icp6D_DUAL(icp6Dminimizer):        # src/slam6d/icp6Ddual.cc, -a 4
    algo = capi.ALGO_DUAL
    want = capi.WANT_MOM2


class icp6D_HELIX(icp6Dminimizer):       # src/slam6d/icp6Dhelix.cc, -a 5
    algo = capi.ALGO_HELIX
    want = capi.WANT_MOM2


class icp6D_LUMEULER(icp6Dminimizer):    # src/slam6d/icp6Dlumeuler.cc, -a 7
    algo = capi.ALGO_LUMEULER
    want = capi.WANT_MOM2


class icp6D_LUMQUAT(icp6Dminimizer):     # src/slam6d/icp6Dlumquat.cc, -a 8
    algo = capi.ALGO_LUMQUAT
    want = capi.WANT_MOM2


class icp6D_QUAT_SCALE(icp6Dminimizer):  # src/slam6d/icp6Dquatscale.cc, -a 9
    algo = capi.ALGO_QUAT_SCALE
    want = capi.WANT_MOM2


# ---------------------------------------------------------------------------------------
# icp6D (include/slam6d/icp6D.h, src/slam6d/icp6D.cc)
# ---------------------------------------------------------------------------------------
class icp6D:
    prefetch_depth = 3      # scans prepared ahead of the one being matched in doICP (own host threads / streams)

    def __init__(self, my_icp6Dminimizer, max_dist_match=25.0, max_num_iterations=50, quiet=False,
                 meta=False, rnd=1, eP=True, anim=-1, epsilonICP=0.0000001, nns_method=0,
                 max_num_metascans=-1):
        if max_dist_match < 0.0:
            raise ValueError("ERROR [ICP6D]: first parameter (max_dist_match) has to be >= 0,")
        if max_num_iterations < 0:
            raise ValueError("ERROR [ICP6D]: second parameter (max_num_iterations)has to be >= 0.")
        self.rnd = rnd
        self.my_icp6Dminimizer = my_icp6Dminimizer
        self.max_dist_match2 = max_dist_match * max_dist_match
        self.max_num_iterations = max_num_iterations
        self.quiet = quiet
        self.meta = meta
        self.max_num_metascans = max_num_metascans
        self.eP = eP
        self.epsilonICP = epsilonICP
        self.nr_pointPair = 0
        self.last = None

    def match(self, PreviousScan, CurrentScan, pairing_mode=0):
        """icp6D::match (icp6D.cc:104-285).  Returns the number of iterations done."""
        if isinstance(CurrentScan, MetaScan):
            return self._match_meta_data(PreviousScan, CurrentScan)
        if self.rnd > 1 and getattr(self, "stepped_rnd", False):
            return self._match_stepped(PreviousScan, CurrentScan, pairing_mode)
        CurrentScan._addFrames("ICP", 0)          # transform(id, ICP, 0), icp6D.cc:109
        if self.max_num_iterations == 0:          # icp6D.cc:112-114
            return 0
        tree = PreviousScan.getSearchTree()
        prm = IcpParams(int(self.my_icp6Dminimizer.getAlgorithmID()), int(pairing_mode),
                        int(self.max_num_iterations), float(self.max_dist_match2),
                        float(self.epsilonICP), 1 if self.quiet else 0)
        res = IcpResult()
        cap = max(1, self.max_num_iterations)
        trace = np.zeros((cap, 18))
        tm = CurrentScan.transMat.copy()
        da = CurrentScan.dalignxf.copy()
        if self.rnd > 1:       # -R: the keep-mask of every iteration is drawn on the host (std::rand, index order), the loop stays resident
            check(lib().tdtk_icp_match_rnd(tree._h, dptr(PreviousScan.dalignxf), CurrentScan.handle, dptr(tm),
                                           dptr(da), C.byref(prm), int(self.rnd), C.byref(res), dptr(trace), cap))
        else:
            check(lib().tdtk_icp_match(tree._h, dptr(PreviousScan.dalignxf), CurrentScan.handle, dptr(tm),
                                       dptr(da), C.byref(prm), C.byref(res), dptr(trace), cap))
        # frames as the reference's loop writes them (anim = -1): one after iteration 0 (transform(alignxf, ICP, 0),
        # icp6D.cc:258-264), one when the loop ends by convergence or at the cap (transform(id, ICP, 0),
        # :266-279) -- none on the "<= 3 pairs" break (:235-243 fall through to the end of the function)
        few_pairs = int(res.last_pairs) <= 3
        if not (few_pairs and res.iterations == 0):
            CurrentScan.transMat = MMult(trace[0, 2:], CurrentScan.transMat)
            CurrentScan._addFrames("ICP", 0)
        CurrentScan.transMat = tm
        CurrentScan.dalignxf = da
        CurrentScan.rPosTheta, CurrentScan.rPos = Matrix4ToEuler(tm)
        if not few_pairs:
            CurrentScan._addFrames("ICP", 0)
        self.nr_pointPair = int(res.last_pairs)
        nrows = min(cap, res.iterations + 1)
        self.last = dict(iterations=res.iterations, converged=bool(res.converged),
                         pairs=int(res.last_pairs), rms=res.last_rms, total_ms=res.total_ms,
                         nn_ms=res.nn_ms, sums_ms=res.sums_ms, trace=trace[:nrows].copy())
        # diagnostics (tdtk_icp_index_hashes): one hash of the correspondence indices per pass of the loop -- asked for only
        # when switched on (nothing on the path of a match otherwise)
        self.last["index_hashes"] = []
        if lib().tdtk_icp_index_hashes(-1):
            hn = C.c_int(0)
            hb = (C.c_uint64 * 1024)()
            check(lib().tdtk_icp_last_hashes(hb, 1024, C.byref(hn)))
            self.last["index_hashes"] = [int(hb[i]) for i in range(min(1024, hn.value))]
        return res.iterations

    def _match_stepped(self, PreviousScan, CurrentScan, pairing_mode=0):
        """The same loop driven from the host, one SearchTree::getPtPairs call per iteration: what
        an unmodified reference icp6D::match does on top of HipSearchTree.  Until round 4 the path of -R (rnd > 1); since
        round 5 the resident loop takes the keep-mask per iteration (tdtk_icp_match_rnd) and this form is kept for the test
        that compares the two (`icp.stepped_rnd = True`)."""
        CurrentScan._addFrames("ICP", 0)
        tree = PreviousScan.getSearchTree()
        algo = self.my_icp6Dminimizer.getAlgorithmID()
        want = self.my_icp6Dminimizer.want
        ret = prev_ret = prev_prev_ret = 0.0
        trace = []
        it = 0
        for it in range(self.max_num_iterations):
            prev_prev_ret, prev_ret = prev_ret, ret
            r = tree.getPtPairs(PreviousScan.dalignxf, CurrentScan.get_xyz_reduced(), None, 0, None, 0, self.rnd,
                                self.max_dist_match2, pairing_mode, want, None, want_pairs=False)
            self.nr_pointPair = r["n"]
            if r["n"] > 3:
                ret, alignxf = self.my_icp6Dminimizer.Align_Parallel(r, CurrentScan.transMat)
            else:
                break
            trace.append(np.concatenate([[r["n"], ret], alignxf]))
            CurrentScan.transform(alignxf, "ICP", 0 if it == 0 else -1)
            if (abs(ret - prev_ret) < self.epsilonICP and abs(ret - prev_prev_ret) < self.epsilonICP) or \
                    it == self.max_num_iterations - 1:
                CurrentScan._addFrames("ICP", 0)
                break
        self.last = dict(iterations=it, converged=it != self.max_num_iterations - 1, pairs=self.nr_pointPair,
                         rms=ret, total_ms=0.0, nn_ms=0.0, trace=np.array(trace))
        return it

    def _match_meta_data(self, PreviousScan, CurrentMeta):
        """icp6D::match with a MetaScan as the DATA scan (ELCH matches MetaScan(first..first+2) against
        MetaScan(last-2..last), elch6Deuler.cc:77-102): Scan::getPtPairsParallel walks the member scans' "xyz reduced"
        (scan.cc:1305-1327), the partial sums are merged (Align_Parallel, icp6Dquat.cc:533-588), and
        MetaScan::transform moves every member.  One batched device call per iteration for all members
        (tdtk_links_pair_sums), merged in the library (tdtk_pair_sums_merge).  QUAT / SVD only (what merges from the
        base block)."""
        algo = int(self.my_icp6Dminimizer.getAlgorithmID())
        if algo not in (ALGO_QUAT, ALGO_SVD):
            raise capi.TdtkError(-5, "a MetaScan as data scan is matched with -a 1 or -a 2 only")
        CurrentMeta.transform(M4identity(), "ICP", 0)                      # icp6D.cc:109
        if self.max_num_iterations == 0:
            return 0
        tree = PreviousScan.getSearchTree()
        members = CurrentMeta.m_scans
        nl = len(members)
        first = (C.c_void_p * nl)(*[tree._h] * nl)
        second = (C.c_void_p * nl)(*[m.handle for m in members])
        dal = np.ascontiguousarray(np.tile(f64(PreviousScan.dalignxf, 16), (nl, 1)))
        ret = prev_ret = prev_prev_ret = 0.0
        trace = []
        it = 0
        for it in range(self.max_num_iterations):
            prev_prev_ret, prev_ret = prev_ret, ret
            parts = (PairSums * nl)()
            check(lib().tdtk_links_pair_sums(nl, first, dptr(dal), second, float(self.max_dist_match2), 0, parts))
            merged = PairSums()
            check(lib().tdtk_pair_sums_merge(nl, parts, C.byref(merged)))
            self.nr_pointPair = int(merged.n)
            if merged.n > 3:
                ret, alignxf = self.my_icp6Dminimizer.Align_Parallel(merged)
            else:
                break
            trace.append(np.concatenate([[merged.n, ret], alignxf]))
            CurrentMeta.transform(alignxf, "ICP", 0 if it == 0 else -1)     # icp6D.cc:258-264, anim = -1
            if (abs(ret - prev_ret) < self.epsilonICP and abs(ret - prev_prev_ret) < self.epsilonICP) or \
                    it == self.max_num_iterations - 1:
                CurrentMeta.transform(M4identity(), "ICP", 0)               # write end pose
                break
        self.last = dict(iterations=it, converged=it != self.max_num_iterations - 1, pairs=self.nr_pointPair, rms=ret,
                         total_ms=0.0, nn_ms=0.0, sums_ms=0.0, trace=np.array(trace))
        return it

    def Point_Point_Error(self, PreviousScan, CurrentScan, max_dist_match, scale_max=None):
        """icp6D::Point_Point_Error (icp6D.cc:293-367) -> (error, number of pairs)."""
        if scale_max is None:
            scale_max = 0.000001          # include/slam6d/icp6D.h default
        err = C.c_double(0.0)
        npairs = C.c_uint64(0)
        check(lib().tdtk_point_point_error(PreviousScan.getSearchTree()._h, dptr(PreviousScan.dalignxf),
                                           CurrentScan.handle, float(max_dist_match), float(scale_max),
                                           C.byref(npairs), C.byref(err)))
        return err.value, int(npairs.value)

    def doICP(self, allScans, pairing_mode=0, prefetch=True):
        """icp6D::doICP (icp6D.cc:374-437): sequential matching against the previous scan, or
        (meta) against a MetaScan of all / the last max_num_metascans processed scans.

        prefetch: while scan i is matched against scan i-1, a second host thread (its own HIP stream; the
        library is thread-safe per handle) uploads scan i+1 and builds its search tree, so the per-scan
        preparation hides behind the previous match instead of adding to it.  The tree is built over
        "xyz reduced original", which no ICP step touches, so the result is unchanged."""
        pool = None
        # (under -R nothing is prepared ahead: the keep-masks come out of the process's std::rand() stream, and worker threads
        # that bring up HIP contexts draw from that stream at times of their own -- adapters/icp_glue.h has the measurement)
        if prefetch and not self.meta and self.rnd <= 1 and len(allScans) > 2:
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(self.prefetch_depth)

        def prep(s):
            _ = s.handle
            s.getSearchTree()

        pending = {}
        try:
            meta_scans, my_MetaScan = [], None
            for i in range(len(allScans)):
                cur = allScans[i]
                if pool is not None:
                    if i in pending:
                        pending.pop(i).result()
                    # a tree build is a chain of dependent adds on a few wavefronts: two of them in flight (scans
                    # i+1, i+2) cost the running match nothing and take the build off the critical path
                    for j in range(i + 1, min(i + 1 + self.prefetch_depth, len(allScans))):
                        if j not in pending and allScans[j]._h is None:
                            pending[j] = pool.submit(prep, allScans[j])
                if i > 0:
                    prev = allScans[i - 1]
                    if self.eP:
                        cur.mergeCoordinatesWithRoboterPosition(prev)
                    self.match(my_MetaScan if self.meta else prev, cur, pairing_mode)
                if self.meta and i != len(allScans) - 1:
                    meta_scans.append(cur)
                    if self.max_num_metascans > 0:
                        while len(meta_scans) > self.max_num_metascans:
                            meta_scans.pop(0)
                    my_MetaScan = MetaScan(meta_scans)
        finally:
            for f in pending.values():
                f.result()
            if pool is not None:
                pool.shutdown()


# ---------------------------------------------------------------------------------------
# Graph (src/slam6d/graph.cc:30-180)
# ---------------------------------------------------------------------------------------
class Graph:
    def __init__(self, nodes=0, cldist2=None, loopsize=None, scans=None, links=None):
        self.frm, self.to = [], []
        self.nrScans = nodes
        if links is not None:
            for a, b in links:
                self.frm.append(int(a)); self.to.append(int(b))
            return
        if cldist2 is not None and nodes > 0:       # graph.cc:107-130 in the library (tdtk_graph_links): chain, then (j, k) j-major
            P = np.array([s.get_rPos() for s in scans[:nodes]], dtype=np.float64).reshape(nodes, 3)
            cap = nodes * 4
            while True:
                frm = np.empty(cap, np.int32); to = np.empty(cap, np.int32); nl = C.c_int(0)
                check(lib().tdtk_graph_links(int(nodes), dptr(P), float(cldist2), int(loopsize), iptr(frm), iptr(to), cap, C.byref(nl)))
                if nl.value <= cap:
                    break
                cap = nl.value
            self.frm, self.to = frm[:nl.value].tolist(), to[:nl.value].tolist()
            self._arrays = (np.ascontiguousarray(frm[:nl.value]), np.ascontiguousarray(to[:nl.value]))
            return
        for i in range(nodes - 1):                  # graph.cc:115-118
            self.frm.append(i); self.to.append(i + 1)

    @classmethod
    def from_netfile(cls, netfile):
        """Graph(const std::string &netfile) (graph.cc:52-74, slam6D -n): "<scans> <links>" then one "from to" pair per
        link; the scan count is NOT taken from the file but accumulated by addLink."""
        g = cls(0, links=[])
        tok = open(netfile).read().split()
        if len(tok) < 2:
            raise RuntimeError("Error while reading network structure")
        nlinks = int(tok[1])
        if len(tok) < 2 + 2 * nlinks:
            raise RuntimeError("Error while reading network structure")        # the reference exit(1)s
        for j in range(nlinks):
            g.addLink(int(tok[2 + 2 * j]), int(tok[3 + 2 * j]))
        return g

    @classmethod
    def chain(cls, nScans, loop=False):
        """Graph(int nScans, bool loop) (graph.cc:84-105): minimally connected, optionally closed"""
        n = nScans if loop else nScans - 1
        g = cls(0, links=[(i, (i + 1 if i != n - 1 else 0) if loop else i + 1) for i in range(n)])
        g.nrScans = nScans
        return g

    def addLink(self, i, j):
        """graph.cc:157-174: an end point seen for the first time counts as a new scan (i == j counts twice)"""
        if not any(f == i or t == i for f, t in zip(self.frm, self.to)):
            self.nrScans += 1
        if not any(f == j or t == j for f, t in zip(self.frm, self.to)):
            self.nrScans += 1
        self.frm.append(int(i)); self.to.append(int(j))
        self._arrays = None

    def getNrScans(self): return self.nrScans
    def getNrLinks(self): return len(self.frm)
    def getLink(self, i, fromTo): return self.frm[i] if fromTo == 0 else self.to[i]


# ---------------------------------------------------------------------------------------
# lum6DEuler (src/slam6d/lum6Deuler.cc) -- single-process form; graphslam.py shards links
# ---------------------------------------------------------------------------------------
def covarianceEuler(first, second, max_dist_match2):
    """lum6DEuler::covarianceEuler (lum6Deuler.cc:94-251) -> (C 6x6, CD 6, m, ss)."""
    Cm = np.zeros(36); CD = np.zeros(6)
    m = C.c_uint64(0); ss = C.c_double(0.0)
    tree = first.getSearchTree()
    check(lib().tdtk_lum_link(tree._h, dptr(first.dalignxf), second.handle, float(max_dist_match2),
                              dptr(Cm), dptr(CD), C.byref(m), C.byref(ss)))
    return Cm.reshape(6, 6), CD, int(m.value), ss.value


def solveSparseCholesky(G, B):
    """graphSlam6D::solveSparseCholesky(GraphMatrix*, B) (graphSlam6D.cc:345-379)."""
    G = f64(G); B = f64(B)
    n = len(B)
    x = np.empty(n)
    check(lib().tdtk_solve_spd(dptr(G), dptr(B), n, dptr(x)))
    return x


def lum_pose_update(scan, Xi):
    """Per-scan pose correction of lum6DEuler::doGraphSlam6D (lum6Deuler.cc:378-448):
    result = Ha^-1 * X_i, new pose = old - result.  Returns (rPos, rPosTheta, |dxyz|)."""
    xa, ya, za = scan.get_rPos()
    tx, ty = scan.get_rPosTheta()[0], scan.get_rPosTheta()[1]
    ctx, stx, cty, sty = math.cos(tx), math.sin(tx), math.cos(ty), math.sin(ty)
    Ha = np.eye(6)
    Ha[0, 4] = -za * ctx + ya * stx
    Ha[0, 5] = ya * cty * ctx + za * stx * cty
    Ha[1, 3] = za
    Ha[1, 4] = -xa * stx
    Ha[1, 5] = -xa * ctx * cty + za * sty
    Ha[2, 3] = -ya
    Ha[2, 4] = xa * ctx
    Ha[2, 5] = -xa * cty * stx - ya * sty
    Ha[3, 5] = sty
    Ha[4, 4] = stx
    Ha[4, 5] = ctx * cty
    Ha[5, 4] = ctx
    Ha[5, 5] = -stx * cty
    result = np.linalg.solve(Ha, Xi)
    rPos = scan.get_rPos() - result[:3]
    rPosTheta = scan.get_rPosTheta() - result[3:]
    return rPos, rPosTheta, float(np.sqrt(result[0] ** 2 + result[1] ** 2 + result[2] ** 2))


class _graphSlam6D_setters:
    def set_mdmll(self, mdmll):
        """graphSlam6D::set_mdmll (graphSlam6D.cc:425-427): slam6D.cc:818-821 runs a second doGraphSlam6D with it"""
        self.max_dist_match2_LUM = mdmll * mdmll

    def set_quiet(self, quiet):
        self.quiet = quiet


class lum6DEuler(_graphSlam6D_setters):
    """lum6DEuler (-G 1).  doGraphSlam6D follows lum6Deuler.cc:314-477; the link loop of
    FillGB3D (lum6Deuler.cc:265-303) is sharded over ranks when a process group is given."""

    def __init__(self, my_icp, mdm=25.0, max_dist_match_LUM=25.0, max_num_iterations=50, quiet=True,
                 epsilonLUM=0.5, group=None):
        self.my_icp = my_icp
        self.max_dist_match2_LUM = max_dist_match_LUM * max_dist_match_LUM
        self.quiet = quiet
        self.epsilonLUM = epsilonLUM
        self.group = group

    def doGraphSlam6D(self, gr, allScans, nrIt, native=True, device=None):
        from .graphslam import lum_iteration, lum_iteration_native
        if gr.getNrScans() <= 0:
            raise RuntimeError("Zero scans in graph")
        ret = float("inf")
        it = 0
        while it < nrIt and ret > self.epsilonLUM:
            if native:
                ret = lum_iteration_native(gr, allScans, self.max_dist_match2_LUM, self.group, device)
            else:
                ret = lum_iteration(gr, allScans, self.max_dist_match2_LUM, self.group, None, device)
            it += 1
        return ret


class lum6DQuat(_graphSlam6D_setters):
    """lum6DQuat (-G 2, src/slam6d/lum6Dquat.cc): LUM with 7 unknowns per scan (translation +
    quaternion).  Links sharded / reduced like lum6DEuler."""

    def __init__(self, my_icp=None, mdm=25.0, max_dist_match_LUM=25.0, max_num_iterations=50, quiet=True,
                 epsilonLUM=0.5, group=None):
        self.my_icp = my_icp
        self.max_dist_match2_LUM = max_dist_match_LUM * max_dist_match_LUM
        self.epsilonLUM = epsilonLUM
        self.group = group

    def doGraphSlam6D(self, gr, allScans, nrIt, device=None):
        from .graphslam import lumquat_iteration
        ret = float("inf")
        it = 0
        while it < nrIt and ret > self.epsilonLUM:
            ret = lumquat_iteration(gr, allScans, self.max_dist_match2_LUM, self.group, device)
            it += 1
        return ret


class ghelix6DQ2(_graphSlam6D_setters):
    """ghelix6DQ2 (-G 3, src/slam6d/ghelix6DQ2.cc): simultaneous registration with the helical-motion
    linearisation; B and bd are zeroed once per doGraphSlam6D call, not per iteration (:329-330)."""

    def __init__(self, my_icp=None, mdm=25.0, max_dist_match_LUM=25.0, max_num_iterations=50, quiet=True,
                 epsilonLUM=0.5, group=None):
        self.my_icp = my_icp
        self.max_dist_match2_LUM = max_dist_match_LUM * max_dist_match_LUM
        self.epsilonLUM = epsilonLUM
        self.group = group

    def doGraphSlam6D(self, gr, allScans, nrIt, device=None):
        from .graphslam import ghelix_iteration
        from .graphslam import graph_state
        state = graph_state(3, gr.getNrScans())
        ret = float("inf")
        it = 0
        while it < nrIt and ret > self.epsilonLUM:
            ret = ghelix_iteration(gr, allScans, self.max_dist_match2_LUM, state, self.group, device)
            it += 1
        return ret


class gapx6D(_graphSlam6D_setters):
    """gapx6D (-G 4, src/slam6d/gapx6D.cc): graph-SLAM with the small-angle (APX) linearisation:
    a 3(n-1) rotation system from per-link second moments, then translations from the link
    Laplacian.  Same link sharding / single all-reduce as lum6DEuler."""

    def __init__(self, my_icp=None, mdm=25.0, max_dist_match_LUM=25.0, max_num_iterations=50, quiet=True,
                 epsilonLUM=0.5, group=None):
        self.my_icp = my_icp
        self.max_dist_match2_LUM = max_dist_match_LUM * max_dist_match_LUM
        self.epsilonLUM = epsilonLUM
        self.group = group

    def doGraphSlam6D(self, gr, allScans, nrIt, device=None):
        from .graphslam import gapx_iteration
        n = gr.getNrScans() - 1
        T = np.zeros(3 * n)                 # not reset between iterations (gapx6D.cc:356,463-468)
        ret = float("inf")
        it = 0
        while it < nrIt and ret > self.epsilonLUM:
            ret = gapx_iteration(gr, allScans, self.max_dist_match2_LUM, T, self.group, device)
            it += 1
        return ret


# ---------------------------------------------------------------------------------------
# loop closing (-L 1): loopSlam6D (include/slam6d/loopSlam6D.h), elch6D::graph_balancer (src/slam6d/elch6D.cc:186-279),
# elch6Deuler::close_loop (src/slam6d/elch6Deuler.cc:44-138)
# ---------------------------------------------------------------------------------------
class loopSlam6D:
    def __init__(self, quiet, my_icp6Dminimizer, mdm, max_num_iterations, rnd=1, eP=True, anim=-1, epsilonICP=0.0000001,
                 nns_method=0):
        self.quiet = quiet
        self.my_icp6D = icp6D(my_icp6Dminimizer, mdm, max_num_iterations, quiet, False, rnd, eP, anim, epsilonICP,
                              nns_method)

    def close_loop(self, allScans, first, last, g):
        raise NotImplementedError


def graph_balancer(nvertices, edges, weights_of_edges, first, last):
    """elch6D::graph_balancer -> weights[nvertices] (tdtk_elch_graph_balancer); edges = [(from, to)]"""
    frm = np.ascontiguousarray([e[0] for e in edges], dtype=np.int32)
    to = np.ascontiguousarray([e[1] for e in edges], dtype=np.int32)
    w = f64(weights_of_edges)
    out = np.zeros(nvertices)
    check(lib().tdtk_elch_graph_balancer(int(nvertices), len(edges), iptr(frm), iptr(to), dptr(w), int(first), int(last),
                                         dptr(out)))
    return out


class elch6Deuler(loopSlam6D):
    """-L 1.  g = the loop-optimisation graph as a list of (from, to) edges over scans 0..n-1 (slam6D.cc:416-428 adds
    (i-1, i) for every matched scan and (first, last) after every closed loop)."""

    def close_loop(self, allScans, first, last, g):
        n = max(max(a, b) for a, b in g) + 1                      # num_vertices(g)
        edges = list(g)
        # one covarianceEuler per edge (elch6Deuler.cc:53-66): all of them in one batched device call
        ne = len(edges)
        firsts = (C.c_void_p * ne)(*[allScans[a].getSearchTree()._h for a, b in edges])
        seconds = (C.c_void_p * ne)(*[allScans[b].handle for a, b in edges])
        dal = np.ascontiguousarray(np.stack([allScans[a].dalignxf for a, b in edges]))
        blocks = np.empty((ne, 42))
        check(lib().tdtk_graph_link_blocks(1, ne, firsts, dptr(dal), seconds, float(self.my_icp6D.max_dist_match2), dptr(blocks)))
        wts = np.empty((6, ne))
        for e in range(ne):
            Cinv = np.empty((6, 6))
            Cm = np.ascontiguousarray(blocks[e, :36].reshape(6, 6))
            check(lib().tdtk_invert(dptr(Cm), 6, dptr(Cinv)))     # C = C.i()
            wts[:, e] = np.abs(np.diag(Cinv))
        weights = [graph_balancer(n, edges, wts[j], first, last) for j in range(6)]
        start = MetaScan([allScans[first], allScans[first + 1], allScans[first + 2]])
        end = MetaScan([allScans[last - 2], allScans[last - 1], allScans[last]])
        for i in range(last - 2, last + 1):
            for j in range(6):
                weights[j][i] = 0.0
        before = np.concatenate([allScans[last].get_rPos(), allScans[last].get_rPosTheta()])
        self.my_icp6D.match(start, end)
        delta = np.concatenate([allScans[last].get_rPos(), allScans[last].get_rPosTheta()]) - before
        self.last_delta = delta
        if not self.quiet:
            print("Delta: " + " ".join("%g" % v for v in delta))
        A1 = np.zeros((n, 16)); A2 = np.zeros((n, 16))
        for i in range(1, n):
            rP = np.array([allScans[i].get_rPos()[k] + delta[k] * (weights[k][i] - weights[k][0]) for k in range(3)])
            rT = np.array([allScans[i].get_rPosTheta()[k] + delta[3 + k] * (weights[3 + k][i] - weights[3 + k][0]) for k in range(3)])
            allScans[i].transformToEuler(rP, rT, "ELCH", 2 if i == n - 1 else 1)


def _qmul(a, b):
    """Hamilton product, (w, x, y, z) (QMult, globals.icc:1112-1117)"""
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3],
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1],
                     a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def _qunit(q):
    """Normalize4 (globals.icc:267-275)"""
    q = np.asarray(q, dtype=np.float64)
    return q / math.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])


def _slerp(qa, qb, t):
    """slerp (globals.icc:1123-1166): qa where the two coincide, the normalised mean where they are opposite"""
    c = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3]
    if abs(c) >= 1.0:
        return np.array(qa, dtype=np.float64)
    half = math.acos(c)
    sn = math.sqrt(1.0 - c * c)
    if abs(sn) < 0.001:
        return _qunit(np.asarray(qa) * 0.5 + np.asarray(qb) * 0.5)
    ra, rb = math.sin((1 - t) * half) / sn, math.sin(t * half) / sn
    return _qunit(np.asarray(qa) * ra + np.asarray(qb) * rb)


class _elchQuatBase(loopSlam6D):
    """What -L 2 / 3 / 4 share: one lum6DQuat::covarianceQuat pass per edge of the loop graph -- all of them in ONE batched
    device call --, C = C.i(), edge weights from |diag C| (seven of them, or three translations + the sum of the four
    quaternion entries), one graph_balancer run per weight set."""

    def _weights(self, allScans, first, last, g, combine):
        n = max(max(a, b) for a, b in g) + 1
        edges = list(g)
        ne = len(edges)
        firsts = (C.c_void_p * ne)(*[allScans[a].getSearchTree()._h for a, b in edges])
        seconds = (C.c_void_p * ne)(*[allScans[b].handle for a, b in edges])
        dal = np.ascontiguousarray(np.stack([allScans[a].dalignxf for a, b in edges]))
        blocks = np.empty((ne, 56))
        check(lib().tdtk_graph_link_blocks(2, ne, firsts, dptr(dal), seconds, float(self.my_icp6D.max_dist_match2), dptr(blocks)))
        nw = 4 if combine else 7
        wts = np.empty((nw, ne))
        for e in range(ne):
            Cinv = np.empty((7, 7))
            Cm = np.ascontiguousarray(blocks[e, :49].reshape(7, 7))
            check(lib().tdtk_invert(dptr(Cm), 7, dptr(Cinv)))
            d = np.abs(np.diag(Cinv))
            if combine:
                wts[:3, e] = d[:3]
                wts[3, e] = d[3] + d[4] + d[5] + d[6]
            else:
                wts[:, e] = d
        return n, [graph_balancer(n, edges, wts[j], first, last) for j in range(nw)]


class elch6Dquat(_elchQuatBase):
    """-L 2 (src/slam6d/elch6Dquat.cc:44-148): like elch6Deuler with the pose as position + quaternion, each of the seven
    components distributed linearly and the quaternion renormalised."""

    def close_loop(self, allScans, first, last, g):
        n, weights = self._weights(allScans, first, last, g, False)
        start = MetaScan([allScans[first], allScans[first + 1], allScans[first + 2]])
        end = MetaScan([allScans[last - 2], allScans[last - 1], allScans[last]])
        for i in range(last - 2, last + 1):
            for j in range(7):
                weights[j][i] = 0.0
        before = np.concatenate([allScans[last].get_rPos(), allScans[last].get_rPosQuat()])
        self.my_icp6D.match(start, end)
        delta = np.concatenate([allScans[last].get_rPos(), allScans[last].get_rPosQuat()]) - before
        self.last_delta = delta
        for i in range(1, n):
            w = np.array([weights[k][i] - weights[k][0] for k in range(7)])
            rP = allScans[i].get_rPos() + delta[:3] * w[:3]
            rQ = _qunit(allScans[i].get_rPosQuat() + delta[3:] * w[3:])
            allScans[i].transformToQuat(rP, rQ, "ELCH", 2 if i == n - 1 else 1)


class elch6DunitQuat(_elchQuatBase):
    """-L 3 (src/slam6d/elch6DunitQuat.cc:45-199): the rotation of the loop error as ONE unit quaternion deltaQ =
    q_after * conj(q_before) of the last scan, applied to scan i blended linearly with weight w_i (and renormalised), the
    whole chain counter-rotated so that scan 0 stays where it is; the three matched scans are put back first."""

    def close_loop(self, allScans, first, last, g):
        n, weights = self._weights(allScans, first, last, g, True)
        start = MetaScan([allScans[first], allScans[first + 1], allScans[first + 2]])
        end = MetaScan([allScans[last - 2], allScans[last - 1], allScans[last]])
        old = [(allScans[k].get_rPos().copy(), allScans[k].get_rPosQuat().copy()) for k in (last, last - 1, last - 2)]
        p_before = allScans[last].get_rPos().copy()
        q1 = allScans[last].get_rPosQuat() * np.array([1.0, -1.0, -1.0, -1.0])
        self.my_icp6D.match(start, end)
        delta = allScans[last].get_rPos() - p_before
        deltaQ = _qmul(allScans[last].get_rPosQuat(), q1)
        self.last_delta = np.concatenate([delta, deltaQ])
        for k, (p, q) in zip((last, last - 1, last - 2), old):            # restore the poses the match moved
            allScans[k].transformToQuat(p, q, "INVALID", -1)
        q0 = allScans[0].get_rPosQuat()
        w0 = weights[3][0]
        s0 = ((1 - w0) * q0 + _qmul(deltaQ, q0) * w0) * np.array([1.0, -1.0, -1.0, -1.0])
        counter = _qmul(q0, _qunit(s0))
        for i in range(1, n):
            rP = allScans[i].get_rPos() + delta * np.array([weights[k][i] - weights[k][0] for k in range(3)])
            qi = allScans[i].get_rPosQuat()
            wi = weights[3][i]
            blended = _qunit((1 - wi) * qi + _qmul(deltaQ, qi) * wi)
            allScans[i].transformToQuat(rP, _qunit(_qmul(counter, blended)), "ELCH", 2 if i == n - 1 else 1)


class elch6Dslerp(_elchQuatBase):
    """-L 4 (src/slam6d/elch6Dslerp.cc:44-184): the loop error as a rigid motion deltaf in the frame of scan `first`,
    its rotation interpolated on the sphere (slerp from the identity by w_i) and its translation scaled per axis; larger
    MetaScans (first-2 .. first+2 against last-2 .. last); the matched scans take the motion of weight w_0 only."""

    def close_loop(self, allScans, first, last, g):
        n, weights = self._weights(allScans, first, last, g, True)
        start = MetaScan([allScans[i] for i in range(first - 2, first + 3) if i >= 0])
        end = MetaScan([allScans[i] for i in range(last - 2, last + 1) if i < n])
        Pl0 = allScans[last].get_transMat().copy()
        self.my_icp6D.match(start, end)
        Pp0 = allScans[last].get_transMat().copy()
        Pf0 = allScans[first].get_transMat().copy()
        Pf0_inv = M4inv(Pf0)
        deltaf = MMult(Pf0_inv, MMult(Pp0, M4inv(MMult(Pf0_inv, Pl0))))
        deltaQ, deltaT = Matrix4ToQuat(deltaf)
        self.last_delta = np.concatenate([deltaT, deltaQ])
        idQ = np.array([1.0, 0.0, 0.0, 0.0])

        def part(i):       # the share of scan i of the loop error
            return QuatToMatrix4(_slerp(idQ, deltaQ, weights[3][i]), np.array([deltaT[k] * weights[k][i] for k in range(3)]))
        delta0 = MMult(Pf0, M4inv(part(0)))
        for i in range(1, n):
            if last - 2 <= i <= last:
                M = MMult(delta0, Pf0_inv)
            else:
                M = MMult(MMult(delta0, part(i)), Pf0_inv)
            allScans[i].transform(M, "ELCH", 2 if i == n - 1 else 1)


def computeGraph6Dautomatic(allScans, clpairs, max_dist_match2_LUM=625.0, group=None, device=None):
    """graphSlam6D::computeGraph6Dautomatic / the graph step of matchGraph6Dautomatic(allScans, nrIt,
    clpairs, loopsize) (src/slam6d/graphSlam6D.cc:82-133, 136-180): a link (j, k) for every ordered pair
    j != k whose closest-point pairing within max_dist_match2_LUM has more than clpairs pairs.  The
    n(n-1) pairings are independent: they go through the batched link entry point, sharded over
    ranks like the LUM links; links are added in the serial build's (j-major) order."""
    from .graphslam import _dist_ready
    n = len(allScans)
    cand = [(j, k) for j in range(n) for k in range(n) if j != k]
    rank, world = 0, 1
    if group is not None or _dist_ready():
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = [i for i, (j, k) in enumerate(cand) if (j + k) % world == rank]
    counts = np.zeros(len(cand))
    if mine:
        nl = len(mine)
        first = (C.c_void_p * nl)(*[allScans[cand[i][0]].getSearchTree()._h for i in mine])
        second = (C.c_void_p * nl)(*[allScans[cand[i][1]].handle for i in mine])
        dal = np.ascontiguousarray(np.stack([allScans[cand[i][0]].dalignxf for i in mine]))
        sums = (PairSums * nl)()
        check(lib().tdtk_links_pair_sums(nl, first, dptr(dal), second, float(max_dist_match2_LUM), 0, sums))
        for q, i in enumerate(mine):
            counts[i] = sums[q].n
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(counts)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        counts = t.cpu().numpy()
    gr = Graph(0, links=[cand[i] for i in range(len(cand)) if counts[i] > clpairs])
    gr.nrScans = n
    gr.pair_counts = {cand[i]: int(counts[i]) for i in range(len(cand))}
    return gr


def matchGraph6Dautomatic_clpairs(my_graphSlam6D, allScans, nrIt, clpairs, loopsize=0):
    """graphSlam6D::matchGraph6Dautomatic(allScans, nrIt, clpairs, loopsize) (graphSlam6D.cc:82-133):
    rebuild the clpairs graph and run one global iteration until the pose change is <= 0.001 or nrIt
    rounds.  Returns the number of rounds."""
    i = 0
    while True:
        i += 1
        gr = computeGraph6Dautomatic(allScans, clpairs, my_graphSlam6D.max_dist_match2_LUM,
                                     getattr(my_graphSlam6D, "group", None))
        ret = my_graphSlam6D.doGraphSlam6D(gr, allScans, 1)
        if not (ret > 0.001 and i < nrIt):
            return i


def matchGraph6Dautomatic(cldist, loopsize, allScans, my_icp6D, meta_icp, my_graphSlam6D, nrIt, epsilonSLAM,
                          mdml, eP=True, max_num_metascans=-1, prefetch=True, my_loopSlam6D=None, mdmll=-1.0, graphDist=0.0):
    """matchGraph6Dautomatic (src/slam6d/slam6D.cc:387-548): sequential ICP (against the predecessor, or with meta_icp
    against a MetaScan of the last max_num_metascans scans, slam6D.cc:436-448), loop detection by pose distance (the
    closest pair (first, last) seen while a loop is being detected), ELCH loop closing when my_loopSlam6D is given (-L,
    slam6D.cc:500-505, 523-527), rounds of { fresh Graph(i+1, cldist^2, loopsize); one doGraphSlam6D iteration } until
    ret <= epsilonSLAM or nrIt rounds, and with mdmll > 0 the closing pass of slam6D.cc:535-547 (-DlastSLAM):
    set_mdmll(mdmll), then such rounds on Graph(n, graphDist^2, loopsize)."""
    cldist2 = cldist * cldist
    metas = []
    n = len(allScans)
    loop_detection = 0
    rounds = 0
    g = []                         # graph for loop optimisation (graph_t g, slam6D.cc:411)
    min_dist, first, last = -1.0, 0, 0

    def global_rounds(nodes, graph_dist2=None):
        nonlocal rounds
        j = 0
        while True:
            gr = Graph(nodes, cldist2 if graph_dist2 is None else graph_dist2, loopsize, allScans)
            ret = my_graphSlam6D.doGraphSlam6D(gr, allScans, 1)
            j += 1
            rounds += 1
            if not (j < nrIt and ret > epsilonSLAM):
                return ret

    # the next scan is uploaded and its tree built on a second host thread while the current one is matched
    # (see icp6D.doICP); scan i+1 is not part of the graph of scans 0..i, so the global rounds do not touch it
    from concurrent.futures import ThreadPoolExecutor
    depth = icp6D.prefetch_depth
    pool = ThreadPoolExecutor(depth) if (prefetch and n > 2) else None

    def prep(s):
        _ = s.handle
        s.getSearchTree()
    pending = {}
    try:
        for i in range(1, n):
            if pool is not None:
                if i in pending:
                    pending.pop(i).result()
                for j in range(i + 1, min(i + 1 + depth, n)):
                    if j not in pending and allScans[j]._h is None:
                        pending[j] = pool.submit(prep, allScans[j])
            g.append((i - 1, i))
            if eP:
                allScans[i].mergeCoordinatesWithRoboterPosition(allScans[i - 1])
            if my_icp6D is not None:
                if meta_icp:
                    metas.append(allScans[i - 1])
                    if max_num_metascans > 0:
                        while len(metas) > max_num_metascans:
                            metas.pop(0)
                    my_icp6D.match(MetaScan(metas), allScans[i])
                else:
                    my_icp6D.match(allScans[i - 1], allScans[i])
            if loop_detection == 1:
                loop_detection = 2
            for j in range(0, i - loopsize):
                d = allScans[j].get_rPos() - allScans[i].get_rPos()
                dist = d[0] * d[0] + d[1] * d[1] + d[2] * d[2]
                if dist < cldist2:
                    loop_detection = 1
                    if min_dist < 0 or dist < min_dist:
                        min_dist, first, last = dist, j, i
            if loop_detection == 2:
                loop_detection = 0
                min_dist = -1.0
                if my_loopSlam6D is not None:
                    my_loopSlam6D.close_loop(allScans, first, last, g)
                    g.append((first, last))
                if my_graphSlam6D is not None and mdml > 0:
                    global_rounds(i + 1)
    finally:
        for f in pending.values():
            f.result()
        if pool is not None:
            pool.shutdown()
    if loop_detection == 1 and my_loopSlam6D is not None:
        my_loopSlam6D.close_loop(allScans, first, last, g)
        g.append((first, last))
    if my_graphSlam6D is not None and mdml > 0.0:
        global_rounds(n)
    if my_graphSlam6D is not None and mdmll > 0.0:              # slam6D.cc:535-547
        my_graphSlam6D.set_mdmll(mdmll)
        global_rounds(n, graphDist * graphDist)
    return rounds
