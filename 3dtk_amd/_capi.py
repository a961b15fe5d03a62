"""ctypes binding of lib3dtk_hip.so (include/tdtk_hip.h).  Plumbing only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "lib3dtk_hip.so")
# The product library and the lab library (csrc/Makefile: the same sources with -DTDTK_LAB -- + the kernels and policies
# that were built, measured and lost, and the environment switches that select them).  Everything loads the product;
# TDTK_LIB=lab in the environment, or `with library("lab"):` around a test that compares a lab variant with the product
# path, selects the other one.
_SOS = {"product": _SO, "lab": os.path.join(_HERE, "lib3dtk_hip_lab.so")}

ALGO_QUAT, ALGO_SVD, ALGO_APX, ALGO_NAPX = 1, 2, 6, 10
ALGO_ORTHO, ALGO_DUAL, ALGO_HELIX, ALGO_LUMEULER, ALGO_LUMQUAT, ALGO_QUAT_SCALE = 3, 4, 5, 7, 8, 9
CLOSEST_POINT, CLOSEST_POINT_ALONG_NORMAL_SIMPLE, CLOSEST_PLANE_SIMPLE = 0, 1, 2
WANT_APX, WANT_NAPX, WANT_LUM, WANT_GAPX, WANT_MOM2 = 1, 2, 4, 8, 16

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


class TdtkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tdtk error %d: %s" % (code, msg))
        self.code = code


class PairSums(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("n", C.c_uint64), ("sum", C.c_double),
                ("centroid_m", C.c_double * 3), ("centroid_d", C.c_double * 3),
                ("Si", C.c_double * 9), ("apx_A", C.c_double * 6), ("apx_B", C.c_double * 3),
                ("napx_A", C.c_double * 21), ("napx_B", C.c_double * 6), ("napx_sum", C.c_double),
                ("lum", C.c_double * 15), ("lum_sumd2", C.c_double),
                ("gapx_MkMkt", C.c_double * 9), ("gapx_DkDkt", C.c_double * 9), ("gapx_MkDkt", C.c_double * 9),
                ("gapx_DkMkt", C.c_double * 9), ("gapx_Ak1", C.c_double * 3), ("gapx_Ak2", C.c_double * 3),
                ("mom_mm", C.c_double * 6), ("mom_dd", C.c_double * 6), ("lum_udot", C.c_double)]


class TreeInfo(C.Structure):
    _fields_ = [("n_points", C.c_uint64), ("n_internal", C.c_uint64), ("n_leaves", C.c_uint64),
                ("max_depth", C.c_uint32), ("max_leaf_points", C.c_uint32),
                ("device_bytes", C.c_uint64), ("build_ms", C.c_double), ("upload_ms", C.c_double)]


class IcpParams(C.Structure):
    _fields_ = [("algo", C.c_int), ("pairing_mode", C.c_int), ("max_num_iterations", C.c_int),
                ("max_dist_match2", C.c_double), ("epsilon_icp", C.c_double), ("quiet", C.c_int)]


class IcpResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("last_pairs", C.c_uint64),
                ("last_rms", C.c_double), ("total_ms", C.c_double), ("nn_ms", C.c_double),
                ("sums_ms", C.c_double)]


# every symbol include/tdtk_hip.h declares (tests check that the library exports all of them)
EXPORTS = [
    "tdtk_last_error", "tdtk_device_count", "tdtk_pool_trim", "tdtk_build_respeculated", "tdtk_version", "tdtk_tree_create", "tdtk_tree_create_from_scan", "tdtk_tree_create_from_scans", "tdtk_scan_mark_original", "tdtk_scan_download_original", "tdtk_tree_destroy",
    "tdtk_tree_get_info", "tdtk_tree_verify", "tdtk_find_closest", "tdtk_find_closest_dev", "tdtk_find_closest_along_dir",
    "tdtk_get_pt_pairs", "tdtk_scan_create", "tdtk_scan_destroy", "tdtk_scan_size",
    "tdtk_scan_transform", "tdtk_scan_download", "tdtk_scan_pairs", "tdtk_align", "tdtk_icp_match", "tdtk_icp_match_rnd",
    "tdtk_lum_link", "tdtk_lum_links", "tdtk_links_pair_sums", "tdtk_lum_update_poses", "tdtk_lum_assemble_solve", "tdtk_point_point_error",
    "tdtk_graph_block_doubles", "tdtk_graph_link_blocks", "tdtk_graph_solve_update", "tdtk_scans_transform2", "tdtk_solve_spd",
    "tdtk_solve_chol_upper", "tdtk_invert", "tdtk_reduce_octree", "tdtk_reduce_octree_nrpts", "tdtk_normals_apx_knn", "tdtk_scan_calc_normals", "tdtk_last_kernel_ms", "tdtk_count_visits",
    "tdtk_comm_unique_id", "tdtk_comm_create", "tdtk_comm_destroy", "tdtk_comm_info", "tdtk_comm_rccl_world", "tdtk_graph_exchange", "tdtk_graph_deal_links",
    "tdtk_graph_iteration", "tdtk_elch_graph_balancer", "tdtk_pair_sums_merge", "tdtk_graph_links",
    "tdtk_last_timings", "tdtk_kernel_timing", "tdtk_visit_counting", "tdtk_visit_counters", "tdtk_measure_bandwidth",
    "tdtk_icp_index_hashes", "tdtk_icp_last_hashes",
    "tdtk_host_tree_layout", "tdtk_host_m4inv", "tdtk_host_mmult",
    "tdtk_host_euler_to_matrix4", "tdtk_host_matrix4_to_euler", "tdtk_host_quat_to_matrix4", "tdtk_host_matrix4_to_quat",
    "tdtk_io_read_uos", "tdtk_io_free", "tdtk_io_read_pose", "tdtk_io_write_frames",
]


def build_extension(force=False):
    """Compile 3dtk_amd/csrc for gfx950 into 3dtk_amd/lib3dtk_hip.so (in-tree)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src, "-j4"], stdout=subprocess.DEVNULL)
    return _SO


_libs = {}
_current = "lab" if os.environ.get("TDTK_LIB") == "lab" else "product"


class library:
    """with library("lab"): ...  -- every lib() call inside goes to that library.  Handles made inside should be released
    inside (a handle that outlives the block is destroyed by the other library, which works -- both free device memory
    the same way -- but is not the intended use)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _current
        self.prev, _current = _current, self.name
        return lib()

    def __exit__(self, *exc):
        global _current
        _current = self.prev


def is_lab():
    return _current == "lab"


def _init_torch_runtime_first():
    """PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  When a process uses both torch
    (device tensors, torch.distributed/RCCL) and this library, torch's runtime has to attach
    to the GPU first; the reverse order leaves torch with "No HIP GPUs are available".  torch is
    plumbing here, so initialise it up front when it is importable (TDTK_NO_TORCH=1 skips)."""
    if os.environ.get("TDTK_NO_TORCH") == "1":
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # torch absent or CPU-only: nothing to order
        pass


def lib():
    """The loaded C-ABI library.  Fails loudly when the HIP extension has not been built."""
    if _current in _libs:
        return _libs[_current]
    so = _SOS[_current]
    if not os.path.exists(so):
        raise TdtkError(-2, "%s is not built (run __graft_entry__.build()); "
                            "there is no CPU fallback" % os.path.basename(so))
    _init_torch_runtime_first()
    L = C.CDLL(so)
    L.tdtk_last_error.restype = C.c_char_p
    L.tdtk_version.restype = C.c_char_p
    L.tdtk_tree_create.argtypes = [_dp, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tdtk_tree_destroy.argtypes = [C.c_void_p]
    L.tdtk_tree_destroy.restype = None
    L.tdtk_tree_get_info.argtypes = [C.c_void_p, C.POINTER(TreeInfo)]
    L.tdtk_tree_verify.argtypes = [C.c_void_p, _u64p]
    L.tdtk_find_closest.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _ip, _dp]
    L.tdtk_find_closest_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_double, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_void_p]
    L.tdtk_find_closest_along_dir.argtypes = [C.c_void_p, _dp, _dp, C.c_size_t, C.c_double, _ip, _dp]
    L.tdtk_get_pt_pairs.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_size_t, C.c_size_t, C.c_int, C.c_int,
                                    C.c_double, C.c_uint32, _dp, _ip, _dp, _dp, _dp, C.POINTER(PairSums)]
    L.tdtk_scan_create.argtypes = [_dp, _dp, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]
    L.tdtk_scan_destroy.argtypes = [C.c_void_p]
    L.tdtk_scan_destroy.restype = None
    L.tdtk_scan_size.argtypes = [C.c_void_p]
    L.tdtk_scan_size.restype = C.c_size_t
    L.tdtk_scan_transform.argtypes = [C.c_void_p, _dp]
    L.tdtk_scan_download.argtypes = [C.c_void_p, _dp, _dp]
    L.tdtk_scan_pairs.argtypes = [C.c_void_p, _dp, C.c_void_p, C.c_int, C.c_double, C.c_uint32, _dp, _ip,
                                  C.POINTER(PairSums)]
    L.tdtk_align.argtypes = [C.c_int, C.POINTER(PairSums), _dp, _dp]
    L.tdtk_icp_match.argtypes = [C.c_void_p, _dp, C.c_void_p, _dp, _dp, C.POINTER(IcpParams),
                                 C.POINTER(IcpResult), _dp, C.c_int]
    L.tdtk_icp_match_rnd.argtypes = [C.c_void_p, _dp, C.c_void_p, _dp, _dp, C.POINTER(IcpParams), C.c_int,
                                     C.POINTER(IcpResult), _dp, C.c_int]
    L.tdtk_lum_link.argtypes = [C.c_void_p, _dp, C.c_void_p, C.c_double, _dp, _dp, _u64p, _dp]
    L.tdtk_lum_links.argtypes = [C.c_int, C.POINTER(C.c_void_p), _dp, C.POINTER(C.c_void_p), C.c_double, _dp,
                                 _dp, _u64p, _dp]
    L.tdtk_lum_update_poses.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_void_p), _dp, _dp]
    L.tdtk_links_pair_sums.argtypes = [C.c_int, C.POINTER(C.c_void_p), _dp, C.POINTER(C.c_void_p), C.c_double,
                                       C.c_uint32, C.POINTER(PairSums)]
    L.tdtk_solve_spd.argtypes = [_dp, _dp, C.c_int, _dp]
    L.tdtk_solve_chol_upper.argtypes = [_dp, _dp, C.c_int, _dp]
    L.tdtk_invert.argtypes = [_dp, C.c_int, _dp]
    L.tdtk_last_kernel_ms.argtypes = [_dp]
    L.tdtk_count_visits.argtypes = [C.c_void_p, _dp, C.c_size_t, C.c_double, _u64p]
    L.tdtk_comm_unique_id.argtypes = [C.c_char_p]
    L.tdtk_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tdtk_comm_destroy.argtypes = [C.c_void_p]
    L.tdtk_comm_destroy.restype = None
    L.tdtk_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u64p]
    L.tdtk_graph_exchange.argtypes = [C.c_void_p, _dp, C.c_size_t]
    L.tdtk_comm_rccl_world.argtypes = [C.c_void_p]
    L.tdtk_graph_deal_links.argtypes = [C.c_int, _ip, _ip, _u64p, C.c_int, C.c_int, _ip]
    L.tdtk_graph_iteration.argtypes = [C.c_int, C.c_void_p, C.c_int, _ip, _ip, C.c_int, _ip, C.POINTER(C.c_void_p), _dp,
                                       C.POINTER(C.c_void_p), C.c_double, C.c_int, _dp, _dp, _dp, _dp,
                                       C.POINTER(C.c_void_p), _dp, _dp, _dp]
    L.tdtk_elch_graph_balancer.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, C.c_int, C.c_int, _dp]
    L.tdtk_pair_sums_merge.argtypes = [C.c_int, C.POINTER(PairSums), C.POINTER(PairSums)]
    L.tdtk_graph_links.argtypes = [C.c_int, _dp, C.c_double, C.c_int, _ip, _ip, C.c_int, C.POINTER(C.c_int)]
    L.tdtk_last_timings.argtypes = [_dp]
    L.tdtk_kernel_timing.argtypes = [C.c_int]
    L.tdtk_visit_counting.argtypes = [C.c_int, C.c_int]
    L.tdtk_visit_counters.argtypes = [C.c_int, _u64p]
    L.tdtk_measure_bandwidth.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int, _dp]
    L.tdtk_icp_index_hashes.argtypes = [C.c_int]
    L.tdtk_icp_last_hashes.argtypes = [_u64p, C.c_int, C.POINTER(C.c_int)]
    L.tdtk_host_tree_layout.argtypes = [_dp, C.c_size_t, C.c_int, _ip, _u64p]
    L.tdtk_host_m4inv.argtypes = [_dp, _dp]
    L.tdtk_host_mmult.argtypes = [_dp, _dp, _dp]
    L.tdtk_host_mmult.restype = None
    for f in ("tdtk_host_euler_to_matrix4", "tdtk_host_matrix4_to_euler", "tdtk_host_quat_to_matrix4", "tdtk_host_matrix4_to_quat"):
        getattr(L, f).argtypes = [_dp, _dp, _dp]
        getattr(L, f).restype = None
    L.tdtk_lum_assemble_solve.argtypes = [C.c_int, _ip, _ip, _dp, _dp, C.c_int, _dp, _dp, _dp]
    L.tdtk_graph_block_doubles.argtypes = [C.c_int]
    L.tdtk_graph_link_blocks.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), _dp, C.POINTER(C.c_void_p), C.c_double, _dp]
    L.tdtk_graph_solve_update.argtypes = [C.c_int, C.c_int, _ip, _ip, _dp, C.c_int, _dp, _dp, _dp, _dp,
                                          C.POINTER(C.c_void_p), _dp, _dp, _dp]
    L.tdtk_scan_mark_original.argtypes = [C.c_void_p]
    L.tdtk_scan_download_original.argtypes = [C.c_void_p, _dp]
    L.tdtk_tree_create_from_scans.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.tdtk_tree_create_from_scan.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.tdtk_point_point_error.argtypes = [C.c_void_p, _dp, C.c_void_p, C.c_double, C.c_double, C.POINTER(C.c_uint64), _dp]
    L.tdtk_scans_transform2.argtypes = [C.c_int, C.POINTER(C.c_void_p), _dp, _dp]
    L.tdtk_reduce_octree.argtypes = [_dp, C.c_size_t, C.c_double, C.c_int, _dp, C.POINTER(C.c_size_t)]
    L.tdtk_reduce_octree_nrpts.argtypes = [_dp, C.c_size_t, C.c_double, C.c_int, C.c_int, C.c_int, _dp, C.POINTER(C.c_size_t)]
    L.tdtk_normals_apx_knn.argtypes = [_dp, C.c_size_t, C.c_int, _dp, C.c_double, C.c_int, _dp, _ip]
    L.tdtk_scan_calc_normals.argtypes = [C.c_void_p, C.c_int, _dp, C.c_double]
    L.tdtk_io_read_uos.argtypes = [C.c_char_p, C.c_double, C.c_double, C.POINTER(_dp), C.POINTER(C.c_size_t)]
    L.tdtk_io_free.argtypes = [C.c_void_p]
    L.tdtk_io_free.restype = None
    L.tdtk_io_read_pose.argtypes = [C.c_char_p, _dp, _dp]
    L.tdtk_io_write_frames.argtypes = [C.c_char_p, _dp, C.POINTER(C.c_int), C.c_size_t, C.c_int]
    _libs[_current] = L
    return L


def check(rc):
    if rc != 0:
        raise TdtkError(rc, lib().tdtk_last_error().decode("utf-8", "replace"))


def device_count():
    return int(lib().tdtk_device_count())


def build_respeculated():
    """tree builds of this process that were redone in order (tdtk_build_respeculated)"""
    f = lib().tdtk_build_respeculated
    f.restype = C.c_uint64
    return int(f())


def pool_trim():
    """give the device memory kept from destroyed trees / scans back to the driver; returns the bytes released"""
    f = lib().tdtk_pool_trim
    f.restype = C.c_size_t
    return int(f())


def version():
    return lib().tdtk_version().decode()


def dptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def iptr(a):
    return a.ctypes.data_as(_ip) if a is not None else None


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
