"""3dtk_amd -- MI355X-native slam6D ICP correspondence + alignment hot path.

This package is a thin host-side mirror of the reference's interface for this path
(``KDtree`` / ``SearchTree.getPtPairs`` / ``icp6Dminimizer`` / ``icp6D`` / ``Graph`` /
``lum6DEuler``; same names, argument meaning and error behaviour) over the C ABI of
``lib3dtk_hip.so`` (include/tdtk_hip.h).  All compute happens in the hand-written HIP
kernels of ``3dtk_amd/csrc``; there is no CPU fallback: if the extension is missing or no
gfx950 device is present, the compute entry points raise.

The directory name starts with a digit, so import it with
``importlib.import_module("3dtk_amd")`` (tests/conftest.py and bench.py do).
"""
# Importing the package changes nothing in the process environment.  A host program with HIP streams of its own (PyTorch,
# RCCL) that wants the link passes of graph-SLAM to overlap should start with GPU_MAX_HW_QUEUES=8 (the HIP runtime maps
# all streams of a process onto 4 hardware queues by default and reads the variable once, when it initialises):
# bench.py does that for itself; INTEGRATION.md, "Streams and hardware queues".

from ._capi import (TdtkError, lib, build_extension, device_count, version, PairSums,  # noqa: F401
                    ALGO_QUAT, ALGO_SVD, ALGO_APX, ALGO_NAPX, CLOSEST_POINT,
                    CLOSEST_POINT_ALONG_NORMAL_SIMPLE, CLOSEST_PLANE_SIMPLE,
                    WANT_APX, WANT_NAPX, WANT_LUM, WANT_GAPX, WANT_MOM2,
                    ALGO_ORTHO, ALGO_DUAL, ALGO_HELIX, ALGO_LUMEULER, ALGO_LUMQUAT, ALGO_QUAT_SCALE)
from .slam6d import (KDtree, Scan, icp6Dminimizer, icp6D_QUAT, icp6D_SVD, icp6D_APX,  # noqa: F401
                     icp6D_ORTHO, icp6D_DUAL, icp6D_HELIX, icp6D_LUMEULER, icp6D_LUMQUAT, icp6D_QUAT_SCALE,
                     icp6D_NAPX, icp6D, Graph, lum6DEuler, lum6DQuat, ghelix6DQ2, gapx6D, QuatToMatrix4, Matrix4ToQuat, M4inv, MMult, M4identity,
                     EulerToMatrix4, Matrix4ToEuler, host_tree_layout, calculateNormalsApxKNN, MetaScan, read_uos, read_pose,
                     openDirectory, closeDirectory, saveFrames, matchGraph6Dautomatic, calcReducedPoints,
                     computeGraph6Dautomatic, matchGraph6Dautomatic_clpairs, prepare_scans, loopSlam6D, elch6Deuler, elch6Dquat, elch6DunitQuat, elch6Dslerp,
                     graph_balancer)
