#!/usr/bin/env python
"""bench.py -- the driver's measurement contract for the slam6D ICP hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched under torch.distributed.run)

N = 1  workload "icp"  = BASELINE.json configs[1]: synthetic 1M-vs-1M point-to-point ICP
       (D = T_gt^-1 (M + N(0,1)), shuffled; -a 1 -d 25 -b 20).
       One step = one full icp6D::match iteration on the resident scan: apply the previous
       alignxf to the 1M data points, 1M kd-tree FindClosest, fused pair sums, D2H of 67 doubles,
       closed-form solve.  value = NN correspondences (queries) per second, whole job.
N > 1  workload "graphslam" = configs[3]: 64 scans x 1M points on a closed loop, one step =
       one lum6DEuler iteration (-G 1): links dealt round-robin to the ranks, one whole-scan
       correspondence pass + two reductions per link, ONE all-reduce of the dense normal
       equations over RCCL, redundant solve, pose update.  Total work is fixed -> "strong".
       (--workload graphslam --gpus 1 gives the 1-GPU point of that curve.)

Rank 0 prints ONE JSON line on stdout, at most 6 KB: the contract's keys + `roofline` + `cpu_baseline` + the key numbers of
the other legs.  Everything else that was measured (the full headline record and every leg: tree_build_1gpu, normals_1gpu,
doicp_small_scans, graphslam_1gpu, c5_shape_1gpu) goes to bench_legs.json beside this file and, one compact line per
leg, to stderr.  What the fields mean is written down once, in DESIGN.md section 6 -- not in the record.
Inputs are resident in HBM when the timed region starts; oracle/ is touched by the cpu_baseline legs only (whose first
sample also serves as a parity spot-check of the GPU indices, outside the timed region).
"""
import argparse
import ctypes as C
import importlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that
# share a queue do not overlap.  The graph-SLAM link passes run on three streams, and PyTorch's process group + RCCL
# bring streams of their own: with 4 queues a LUM iteration of configs[3] takes 15.9 ms, with 8 it takes 13.3 (measured
# under torch.distributed.run).  Read when the runtime initialises, so it is set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
L2_PEAK_GBS = 34500.0      # MI355X_MICROARCH.md: aggregate L2 bandwidth of the 8 XCDs
PMC_ROUND = "r06"
PMC_ROUND_C5 = "r06"
LINE_MAX = 6144           # the driver parses the last stdout line; round 5's 21.7 KB line did not parse
NUM_SIMD = 1024            # 256 CUs x 4 SIMDs


def pmc_file_for(steps, warmup):
    """The committed summary of the rocprofv3 --pmc passes over THIS command line (tools/profile_bench.sh <tag> <steps>
    <warmup> -> tools/summarize_profiles.py): one file per (steps, warmup), because the timed iterations of a 20-step run
    right behind the initial pose are not those of a 100-step run."""
    return "%s_pmc_bench_s%d_w%d.json" % (PMC_ROUND, int(steps), int(warmup))


def pmc_kernel(kernel, fname, steps=None, warmup=None):
    """Per-launch PMC averages of `kernel` over the timed region, or None.  When steps / warmup are given the summary is
    REFUSED unless it was taken with exactly those (counters and kernel times of different runs are not combined)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", fname)))
        if steps is not None and (int(d.get("steps", -1)) != int(steps) or int(d.get("warmup", -1)) != int(warmup)):
            return None
        return d["kernels"][kernel]
    except Exception:
        return None


def pmc_traffic_bytes(k):
    """HBM-side (fabric) bytes per launch: FETCH_SIZE doubled per the guide's gfx950 correction (calibrated here on
    k_transform's known 24 MB stream) + WRITE_SIZE, both reported in KiB."""
    if not k or "FETCH_SIZE_KiB" not in k or "WRITE_SIZE_KiB" not in k:
        return None
    return (2.0 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"]) * 1024.0


def measured_bandwidth(local):
    """stream-copy and L2-resident read bandwidth of THIS box (tdtk_measure_bandwidth), GB/s"""
    tdtk = importlib.import_module("3dtk_amd")
    g = C.c_double(0.0)
    out = {}
    for name, kind, nbytes in (("hbm_copy", 0, 1 << 30), ("l2_read", 1, 16 << 20)):
        rc = tdtk.lib().tdtk_measure_bandwidth(int(local), kind, nbytes, 5, C.byref(g))
        out[name] = g.value if rc == 0 else None
    return out


class visit_counting:
    """with visit_counting(dev, mode) as vc: ... ; vc.read() -> (internal nodes, buckets, bucket points, queries) of the
    FindClosest launches issued inside.  mode 1: the launches' own walk (warm start, deferred quick check);
    mode 2: the REFERENCE's walk over the same queries (every search cold, kdTreeImpl.h:345-383) -- SURVEY 8(d)'s
    n_int / n_pts, the figure `roofline.bytes_per_query` is made of.  Results, and so a loop's path, are the same."""

    def __init__(self, dev, mode=1):
        self.dev = int(dev)
        self.mode = int(mode)
        self.L = importlib.import_module("3dtk_amd").lib()

    def __enter__(self):
        self.L.tdtk_visit_counting(self.dev, self.mode)
        return self

    def read(self):
        c = (C.c_uint64 * 8)()
        self.L.tdtk_visit_counters(self.dev, c)
        return int(c[0]), int(c[1]), int(c[2]), int(c[3])

    def read_ann(self):
        c = (C.c_uint64 * 8)()
        self.L.tdtk_visit_counters(self.dev, c)
        return int(c[4]), int(c[5]), int(c[6])

    def __exit__(self, *exc):
        self.L.tdtk_visit_counting(self.dev, 0)


class kernel_timing:
    """with kernel_timing(): ... -- HIP events around the library's search / pair-sum launches (tdtk_kernel_timing):
    off in the product because the events cost ~10 us per ICP iteration; `icp.last["nn_ms"]` etc. are 0 outside."""

    def __enter__(self):
        self.L = importlib.import_module("3dtk_amd").lib()
        self.was = self.L.tdtk_kernel_timing(1)
        return self

    def __exit__(self, *exc):
        self.L.tdtk_kernel_timing(self.was)


def visits(counts):
    c_int, c_leaf, c_pts, nq = counts
    nq = max(1, nq)
    return {"internal": c_int / nq, "leaves": c_leaf / nq, "points": c_pts / nq}


def counter_bounds(pk, k_ms, comp_bytes):
    """The utilisations that can rank kernels (all < 1) from a committed rocprofv3 --pmc summary `pk` of the same launch
    (DESIGN.md section 6 defines each): fabric bytes against compulsory bytes and the HBM peak, L2 hit rate, how busy each
    CU's vector L1 was, the share of VALU lane-slots that did work, the share of a wave's cycles spent waiting."""
    b = {"compulsory_hbm": {"bytes": comp_bytes, "frac": comp_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}}
    traffic = pmc_traffic_bytes(pk)
    if traffic:
        b["hbm_traffic_pmc"] = {"bytes": traffic, "frac": traffic / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "refetch_factor": traffic / comp_bytes, "write_bytes": pk["WRITE_SIZE_KiB"] * 1024.0}
    if not pk:
        return b
    hit, miss = pk.get("TCC_HIT_sum"), pk.get("TCC_MISS_sum")
    if hit is not None and miss is not None:
        b["l2_hit_rate"] = hit / max(1.0, hit + miss)
    if pk.get("GRBM_GUI_ACTIVE") and pk.get("SQ_ACTIVE_INST_VALU"):
        cyc = pk["GRBM_GUI_ACTIVE"] / 8.0                    # the counter sums the 8 XCDs
        b["valu_busy"] = pk["SQ_ACTIVE_INST_VALU"] * 4.0 / (NUM_SIMD * cyc)
        if pk.get("SQ_THREAD_CYCLES_VALU"):
            b["lane_efficiency"] = pk["SQ_THREAD_CYCLES_VALU"] / (pk["SQ_ACTIVE_INST_VALU"] * 64.0)
        if pk.get("TCP_GATE_EN2_sum"):
            b["vector_l1_busy"] = pk["TCP_GATE_EN2_sum"] / 256.0 / cyc
        if pk.get("TCP_PENDING_STALL_CYCLES_sum"):
            b["vector_l1_stalled_on_fills"] = pk["TCP_PENDING_STALL_CYCLES_sum"] / 256.0 / cyc
        if pk.get("TCP_TCC_READ_REQ_LATENCY_sum") and pk.get("TCP_TCC_READ_REQ_sum"):
            b["l1_miss_latency_cycles"] = pk["TCP_TCC_READ_REQ_LATENCY_sum"] / pk["TCP_TCC_READ_REQ_sum"]
    if pk.get("SQ_WAVE_CYCLES") and pk.get("SQ_WAIT_ANY"):
        b["wave_wait_share"] = pk["SQ_WAIT_ANY"] / pk["SQ_WAVE_CYCLES"]
    if pk.get("SQ_INSTS_VMEM_WR"):
        b["vmem_write_wave_instructions"] = pk["SQ_INSTS_VMEM_WR"]
    return b


def search_roofline(kernel, k_ms, nq_per_launch, counts_ref, counts_walked, comp_bytes, pk, bw, pmc_file):
    """The `roofline` object of a search launch.  achieved = SURVEY 8(d)'s ALGORITHMIC bytes per launch -- 24 B query +
    64 B per internal node + 24 B per bucket point + 4 B index, with the node / point counts of the REFERENCE's walk
    (kdTreeImpl.h:345-383, cold, radius maxdist2) over exactly the queries of the timed launches (`visits_per_query`) --
    / average launch duration (HIP events).  `visits_walked` = what the kernel's own walk visited (warm start, deferred
    quick check) and its byte ratio to the reference's; `bounds` = the utilisations from the committed counter summary."""
    v = visits(counts_ref)
    bq = algorithmic_bytes_per_query(v["internal"], v["points"])
    achieved = bq * nq_per_launch / (k_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(pk), "kernel_ms": k_ms,
         "bytes_per_query": bq, "queries_per_launch": nq_per_launch, "visits_per_query": v,
         "nn_per_s_kernel_only": nq_per_launch / (k_ms * 1e-3)}
    if counts_walked is not None:
        w = visits(counts_walked)
        w["bytes_ratio"] = algorithmic_bytes_per_query(w["internal"], w["points"]) / bq
        r["visits_walked"] = w
    b = counter_bounds(pk, k_ms, comp_bytes)
    b["peak_measured_copy_GBs"] = bw.get("hbm_copy")
    b["peak_measured_l2_GBs"] = bw.get("l2_read")
    b["source"] = {"file": ("profiles/" + pmc_file) if pk else None}
    r["bounds"] = b
    return r


def algorithmic_bytes_per_query(n_int, n_pts):
    """24 B query + 64 B per internal node visited + 24 B per leaf point tested +
    4 B index out (properties of tree/query/radius, independent of the implementation)."""
    return 24.0 + 64.0 * n_int + 24.0 * n_pts + 4.0


# --------------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------------
def make_icp_pair(n, seed=42):
    rng = np.random.default_rng(seed)
    m = rng.uniform(-1000.0, 1000.0, (n, 3))
    tdtk = importlib.import_module("3dtk_amd")
    T = tdtk.EulerToMatrix4([10.0, -5.0, 3.0], [0.02, -0.03, 0.05])
    Tinv = tdtk.M4inv(T)
    R = np.array([[Tinv[0], Tinv[4], Tinv[8]], [Tinv[1], Tinv[5], Tinv[9]], [Tinv[2], Tinv[6], Tinv[10]]])
    d = (m + rng.normal(0.0, 1.0, m.shape))[rng.permutation(n)]
    d = d @ R.T + Tinv[12:15]
    return m, np.ascontiguousarray(d), T


def make_graphslam_scans(nscans, npts, seed=7):
    """one world cloud in a 4000x4000x2000 box, scans = points within range
    1500 of poses on a closed circle (radius 800, tangential heading), exactly npts each, in the
    scan frame, + N(0,1) noise; initial poses = truth + accumulated odometry drift."""
    rng = np.random.default_rng(seed)
    nworld = 4 * npts
    world = np.empty((nworld, 3))
    world[:, 0] = rng.uniform(-2000, 2000, nworld)
    world[:, 1] = rng.uniform(-1000, 1000, nworld)     # y is "up" in 3DTK's left-handed frame
    world[:, 2] = rng.uniform(-2000, 2000, nworld)
    tdtk = importlib.import_module("3dtk_amd")
    out = []
    drift_p = np.zeros(3); drift_t = 0.0
    for k in range(nscans):
        ang = 2 * math.pi * k / nscans
        pos = np.array([800 * math.cos(ang), 0.0, 800 * math.sin(ang)])
        theta = np.array([0.0, -ang, 0.0])
        d = world - pos
        sel = np.flatnonzero(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2] < 1500.0 ** 2)
        if len(sel) < npts:
            raise RuntimeError("world cloud too sparse for %d points per scan" % npts)
        sel = rng.choice(sel, npts, replace=False)
        T = tdtk.EulerToMatrix4(pos, theta)
        Ti = tdtk.M4inv(T)
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        loc = world[sel] @ R.T + Ti[12:15] + rng.normal(0.0, 1.0, (npts, 3))
        if k > 0:
            drift_p = drift_p + rng.normal(0.0, 0.3, 3)
            drift_t = drift_t + rng.normal(0.0, math.radians(0.01))
        out.append((pos + drift_p, theta + np.array([0.0, drift_t, 0.0]), np.ascontiguousarray(loc)))
    return out


def make_city(rng, n):
    """points on a ground plane and on the walls of random boxes, 4000 x 4000 footprint, y up (3DTK's frame): the world of the
    configs[4]-shaped scans (bremen_city itself is not on the box; SURVEY 8(d) allows the substitute)"""
    ng = n // 3
    g = np.empty((ng, 3)); g[:, 0] = rng.uniform(-2000, 2000, ng); g[:, 1] = 0.0; g[:, 2] = rng.uniform(-2000, 2000, ng)
    nb = 160
    cx, cz = rng.uniform(-1900, 1900, nb), rng.uniform(-1900, 1900, nb)
    sx, sz, h = rng.uniform(40, 160, nb), rng.uniform(40, 160, nb), rng.uniform(60, 400, nb)
    nw = n - ng
    b = rng.integers(0, nb, nw); face = rng.integers(0, 4, nw)
    u, v = rng.uniform(-1, 1, nw), rng.uniform(0, 1, nw)
    w = np.empty((nw, 3))
    w[:, 1] = v * h[b]
    xs = np.where(face == 0, cx[b] - sx[b], np.where(face == 1, cx[b] + sx[b], cx[b] + u * sx[b]))
    zs = np.where(face >= 2, np.where(face == 2, cz[b] - sz[b], cz[b] + sz[b]), cz[b] + u * sz[b])
    w[:, 0] = xs; w[:, 2] = zs
    return np.concatenate([g, w])


def make_c5_scans(nscans, npts, loop=13, seed=55):
    """configs[4]'s shape (bremen_city reduced: ~10M points per scan, 13 scans): the first `nscans` of `loop` scanner poses on a
    circle of radius 500 over a synthetic city of 3 * npts surface points; a scan = the npts points nearest to its scanner
    (horizontal range), in the scan frame, + N(0, 0.3) noise; initial poses = truth + accumulated odometry drift.  The same
    construction as tests/test_gpu_configs.py::test_config5_ten_million_point_scans_with_normals (which runs all 13)."""
    rng = np.random.default_rng(seed)
    tdtk = importlib.import_module("3dtk_amd")
    W = make_city(rng, 3 * npts)
    out = []
    drift_p, drift_t = np.zeros(3), 0.0
    for k in range(nscans):
        ang = 2 * math.pi * k / loop
        pos = np.array([500 * math.cos(ang), 150.0, 500 * math.sin(ang)])
        th = np.array([0.0, -ang, 0.0])
        d2 = (W[:, 0] - pos[0]) ** 2 + (W[:, 2] - pos[2]) ** 2
        sel = np.argpartition(d2, npts)[:npts]
        Ti = tdtk.M4inv(tdtk.EulerToMatrix4(pos, th))
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        loc = W[sel] @ R.T + Ti[12:15] + rng.normal(0.0, 0.3, (npts, 3))
        if k > 0:
            drift_p = drift_p + rng.normal(0.0, 0.4, 3); drift_t += rng.normal(0.0, 0.0004)
        out.append((pos + drift_p, th + np.array([0.0, drift_t, 0.0]), np.ascontiguousarray(loc)))
    return out


# --------------------------------------------------------------------------------------------
def make_small_scans(nscans=16, raw=400000, seed=11):
    """SURVEY 8(d) C3's shape (hannover1 -s 1 -e 65 -r 10 -i 100 -d 75: a vehicle's scans of a street scene, ~400K raw
    points each, octree-reduced with 10 cm voxels to ~40K; the dataset itself is not on the box): a ground plane, walls and
    boxes, sampled where a scanner on a straight drive sees them (range 1150), in the scan frame with N(0, 0.5) noise; the
    initial poses carry accumulated odometry drift."""
    rng = np.random.default_rng(seed)
    tdtk = importlib.import_module("3dtk_amd")
    out = []
    drift = np.zeros(3); drift_t = 0.0
    for k in range(nscans):
        pos = np.array([60.0 * k, 0.0, 0.0])
        theta = np.array([0.0, 0.002 * k, 0.0])
        n_g = raw * 6 // 10
        n_w = raw - n_g
        # ground: denser near the scanner (1 / r falloff of a rotating scanner), y = 0
        r = 1150.0 * rng.uniform(0.02, 1.0, n_g) ** 1.5
        a = rng.uniform(0, 2 * math.pi, n_g)
        g = np.stack([pos[0] + r * np.cos(a), np.zeros(n_g), r * np.sin(a)], 1)
        # walls along the street (z = +-220) and cross walls every 400 units, 0 .. 120 high
        w = np.empty((n_w, 3))
        side = rng.integers(0, 3, n_w)
        along = pos[0] + rng.uniform(-1150, 1150, n_w)
        w[:, 0] = np.where(side < 2, along, np.round(along / 400.0) * 400.0)
        w[:, 1] = rng.uniform(0, 120, n_w)
        w[:, 2] = np.where(side == 0, 220.0, np.where(side == 1, -220.0, rng.uniform(-220, 220, n_w)))
        world_pts = np.concatenate([g, w]) + rng.normal(0.0, 0.5, (raw, 3))
        T = tdtk.EulerToMatrix4(pos, theta)
        Ti = tdtk.M4inv(T)
        R = np.array([[Ti[0], Ti[4], Ti[8]], [Ti[1], Ti[5], Ti[9]], [Ti[2], Ti[6], Ti[10]]])
        loc = world_pts @ R.T + Ti[12:15]
        if k > 0:
            drift = drift + rng.normal(0.0, 0.4, 3) + np.array([0.5, 0.0, 0.2])
            drift_t = drift_t + rng.normal(0.0, math.radians(0.02))
        out.append((pos + drift, theta + np.array([0.0, drift_t, 0.0]), np.ascontiguousarray(loc)))
    return out


def bench_small_scans(args, local):
    """`doicp_small_scans`: the regime of real data sets -- many scans of a few ten thousand reduced points,
    where a scan costs its preparation (reduction, upload + ordering, tree build) as much as its match.  16 scans of 400K raw
    points, `-r 10` on the device, then icp6D::doICP (-i 100 -d 75 --epsICP 1e-5) with the next scans prepared ahead; per scan:
    reduce / prepare (upload + ordering + tree) / match, and the wall time of the whole doICP."""
    tdtk = importlib.import_module("3dtk_amd")
    nscans = 16
    raw = make_small_scans(nscans)
    # octree reduction on the device (Scan::calcReducedPoints, -r 10)
    red, t_red = [], []
    for (p, th, loc) in raw:
        tdtk.calcReducedPoints(loc[:1000], 10.0, device=local)          # (first call of the process: code objects)
        t0 = time.perf_counter(); r = tdtk.calcReducedPoints(loc, 10.0, device=local); t_red.append(time.perf_counter() - t0)
        red.append(r)
    npts = [len(r) for r in red]

    def run(prefetch):
        S = [tdtk.Scan(p, th, r, device=local) for (p, th, _), r in zip(raw, red)]
        icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), 75.0, 100, quiet=True, epsilonICP=1e-5)
        its = []
        orig = icp.match

        def rec(a, b, pm=0):
            t0 = time.perf_counter(); it = orig(a, b, pm); its.append((it, time.perf_counter() - t0, icp.last["total_ms"])); return it
        icp.match = rec
        t0 = time.perf_counter(); icp.doICP(S, prefetch=prefetch); wall = time.perf_counter() - t0
        builds = [s.getSearchTree().info()["build_ms"] for s in S[:-1]]
        poses = np.stack([s.transMat for s in S])
        for s in S:
            s.release()
        return wall, its, builds, poses
    run(True)                                                              # warm (first builds of these sizes: arenas, pools)
    wall0, its0, builds0, poses0 = run(False)
    wall1, its1, builds1, poses1 = run(True)
    assert np.array_equal(poses0, poses1), "doICP result depends on the prefetch"
    iters = [it for it, _, _ in its1]
    match_ms = [dt * 1e3 for _, dt, _ in its0]
    loop_ms = [lm for _, _, lm in its0]          # the library's own clock around the resident loop (tdtk_icp_match: no preparation in it)
    out = {"scans": nscans, "raw_points_per_scan": len(raw[0][2]), "reduced_points_per_scan": {"min": min(npts), "mean": float(np.mean(npts)), "max": max(npts)},
           "voxel": 10.0, "max_dist_match": 75.0, "max_iterations": 100, "epsilonICP": 1e-5,
           "per_scan_ms": {"reduce": float(np.mean(t_red)) * 1e3, "tree_build": float(np.mean(builds0)),
                           "match": float(np.mean(match_ms)), "doICP_wall_nothing_ahead": wall0 * 1e3 / (nscans - 1),
                           "doICP_wall_three_ahead": wall1 * 1e3 / (nscans - 1)},
           "iterations_per_match": {"mean": float(np.mean(iters)), "max": int(max(iters))},
           "ms_per_iteration": float(np.sum(match_ms) / max(1, sum(it + 1 for it in iters))),
           "loop_ms_per_match": float(np.mean(loop_ms)),
           "loop_us_per_iteration": float(1e3 * np.sum(loop_ms) / max(1, sum(it + 1 for it in iters)))}
    if not args.no_cpu:
        from oracle import orc
        if orc.have_ref():
            # the reference's own TUs on the host: KDtreeIndexed's constructor + full OpenMP-branch ICP iterations of one pair
            a, b = red[0], red[1]
            Ta = tdtk.EulerToMatrix4(raw[0][0], raw[0][1]); Tb = tdtk.EulerToMatrix4(raw[1][0], raw[1][1])
            ga = a.copy(); orc.transform_points(Ta, ga)
            gb = b.copy(); orc.transform_points(Tb, gb)
            t0 = time.perf_counter(); tree = orc.RefTree(ga, 20); tb = time.perf_counter() - t0
            threads = min(int(orc.ref().ref_host_threads()), 16)
            nit = max(2, int(round(np.mean(iters))))
            t0 = time.perf_counter(); tree.icp_iterations(np.eye(4).reshape(16), gb, 75.0 ** 2, threads, nit); tm = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": (tb + tm) * 1e3, "unit": "ms per scan (tree build + match)", "cores": threads, "kind": "reference",
                                   "tree_build_ms": tb * 1e3, "match_ms": tm * 1e3,
                                   "sample": "scans 0/1: KDtreeIndexed ctor over %d points + %d OpenMP-branch iterations" % (len(ga), nit)}
    return out


class _stdout_to_stderr:
    """fd-level redirect: RCCL prints its version banner to stdout when the communicator comes up
    (NCCL_DEBUG=VERSION is set on the pool); stdout must carry the one JSON line only."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def dist_setup(ngpus):
    import torch
    rank, world, local = 0, 1, 0
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:
        import torch.distributed as dist
        rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
        local = int(os.environ.get("LOCAL_RANK", rank))
        if os.environ.get("TDTK_BENCH_BACKEND") == "gloo":
            # test rig only (tests/test_gpu_parity.py): several ranks sharing the GPUs that exist, exchange over gloo
            local = local % torch.cuda.device_count()
            torch.cuda.set_device(local)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local)
            with _stdout_to_stderr():
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
                dist.barrier()                      # brings the communicator (and its banner) up now
                torch.cuda.synchronize()
    else:
        torch.cuda.set_device(0)
    if world != ngpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (ngpus, world))
    return rank, world, local


def barrier_sync(world):
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, local):
    if world == 1:
        return x
    import torch
    import torch.distributed as dist
    on_gpu = dist.get_backend() != "gloo"
    t = torch.tensor([x], dtype=torch.float64, device=torch.device("cuda", local) if on_gpu else None)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline_nn(model, queries, maxd2, budget_s=8.0, check=None, full_iter=None):
    """The reference's KDtreeIndexed::FindClosest (oracle/_ref, kind "reference") or the C
    restatement (kind "port") on this box's host cores, bounded sample of the same workload."""
    from oracle import orc
    if orc.have_ref():
        kind, tree = "reference", orc.RefTree(model, 20)
        threads = min(int(orc.ref().ref_host_threads()), 512)
        run = lambda q, nt: tree.find_closest(q, maxd2, nt)
    else:
        kind, tree = "port", orc.Tree(model, 20)
        threads = int(orc.lib().orc_max_threads())
        run = lambda q, nt: tree.find_closest(q, maxd2, nt)
    first = run(queries[:20000], threads)               # warm; doubles as the parity spot-check of the GPU indices
    first = first[0] if isinstance(first, tuple) else first
    if check is not None:
        assert np.array_equal(np.asarray(first), np.asarray(check)), "parity spot-check failed"
    t0 = time.perf_counter(); run(queries[:200000], 1); t1 = time.perf_counter() - t0
    one = 200000 / t1
    reps, t_all = 0, 0.0
    while t_all < budget_s and reps < 400:
        t0 = time.perf_counter(); run(queries, threads); t_all += time.perf_counter() - t0; reps += 1
    allc = reps * len(queries) / t_all
    out = {"value": allc, "unit": "NN correspondences/s", "cores": threads, "kind": kind,
           "one_thread_value": one, "nn_only_value": allc,
           "sample": "%d FindClosest passes over the same %d queries, %d threads; NN only" % (reps, len(queries), threads)}
    if kind == "reference" and full_iter is not None:
        # the same unit of work as `value` of the GPU line: FULL icp6D::match iterations of the OpenMP branch
        # (icp6D.cc:129-222) -- per-thread getPtPairs chunks incl. the critical-section push_back of 208-byte
        # PtPairs, the Si pass, icp6D_QUAT::Align_Parallel, serial transformReduced -- every piece the reference's
        # own compiled code (oracle/ref_driver.cc: ref_icp_iterations), T = OPENMP_NUM_THREADS = the threads used
        data0, dal = full_iter
        best = None
        for T in sorted({min(threads, 16), min(threads, 64), threads}):
            t0 = time.perf_counter(); _, tr = tree.icp_iterations(dal, data0, maxd2, T, 2); dtT = (time.perf_counter() - t0) / 2
            if best is None or dtT < best[1]:
                best = (T, dtT, tr)
        T, dtT, tr = best
        out.update({"value": len(data0) / dtT, "cores": T, "ms_per_iteration": dtT * 1e3,
                    "first_iteration_pairs": int(tr[0, 0]), "first_iteration_rms": float(tr[0, 1]),
                    "sample": "2 full icp6D::match iterations (OpenMP branch) of the same %d-vs-%d pair from the initial pose, "
                              "best of 16 / 64 / %d threads; nn_only_value: %d FindClosest passes on %d threads"
                              % (len(data0), len(model), threads, reps, threads)})
    return out


# --------------------------------------------------------------------------------------------
def bench_icp(args, rank, world, local):
    tdtk = importlib.import_module("3dtk_amd")
    n = args.points
    m, d, T = make_icp_pair(n)
    model = tdtk.Scan([0, 0, 0], [0, 0, 0], m, device=local)
    data = tdtk.Scan([0, 0, 0], [0, 0, 0], d, device=local)
    tree = model.getSearchTree()
    info = tree.info()
    _ = data.handle
    mini = tdtk.icp6D_QUAT(True)
    # warm-up: W untimed iterations of the same loop (also brings the pose close to T)
    icp_w = tdtk.icp6D(mini, 25.0, max(1, args.warmup), quiet=True, epsilonICP=-1.0)
    icp_w.match(model, data)
    icp = tdtk.icp6D(mini, 25.0, args.steps, quiet=True, epsilonICP=-1.0)   # eps < 0: exactly K steps
    barrier_sync(world)
    t0 = time.perf_counter()
    it = icp.match(model, data)
    barrier_sync(world)
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, world, local)
    steps = it + 1
    assert steps == args.steps, (steps, args.steps)
    last = icp.last
    pose_err = float(np.abs(data.get_transMat() - T).max())
    # Device time of the timed launches.  The library's HIP events around its search and pair-sum launches are a
    # profiling switch that is off in the product (four marker packets, two waits and two read-outs cost ~10 us of an
    # iteration), so the timed region above ran without them; the same W + K iterations are run again on a second
    # data scan with the events on (same launches: the loop is deterministic, the final RMS is compared).
    rep_t = tdtk.Scan([0, 0, 0], [0, 0, 0], d, device=local)
    _ = rep_t.handle
    tdtk.icp6D(mini, 25.0, max(1, args.warmup), quiet=True, epsilonICP=-1.0).match(model, rep_t)
    icp_t = tdtk.icp6D(mini, 25.0, args.steps, quiet=True, epsilonICP=-1.0)
    with kernel_timing():
        tt0 = time.perf_counter(); icp_t.match(model, rep_t); dt_ev = time.perf_counter() - tt0
    assert icp_t.last["rms"] == last["rms"] and icp_t.last["pairs"] == last["pairs"], "timing replay diverged"
    last_t = icp_t.last
    del rep_t

    # Algorithmic bytes of exactly the timed launches: the loop is deterministic (the results of a search do not depend on
    # how it walks), so further data scans run through the same W + K iterations with the instrumented kernels see the
    # queries the timed launches saw -- checked through the final RMS.  Counted twice: the REFERENCE's walk (mode 2:
    # every search cold, SURVEY 8(d)'s n_int / n_pts, what `bytes_per_query` is made of) and the launches' own walk.
    def counted_replay(mode):
        rep = tdtk.Scan([0, 0, 0], [0, 0, 0], d, device=local)
        _ = rep.handle
        tdtk.icp6D(mini, 25.0, max(1, args.warmup), quiet=True, epsilonICP=-1.0).match(model, rep)
        icp_c = tdtk.icp6D(mini, 25.0, args.steps, quiet=True, epsilonICP=-1.0)
        with visit_counting(local, mode) as vc:
            icp_c.match(model, rep)
            cnt = vc.read()
        assert icp_c.last["rms"] == last["rms"] and icp_c.last["pairs"] == last["pairs"], "counting replay diverged"
        assert cnt[3] == n * steps, cnt
        rep.release()
        return cnt
    counts_ref = counted_replay(2)
    counts_own = counted_replay(1)
    cur = data.get_xyz_reduced()
    k_ms = last_t["nn_ms"] / steps                        # HIP-event time of k_search, per launch
    sums_ms = last_t["sums_ms"] / steps                   # ... and of the pair-sum kernels behind it
    bw = measured_bandwidth(local)
    pfile = pmc_file_for(steps, args.warmup)
    pk = pmc_kernel("k_search [timed region]", pfile, steps, args.warmup)
    sums_inside = 262144 <= n < 256 * 7168   # (one generation of waves: each adds up its own slab, FUSE 3)
    # compulsory bytes: the tree once (64 B per node, 32 B per point) + per query x,y,z read and written back (fused
    # transform), the hit written and read back (warm start) and, with the sums inside, query + hit + hit point again
    comp = info["n_internal"] * 64 + n * 32 + n * (24 + 24 + 4 + 4 + (60 if sums_inside else 0))
    roof = search_roofline("k_search_refill (search + fused transform" + (" + pair sums" if sums_inside else "") + ")",
                           k_ms, n, counts_ref, counts_own, comp, pk, bw, pfile)

    # the same 1M queries through the host-buffer entry point (H2D of queries, in-call binning,
    # search, D2H of indices + distances): the PCIe-inclusive rate -- reported, never the `value`
    t_h = []
    for _ in range(3):
        th0 = time.perf_counter(); tree.FindClosestBatch(cur, 625.0); t_h.append(time.perf_counter() - th0)
    host_path = {"value": n / min(t_h), "unit": "NN correspondences/s", "ms": min(t_h) * 1e3}
    prep = {}
    tp0 = time.perf_counter(); t2_ = tdtk.KDtree(m, 20, device=local); prep["tree_create_ms"] = (time.perf_counter() - tp0) * 1e3
    tp0 = time.perf_counter(); s2_ = tdtk.Scan([0, 0, 0], [0, 0, 0], d, device=local); _ = s2_.handle
    prep["scan_create_ms"] = (time.perf_counter() - tp0) * 1e3
    del t2_, s2_
    # the model scan's own tree again, now that the process is warm (the first build of a process also pays for the
    # scratch arena, the code objects and rocPRIM's first use: reported as first_build_ms)
    warm = []
    for _ in range(3):
        s3_ = tdtk.Scan([0, 0, 0], [0, 0, 0], m, device=local)
        warm.append(s3_.getSearchTree().info()["build_ms"])
        del s3_
    first_build_ms = info["build_ms"]
    info = dict(info, build_ms=min(warm))

    # Does the loop that was timed converge to the pose the data was generated with?  The timed K steps alone need not
    # (20 steps from the initial pose do not): the same loop is continued, untimed, with the reference's stopping rule
    # (--epsICP 1e-5) and the pose it ends at is compared with T_gt.  Noise of sigma = 1 on 1e6 points pins the pose to
    # ~1e-3; the assertion leaves a factor of ten.
    icp_f = tdtk.icp6D(mini, 25.0, 400, quiet=True, epsilonICP=1e-5)
    more = icp_f.match(model, data) + 1
    pose_err_converged = float(np.abs(data.get_transMat() - T).max())
    convergence = {"further_iterations": more, "pose_max_abs_err": pose_err_converged, "rms": icp_f.last["rms"]}
    assert pose_err_converged < 2e-2, "ICP did not converge to the generating pose: %r" % (convergence,)

    out = {
        "metric": "NN correspondences/sec (1M-vs-1M pairwise ICP, full iteration)",
        "value": n * steps / dt, "unit": "NN correspondences/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[1]: synthetic %d-vs-%d uniform pair, point-to-point ICP -a 1 -d 25 -b 20, 1xMI355X" % (n, n),
                   "points": n, "bucket": 20, "max_dist_match": 25.0, "minimizer": "QUAT",
                   "tree": {"internal": info["n_internal"], "leaves": info["n_leaves"], "depth": info["max_depth"]},
                   "tree_build_ms": info["build_ms"], "tree_upload_ms": info["upload_ms"]},
        "icp_iters_per_s": steps / dt,
        "pairs_last": last["pairs"], "rms_last": last["rms"], "pose_max_abs_err": pose_err, "convergence": convergence,
        "host_buffer_path": host_path, "per_scan_preparation": prep,
        "pair_sums_ms": sums_ms, "outside_kernels_ms": dt * 1e3 / steps - k_ms - sums_ms,
        "ms_per_step_with_kernel_events": dt_ev * 1e3 / steps,
        # for whoever reads a kernel trace of this command: the timed region is search launches [first, first + steps)
        "search_launches_before_timed_region": max(1, args.warmup),
        "roofline": roof,
    }
    # tree build (A1) on the device, against the bytes its levels move: every level streams the 24-B points + 8 B of
    # permutation / keys in and out (64 B per point and level)
    levels = info["max_depth"]
    tb_bytes = float(levels) * n * 64.0
    out["tree_build_1gpu"] = {"ms": info["build_ms"], "first_build_ms": first_build_ms, "points": n, "levels": levels,
                              "roofline": {"bound": "hbm", "kernel": "device tree build (level passes + subtree finisher)",
                                           "achieved": tb_bytes / (info["build_ms"] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                           "unit": "GB/s", "frac": tb_bytes / (info["build_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                           "traffic": None}}
    if rank == 0 and not args.no_cpu:
        gi, _ = tree.FindClosestBatch(cur[:20000], 625.0)
        out["cpu_baseline"] = cpu_baseline_nn(m, cur, 625.0, check=gi, full_iter=(d, model.dalignxf))
    if world == 1 and not args.no_normals:
        # Scan::calcNormals (k = 10, eps = 1.0) on the resident data scan: tree build + k-NN + PCA
        t_n, k_n = [], []
        tm4 = (C.c_double * 4)()
        for _ in range(4):
            tn0 = time.perf_counter(); data.calcNormals(); t_n.append(time.perf_counter() - tn0)
            tdtk.lib().tdtk_last_timings(tm4); k_n.append(tm4[2])
        with visit_counting(local) as vc:
            data.calcNormals()
            a_split, a_leaf, a_q = vc.read_ann()
        kn_ms = float(np.mean(k_n[1:]))
        # algorithmic bytes per point: the point in (24) + 32 B per splitting node + 24 B per leaf point tested +
        # the k neighbours gathered for mean / covariance (24 B each) + the normal out (24)
        bp = 24.0 + 32.0 * a_split / a_q + 24.0 * a_leaf / a_q + 24.0 * 10 + 24.0
        ach = bp * n / (kn_ms * 1e-3) / 1e9
        pk = pmc_kernel("k_ann_normals<10>", "r01_normals_pmc.json")   # tools/profile_normals.sh
        out["normals_1gpu"] = {"value": n / min(t_n), "unit": "points/s", "ms": min(t_n) * 1e3, "points": n, "k": 10, "eps": 1.0,
                               "roofline": {"bound": "hbm", "kernel": "k_ann_normals", "achieved": ach, "peak": HBM_PEAK_GBS,
                                            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(pk),
                                            "kernel_ms": kn_ms, "bytes_per_point": bp,
                                            "visits_per_point": {"split_nodes": a_split / a_q, "leaf_points": a_leaf / a_q},
                                            "whole_call_ms": min(t_n) * 1e3}}
        if not args.no_cpu:
            from oracle import orc as _orc
            ns = min(n, 100000)
            which = "ref" if _orc.have_ref() else "oracle"
            tc0 = time.perf_counter(); _orc.normals_apx_knn(cur[:ns], 10, [0.0, 0.0, 0.0], 1.0, which); tc = time.perf_counter() - tc0
            out["normals_1gpu"]["cpu_baseline"] = {"value": ns / tc, "unit": "points/s", "cores": 1,
                                                   "kind": "reference" if which == "ref" else "port",
                                                   "sample": "first %d points of the scan (serial calcNormals)" % ns}
    if world == 1 and not args.no_small_scans:
        out["doicp_small_scans"] = bench_small_scans(args, local)
    if world == 1 and args.workload == "auto" and not args.no_graphslam_base:
        # the 1-GPU point of the graph-SLAM strong-scaling curve (the N>1 runs of this script measure
        # configs[3]); reported beside the headline so scaling can be read against the same workload
        del model, data, tree
        import copy
        ga = copy.copy(args); ga.steps, ga.warmup = 10, 3
        g1 = bench_graphslam(ga, rank, world, local)
        out["graphslam_1gpu"] = {k: g1[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "lum_iters_per_s", "roofline", "sharded_step_rehearsal")}
        out["graphslam_1gpu"]["workload"] = g1["config"]["workload"]
    if world == 1 and args.workload == "auto" and not args.no_c5:
        try:
            del model, data, tree
        except NameError:
            pass
        out["c5_shape_1gpu"] = bench_c5(args, local)
    return out


def rehearse_shards(tdtk, gs, scans, nscans, npts, full_ms, local):
    """One GPU: what an N-rank step of the sharded LUM iteration would cost -- a PREDICTION, no multi-GPU box is available to
    the build session.  For world in 1, 2, 4, 8 the links are dealt exactly as tdtk_graph_deal_links does; every rank's share
    is timed on this GPU (tdtk_lum_links over the share, best of 3, with the previous round's scan moves queued on every scan
    as tdtk_graph_iteration leaves them: since round 4 a share carries out the moves of the scans it reads) and the slowest
    share taken; the rest of a step (graph, table set-up, exchange of 28 KB, solve, pose update) = this run's measured
    1-GPU step minus the same measurement of all links."""
    capi = importlib.import_module("3dtk_amd._capi")
    L = capi.lib()
    g = tdtk.Graph(nscans, 500.0 ** 2, 20, scans)
    hs_all = (C.c_void_p * (nscans - 1))(*[scans[k].handle for k in range(1, nscans)])
    wig = np.ascontiguousarray(np.tile(tdtk.EulerToMatrix4([1e-4, -1e-4, 1e-4], [1e-7, -1e-7, 1e-7]), (nscans - 1, 1)))
    wig_inv = np.ascontiguousarray(np.stack([tdtk.M4inv(m) for m in wig]))

    def time_links(idx):
        nl = len(idx)
        if nl == 0:
            return 0.0
        first = (C.c_void_p * nl)(*[scans[g.getLink(i, 0)].getSearchTree()._h for i in idx])
        second = (C.c_void_p * nl)(*[scans[g.getLink(i, 1)].handle for i in idx])
        dal = np.ascontiguousarray(np.stack([scans[g.getLink(i, 0)].dalignxf for i in idx]))
        Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
        best = 1e9
        for _ in range(3):
            capi.check(L.tdtk_scans_transform2(nscans - 1, hs_all, capi.dptr(wig), capi.dptr(wig_inv)))
            t0 = time.perf_counter()
            capi.check(L.tdtk_lum_links(nl, first, capi.dptr(dal), second, 625.0, capi.dptr(Cm), capi.dptr(CD), m, capi.dptr(ss)))
            best = min(best, time.perf_counter() - t0)
        return best * 1e3
    all_ms = time_links(list(range(g.getNrLinks())))
    rest = max(0.0, full_ms - all_ms)
    pred = {}
    for world in (1, 2, 4, 8):
        shares = [gs.shard_links(g, r, world, scans) for r in range(world)]
        per = [time_links(sh) for sh in shares]
        pred[str(world)] = {"links_per_rank": [len(x) for x in shares], "slowest_share_ms": max(per), "rest_ms": rest,
                            "predicted_step_ms": max(per) + rest}
    return {"kind": "prediction (one-GPU rehearsal, not measured on N GPUs)", "all_links_ms": all_ms, "rest_ms": rest, "by_world": pred}


def bench_graphslam(args, rank, world, local):
    tdtk = importlib.import_module("3dtk_amd")
    gs = importlib.import_module("3dtk_amd.graphslam")
    sl = importlib.import_module("3dtk_amd.slam6d")
    import torch
    nscans, npts = args.scans, args.points
    raw = make_graphslam_scans(nscans, npts)
    scans = [tdtk.Scan(p, th, loc, device=local) for (p, th, loc) in raw]
    del raw
    g0 = tdtk.Graph(nscans, 500.0 ** 2, 20, scans)
    nlinks = g0.getNrLinks()
    mine = gs.shard_links(g0, rank, world, scans)
    # materialise what this rank touches (trees of the link sources, points of the link targets),
    # several at a time: a tree build keeps only a few wavefronts busy
    need_tree = sorted({g0.getLink(i, 0) for i in mine})
    need_pts = sorted({g0.getLink(i, 1) for i in mine} - set(need_tree))
    tdtk.prepare_scans([scans[k] for k in need_tree], trees=True, threads=8)
    tdtk.prepare_scans([scans[k] for k in need_pts], trees=False, threads=8)
    import torch.distributed as tdist
    on_dist = tdist.is_available() and tdist.is_initialized()
    dev = torch.device("cuda", local) if (on_dist and tdist.get_backend() != "gloo") else None
    # N > 1 over RCCL: the library's own communicator (tdtk_comm_*, ncclAllReduce inside tdtk_graph_iteration);
    # torch.distributed only hands the 128-byte unique id to the ranks and provides the barriers around the timed
    # region.  TDTK_FORCE_ALLREDUCE=1 runs the same path with a 1-rank communicator.  The gloo rig of the test
    # suite (several ranks sharing one GPU, which RCCL refuses) keeps the torch.distributed exchange.
    comm = None
    with _stdout_to_stderr():          # RCCL prints its version banner when a communicator comes up
        if on_dist and world > 1 and dev is not None:
            comm = gs.NativeComm(rank, world, local, gs.torch_id_bcast(dev))
        elif world == 1 and os.environ.get("TDTK_FORCE_ALLREDUCE") == "1":
            comm = gs.NativeComm(0, 1, local)
    use_torch_exchange = on_dist and world > 1 and dev is None
    nn_ms = [0.0]
    # one pair of HIP events per step around the step's last search launch (the roofline's kernel time): on for this
    # leg -- two markers in a step of ~12 ms, unlike the four per 0.25 ms of the ICP loop
    tdtk.lib().tdtk_kernel_timing(1)

    def step():
        gr = tdtk.Graph(nscans, 500.0 ** 2, 20, scans)   # slam6D.cc:525-532: fresh Graph + 1 LUM iteration
        if use_torch_exchange:
            r = gs.lum_iteration_native(gr, scans, 625.0, None, None)
        else:
            r = gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, scans, 625.0, comm)
        ms = C.c_double(0.0)
        tdtk.lib().tdtk_last_kernel_ms(C.byref(ms))       # the last link's k_search of this rank
        nn_ms[0] += ms.value
        return r, gr.getNrLinks()

    for _ in range(args.warmup):
        step()
    nn_ms[0] = 0.0
    barrier_sync(world)
    t0 = time.perf_counter()
    links_done = 0
    for _ in range(args.steps):
        ts = time.perf_counter()
        ret, nl = step()
        links_done += nl
        if os.environ.get("TDTK_BENCH_VERBOSE"):
            print("step %.2f ms ret %.4f" % ((time.perf_counter() - ts) * 1e3, ret), file=sys.stderr)
    barrier_sync(world)
    dt = max_over_ranks(time.perf_counter() - t0, world, local)
    tdtk.lib().tdtk_kernel_timing(0)
    queries = links_done * npts                         # one whole-scan NN pass per link
    k_ms = nn_ms[0] / max(1, args.steps)                # one sampled k_search launch per step
    # algorithmic bytes of this rank's link passes: two more steps with the instrumented kernels (poses have converged to
    # ~1e-3 per step by now, so they visit what the timed steps visited to within a fraction of a percent): the reference's
    # walk (mode 2, every search cold: SURVEY 8(d)) and the launches' own (warm start from the link's previous hits)
    with visit_counting(local, 2) as vc:
        step()
        counts_ref = vc.read()
    with visit_counting(local, 1) as vc:
        step()
        counts_own = vc.read()
    my_links = max(1, len(gs.shard_links(tdtk.Graph(nscans, 500.0 ** 2, 20, scans), rank, world, scans)))
    links_sums_inside = npts >= 262144
    # All link passes of a rank go out in launches of up to 128 links (k_search_refill_multi); the HIP events sit around the
    # LAST launch of a step.  achieved = algorithmic bytes of that launch / its duration; beside it the aggregate over
    # the whole step (bytes of all this rank's link searches / wall time of the step, exchange, solve and pose update
    # included in the denominator).
    batch = 128
    batched = batch > 1 and my_links > 1 and npts >= 262144
    groups = (my_links + batch - 1) // batch if batched else my_links
    last_links = my_links - batch * (groups - 1) if batched else 1
    pk = pmc_kernel("k_search (several links per launch)" if batched else "k_search", PMC_ROUND + "_graphslam_pmc.json")
    if pk and batched and groups != 1:
        pk = None                                             # (the committed passes are of one launch holding every link)
    # compulsory: every tree walked once (64 B per node + 32 B per point + 12 B of fp32 shadow) + per query x,y,z read, hit written
    # and read back, and for the sums query + hit + hit point again
    trees = len({g0.getLink(i, 0) for i in mine}) if len(mine) else 1
    info0 = scans[g0.getLink(mine[0], 0)].getSearchTree().info() if len(mine) else {"n_internal": 0}
    comp = trees * (info0["n_internal"] * 64 + npts * 44) * last_links / my_links + last_links * npts * (24 + 4 + 4 + (60 if links_sums_inside else 0))
    bw = measured_bandwidth(local) if rank == 0 else {}
    kname = ("k_search_refill_multi (%d links per launch%s)" % (last_links, ", sums inside" if links_sums_inside else "")) if batched else "k_search (link passes on streams side by side)"
    roof = search_roofline(kname, k_ms if k_ms > 0 else dt * 1e3 / args.steps, last_links * npts, counts_ref, counts_own, comp, pk, bw, PMC_ROUND + "_graphslam_pmc.json")
    roof.update({"links_in_that_launch": last_links, "launches_per_step": groups, "links_this_rank": my_links})
    agg = roof["bytes_per_query"] * my_links * npts / (dt / args.steps) / 1e9
    roof["whole_step"] = {"achieved": agg, "frac": agg / HBM_PEAK_GBS}
    if roof["traffic"]:
        roof["bounds"]["hbm_traffic_pmc"]["GB_per_link"] = roof["traffic"] / max(1, last_links) / 1e9
    exchange = ("RCCL ncclAllReduce inside tdtk_graph_iteration, %d collectives issued" % comm.n_allreduce()) if comm is not None \
        else ("torch.distributed gloo (test rig)" if use_torch_exchange else "none (one rank)")
    # what actually ran, for whoever reads the N > 1 line: the size of the communicator as RCCL itself reports it
    # (ncclCommCount; NativeComm / tdtk_comm_create refuse anything but the world asked for) and every rank's share of the links
    owners = gs.link_owners(tdtk.Graph(nscans, 500.0 ** 2, 20, scans), world, scans)
    links_per_rank = [int((owners == r).sum()) for r in range(world)]
    rccl_world = comm.rccl_world if comm is not None else None
    if comm is not None and rccl_world != args.gpus:
        raise SystemExit("bench.py: RCCL communicator has %d ranks but --gpus %d" % (rccl_world, args.gpus))
    rehearsal = None
    if world == 1 and comm is None and not getattr(args, "no_rehearsal", False):
        rehearsal = rehearse_shards(tdtk, gs, scans, nscans, npts, dt * 1e3 / args.steps, local)
    if comm is not None:
        barrier_sync(world)
        comm.close()                  # ncclCommDestroy now, on every rank together, not at interpreter shutdown
    return {
        "metric": "NN correspondences/sec (graph-SLAM lum6DEuler iteration, links sharded)",
        "value": queries / dt, "unit": "NN correspondences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "configs[3]: synthetic %d scans x %d pts, graph-SLAM -G 1 (lum6DEuler), one iteration per step, "
                               "%d links over %d ranks, one fp64 all-reduce of %d doubles" % (nscans, npts, nlinks, world, 42 * nlinks),
                   "scans": nscans, "points": npts, "links": nlinks},
        "lum_iters_per_s": args.steps / dt, "last_ret": ret, "sharded_step_rehearsal": rehearsal,
        "exchange": exchange, "rccl_world": rccl_world, "links_per_rank": links_per_rank,
        "roofline": roof,
    }


def bench_c5(args, local):
    """`c5_shape_1gpu`: the regime of BASELINE.json configs[4] (bremen_city reduced, ~10M points per scan, lum6DEuler): the one
    workload whose trees (a 10M-point tree: 0.34 GB of point slots + 0.13 GB of shadows + 74 MB of nodes) do not fit the
    256 MB Infinity Cache, i.e. where HBM bandwidth is a real bound.  `--c5-scans` (default 4) of the 13 scans of
    tests/test_gpu_configs.py::test_config5_... (which checks all 13 against the oracle); per scan: upload + ordering, tree
    build, calcNormals; one whole-scan correspondence pass (10M queries against a 10M-point tree, cold: Scan::getPtPairs),
    ICP iterations at that size (device-resident icp6D::match, warm start from the second on), one lum6DEuler round over
    the chain + closure links (lum6Deuler.cc:94-251 per link); the reference's TUs on the host beside it."""
    tdtk = importlib.import_module("3dtk_amd")
    gs = importlib.import_module("3dtk_amd.graphslam")
    capi = importlib.import_module("3dtk_amd._capi")
    L = tdtk.lib()
    nscans, npts = max(2, int(args.c5_scans)), int(args.c5_points)
    maxd2 = 100.0                                         # -d 10 (the test's value: ~8 point spacings on the surfaces)
    tg0 = time.perf_counter(); raw = make_c5_scans(nscans, npts); t_gen = time.perf_counter() - tg0
    capi.pool_trim()
    tm4 = (C.c_double * 4)()
    S, t_up, t_tree, infos = [], [], [], []
    for (p, th, loc) in raw:
        s = tdtk.Scan(p, th, loc, device=local)
        t0 = time.perf_counter(); _ = s.handle; t_up.append((time.perf_counter() - t0) * 1e3)
        t0 = time.perf_counter(); kd = s.getSearchTree(); t_tree.append((time.perf_counter() - t0) * 1e3)
        infos.append(kd.info())
        S.append(s)
    info = infos[0]
    # the tree again on a warm process (the first build of a size pays for arenas and pools)
    warm_build = []
    for _ in range(2):
        t2 = tdtk.KDtree.from_scan(S[0].handle, S[0].n, 20, local); warm_build.append(t2.info()["build_ms"]); del t2
    bw = measured_bandwidth(local)
    pfile = PMC_ROUND_C5 + "_c5_pmc.json"
    # what a search launch must touch at least: the tree's records + the query stream
    slots = info.get("n_point_slots", int(npts * 1.07))
    tree_bytes = info["n_internal"] * (64 + 48) + slots * (32 + 6)          # the single-pass kernel filters on the 16-bit shadow (6 B per slot)
    tree_bytes_links = info["n_internal"] * (64 + 48) + slots * (32 + 12)   # the several-links launch on the fp32 groups (12 B per slot)
    out = {"scans": nscans, "points_per_scan": npts, "max_dist_match2": maxd2, "generation_s": t_gen,
           "tree": {"internal": info["n_internal"], "leaves": info["n_leaves"], "depth": info["max_depth"], "device_bytes": info["device_bytes"]},
           "per_scan_ms": {"upload_and_ordering": {"first": t_up[0], "mean_rest": float(np.mean(t_up[1:]))},
                           "tree_build_device": {"each": [i["build_ms"] for i in infos], "warm": min(warm_build)},
                           "tree_create_wall": {"first": t_tree[0], "mean_rest": float(np.mean(t_tree[1:]))}}}
    lv = info["max_depth"]
    tb_bytes = float(lv) * npts * 64.0
    out["tree_build_roofline"] = {"bound": "hbm", "kernel": "device tree build (level passes + subtree finisher)", "achieved": tb_bytes / (min(warm_build) * 1e-3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": tb_bytes / (min(warm_build) * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}
    # ---- one whole-scan correspondence pass, cold (Scan::getPtPairs: no warm start, sums by k_accum behind the search)
    a, b = S[0], S[1]
    tdtk.Scan.getPtPairs(a, b, 0, 0, maxd2)
    walls, ks, ss = [], [], []
    with kernel_timing():
        for _ in range(max(2, args.c5_reps)):
            t0 = time.perf_counter(); r = tdtk.Scan.getPtPairs(a, b, 0, 0, maxd2); walls.append((time.perf_counter() - t0) * 1e3)
            L.tdtk_last_timings(tm4); ks.append(tm4[0]); ss.append(tm4[1])
    with visit_counting(local, 2) as vc:               # (a cold pass: the launch's own walk IS the reference's)
        rc_ = tdtk.Scan.getPtPairs(a, b, 0, 0, maxd2)
        counts = vc.read()
    assert rc_["n"] == r["n"] and counts[3] == npts, (rc_["n"], r["n"], counts)
    k_ms = float(np.mean(ks)); sums_ms = float(np.mean(ss))
    comp = tree_bytes + npts * (24 + 4)
    kname = "k_search [whole-scan pass, 10M queries]"
    roof = search_roofline("k_search_refill (whole-scan pass, cold)", k_ms, npts, counts, None, comp, pmc_kernel(kname, pfile), bw, pfile)
    out["whole_scan_pass"] = {"value": npts / (min(walls) * 1e-3), "unit": "NN correspondences/s", "ms": min(walls), "k_search_ms": k_ms,
                              "pair_sums_ms": sums_ms, "pairs": int(r["n"]), "roofline": roof}
    # ---- ICP at this size: K iterations of the resident loop (transform fused, warm start from the second iteration)
    d_icp = tdtk.Scan(raw[1][0], raw[1][1], raw[1][2], device=local); _ = d_icp.handle
    K = max(2, args.c5_icp_iters)
    icp = tdtk.icp6D(tdtk.icp6D_QUAT(True), math.sqrt(maxd2), K, quiet=True, epsilonICP=-1.0)
    with kernel_timing():
        t0 = time.perf_counter(); it = icp.match(a, d_icp); dt_icp = time.perf_counter() - t0
    last = dict(icp.last)
    cnt_icp = {}
    for mode in (2, 1):                                  # the reference's walk, then the launches' own (same loop, same results)
        d_cnt = tdtk.Scan(raw[1][0], raw[1][1], raw[1][2], device=local); _ = d_cnt.handle
        icp_c = tdtk.icp6D(tdtk.icp6D_QUAT(True), math.sqrt(maxd2), K, quiet=True, epsilonICP=-1.0)
        with visit_counting(local, mode) as vc:
            icp_c.match(a, d_cnt)
            cnt_icp[mode] = vc.read()
        assert icp_c.last["rms"] == last["rms"] and icp_c.last["pairs"] == last["pairs"], "counting replay diverged"
        d_cnt.release()
    ki_ms = last["nn_ms"] / (it + 1)
    kname_i = "k_search [icp6D::match at 10M]"
    out["icp_10M"] = {"iterations": it + 1, "ms_per_iteration": dt_icp * 1e3 / (it + 1), "value": npts * (it + 1) / dt_icp, "unit": "NN correspondences/s",
                      "k_search_ms": ki_ms, "pair_sums_ms": last["sums_ms"] / (it + 1), "pairs_last": last["pairs"], "rms_last": last["rms"],
                      "roofline": search_roofline("k_search_refill (tdtk_icp_match at 10M: fused transform, warm start)", ki_ms, npts,
                                                  cnt_icp[2], cnt_icp[1], comp + npts * 48, pmc_kernel(kname_i, pfile), bw, pfile)}
    del d_icp
    # ---- calcNormals at this size (the -z / point-to-plane leg of configs[4])
    t_n, k_n = [], []
    for _ in range(3):
        t0 = time.perf_counter(); b.calcNormals(); t_n.append((time.perf_counter() - t0) * 1e3)
        L.tdtk_last_timings(tm4); k_n.append(tm4[2])
    with visit_counting(local) as vc:
        b.calcNormals()
        a_split, a_leaf, a_q = vc.read_ann()
    kn_ms = float(np.mean(k_n[1:]))
    bp = 24.0 + 32.0 * a_split / max(1, a_q) + 24.0 * a_leaf / max(1, a_q) + 24.0 * 10 + 24.0
    pkn = pmc_kernel("k_ann_normals<10>", pfile)
    out["normals"] = {"value": npts / (min(t_n) * 1e-3), "unit": "points/s", "ms": min(t_n), "kernel_ms": kn_ms, "ann_tree_build_ms": min(t_n) - kn_ms,
                      "roofline": {"bound": "hbm", "kernel": "k_ann_normals", "achieved": bp * npts / (kn_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": bp * npts / (kn_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(pkn), "bytes_per_point": bp,
                                   "visits_per_point": {"split_nodes": a_split / max(1, a_q), "leaf_points": a_leaf / max(1, a_q)}}}
    # ---- one lum6DEuler round: chain links + closures (every pair of scans further apart than one step)
    links = [(i, i + 1) for i in range(nscans - 1)] + [(i, j) for i in range(nscans) for j in range(i + 2, nscans)]
    links = links[:max(nscans - 1, args.c5_links)] if args.c5_links else links
    def step():
        gr = tdtk.Graph(nscans, links=links)
        return gs.graph_iteration_comm(gs.GRAPH_LUMEULER, gr, S, maxd2, None)
    L.tdtk_kernel_timing(1)
    step()
    t_round, k_round = [], []
    for _ in range(max(1, args.c5_rounds)):
        t0 = time.perf_counter(); ret = step(); t_round.append((time.perf_counter() - t0) * 1e3)
        ms = C.c_double(0.0); L.tdtk_last_kernel_ms(C.byref(ms)); k_round.append(ms.value)
    L.tdtk_kernel_timing(0)
    with visit_counting(local, 2) as vc:
        step()
        cnt_l = vc.read()
    with visit_counting(local, 1) as vc:
        step()
        cnt_l_own = vc.read()
    nl = len(links)
    kl_ms = float(np.mean(k_round))
    kname_l = "k_search (several links per launch)"
    trees_walked = len({f for f, _ in links})
    comp_l = trees_walked * tree_bytes_links + nl * npts * (24 + 4 + 60)
    out["lum_round"] = {"links": nl, "ms": min(t_round), "value": nl * npts / (min(t_round) * 1e-3), "unit": "NN correspondences/s", "last_ret": ret,
                        "link_launch_ms": kl_ms,
                        "roofline": search_roofline("k_search_refill_multi (%d links in one launch, sums inside)" % nl, kl_ms, nl * npts,
                                                    cnt_l, cnt_l_own, comp_l, pmc_kernel(kname_l, pfile), bw, pfile)}
    # ---- the reference on the host: its own TUs (oracle/_ref) on a stated sample
    if not args.no_cpu:
        from oracle import orc
        if orc.have_ref():
            model = a.xyz_reduced_original
            q = b.get_xyz_reduced()
            Ai, ok = orc.m4inv(a.dalignxf)
            q = np.ascontiguousarray(q @ np.array([[Ai[0], Ai[4], Ai[8]], [Ai[1], Ai[5], Ai[9]], [Ai[2], Ai[6], Ai[10]]]).T + Ai[12:15]) if not np.array_equal(Ai, np.eye(4).reshape(16)) else q
            t0 = time.perf_counter(); rt = orc.RefTree(model, 20); tb = time.perf_counter() - t0
            threads = min(int(orc.ref().ref_host_threads()), 512)
            ns = min(npts, 2_000_000)
            sel = np.ascontiguousarray(q[:: max(1, npts // ns)][:ns])
            gi, _ = a.getSearchTree().FindClosestBatch(sel[:20000], maxd2)
            ri = rt.find_closest(sel[:20000], maxd2, threads)
            ri = ri[0] if isinstance(ri, tuple) else ri
            assert np.array_equal(np.asarray(ri), np.asarray(gi)), "parity spot-check against the reference's KDtreeIndexed failed"
            t0 = time.perf_counter(); rt.find_closest(sel, maxd2, threads); tq = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns / tq, "unit": "NN correspondences/s", "cores": threads, "kind": "reference", "tree_build_s": tb,
                                   "sample": "KDtreeIndexed over scan 0 (%d points) + FindClosest of every %d-th query of scan 1 (%d)" % (npts, max(1, npts // ns), ns)}
            del rt
    for s in S:
        s.release()
    capi.pool_trim()
    return out


# --------------------------------------------------------------------------------------------
# the record: one compact stdout line, everything else to bench_legs.json + one stderr line per leg
# --------------------------------------------------------------------------------------------
LEG_KEYS = ("tree_build_1gpu", "normals_1gpu", "doicp_small_scans", "graphslam_1gpu", "c5_shape_1gpu")
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data")


def _sig(x, digits=6):
    """floats of the secondary fields at `digits` significant digits (the contract's own numbers are never rounded)"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if math.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def compact_roofline(r):
    """the `roofline` block of the stdout line: the contract's seven fields + what its `frac` is made of + the six
    utilisations from the counter summary (DESIGN.md section 6 says what each is)"""
    if not r:
        return None
    b = r.get("bounds", {})
    out = {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "bytes_per_query")}
    out["visits_per_query"] = {k: _get(r, "visits_per_query", k) for k in ("internal", "points")}
    if r.get("visits_walked"):
        out["visits_walked"] = {k: r["visits_walked"].get(k) for k in ("internal", "points", "bytes_ratio")}
    out["bounds"] = {"compulsory_hbm_frac": _get(b, "compulsory_hbm", "frac"), "hbm_traffic_pmc_frac": _get(b, "hbm_traffic_pmc", "frac"),
                     "refetch_factor": _get(b, "hbm_traffic_pmc", "refetch_factor"), "l2_hit_rate": b.get("l2_hit_rate"),
                     "vector_l1_busy": b.get("vector_l1_busy"), "lane_efficiency": b.get("lane_efficiency"),
                     "measured_copy_GBs": b.get("peak_measured_copy_GBs")}
    out["source"] = _get(b, "source", "file")
    for k in ("links_in_that_launch", "launches_per_step"):
        if k in r:
            out[k] = r[k]
    return _sig(out)


def leg_numbers(res):
    """the key number(s) of every other leg, for the stdout line (the legs themselves: bench_legs.json)"""
    g, c5, sm = res.get("graphslam_1gpu") or {}, res.get("c5_shape_1gpu") or {}, res.get("doicp_small_scans") or {}
    out = {
        "tree_build_1M_ms": _get(res, "tree_build_1gpu", "ms"),
        "normals_1M_ms": _get(res, "normals_1gpu", "ms"),
        "small_scans_loop_us_per_iteration": sm.get("loop_us_per_iteration"),
        "small_scans_match_ms": _get(sm, "per_scan_ms", "match"),
        "graphslam_1gpu_ms_per_step": g.get("ms_per_step"),
        "graphslam_link_launch_ms": _get(g, "roofline", "kernel_ms"),
        "graphslam_roofline_frac": _get(g, "roofline", "frac"),
        "graphslam_GB_per_link": _get(g, "roofline", "bounds", "hbm_traffic_pmc", "GB_per_link"),
        "graphslam_predicted_step_ms_8_ranks": _get(g, "sharded_step_rehearsal", "by_world", "8", "predicted_step_ms"),
        "c5_tree_build_10M_ms": _get(c5, "per_scan_ms", "tree_build_device", "warm"),
        "c5_whole_scan_pass_k_search_ms": _get(c5, "whole_scan_pass", "k_search_ms"),
        "c5_whole_scan_pass_frac": _get(c5, "whole_scan_pass", "roofline", "frac"),
        "c5_whole_scan_pass_hbm_traffic_frac": _get(c5, "whole_scan_pass", "roofline", "bounds", "hbm_traffic_pmc", "frac"),
        "c5_icp_10M_ms_per_iteration": _get(c5, "icp_10M", "ms_per_iteration"),
        "c5_normals_10M_ms": _get(c5, "normals", "ms"),
        "c5_link_launch_ms": _get(c5, "lum_round", "link_launch_ms"),
    }
    return _sig({k: v for k, v in out.items() if v is not None}, 5)


def build_line(res):
    """The ONE stdout line (<= LINE_MAX bytes) from a full result dict: the contract's keys, a short `config`, the compact
    `roofline`, `cpu_baseline` and the legs' key numbers.  Strict JSON (a NaN / Infinity anywhere raises)."""
    line = {k: res.get(k) for k in CONTRACT_KEYS}
    cfg = res.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "points", "scans", "links", "bucket", "max_dist_match", "minimizer") if k in cfg}
    for k in ("icp_iters_per_s", "lum_iters_per_s", "pairs_last", "rms_last", "pose_max_abs_err", "pair_sums_ms", "outside_kernels_ms",
              "last_ret", "exchange", "rccl_world", "links_per_rank", "search_launches_before_timed_region"):
        if res.get(k) is not None:
            line[k] = _sig(res[k])
    if res.get("convergence"):
        line["converged_pose_max_abs_err"] = _sig(res["convergence"]["pose_max_abs_err"])
    line["roofline"] = compact_roofline(res.get("roofline"))
    cb = res.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = _sig({k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "one_thread_value", "nn_only_value", "ms_per_iteration") if k in cb})
        line["cpu_baseline"]["value"] = cb["value"]
    legs = leg_numbers(res)
    if legs:
        line["legs"] = legs
        line["legs_file"] = "bench_legs.json"
    text = json.dumps(line, allow_nan=False, separators=(",", ":"))
    if len(text) >= LINE_MAX:
        raise RuntimeError("bench.py: the stdout line is %d bytes (limit %d): move fields to bench_legs.json" % (len(text), LINE_MAX))
    return text


def emit(res):
    """bench_legs.json (the full record), one compact line per leg on stderr, the ONE line on stdout (last)."""
    full = json.dumps(res, allow_nan=False, indent=1, default=lambda o: o.item() if hasattr(o, "item") else str(o))
    try:
        with open(os.path.join(ROOT, "bench_legs.json"), "w") as f:
            f.write(full + "\n")
    except OSError as e:
        print("bench.py: bench_legs.json not written: %s" % e, file=sys.stderr)
    for k in LEG_KEYS:
        if res.get(k) is not None:
            print(json.dumps({"leg": k, **_sig(res[k], 5)}, separators=(",", ":"), default=lambda o: o.item() if hasattr(o, "item") else str(o)), file=sys.stderr)
    sys.stderr.flush()
    print(build_line(res))
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 100 (icp) / 10 (graphslam)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 10 (icp) / 3 (graphslam)")
    ap.add_argument("--workload", choices=["auto", "icp", "graphslam", "c5"], default="auto")
    ap.add_argument("--points", type=int, default=1000000)
    ap.add_argument("--scans", type=int, default=64)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-normals", action="store_true", help="N=1 only: skip the calcNormals measurement")
    ap.add_argument("--no-small-scans", action="store_true", help="N=1 only: skip the doicp_small_scans leg")
    ap.add_argument("--no-rehearsal", action="store_true",
                    help="graph-SLAM at N=1: skip the one-GPU rehearsal of the sharded step (the profiles: every search dispatch is then a full step's)")
    ap.add_argument("--no-c5", action="store_true", help="N=1 only: skip the configs[4]-shape leg (10M-point scans)")
    ap.add_argument("--c5-scans", type=int, default=4, help="scans of the configs[4]-shape leg (of the 13 of the full-size test)")
    ap.add_argument("--c5-points", type=int, default=10000000)
    ap.add_argument("--c5-reps", type=int, default=4, help="whole-scan passes timed")
    ap.add_argument("--c5-icp-iters", type=int, default=6)
    ap.add_argument("--c5-rounds", type=int, default=3, help="lum6DEuler rounds timed")
    ap.add_argument("--c5-links", type=int, default=0, help="0: chain + all closures among the scans")
    ap.add_argument("--no-graphslam-base", action="store_true",
                    help="N=1 only: skip the extra 1-GPU graph-SLAM measurement (the base of the N>1 curve)")
    args = ap.parse_args()
    capi = importlib.import_module("3dtk_amd._capi")
    if not os.path.exists(os.path.join(ROOT, "3dtk_amd", "lib3dtk_hip.so")):
        capi.build_extension()
    rank, world, local = dist_setup(args.gpus)
    wl = args.workload
    if wl == "auto":
        wl = "icp" if world == 1 else "graphslam"
    if wl == "c5" and args.steps is None:
        args.steps, args.warmup = 1, 0
    # explicit --steps / --warmup are always honoured; the defaults depend on the workload
    # (a LUM step is ~84 whole-scan passes, an ICP step is one)
    if args.steps is None:
        args.steps = 100 if wl == "icp" else 10
    if args.warmup is None:
        args.warmup = 10 if wl == "icp" else 3
    if wl == "c5":
        if world != 1:
            raise SystemExit("bench.py: --workload c5 is a one-GPU leg")
        c5 = bench_c5(args, local)
        # (a leg on its own: the line carries the leg's whole-scan pass as its headline so that it stays a valid record)
        p_ = c5["whole_scan_pass"]
        res = {"metric": "NN correspondences/sec (configs[4] shape, one cold whole-scan pass of 10M queries)", "value": p_["value"],
               "unit": p_["unit"], "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": p_["ms"], "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "configs[4] shape: %d synthetic city scans x %d points" % (c5["scans"], c5["points_per_scan"]), "points": c5["points_per_scan"]},
               "roofline": p_["roofline"], "cpu_baseline": c5.get("cpu_baseline"), "c5_shape_1gpu": c5}
    else:
        res = bench_icp(args, rank, world, local) if wl == "icp" else bench_graphslam(args, rank, world, local)
    if rank == 0:
        emit(res)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
