"""GPU box (1 GPU): predict the N-rank step time of the sharded LUM iteration.

For world in 1,2,4,8 the links are dealt exactly as graphslam.shard_links does; every rank's share
is timed on this one GPU (batched tdtk_lum_links, best of 3) and the slowest share is taken.  The
rest of the step (Graph construction, all-reduce through a 1-rank NCCL group -- H2D, collective,
D2H of the same 1.1 MB --, SPD solve, pose update) is timed as the full iteration minus its link
part.  No multi-GPU box is available to the build session; this is the evidence the N>1 numbers
in DESIGN.md rest on."""
import importlib, os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
import bench
t = importlib.import_module("3dtk_amd"); gs = importlib.import_module("3dtk_amd.graphslam")
capi = importlib.import_module("3dtk_amd._capi")
ns, npts = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
raw = bench.make_graphslam_scans(ns, npts)
scans = [t.Scan(p, th, loc) for (p, th, loc) in raw]
g = t.Graph(ns, 500.0 ** 2, 20, scans)
nl_all = g.getNrLinks()
for i in range(nl_all):
    scans[g.getLink(i, 0)].getSearchTree(); _ = scans[g.getLink(i, 1)].handle
L = capi.lib()


hs_all = (C.c_void_p * (ns - 1))(*[scans[k].handle for k in range(1, ns)])
_wig = np.ascontiguousarray(np.tile(t.EulerToMatrix4([1e-4, -1e-4, 1e-4], [1e-7, -1e-7, 1e-7]), (ns - 1, 1)))
_wig_inv = np.ascontiguousarray(np.stack([t.M4inv(m) for m in _wig]))


def queue_moves():
    """what the pose update of the previous round leaves behind: two in-place transforms queued on every scan but the
    first (round 4: carried out by the link passes that read the scan, so they belong to the share's time)"""
    capi.check(L.tdtk_scans_transform2(ns - 1, hs_all, capi.dptr(_wig), capi.dptr(_wig_inv)))


def time_links(idx):
    nl = len(idx)
    first = (C.c_void_p * nl)(*[scans[g.getLink(i, 0)].getSearchTree()._h for i in idx])
    second = (C.c_void_p * nl)(*[scans[g.getLink(i, 1)].handle for i in idx])
    dal = np.ascontiguousarray(np.stack([scans[g.getLink(i, 0)].dalignxf for i in idx]))
    Cm = np.empty((nl, 36)); CD = np.empty((nl, 6)); m = (C.c_uint64 * nl)(); ss = np.empty(nl)
    best = 1e9
    for _ in range(3):
        queue_moves()
        t0 = time.perf_counter()
        capi.check(L.tdtk_lum_links(nl, first, capi.dptr(dal), second, 625.0, capi.dptr(Cm), capi.dptr(CD), m, capi.dptr(ss)))
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


dev = torch.device("cuda", 0)
full = []
for _ in range(4):
    t0 = time.perf_counter()
    gr = t.Graph(ns, 500.0 ** 2, 20, scans)
    gs.lum_iteration_native(gr, scans, 625.0, None, dev)
    torch.cuda.synchronize()
    full.append((time.perf_counter() - t0) * 1e3)
full_ms = min(full[1:])
links_ms = time_links(list(range(nl_all)))
rest = full_ms - links_ms
print("links %d, full iteration %.2f ms, all links %.2f ms, rest (graph + all-reduce + solve + pose update) %.2f ms"
      % (nl_all, full_ms, links_ms, rest))
base = None
pred = {}
for world in (1, 2, 4, 8):
    shares = [gs.shard_links(g, r, world) for r in range(world)]
    per = [time_links(s) if len(s) else 0.0 for s in shares]
    step = max(per) + rest
    base = base or step
    pred[str(world)] = {"links_per_rank": [len(x) for x in shares], "slowest_share_ms": round(max(per), 3), "rest_ms": round(rest, 3),
                        "predicted_step_ms": round(step, 3)}
    print("world %d: links/rank %s  slowest share %.2f ms  predicted step %.2f ms  speedup %.2f  efficiency %.0f%%"
          % (world, [len(s) for s in shares], max(per), step, base / step, 100 * base / step / world))
import json
print("PREDICTION " + json.dumps(pred))
dist.destroy_process_group()
