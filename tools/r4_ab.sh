#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4h
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals > gpurun_out/r4h/b$i.json 2>gpurun_out/r4h/b$i.err
python -c "import json;d=json.load(open('gpurun_out/r4h/b$i.json'));print('s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tree or build or verify or edge or octree or config1 or dat_" 2>&1 | tail -4
TDTK_LIB=lab TDTK_BUILD_TRACE=1 python /dev/stdin <<'PY' 2>&1 | tail -12
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "dat_scans.npz"))
pts = z["scan000"]
for rep in range(3):
    t0 = time.perf_counter(); kd = t.KDtree(pts, 20); print("tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), kd.verify(), flush=True)
u = np.random.default_rng(1).uniform(-100, 100, (40000, 3))
for rep in range(3):
    t0 = time.perf_counter(); kd = t.KDtree(u, 20); print("uniform 40K tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), kd.verify(), flush=True)
u = np.random.default_rng(2).uniform(-100, 100, (1000000, 3))
for rep in range(3):
    t0 = time.perf_counter(); kd = t.KDtree(u, 20); print("uniform 1M tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"), kd.verify(), flush=True)
PY
