#!/bin/bash
# lab: k_fin_wave stopped after k - 1 levels (TDTK_FW_DEBUG=k; the trees are not valid): where a subtree's time goes
cd "$GRAFT_REPO_ROOT"
cat > /tmp/fwdbg.py <<'PY'
import importlib, os, sys, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
rng = np.random.default_rng(7); pts = rng.uniform(-1000.0, 1000.0, (10000000, 3))
for rep in range(3):
    try: kd = t.KDtree(pts, 20)
    except Exception as e: pass
PY
for k in 1 0; do
  rm -rf gpurun_out/fw; mkdir -p gpurun_out/fw
  (cd /tmp; export TMPDIR=/tmp; TDTK_LIB=lab TDTK_FW_DEBUG=$k timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/fw -o p -- python /tmp/fwdbg.py > /dev/null 2>&1)
  f=$(find gpurun_out/fw -name "p_kernel_stats.csv" | head -1)
  echo "levels k=$k: $(grep k_fin_wave $f | cut -d, -f1-5)"
done
