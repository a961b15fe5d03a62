#!/bin/bash
# GPU box: kernel trace of one 1M-point tree build (the last of a few), per-launch durations of the build kernels in order
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/tt
cat > /tmp/tt.py <<'PY'
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
import bench
m, d, T = bench.make_icp_pair(1000000)
for rep in range(3):
    t0 = time.perf_counter(); kd = t.KDtree(m, 20); print("tree %.2f ms" % ((time.perf_counter() - t0) * 1e3))
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tt -o p -- python /tmp/tt.py 2>&1 | tail -4
python - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/tt/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last build: from the last k_init on
last = max(i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("tdtk::k_init") or "k_init" in r["Kernel_Name"])
sel = rows[last:]
t0 = int(sel[0]["Start_Timestamp"])
lvl = -1
for r in sel:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("tdtk::", "")
    if "rocprim" in n: short = "rocprim:" + ("init" if "init_lookback" in n else "scan") + (":BSum" if "BSum" in n else ":BPre" if "BPre" in n else ":u64" if "unsigned long" in n else ":u32")
    if short.startswith("k_measure"): lvl += 1
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    st = (int(r["Start_Timestamp"]) - t0) / 1e3
    if dur > 3:
        print("L%02d  +%8.1f us  %7.1f us  %s" % (lvl, st, dur, short[:60]))
print("total span %.1f us" % ((int(sel[-1]["End_Timestamp"]) - t0) / 1e3))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/tt
