#!/bin/bash
# GPU box, round 4: kernel trace of the 81K-point tree build (bundled scan), every launch in order with gaps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4t
cat > /tmp/tt.py <<'PY'
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
t = importlib.import_module("3dtk_amd")
z = np.load(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests", "golden", "dat_scans.npz"))
pts = z["scan000"]
for rep in range(6):
    t0 = time.perf_counter(); kd = t.KDtree(pts, 20); print("tree %.3f ms" % ((time.perf_counter() - t0) * 1e3), kd.info().get("max_depth"))
PY
python /tmp/tt.py 2>&1 | tail -6 | tee gpurun_out/r4t/plain.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4t/tt -o p -- python /tmp/tt.py 2>&1 | tail -4
python - <<'PY' | tee $GRAFT_REPO_ROOT/gpurun_out/r4t/trace.txt
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r4t/tt/**/p_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = max(i for i, r in enumerate(rows) if "k_init" in r["Kernel_Name"])
sel = rows[last:]
t0 = int(sel[0]["Start_Timestamp"])
lvl = -1
prev_end = t0
busy = 0.0
for r in sel:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("tdtk::", "")
    if "rocprim" in n: short = "rocprim:" + ("init" if "init_lookback" in n else "scan")
    if short.startswith("k_measure"): lvl += 1
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    st = (int(r["Start_Timestamp"]) - t0) / 1e3
    gap = (int(r["Start_Timestamp"]) - prev_end) / 1e3
    prev_end = max(prev_end, int(r["End_Timestamp"]))
    busy += dur
    print("L%02d  +%8.1f us  gap %6.1f  dur %7.1f us  %s" % (lvl, st, gap, dur, short[:50]))
print("total span %.1f us, sum of durations %.1f us, launches %d" % ((int(sel[-1]["End_Timestamp"]) - t0) / 1e3, busy, len(sel)))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r4t/tt
