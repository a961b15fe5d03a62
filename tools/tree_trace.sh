#!/bin/bash
N=${1:-81000}
cd "$GRAFT_REPO_ROOT"; rm -rf gpurun_out/tt; mkdir -p gpurun_out/tt
python tools/tree_trace.py $N 2>&1 | tail -1
TDTK_LIB=lab TDTK_BUILD_TRACE=2 python tools/tree_trace.py $N 2>&1 | tail -4
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tt -o p -- python $GRAFT_REPO_ROOT/tools/tree_trace.py $N 2>&1 | tail -1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tt/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last build: from the last k_init on
names = [r["Kernel_Name"] for r in rows]
last = max(i for i, n in enumerate(names) if "k_init" in n)
# walk back to the scan-creation kernels of that build? just print from 12 kernels before k_init
start = max(0, last - 3)
t0 = int(rows[start]["Start_Timestamp"])
prev = None
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tdtk::", "")[:46]
    print("%8.1f  +%6.1f  dur %6.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3 if prev else 0, (e - s) / 1e3, nm))
    prev = e
PY
