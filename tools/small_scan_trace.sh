#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/sst; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/sst -o p -- python $GRAFT_REPO_ROOT/tools/small_scan_trace.py 2>&1 | tail -4
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/sst/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-170:]
prev_end = None
out = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void tdtk::", "")[:40]
    out.append((name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
for o in out[-24:]:
    print("%-42s dur %7.2f us  gap before %7.2f us" % o)
import collections
d = collections.defaultdict(list); g = collections.defaultdict(list)
for n, du, ga in out[20:]:
    d[n].append(du); g[n].append(ga)
for n in d:
    print("%-42s n=%d dur mean %.2f  gap-before mean %.2f" % (n, len(d[n]), sum(d[n]) / len(d[n]), sum(g[n]) / len(g[n])))
PY
