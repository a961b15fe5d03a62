cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sp -o p -- python $GRAFT_REPO_ROOT/tools/r6_solve_probe.py 2>&1 | grep status
python - <<'PY'
import csv
rows = sorted(csv.DictReader(open("/tmp/sp/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_final_solve" in r["Kernel_Name"]]
a, b = sorted(d[:200]), sorted(d[200:400])
print("full arithmetic: median %.2f us min %.2f | early exit: median %.2f us min %.2f" % (a[100], a[0], b[100], b[0]))
p = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "k_loop_publish" in r["Kernel_Name"])
print("k_loop_publish median %.2f us" % p[len(p) // 2])
PY
