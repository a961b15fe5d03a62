# GPU box: kernel trace of the small-scan ICP iteration inside the host-free loop (tools/small_iter_probe.py): durations and gaps
cd /tmp; export TMPDIR=/tmp
TDTK_ICP_DEVICE_LOOP=${1:-1} timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -o p -- python $GRAFT_REPO_ROOT/tools/small_iter_probe.py > /dev/null 2>&1
python - <<'PY'
import csv, collections
rows = sorted(csv.DictReader(open("/tmp/lt/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
agg = collections.defaultdict(list)
prev_end = None; prev_name = None
gaps = collections.defaultdict(list)
seq = []
for r in rows:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("tdtk::", "")[:60]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    agg[n].append((e - s) / 1e3)
    if prev_end is not None and (s - prev_end) < 100000:
        gaps[(prev_name, n)].append((s - prev_end) / 1e3)
    seq.append((n, s, e))
    prev_end, prev_name = e, n
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:8]:
    v2 = sorted(v); print("%-62s calls %6d median %8.2f us min %8.2f" % (k, len(v), v2[len(v2) // 2], v2[0]))
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:6]:
    v = sorted(v); print("gap %-40s -> %-40s n %5d median %6.2f us" % (k[0][:40], k[1][:40], len(v), v[len(v) // 2]))
# a stretch of the first match's launches
i0 = next(i for i, x in enumerate(seq) if "k_search_g8" in x[0])
t0 = seq[i0][1]
for n, s, e in seq[i0:i0 + 14]:
    print("%-50s start %8.2f us  dur %6.2f" % (n[:50], (s - t0) / 1e3, (e - s) / 1e3))
PY
