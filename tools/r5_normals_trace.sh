#!/usr/bin/env bash
# GPU box: kernel trace of the resident calcNormals at 1M (round 5: with the mid-cell finisher), per-kernel totals of ONE call printed
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/r5norm"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
N="${1:-1000000}"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/r5n -o p -- python $GRAFT_REPO_ROOT/tools/normals_probe.py --n $N --reps 2 > "$OUT/normals_$N.log" 2> "$OUT/normals.err"
tail -3 "$OUT/normals_$N.log"
python - <<'P' > "$OUT/trace_$N.txt"
import csv, collections, glob
f = glob.glob('/tmp/r5n/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last call = from the last k_ann_init to the end
idx = [i for i, r in enumerate(rows) if 'k_ann_init' in r['Kernel_Name']]
rows = rows[idx[-1]:]
t0 = int(rows[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in rows)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('tdtk::', '')[:60]
    agg[n][0] += 1; agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("one calcNormals call: %d launches, %.1f us from first start to last end, %.1f us inside kernels" % (len(rows), (t1 - t0) / 1e3, sum(v[1] for v in agg.values())))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %4d %9.1f us" % (k, v[0], v[1]))
P
cat "$OUT/trace_$N.txt"
