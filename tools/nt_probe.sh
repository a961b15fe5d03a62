#!/usr/bin/env bash
# GPU box: A/B of an environment switch on the ICP loop -- wall time, then FETCH_SIZE / WRITE_SIZE+TCC of the search kernel.
#   usage: tools/nt_probe.sh <tag> VAR v0 v1 [points]
set -u
TAG="$1"; VAR="$2"; V0="$3"; V1="$4"; N="${5:-1000000}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
for v in $V0 $V1 $V0 $V1; do env $VAR=$v python $GRAFT_REPO_ROOT/tools/icp_probe.py $N 100 10; done 2>&1 | grep "^n=" | tee "$OUT/time.log"
for v in $V0 $V1; do
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    D="/tmp/ntp_${v}_${pass%% *}"; rm -rf "$D"
    env $VAR=$v timeout 300 rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$D" -o p -- python $GRAFT_REPO_ROOT/tools/icp_probe.py $N 100 10 > /dev/null 2>&1
    python - "$D" "$VAR=$v" <<'PY' | tee -a "$OUT/pmc.log"
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_search" not in k: continue
        a = acc[(k.split("(")[0][:60], r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%s  %-50s %-14s avg %.1f over %d launches" % (tag, k, c, v / n, n))
PY
  done
done
