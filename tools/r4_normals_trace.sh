#!/usr/bin/env bash
# GPU box: kernel trace of calcNormals (stats only), summary printed
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/r4norm"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/n" -o p -- python $GRAFT_REPO_ROOT/tools/normals_probe.py --reps 5 > "$OUT/normals.log" 2> "$OUT/normals.err"
tail -5 "$OUT/normals.log"
f=$(find "$OUT/n" -name "*kernel_stats.csv" | head -1)
head -30 "$f" | cut -c1-150
cp "$f" "$OUT/kernel_stats.csv"; rm -rf "$OUT/n"
