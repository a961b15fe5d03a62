#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace of Scan::calcNormals on a 1M-point scan (tools/normals_probe.py: 5 runs through
# the host-buffer entry point + 5 on the resident scan).
#   usage: tools/profile_normals.sh <tag>   -> gpurun_out/<tag>/normals/...
set -u
TAG="${1:-prof}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/tools/normals_probe.py --reps 5"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/normals" -o p -- $CMD > "$OUT/normals.log" 2> "$OUT/normals.err"
ls "$OUT/normals"
