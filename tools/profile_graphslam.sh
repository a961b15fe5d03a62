#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace of the 1-GPU graph-SLAM workload (64 x 1M, 84 links).
#   usage: tools/profile_graphslam.sh <tag>   -> gpurun_out/<tag>/gs/...
set -u
TAG="${1:-prof}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/bench.py --workload graphslam --steps 10 --warmup 3"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gs" -o p -- $CMD > "$OUT/gs.json" 2> "$OUT/gs.err"
ls "$OUT/gs"
