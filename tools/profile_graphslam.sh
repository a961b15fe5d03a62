#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace + HBM-traffic counters of the 1-GPU graph-SLAM workload (64 x 1M, 84 links).
#   usage: tools/profile_graphslam.sh <tag>   -> gpurun_out/<tag>/{gs,gs_fetch,gs_write}/...
set -u
TAG="${1:-prof}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/bench.py --workload graphslam --steps 10 --warmup 3"
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gs" -o p -- $CMD > "$OUT/gs.json" 2> "$OUT/gs.err"
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/gs_fetch" -o p -- $CMD > "$OUT/gs_fetch.json" 2> "$OUT/gs_fetch.err"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/gs_write" -o p -- $CMD > "$OUT/gs_write.json" 2> "$OUT/gs_write.err"
ls "$OUT/gs"
