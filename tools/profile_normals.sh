#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace of Scan::calcNormals on a 1M-point scan (tools/normals_probe.py: 5 runs through
# the host-buffer entry point + 5 on the resident scan).
#   usage: tools/profile_normals.sh <tag>   -> gpurun_out/<tag>/normals/...
set -u
TAG="${1:-prof}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/tools/normals_probe.py --reps 5"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/normals" -o p -- $CMD > "$OUT/normals.log" 2> "$OUT/normals.err"
# counters in their own passes (kernel trace only)
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/normals_fetch" -o p -- $CMD > /dev/null 2> "$OUT/normals_fetch.err"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/normals_write" -o p -- $CMD > /dev/null 2> "$OUT/normals_write.err"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/normals_sq" -o p -- $CMD > /dev/null 2> "$OUT/normals_sq.err"
ls "$OUT/normals" "$OUT/normals_fetch"
