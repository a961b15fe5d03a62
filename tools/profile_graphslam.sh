#!/usr/bin/env bash
# GPU box: rocprofv3 kernel trace + HBM-traffic counters of the 1-GPU graph-SLAM workload (64 x 1M, 84 links).
#   usage: tools/profile_graphslam.sh <tag>   -> gpurun_out/<tag>/{gs,gs_fetch,gs_write}/...
set -u
TAG="${1:-prof}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="${GS_CMD:-python $GRAFT_REPO_ROOT/bench.py --workload graphslam --steps 10 --warmup 3 --no-rehearsal}"
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/gs" -o p -- $CMD > "$OUT/gs.json" 2> "$OUT/gs.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/gs_fetch" -o p -- $CMD > "$OUT/gs_fetch.json" 2> "$OUT/gs_fetch.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/gs_write" -o p -- $CMD > "$OUT/gs_write.json" 2> "$OUT/gs_write.err"
# round 4: the issue side of the link launch (each pass its own run, kernel-trace only)
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/gs_sq1" -o p -- $CMD > "$OUT/gs_sq1.json" 2> "$OUT/gs_sq1.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/gs_sq2" -o p -- $CMD > "$OUT/gs_sq2.json" 2> "$OUT/gs_sq2.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum --output-format csv -d "$OUT/gs_tcp" -o p -- $CMD > "$OUT/gs_tcp.json" 2> "$OUT/gs_tcp.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d "$OUT/gs_tcp2" -o p -- $CMD > "$OUT/gs_tcp2.json" 2> "$OUT/gs_tcp2.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --pmc TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --output-format csv -d "$OUT/gs_tcp3" -o p -- $CMD > "$OUT/gs_tcp3.json" 2> "$OUT/gs_tcp3.err"
ls "$OUT/gs"
