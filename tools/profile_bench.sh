#!/usr/bin/env bash
# GPU box: rocprofv3 passes over ONE bench command.  Each pass has its own timeout; counters are collected in their own
# runs (kernel-trace only), never with other trace domains.
#   usage: tools/profile_bench.sh <tag> [steps] [warmup]     -> gpurun_out/<tag>/{stats,fetch,write,sq1,sq2,sq3,tcc}/...
# The summary (tools/summarize_profiles.py <tag> <prefix>) is keyed by the steps / warmup of the command, and bench.py
# only uses a summary whose steps / warmup equal its own.
set -u
TAG="${1:-prof}"; STEPS="${2:-100}"; WARM="${3:-10}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-c5"
echo "$CMD" > "$OUT/command.txt"
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $CMD > "$OUT/stats.json" 2> "$OUT/stats.err"
pass() { name="$1"; shift; timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed"; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
pass sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
pass tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
# (at most four counters of the TCP block per pass: a fifth makes rocprofv3 abort -- 'exceeds the capabilities of the hardware' -- and hang)
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass tcp3 TCP_TCP_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
find "$OUT" -name "*.csv" | sort
