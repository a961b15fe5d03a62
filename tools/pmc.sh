#!/usr/bin/env bash
# GPU box: PMC passes (each in its own rocprofv3 run, no tracing domains besides kernel-trace)
# usage: tools/pmc.sh <outdir-under-gpurun_out> <kbench mode> [reps]
set -u
OUT="$GRAFT_REPO_ROOT/gpurun_out/$1"; MODE="$2"; REPS="${3:-6}"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
run() { name="$1"; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python "$GRAFT_REPO_ROOT/tools/kbench.py" "$REPS" "$MODE" > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
run sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
run ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python "$GRAFT_REPO_ROOT/tools/kbench.py" 20 "$MODE" > "$OUT/stats.log" 2>&1
find "$OUT" -name "*.csv" | head -40
