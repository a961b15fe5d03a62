#!/usr/bin/env bash
# GPU box: where does a vector-memory access of k_search wait?  TA / TCP / UTCL1 counters, each group in its own pass.
set -u
# (at most four counters of one block per pass: a fifth TCP counter makes rocprofv3 abort -- 'exceeds the capabilities of
# the hardware to collect' -- and the process then hangs until its timeout)
TAG="${1:-r4diag}"
OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal"
cd /tmp; export TMPDIR=/tmp
pass() { name="$1"; shift; timeout -s KILL 100 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed"; }
pass ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pass tcp1 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum
pass tcp2 TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass tcp3 TCP_TOTAL_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
pass td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE
python "$GRAFT_REPO_ROOT/tools/r4_tcp_diag.py" "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
