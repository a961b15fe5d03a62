#!/usr/bin/env bash
# GPU box: rocprofv3 passes over the configs[4]-shape leg alone (python bench.py --workload c5 --no-cpu): kernel trace + stats,
# then the counters, each pass its own run under its own timeout (kernel-trace only, never with other trace domains).
#   usage: tools/profile_c5.sh <tag> [extra bench args]   -> /tmp/<tag>/{c5,c5_fetch,...}/ (raw CSVs: too big for gpurun_out's 64 MiB),
#          summarised on the box by tools/summarize_c5_profile.py into gpurun_out/<tag>/ (copy those files to profiles/)
set -u
TAG="${1:-c5prof}"; shift || true
OUT="/tmp/$TAG"; rm -rf "$OUT"; mkdir -p "$OUT" "$GRAFT_REPO_ROOT/gpurun_out/$TAG"
CMD="python $GRAFT_REPO_ROOT/bench.py --workload c5 --no-cpu $*"
echo "python bench.py --workload c5 --no-cpu $*" > "$OUT/command.txt"
cd /tmp; export TMPDIR=/tmp
timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/c5" -o p -- $CMD > "$OUT/c5.json" 2> "$OUT/c5.err"
pass() { name="$1"; shift; timeout -s KILL 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed"; }
pass c5_fetch FETCH_SIZE
pass c5_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pass c5_sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pass c5_sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pass c5_tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCC_REQ_sum
pass c5_tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass c5_tcp3 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum
pass c5_ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
cd "$GRAFT_REPO_ROOT"
python tools/summarize_c5_profile.py "$OUT" "${C5_PREFIX:-r05}" "gpurun_out/$TAG"
cp "$OUT"/*.err "gpurun_out/$TAG/" 2>/dev/null; for f in gpurun_out/$TAG/*.err; do tail -c 2000 "$f" > "$f.tail"; rm -f "$f"; done
ls -la "gpurun_out/$TAG"
