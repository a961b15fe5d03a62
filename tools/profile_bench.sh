#!/usr/bin/env bash
# GPU box: rocprofv3 passes over the default bench command.  Each pass has its own timeout;
# counters are collected in their own runs (kernel-trace only), never with other trace domains.
#   usage: tools/profile_bench.sh <tag>          -> gpurun_out/<tag>/{stats,fetch,write}/...
set -u
TAG="${1:-prof}"; OUT="$GRAFT_REPO_ROOT/gpurun_out/$TAG"; mkdir -p "$OUT"
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-graphslam-base"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- $CMD > "$OUT/stats.json" 2> "$OUT/stats.err"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- $CMD > "$OUT/fetch.json" 2> "$OUT/fetch.err"
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/write" -o p -- $CMD > "$OUT/write.json" 2> "$OUT/write.err"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/sq1" -o p -- $CMD > "$OUT/sq1.json" 2> "$OUT/sq1.err"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq2" -o p -- $CMD > "$OUT/sq2.json" 2> "$OUT/sq2.err"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq3" -o p -- $CMD > "$OUT/sq3.json" 2> "$OUT/sq3.err"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/tcc" -o p -- $CMD > "$OUT/tcc.json" 2> "$OUT/tcc.err"
find "$OUT" -name "*.csv" | sort
