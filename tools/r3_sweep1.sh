#!/bin/bash
# GPU box, round 3, first session: parity of the spill-free kernel, occupancy sweep (dynamic LDS as the limiter), eight
# bucket points per trip, TA / TCP counters of the search kernel.   usage: tools/r3_sweep1.sh <tag>
cd "$(dirname "$0")/.."
TAG="${1:-r3a}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
{
echo "== baseline (driver args 20/5, then 100/10)"
N=1000000 K=20 W=5 run A=1
N=1000000 K=100 W=10 run A=1
N=4000000 K=30 W=5 run A=1
echo "== occupancy by dynamic LDS (4M: W=5 free, 4, 3, 2), then 1M"
for d in 0 13312 20480 33792; do N=4000000 K=30 W=5 run TDTK_OCC_LDS=$d; done
for d in 13312 20480 33792; do N=1000000 K=100 W=10 run TDTK_OCC_LDS=$d; done
echo "== bucket points per trip 8"
N=1000000 K=100 W=10 run TDTK_BUCKET_PTS=8
N=1000000 K=20 W=5 run TDTK_BUCKET_PTS=8
N=4000000 K=30 W=5 run TDTK_BUCKET_PTS=8
N=1000000 K=100 W=10 run TDTK_BUCKET_PTS=8 TDTK_REFILL_QPW=256
echo "== slab length with the spill-free kernel"
for q in 192 256; do N=1000000 K=100 W=10 run TDTK_REFILL_QPW=$q; done
} > "$OUT/sweep.log" 2>&1
cat "$OUT/sweep.log"
# counters: each pass its own rocprofv3 run, kernel-trace only
cd /tmp; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/tools/icp_probe.py 1000000 100 10"
pmc() { name="$1"; shift; timeout 240 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- $CMD > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
pmc ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum GRBM_GUI_ACTIVE
pmc ta2 TA_BUSY_avr TA_BUSY_max TA_BUFFER_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
pmc tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
pmc td TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TCP_TD_TCP_STALL_CYCLES_sum
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD
pmc sq2 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM
pmc sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
pmc sq4 SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAVES_EQ_64 SQ_INSTS_VALU_MFMA_I8
cd "$GRAFT_REPO_ROOT"
python tools/r3_pmc_summary.py "$OUT" > "$OUT/pmc_summary.txt" 2>&1
cat "$OUT/pmc_summary.txt"
find "$OUT" -name "*.csv" -size +2000k -delete
