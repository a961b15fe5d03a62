#!/bin/bash
# GPU box, round 3: slab length / refill threshold of the persistent-lane kernel with the bucket groups in (122 registers, 4 waves per SIMD)
cd "$(dirname "$0")/.."
TAG="${1:-r3e}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
{
for q in 256 288 320; do for th in 8 16 32; do N=1000000 K=20 W=5 run TDTK_REFILL_QPW=$q TDTK_REFILL_THRESH=$th; done; done
N=1000000 K=20 W=5 run TDTK_COST_ORDER=0
N=1000000 K=20 W=5 run TDTK_WARM_START=0
N=1000000 K=100 W=10 run TDTK_REFILL_THRESH=8
N=1000000 K=100 W=10 run TDTK_REFILL_THRESH=32
} > "$OUT/sweep.log" 2>&1
cat "$OUT/sweep.log"
