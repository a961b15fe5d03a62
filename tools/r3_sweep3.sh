#!/bin/bash
# GPU box, round 3: bucket groups (fp32 shadow filter, whole bucket per round trip) against the fp64 bucket scan
cd "$(dirname "$0")/.."
TAG="${1:-r3c}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
{
N=1000000 K=20 W=5 run A=1
N=1000000 K=100 W=10 run A=1
N=1000000 K=20 W=5 run TDTK_BUCKET_GROUPS=0
N=1000000 K=100 W=10 run TDTK_BUCKET_GROUPS=0
N=4000000 K=30 W=5 run A=1
N=4000000 K=30 W=5 run TDTK_BUCKET_GROUPS=0
N=300000 K=100 W=10 run A=1
N=300000 K=100 W=10 run TDTK_BUCKET_GROUPS=0
for q in 224 320; do N=1000000 K=100 W=10 run TDTK_REFILL_QPW=$q; done
} > "$OUT/sweep.log" 2>&1
cat "$OUT/sweep.log"
