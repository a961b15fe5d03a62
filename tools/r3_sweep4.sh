#!/bin/bash
# GPU box, round 3: two tree levels per round trip (fat node records) against one
cd "$(dirname "$0")/.."
TAG="${1:-r3d}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
{
N=1000000 K=20 W=5 run A=1
N=1000000 K=20 W=5 run TDTK_FAT_NODES=0
N=1000000 K=100 W=10 run A=1
N=1000000 K=100 W=10 run TDTK_FAT_NODES=0
N=4000000 K=30 W=5 run A=1
N=4000000 K=30 W=5 run TDTK_FAT_NODES=0
N=300000 K=100 W=10 run A=1
N=300000 K=100 W=10 run TDTK_FAT_NODES=0
} > "$OUT/sweep.log" 2>&1
cat "$OUT/sweep.log"
