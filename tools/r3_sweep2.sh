#!/bin/bash
# GPU box, round 3: what is the search kernel sensitive to?  One more load per bucket point (TDTK_BUCKET_PTS=41), eight more
# fp64 VALU instructions per bucket point (42), against the same kernel at the same occupancy (four waves per SIMD).
cd "$(dirname "$0")/.."
TAG="${1:-r3b}"; OUT="$PWD/gpurun_out/$TAG"; mkdir -p "$OUT"
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
{
for N in 4000000 1000000; do
  export N; K=30; W=5; [ $N = 1000000 ] && K=100 && W=10; export K W
  run TDTK_OCC_LDS=13312 TDTK_REFILL_QPW=256
  run TDTK_BUCKET_PTS=41 TDTK_REFILL_QPW=256
  run TDTK_BUCKET_PTS=42 TDTK_REFILL_QPW=256
done
} > "$OUT/sweep.log" 2>&1
cat "$OUT/sweep.log"
