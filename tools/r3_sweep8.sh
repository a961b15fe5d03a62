#!/bin/bash
# GPU box: (1) how long one link of a dependent fp64 add chain takes on a lone wave, by enabled lanes;
# (2) what one more 16-byte load per node visit costs the persistent-lane search kernel (TDTK_BUCKET_PTS=43)
cd "$(dirname "$0")/.."
./tools/micro/add_chain
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
for rep in 1 2; do
  echo "== base"; N=1000000 K=20 W=5 run A=1; N=1000000 K=100 W=10 run A=1
  echo "== +1 load per node visit"; N=1000000 K=20 W=5 run TDTK_BUCKET_PTS=43; N=1000000 K=100 W=10 run TDTK_BUCKET_PTS=43
done
