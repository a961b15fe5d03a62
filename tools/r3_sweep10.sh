#!/bin/bash
# GPU box: capped node walks of the persistent-lane kernel -- the walk of a round ends when at most TDTK_CAP_WALK lanes
# are still walking while at least TDTK_CAP_LEAF hold a bucket
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
for cfg in "0 24" "1 24" "2 24" "4 24" "8 24" "16 24" "4 8" "4 40" "8 40" "16 40" "32 16" "0 24"; do
  set -- $cfg
  N=1000000 K=20 W=5 run TDTK_CAP_WALK=$1 TDTK_CAP_LEAF=$2
done
for cfg in "0 24" "4 24" "8 24"; do
  set -- $cfg
  N=1000000 K=100 W=10 run TDTK_CAP_WALK=$1 TDTK_CAP_LEAF=$2
  N=4000000 K=30 W=5 run TDTK_CAP_WALK=$1 TDTK_CAP_LEAF=$2
done
