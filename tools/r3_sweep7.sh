#!/bin/bash
# GPU box: the ordering key of the slab hand-out with / without the node visits in it (library variants built beforehand)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
for rep in 1 2; do for v in base nonodes; do
  cp 3dtk_amd/variants_$v.so 3dtk_amd/lib3dtk_hip.so; echo "== $v"
  N=1000000 K=20 W=5 run V=$v; N=1000000 K=100 W=10 run V=$v; [ $rep = 1 ] && N=4000000 K=30 W=5 run V=$v
done; done
cp 3dtk_amd/variants_base.so 3dtk_amd/lib3dtk_hip.so
