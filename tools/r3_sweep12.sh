#!/bin/bash
# GPU box: kernels.hip compiled with other optimisation / scheduling flags (variants built beforehand)
cd "$(dirname "$0")/.."
cp 3dtk_amd/lib3dtk_hip.so /tmp/base.so
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1 | sed 's/ sums.*//'; }
for rep in 1 2; do for v in base O2 nomisched nopost; do
  [ $v = base ] && cp /tmp/base.so 3dtk_amd/lib3dtk_hip.so || cp 3dtk_amd/variants_$v.so 3dtk_amd/lib3dtk_hip.so
  echo "== $v"; N=1000000 K=20 W=5 run V=$v; N=1000000 K=100 W=10 run V=$v
done; done
cp /tmp/base.so 3dtk_amd/lib3dtk_hip.so
