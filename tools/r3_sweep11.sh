#!/bin/bash
# GPU box: the capped node walk against the kernel without it, same box (3dtk_amd/variants_orig.so built from the parent commit)
cd "$(dirname "$0")/.."
cp 3dtk_amd/lib3dtk_hip.so /tmp/new.so
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1; }
for rep in 1 2; do
  cp 3dtk_amd/variants_orig.so 3dtk_amd/lib3dtk_hip.so; echo "== original"
  N=1000000 K=20 W=5 run A=1; N=1000000 K=100 W=10 run A=1
  cp /tmp/new.so 3dtk_amd/lib3dtk_hip.so; echo "== new"
  for cfg in "0 24" "16 24" "32 16" "48 8" "24 16"; do set -- $cfg
    N=1000000 K=20 W=5 run TDTK_CAP_WALK=$1 TDTK_CAP_LEAF=$2; N=1000000 K=100 W=10 run TDTK_CAP_WALK=$1 TDTK_CAP_LEAF=$2
  done
done
cp /tmp/new.so 3dtk_amd/lib3dtk_hip.so
