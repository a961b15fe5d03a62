#!/bin/bash
# GPU box: clusters of walked pieces in k_big_stitch -- pieces per LDS stage (BIG_CL) x stages requested ahead (BIG_NQ);
# library variants built beforehand (3dtk_amd/variants_cl*_nq*.so)
cd "$(dirname "$0")/.."
cp 3dtk_amd/lib3dtk_hip.so /tmp/base.so
for v in base cl2_nq6 cl4_nq5 cl1_nq8; do
  [ $v = base ] && cp /tmp/base.so 3dtk_amd/lib3dtk_hip.so || cp 3dtk_amd/variants_$v.so 3dtk_amd/lib3dtk_hip.so
  echo "== $v"; python tools/tree_probe.py 2>&1 | tail -1
done
cp /tmp/base.so 3dtk_amd/lib3dtk_hip.so
