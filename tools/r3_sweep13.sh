#!/bin/bash
# GPU box: which kernel a small batch should get now that every kernel filters buckets by groups (variant 4: one query
# per lane, 9 / 10 / 11: eight / four / sixteen lanes per query, 20: persistent lanes)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} ${W:-10} 2>&1 | tail -1 | sed 's/ sums.*//'; }
for n in 20000 50000 81360 120000 200000 300000; do
  for v in 10 9 4 20; do N=$n K=200 W=20 run TDTK_SEARCH_VARIANT=$v; done
done
