#!/bin/bash
# GPU box: the single-loop ("if-if") kernel (variant 40) against the nested-loop persistent-lane kernel (20)
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} 10 2>&1 | tail -1; }
for N in ${SIZES:-1000000 4000000 300000}; do
  export N
  run TDTK_SEARCH_VARIANT=20
  run TDTK_SEARCH_VARIANT=40
  for q in 128 160 192 256 320; do run TDTK_SEARCH_VARIANT=40 TDTK_REFILL_QPW=$q; done
  for th in 8 32; do run TDTK_SEARCH_VARIANT=40 TDTK_REFILL_THRESH=$th; done
  run TDTK_SEARCH_VARIANT=40 TDTK_REFILL_THRESH=8 TDTK_REFILL_QPW=256
  run TDTK_SEARCH_VARIANT=40 TDTK_REFILL_THRESH=32 TDTK_REFILL_QPW=256
done
