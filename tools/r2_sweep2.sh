#!/bin/bash
# GPU box: work-queue kernel (variant 30) against the static persistent-lane kernel (20), ICP loop at 1M / 4M
cd "$(dirname "$0")/.."
run() { env "$@" timeout 300 python tools/icp_probe.py ${N:-1000000} ${K:-100} 10 2>&1 | tail -1; }
for N in ${SIZES:-1000000 4000000}; do
  export N
  run TDTK_FUSE_SUMS=0
  for q in 160 192 224; do run TDTK_FUSE_SUMS=0 TDTK_REFILL_QPW=$q; run TDTK_FUSE_SUMS=0 TDTK_REFILL_QPW=$q TDTK_REFILL_THRESH=32; done
  for w in 4 5 6 7 8; do run TDTK_SEARCH_VARIANT=30 TDTK_STREAM_WPS=$w; done
  for sl in 32 128 256; do run TDTK_SEARCH_VARIANT=30 TDTK_STREAM_SLAB=$sl; done
  for th in 8 32; do run TDTK_SEARCH_VARIANT=30 TDTK_REFILL_THRESH=$th; done
  run TDTK_SEARCH_VARIANT=30 TDTK_REFILL_THRESH=32 TDTK_STREAM_SLAB=128
  run TDTK_SEARCH_VARIANT=30 TDTK_REFILL_THRESH=8 TDTK_STREAM_SLAB=32
done
