#!/usr/bin/env bash
# round 5: the 16-bit bucket shadow (product build) against the fp32 groups (lab library built with LABFLAGS=-DTDTK_BUCKET_FP32):
# 1M-vs-1M ICP at both argument sets, slab lengths for the five waves per SIMD the q16 kernel leaves room for, the 10M pair.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5e; L=gpurun_out/r5e/ab.log
run() { env "$@" timeout 300 python tools/icp_probe.py 1000000 20 5 2>&1 | grep "^n=" >> $L; env "$@" timeout 300 python tools/icp_probe.py 1000000 100 10 2>&1 | grep "^n=" >> $L; }
for r in 1 2; do run TDTK_LIB=product; run TDTK_LIB=lab; done
for q in 192 224 160; do run TDTK_LIB=lab TDTK_REFILL_QPW=$q; done
c5() { env "$@" timeout 300 python tools/c5_probe.py 2>&1 | grep C5PROBE >> $L; }
c5 TDTK_LIB=product; c5 TDTK_LIB=lab
cat $L
