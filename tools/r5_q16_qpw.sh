#!/usr/bin/env bash
# round 5: slab length of the single-pass launch for the q16 kernel (94 VGPRs: five waves per SIMD fit) -- product library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5f; L=gpurun_out/r5f/qpw2.log
run() { env "$@" timeout 300 python tools/icp_probe.py 1000000 20 5 2>&1 | grep "^n=" >> $L; env "$@" timeout 300 python tools/icp_probe.py 1000000 100 10 2>&1 | grep "^n=" >> $L; }
run A=0; for q in 200 208 216 224 232 240 192; do run TDTK_REFILL_QPW_X=$q; done; run A=0
cat $L
