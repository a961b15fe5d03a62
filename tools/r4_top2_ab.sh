#!/bin/bash
# what of the upper-levels-in-LDS result was the workgroup size: plain 1024-thread workgroups, staged 1024, and seven levels
# staged by ordinary 128-thread workgroups (lab library)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4top2; mkdir -p $O
TDTK_LIB=lab TDTK_TOP_BLOCK=128 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "k5 or pair_sums_inside or full_size_icp or lattice" 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals --no-small-scans --no-rehearsal 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(\"$1: ms_per_step %.4f kernel %.4f\" % (d[\"ms_per_step\"], d[\"roofline\"][\"kernel_ms\"]))"; }
export TDTK_LIB=lab
for rep in 1 2; do
run "default 128"
TDTK_TOP_BLOCK=128 run "128 threads, 7 levels in LDS"
TDTK_TOP_BLOCK=1024 TDTK_TOP_LEVELS=0 run "1024 threads, nothing staged"
TDTK_TOP_BLOCK=1024 run "1024 threads, 10 levels in LDS"
done
