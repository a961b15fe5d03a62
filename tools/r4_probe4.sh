#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4d
python tools/gs_step_probe.py 2>&1 | tail -8 | tee gpurun_out/r4d/step.log
bash tools/r4_tree_trace.sh 2>&1 | tail -150 > gpurun_out/r4d/tree.log; tail -3 gpurun_out/r4d/tree.log
