#!/usr/bin/env bash
# round 5: the slab geometry of the single-pass persistent-lane launch at 10M queries (lab library): does an XCD whose waves
# sweep its eighth of the scan window by window (several pieces per slab; one resident generation) keep the tree in its L2?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r5c
run() { env TDTK_LIB=lab "$@" timeout 300 python tools/c5_probe.py 2>&1 | grep C5PROBE >> gpurun_out/r5c/sweep.log; }
run
run TDTK_REFILL_PHASES=5
run TDTK_REFILL_PHASES=2
run TDTK_REFILL_QPW=2560
run TDTK_REFILL_QPW=2560 TDTK_REFILL_PHASES=10
run TDTK_REFILL_QPW=2560 TDTK_REFILL_PHASES=20
run TDTK_REFILL_QPW=1280 TDTK_REFILL_PHASES=5
run TDTK_REFILL_QPW=1280 TDTK_REFILL_PHASES=10
run TDTK_REFILL_QPW=2560 TDTK_REFILL_PHASES=10 TDTK_REFILL_THRESH=16
run TDTK_REFILL_QPW=256
cat gpurun_out/r5c/sweep.log
