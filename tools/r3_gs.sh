#!/bin/bash
# GPU box, round 3: graph-SLAM link passes -- slab pieces (L2 working set per XCD), then the profile + traffic counters,
# summarised on the box (raw CSVs do not travel)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
python tools/gs_knobs_probe.py default TDTK_LINK_PHASES=2 TDTK_LINK_PHASES=4 TDTK_LINK_PHASES=8 TDTK_LINK_PHASES=4,TDTK_REFILL_QPW=1024 TDTK_LINK_PHASES=8,TDTK_REFILL_QPW=1024 2>&1 | tail -13 | tee gpurun_out/keep/gs_phases.log
for ph in 1 4; do
  export TDTK_LINK_PHASES=$ph
  bash tools/profile_graphslam.sh gsprof_ph$ph > gpurun_out/keep/gsprof_ph$ph.log 2>&1
  python tools/summarize_graphslam_profile.py gsprof_ph$ph r03_ph$ph > gpurun_out/keep/gsprof_ph$ph.summary.txt 2>&1
  tail -3 gpurun_out/gsprof_ph$ph/gs.err > gpurun_out/keep/gsprof_ph$ph.err.tail
  rm -rf gpurun_out/gsprof_ph$ph
done
cp profiles/r03_ph* gpurun_out/keep/
ls gpurun_out/keep
