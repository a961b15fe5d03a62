#!/bin/bash
# GPU box, round 4: the graph-SLAM profiles only (full graph + a rank's share)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
bash tools/profile_graphslam.sh gsprof > gpurun_out/keep/gsprof.log 2>&1
GS_LINKS=84 python tools/summarize_graphslam_profile.py gsprof r04 > gpurun_out/keep/gsprof.summary.txt 2>&1
rm -rf gpurun_out/gsprof
GS_CMD="python $GRAFT_REPO_ROOT/tools/gs_share_run.py 8 0" bash tools/profile_graphslam.sh gsshare > gpurun_out/keep/gsshare.log 2>&1
GS_LINKS=11 GS_SUFFIX=_share11 GS_CMD_LABEL="python tools/gs_share_run.py 8 0  (rank 0 of 8's 11 links, scan moves queued)" python tools/summarize_graphslam_profile.py gsshare r04 > gpurun_out/keep/gsshare.summary.txt 2>&1
rm -rf gpurun_out/gsshare
cp profiles/r04_graphslam* gpurun_out/keep/ 2>/dev/null
tail -30 gpurun_out/keep/gsprof.summary.txt; tail -30 gpurun_out/keep/gsshare.summary.txt
