#!/bin/bash
# GPU box, round 4 (final library): the graph-SLAM profile set (full graph + a rank's share), with the vector L1's counters,
# summarised ON the box into profiles/
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep2
bash tools/profile_graphslam.sh gsprof > gpurun_out/keep2/gsprof.log 2>&1
GS_LINKS=84 python tools/summarize_graphslam_profile.py gsprof r04 > gpurun_out/keep2/gsprof.summary.txt 2>&1
rm -rf gpurun_out/gsprof
GS_CMD="python $GRAFT_REPO_ROOT/tools/gs_share_run.py 8 0" bash tools/profile_graphslam.sh gsshare > gpurun_out/keep2/gsshare.log 2>&1
GS_LINKS=11 GS_SUFFIX=_share11 GS_CMD_LABEL="python tools/gs_share_run.py 8 0  (rank 0 of 8's 11 links, scan moves queued)" python tools/summarize_graphslam_profile.py gsshare r04 > gpurun_out/keep2/gsshare.summary.txt 2>&1
rm -rf gpurun_out/gsshare
cp profiles/r04_graphslam* gpurun_out/keep2/ 2>/dev/null
python bench.py --workload graphslam --no-cpu > gpurun_out/keep2/bench_gs.json 2> gpurun_out/keep2/bench_gs.err
ls -la gpurun_out/keep2
