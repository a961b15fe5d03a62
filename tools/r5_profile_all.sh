#!/bin/bash
# GPU box, round 5 (final library): every profile set the bench line cites, summarised ON the box (the raw CSVs are too big to
# travel) into gpurun_out/keep5/ -- copy its r05_* files to profiles/.
#   ICP bench at both argument sets (kernel trace + PMC passes), graph-SLAM (64 x 1M), the configs[4]-shape leg.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep5
for cfg in "20 5" "100 10"; do
  set -- $cfg
  tag="prof_s$1_w$2"
  bash tools/profile_bench.sh $tag $1 $2 > gpurun_out/keep5/$tag.log 2>&1
  python tools/summarize_profiles.py $tag r05 > gpurun_out/keep5/$tag.summary.txt 2>&1
  rm -rf gpurun_out/$tag
done
bash tools/profile_graphslam.sh r5gs > gpurun_out/keep5/gs.log 2>&1
python tools/summarize_graphslam_profile.py r5gs r05 > gpurun_out/keep5/gs.summary.txt 2>&1
rm -rf gpurun_out/r5gs
cp profiles/r05_* gpurun_out/keep5/ 2>/dev/null
bash tools/profile_c5.sh r5c5final > gpurun_out/keep5/c5.log 2>&1
cp gpurun_out/r5c5final/r05_c5_* gpurun_out/keep5/ 2>/dev/null
ls -la gpurun_out/keep5 | head -60
