#!/bin/bash
# GPU box, round 4: every profile of the round, summarised ON the box into profiles/ (copied back through gpurun_out/keep)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
for cfg in "20 5" "100 10"; do
  set -- $cfg
  tag="prof_s$1_w$2"
  bash tools/profile_bench.sh $tag $1 $2 > gpurun_out/keep/$tag.log 2>&1
  python tools/summarize_profiles.py $tag r04 > gpurun_out/keep/$tag.summary.txt 2>&1
  rm -rf gpurun_out/$tag
done
bash tools/profile_graphslam.sh gsprof > gpurun_out/keep/gsprof.log 2>&1
GS_LINKS=84 python tools/summarize_graphslam_profile.py gsprof r04 > gpurun_out/keep/gsprof.summary.txt 2>&1
rm -rf gpurun_out/gsprof
GS_CMD="python $GRAFT_REPO_ROOT/tools/gs_share_run.py 8 0" bash tools/profile_graphslam.sh gsshare > gpurun_out/keep/gsshare.log 2>&1
GS_LINKS=11 GS_SUFFIX=_share11 GS_CMD_LABEL="python tools/gs_share_run.py 8 0  (rank 0 of 8's 11 links, scan moves queued)" python tools/summarize_graphslam_profile.py gsshare r04 > gpurun_out/keep/gsshare.summary.txt 2>&1
rm -rf gpurun_out/gsshare
cp profiles/r04_* gpurun_out/keep/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/keep/r04_bench_n1_driver_args.json 2> gpurun_out/keep/bench_driver_args.err
python bench.py > gpurun_out/keep/r04_bench_n1.json 2> gpurun_out/keep/bench_default.err
python tools/lane_fill_probe.py > gpurun_out/keep/r04_lane_fill.txt 2>&1
ls -la gpurun_out/keep | head -60
