#!/bin/bash
# GPU box: profile the bench command at the driver's arguments and at the script's defaults, and the 1-GPU graph-SLAM
# workload; summarise ON the box (the raw rocprofv3 CSVs are far beyond what gpurun copies back); keep the summaries and
# the bench lines under gpurun_out/keep/
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
for cfg in "20 5" "100 10"; do
  set -- $cfg
  tag="prof_s$1_w$2"
  bash tools/profile_bench.sh $tag $1 $2 > gpurun_out/keep/$tag.log 2>&1
  python tools/summarize_profiles.py $tag r03 > gpurun_out/keep/$tag.summary.txt 2>&1
  rm -rf gpurun_out/$tag
done
bash tools/profile_graphslam.sh gsprof > gpurun_out/keep/gsprof.log 2>&1
python tools/summarize_graphslam_profile.py gsprof r03 > gpurun_out/keep/gsprof.summary.txt 2>&1
rm -rf gpurun_out/gsprof
cp profiles/r03_* gpurun_out/keep/ 2>/dev/null      # the summaries just written (before the bench lines: stale copies of those are in profiles/ too)
python bench.py --steps 20 --warmup 5 > gpurun_out/keep/r03_bench_n1_driver_args.json 2> gpurun_out/keep/bench_driver_args.err
python bench.py > gpurun_out/keep/r03_bench_n1.json 2> gpurun_out/keep/bench_default.err
ls -la gpurun_out/keep | head -40
