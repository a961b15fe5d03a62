#!/bin/bash
# GPU box: profile the bench command at the driver's arguments and at the script's defaults, summarise ON the box (the raw
# rocprofv3 CSVs are far beyond what gpurun copies back), keep the summaries under gpurun_out/keep/
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
for cfg in "20 5" "100 10"; do
  set -- $cfg
  tag="prof_s$1_w$2"
  bash tools/profile_bench.sh $tag $1 $2 > gpurun_out/keep/$tag.log 2>&1
  python tools/summarize_profiles.py $tag r03 > gpurun_out/keep/$tag.summary.txt 2>&1
  tail -5 gpurun_out/$tag/stats.err > gpurun_out/keep/$tag.stats.err.tail 2>/dev/null
  rm -rf gpurun_out/$tag
done
cp profiles/r03_* gpurun_out/keep/ 2>/dev/null
ls -la gpurun_out/keep
