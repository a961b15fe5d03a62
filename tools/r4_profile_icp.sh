#!/bin/bash
# GPU box, round 4 (final library): the ICP bench's profile set for both command lines, summarised ON the box into profiles/
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
for cfg in "20 5" "100 10"; do
  set -- $cfg
  tag="prof_s$1_w$2"
  bash tools/profile_bench.sh $tag $1 $2 > gpurun_out/keep/$tag.log 2>&1
  python tools/summarize_profiles.py $tag r04 > gpurun_out/keep/$tag.summary.txt 2>&1
  rm -rf gpurun_out/$tag
done
cp profiles/r04_* gpurun_out/keep/ 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/keep/r04_bench_n1_driver_args.json 2> gpurun_out/keep/bench_driver_args.err
python bench.py > gpurun_out/keep/r04_bench_n1.json 2> gpurun_out/keep/bench_default.err
ls -la gpurun_out/keep | head -40
