#!/bin/bash
# GPU box, round 4 (final library): the graph-SLAM profile set of the full graph, with the vector L1's counters, summarised ON the box
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep2
bash tools/profile_graphslam.sh gsprof > gpurun_out/keep2/gsprof.log 2>&1
GS_LINKS=84 python tools/summarize_graphslam_profile.py gsprof r04 > gpurun_out/keep2/gsprof.summary.txt 2>&1
rm -rf gpurun_out/gsprof
cp profiles/r04_graphslam_pmc.json profiles/r04_graphslam_kernel_stats.csv profiles/r04_graphslam_bench_under_rocprof.json gpurun_out/keep2/ 2>/dev/null
python bench.py --workload graphslam --no-cpu --no-rehearsal > gpurun_out/keep2/bench_gs.json 2> gpurun_out/keep2/bench_gs.err
ls -la gpurun_out/keep2
