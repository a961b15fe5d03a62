#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/keep
bash tools/profile_graphslam.sh gsprof > gpurun_out/keep/gsprof.log 2>&1
GS_LINKS=84 python tools/summarize_graphslam_profile.py gsprof r04 > gpurun_out/keep/gsprof.summary.txt 2>&1
rm -rf gpurun_out/gsprof
cp profiles/r04_graphslam_pmc.json profiles/r04_graphslam_kernel_stats.csv profiles/r04_graphslam_bench_under_rocprof.json gpurun_out/keep/ 2>/dev/null
python -c "
import json;d=json.load(open('profiles/r04_graphslam_pmc.json'));print(json.dumps(d['derived'],indent=1)); print(d['kernels']['k_search (several links per launch)']['dispatches_gs_sq1'], d['kernels']['k_search (several links per launch)']['SQ_WAVES'])"
head -8 profiles/r04_graphslam_kernel_stats.csv
