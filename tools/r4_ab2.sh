#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4i
for i in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals > gpurun_out/r4i/b$i.json 2>gpurun_out/r4i/b$i.err
python -c "import json;d=json.load(open('gpurun_out/r4i/b$i.json'));print('s20 ms_per_step %.4f k_ms %.4f value %.3e tree %.3f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['tree_build_1gpu']['ms']))"
done
