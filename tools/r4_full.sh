#!/bin/bash
# GPU box: the whole GPU tier + smoke + the driver's bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4full
timeout 3000 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r4full/pytest.log 2>&1; tail -25 gpurun_out/r4full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/r4full/bench_s20.json 2> gpurun_out/r4full/bench_s20.err; python -c "
import json;d=json.load(open('gpurun_out/r4full/bench_s20.json'));print('value %.3e ms/step %.4f k_ms %.4f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])); print('graphslam', d.get('graphslam_1gpu',{}).get('ms_per_step'))"
