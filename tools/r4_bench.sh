#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4j
( time python bench.py --steps 20 --warmup 5 > gpurun_out/r4j/bench_s20.json 2> gpurun_out/r4j/bench_s20.err ) 2>&1 | tail -3
tail -3 gpurun_out/r4j/bench_s20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4j/bench_s20.json'))
print('value %.3e ms/step %.4f k_ms %.4f frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac']))
print('tree', d['tree_build_1gpu']['ms'], 'cpu', d.get('cpu_baseline',{}).get('value'))
print('small', json.dumps(d.get('doicp_small_scans'))[:1500])
g=d.get('graphslam_1gpu',{})
print('graphslam', g.get('ms_per_step'), json.dumps(g.get('sharded_step_rehearsal'))[:900])
PY
