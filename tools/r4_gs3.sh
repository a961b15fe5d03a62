#!/bin/bash
# GPU box, round 4: batched pre-pass, glue + ELCH tests
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4c
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lazy or lum_links or config4 or slam_glue or config1 or icp_glue or elch or lum6DQuat" > gpurun_out/r4c/pytest.log 2>&1; tail -15 gpurun_out/r4c/pytest.log
for lz in 1 0 1; do
  TDTK_LAZY_MOVES=$lz timeout 600 python bench.py --workload graphslam --steps 10 --warmup 3 > gpurun_out/r4c/gs_lazy$lz.json 2> gpurun_out/r4c/gs_lazy$lz.err
  python -c "import json;d=json.load(open('gpurun_out/r4c/gs_lazy$lz.json'));print('lazy=$lz ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])"
done
timeout 900 python tools/gs_shard_probe.py > gpurun_out/r4c/shard.log 2>&1; tail -3 gpurun_out/r4c/shard.log
