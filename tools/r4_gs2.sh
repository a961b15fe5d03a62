#!/bin/bash
# GPU box, round 4: lazy scan moves as a per-wave slab pre-pass -- parity, the 1-GPU round with and without, the 8-rank rehearsal
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "lazy or lum_links or company or config4 or rank_count or clpairs or gapx or lum6DQuat or two_ranks or slam_glue or match_graph or config1 or icp_glue" > gpurun_out/r4b/pytest.log 2>&1; tail -5 gpurun_out/r4b/pytest.log
for lz in 1 0 1 0; do
  TDTK_LAZY_MOVES=$lz timeout 600 python bench.py --workload graphslam --steps 10 --warmup 3 > gpurun_out/r4b/gs_lazy$lz.json 2> gpurun_out/r4b/gs_lazy$lz.err
  python -c "import json;d=json.load(open('gpurun_out/r4b/gs_lazy$lz.json'));print('lazy=$lz ms_per_step',d['ms_per_step'],'kernel_ms',d['roofline']['kernel_ms'])"
done
timeout 900 python tools/gs_shard_probe.py > gpurun_out/r4b/shard.log 2>&1; tail -7 gpurun_out/r4b/shard.log
