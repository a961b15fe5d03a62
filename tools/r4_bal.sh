#!/bin/bash
# GPU box, round 4: slabs of equal cost in the ICP loop -- parity, then the bench line with and without
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4f
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "icp or alternative or dat_ or fused or k5 or K5 or minimizer or glue or warm or cost" > gpurun_out/r4f/pytest.log 2>&1; tail -4 gpurun_out/r4f/pytest.log
for b in 1 0 1 0; do
  TDTK_BALANCE=$b python bench.py --steps 20 --warmup 5 --no-cpu --no-graphslam-base --no-normals > gpurun_out/r4f/b20_$b.json 2>gpurun_out/r4f/b20_$b.err
  python -c "import json;d=json.load(open('gpurun_out/r4f/b20_$b.json'));print('balance=$b s20 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
for b in 1 0; do
  TDTK_BALANCE=$b python bench.py --no-cpu --no-graphslam-base --no-normals > gpurun_out/r4f/b100_$b.json 2>gpurun_out/r4f/b100_$b.err
  python -c "import json;d=json.load(open('gpurun_out/r4f/b100_$b.json'));print('balance=$b s100 ms_per_step %.4f k_ms %.4f value %.3e' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']))"
done
