#!/bin/bash
# the whole GPU tier + smoke + the bench line at the driver's arguments
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4full; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py --steps 20 --warmup 5 > $O/bench_s20.json 2> $O/bench_s20.err; tail -c 600 $O/bench_s20.json
