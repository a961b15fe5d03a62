#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4fuzz2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lopsided or tree or randomized" > $O/tests.log 2>&1; tail -3 $O/tests.log
for seed in 9101 9102 9103; do timeout 400 python tools/fuzz_parity.py --seconds 200 --seed $seed > $O/small_$seed.log 2>&1; tail -2 $O/small_$seed.log | grep -v amdgpu; done
timeout 500 python tools/fuzz_parity.py --big --seconds 300 --seed 9201 > $O/big.log 2>&1; tail -2 $O/big.log | grep -v amdgpu
