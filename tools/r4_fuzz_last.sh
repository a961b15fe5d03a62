#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4fuzz9; mkdir -p $O
for seed in 8101 8102 8103; do timeout 400 python tools/fuzz_parity.py --seconds 200 --seed $seed > $O/small_$seed.log 2>&1; grep -v amdgpu $O/small_$seed.log | tail -2; done
for seed in 8201 8202; do timeout 500 python tools/fuzz_parity.py --big --seconds 250 --seed $seed > $O/big_$seed.log 2>&1; grep -v amdgpu $O/big_$seed.log | tail -2; done
timeout 400 python tools/fuzz_graph.py --seconds 200 --seed 8301 > $O/graph.log 2>&1; grep -v amdgpu $O/graph.log | tail -2
