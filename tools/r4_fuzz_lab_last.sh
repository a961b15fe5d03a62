#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4fuzz10; mkdir -p $O; export TDTK_LIB=lab
for seed in 8401 8402; do timeout 400 python tools/fuzz_parity.py --seconds 220 --seed $seed > $O/small_$seed.log 2>&1; grep -v amdgpu $O/small_$seed.log | tail -2; done
for seed in 8501 8502; do timeout 500 python tools/fuzz_parity.py --big --seconds 250 --seed $seed > $O/big_$seed.log 2>&1; grep -v amdgpu $O/big_$seed.log | tail -2; done
timeout 400 python tools/fuzz_graph.py --seconds 220 --seed 8601 > $O/graph.log 2>&1; grep -v amdgpu $O/graph.log | tail -2
