#!/bin/bash
# the lab library's arena guards: its tests, then the differential runs with every tree built under them
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r4guard; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lab or alternative or lopsided or normals" > $O/tests.log 2>&1; tail -3 $O/tests.log
TDTK_LIB=lab timeout 400 python tools/fuzz_parity.py --seconds 200 --seed 4401 > $O/small.log 2>&1; grep -v amdgpu $O/small.log | tail -3
TDTK_LIB=lab timeout 400 python tools/fuzz_parity.py --big --seconds 200 --seed 9401 > $O/big.log 2>&1; grep -v amdgpu $O/big.log | tail -3
TDTK_LIB=lab timeout 300 python tools/fuzz_graph.py --seconds 100 --seed 502 > $O/graph.log 2>&1; grep -v amdgpu $O/graph.log | tail -3
