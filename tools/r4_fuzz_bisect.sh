#!/bin/bash
cd "$GRAFT_REPO_ROOT"; F=tests/golden/fuzz_case_dynamic_range_20000.npz
python tools/r4_fuzz_repro.py $F 2>&1 | tail -1
TDTK_BUILD_SPEC=0 python tools/r4_fuzz_repro.py $F 2>&1 | tail -1
TDTK_LIB=lab python tools/r4_fuzz_repro.py $F 2>&1 | tail -1
timeout 400 python tools/fuzz_parity.py --seconds 240 --seed 4401 2>&1 | tail -2
