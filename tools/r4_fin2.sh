#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "tree or build or verify or edge or octree or config1 or dat_ or meta or k5 or icp_glue or normals" 2>&1 | tail -3
bash tools/r4_fin_ab.sh 2>&1 | grep FINISH
