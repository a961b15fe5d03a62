#!/bin/bash
# round 6: the tree build's new pieces one at a time (lab switches), dat/ scan + 1M uniform (tools/tree_probe.py)
cd "$GRAFT_REPO_ROOT"
for kv in NONE=1 TDTK_BUILD_FINWAVE=0 TDTK_BUILD_FINHALF=0 TDTK_BUILD_CHAINFROM=99 TDTK_BUILD_STREAMS=3 TDTK_BUILD_PART=0; do
  echo "== $kv"; env TDTK_LIB=lab $kv timeout 60 python tools/tree_probe.py 2>&1 | tail -2
done
