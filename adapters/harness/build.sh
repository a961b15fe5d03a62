#!/usr/bin/env bash
# Builds the executed reference-side binding: adapters/hip_search_tree.cc + harness against the reference's own
# headers (where a checkout exists) and lib3dtk_hip.so -> oracle/_ref/hip_search_tree_harness (git-ignored, travels
# to the GPU box).  usage: adapters/harness/build.sh [REF=/root/reference]
set -euo pipefail
REF="${1:-${REF:-/root/reference}}"
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/oracle/_ref"
if [ ! -d "$REF/include/slam6d" ]; then
  echo "harness/build.sh: reference checkout not found at $REF (skipping)" >&2
  exit 3
fi
mkdir -p "$OUT/inc/slam6d"
# the adapter header is meant to be dropped into include/slam6d/ of the reference: give the compiler that layout
# without touching the checkout
ln -sf "$ROOT/adapters/hip_search_tree.h" "$OUT/inc/slam6d/hip_search_tree.h"
g++ -std=c++17 -O2 -w -I"$OUT/inc" -I"$REF/include" -I"$ROOT/include" \
    "$ROOT/adapters/hip_search_tree.cc" "$HERE/hip_search_tree_harness.cc" \
    -L"$ROOT/3dtk_amd" -l3dtk_hip -Wl,-rpath,'$ORIGIN/../../3dtk_amd' -Wl,-rpath,/opt/rocm/lib \
    -o "$OUT/hip_search_tree_harness"
echo "built $OUT/hip_search_tree_harness"
