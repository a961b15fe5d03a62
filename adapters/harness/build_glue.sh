#!/usr/bin/env bash
# Builds the executed ICP glue harness (adapters/icp_glue.h instantiated with a minimal scan type; no reference header
# involved) -> adapters/harness/_bin/icp_glue_harness (git-ignored, travels to the GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_bin"
g++ -std=c++17 -O2 -Wall -pthread -I"$ROOT/include" "$HERE/icp_glue_harness.cc" \
    -L"$ROOT/3dtk_amd" -l3dtk_hip -Wl,-rpath,'$ORIGIN/../../../3dtk_amd' -Wl,-rpath,/opt/rocm/lib \
    -o "$HERE/_bin/icp_glue_harness"
echo "built $HERE/_bin/icp_glue_harness"
