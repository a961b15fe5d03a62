#!/usr/bin/env bash
# Builds the executed glue harnesses (adapters/icp_glue.h, graph_slam_glue.h and slam6d_glue.h instantiated with minimal
# scan types; no reference header involved) -> adapters/harness/_bin/{icp,slam}_glue_harness (git-ignored, travel to the
# GPU box).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_bin"
for h in icp_glue_harness slam_glue_harness; do
  g++ -std=c++17 -O2 -Wall -pthread -I"$ROOT/include" "$HERE/$h.cc" \
      -L"$ROOT/3dtk_amd" -l3dtk_hip -Wl,-rpath,'$ORIGIN/../../../3dtk_amd' -Wl,-rpath,/opt/rocm/lib \
      -o "$HERE/_bin/$h"
  echo "built $HERE/_bin/$h"
done
