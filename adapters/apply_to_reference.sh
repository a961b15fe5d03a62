#!/usr/bin/env bash
# Puts the reference-side binding into a 3DTK checkout:
#   adapters/apply_to_reference.sh <3dtk-checkout> [--check]
# copies hip_search_tree.{h,cc}, icp6D_hip.h, graphSlam6D_hip.h, slam6D_hip.h, the three glue headers, normals_hip.h and tdtk_hip.h to where the reference
# keeps such files and applies adapters/reference.patch (enum HipKD, `case HipKD:` in
# BasicScan::createSearchTreePrivate, Scan::transformMatrixAndFrames / hipResident, icp6D_hip at the four icp6D
# construction sites of slam6D.cc, the WITH_HIP_ICP CMake option).  --check only verifies that the patch applies.
# Then: cmake -DWITH_HIP_ICP=ON -DTDTK_HIP_DIR=<repo>/3dtk_amd -DTDTK_HIP_INCLUDE_DIR=<repo>/include ... ; bin/slam6D -t 4
set -euo pipefail
REF="$1"; MODE="${2:-apply}"
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ "$MODE" = "--check" ]; then
  (cd "$REF" && patch -p1 --dry-run --force < "$HERE/reference.patch")
  exit 0
fi
(cd "$REF" && patch -p1 --force < "$HERE/reference.patch")
cp "$HERE/hip_search_tree.h" "$HERE/icp6D_hip.h" "$HERE/icp_glue.h" "$HERE/graphSlam6D_hip.h" "$HERE/graph_slam_glue.h" "$HERE/slam6d_glue.h" "$HERE/slam6D_hip.h" "$HERE/normals_hip.h" "$REF/include/slam6d/"
cp "$HERE/../include/tdtk_hip.h" "$REF/include/"
cp "$HERE/hip_search_tree.cc" "$REF/src/slam6d/"
echo "lib3dtk_hip binding installed into $REF"
