#!/usr/bin/env bash
# TEST INFRASTRUCTURE: build the reference's own translation units for the hot
# path, from where they lie under $REF, into oracle/_ref/libref3dtk.so.
#   usage: oracle/build_ref.sh [REF=/root/reference]
# Nothing is copied into the repository: objects and the .so go to oracle/_ref/
# (git-ignored; it travels to the GPU box with the snapshot).  No stand-in
# headers: only TUs that compile as-is with this image's g++ are built
# (kdIndexed.cc, icp6Dquat.cc, icp6Dsvd.cc + vendored newmat, icp6Dapx.cc,
# icp6Dnapx.cc, icp6Dortho.cc, icp6Ddual.cc, icp6Dhelix.cc, icp6Dlumeuler.cc,
# icp6Dlumquat.cc, icp6Dquatscale.cc, pointfilter.cc; the vendored ANN library).  Flags mirror the reference CMakeLists.txt:306-342 (-O3,
# OpenMP, no -march, no -ffast-math).
set -euo pipefail
REF="${1:-${REF:-/root/reference}}"
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF/src/slam6d" ]; then
  echo "build_ref.sh: reference checkout not found at $REF (skipping)" >&2
  exit 3
fi
mkdir -p "$OUT/obj"
# OPENMP_NUM_THREADS sizes Align_Parallel's arrays; MAX_OPENMP_NUM_THREADS sizes
# the static KDParams array indexed by threadNum (kdTreeImpl.h:211-219).
F="-std=c++17 -O3 -fPIC -fopenmp -DOPENMP -DOPENMP_NUM_THREADS=8 -DMAX_OPENMP_NUM_THREADS=512 -w"
INC="-I$REF/include -I$REF/3rdparty/newmat/newmat-10"
for f in kdIndexed icp6Dquat icp6Dsvd icp6Dapx icp6Dnapx icp6Dortho icp6Ddual icp6Dhelix icp6Dlumeuler \
         icp6Dlumquat icp6Dquatscale pointfilter; do
  if [ ! -f "$OUT/obj/$f.o" ] || [ "$REF/src/slam6d/$f.cc" -nt "$OUT/obj/$f.o" ]; then
    g++ $F $INC -c "$REF/src/slam6d/$f.cc" -o "$OUT/obj/$f.o" &
  fi
done
NM="$REF/3rdparty/newmat/newmat-10/newmat"
# list = 3rdparty/newmat/CMakeLists.txt
for f in newmat1 newmat2 newmat3 newmat4 newmat5 newmat6 newmat7 newmat8 newmatex bandmat submat \
         myexcept cholesky evalue fft hholder jacobi newfft sort svd newmatrm newmat9; do
  if [ ! -f "$OUT/obj/nm_$f.o" ]; then
    g++ -O2 -fPIC -w -c "$NM/$f.cpp" -o "$OUT/obj/nm_$f.o" &
  fi
done
# vendored ANN 1.1.1 (list = 3rdparty/ann/CMakeLists.txt): the k-NN under Scan::calcNormals (normals.cc:35-111)
ANN="$REF/3rdparty/ann/ann_1.1.1_modified"
for f in ANN brute kd_tree kd_util kd_split kd_dump kd_search kd_pr_search kd_fix_rad_search bd_tree bd_search \
         bd_pr_search bd_fix_rad_search perf; do
  if [ ! -f "$OUT/obj/ann_$f.o" ]; then
    g++ -O3 -fPIC -w -I"$ANN/include" -c "$ANN/src/$f.cpp" -o "$OUT/obj/ann_$f.o" &
  fi
done
wait
g++ $F $INC -c "$HERE/ref_driver.cc" -o "$OUT/obj/ref_driver.o"
g++ $F $INC -I"$ANN/include" -c "$HERE/ref_ann_driver.cc" -o "$OUT/obj/ref_ann_driver.o"
g++ -shared -fopenmp -o "$OUT/libref3dtk.so" "$OUT"/obj/*.o
echo "built $OUT/libref3dtk.so"
