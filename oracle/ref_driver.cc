/*
 * ref_driver.cc -- TEST INFRASTRUCTURE ONLY (our code, not reference code).
 *
 * A thin extern "C" face over the *reference's own* translation units
 * (src/slam6d/kdIndexed.cc, icp6Dquat.cc, icp6Dsvd.cc + vendored newmat,
 * icp6Dapx.cc, icp6Dnapx.cc, icp6Dortho.cc, icp6Ddual.cc, icp6Dhelix.cc,
 * icp6Dlumeuler.cc, icp6Dlumquat.cc, icp6Dquatscale.cc), compiled where they lie under $REF by
 * oracle/build_ref.sh into oracle/_ref/libref3dtk.so.  Nothing from the
 * reference is copied into this repository; this file only #includes the
 * reference headers at build time.  Used to (a) pin oracle/oracle.c and the
 * numpy minimizer restatements, (b) generate tests/golden fixtures, (c) serve as
 * bench.py's cpu_baseline of kind "reference".
 *
 * Not built here: searchTree.cc / scan.cc / icp6D.cc / lum6Deuler.cc need
 * Boost and SuiteSparse, which this image lacks -> those are restated in
 * oracle/ and pinned through the pieces below (see DESIGN.md "Oracle").
 */
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#include <omp.h>

#include "slam6d/kdIndexed.h"
#include "slam6d/icp6Dquat.h"
#include "slam6d/icp6Dsvd.h"
#include "slam6d/icp6Dapx.h"
#include "slam6d/icp6Dnapx.h"
#include "slam6d/icp6Dortho.h"
#include "slam6d/icp6Ddual.h"
#include "slam6d/icp6Dhelix.h"
#include "slam6d/icp6Dlumeuler.h"
#include "slam6d/icp6Dlumquat.h"
#include "slam6d/icp6Dquatscale.h"

struct RefTree {
  std::vector<double*> ptrs;
  KDtreeIndexed* tree;
};

static void fill_pairs(std::vector<PtPair>& v, size_t n, const double* p1, const double* p2,
                       const double* nrm)
{
  v.reserve(n);
  for (size_t i = 0; i < n; i++) {
    double a[3] = { p1[3 * i], p1[3 * i + 1], p1[3 * i + 2] };
    double b[3] = { p2[3 * i], p2[3 * i + 1], p2[3 * i + 2] };
    if (nrm) {
      double c[3] = { nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2] };
      v.push_back(PtPair(a, b, c));
    } else {
      v.push_back(PtPair(a, b));
    }
  }
}

extern "C" {

int ref_openmp_num_threads(void) { return OPENMP_NUM_THREADS; }
int ref_max_openmp_num_threads(void) { return MAX_OPENMP_NUM_THREADS; }
int ref_host_threads(void) { return omp_get_max_threads(); }

/* KDtreeIndexed(double**, size_t, int): same KDTreeImpl::create / _FindClosest
 * template as KDtree, index-returning (include/slam6d/kdIndexed.h).          */
void* ref_kdi_create(const double* xyz, size_t M, int bucket)
{
  RefTree* t = new RefTree;
  t->ptrs.resize(M);
  for (size_t i = 0; i < M; i++) t->ptrs[i] = const_cast<double*>(xyz + 3 * i);
  t->tree = new KDtreeIndexed(t->ptrs.data(), M, bucket);
  return t;
}

void ref_kdi_destroy(void* h)
{
  RefTree* t = static_cast<RefTree*>(h);
  delete t->tree;
  delete t;
}

/* batched FindClosest; threadNum = omp thread id exactly like the reference's
 * OpenMP callers (icp6D.cc:159-166).  idx = -1 when none.                     */
void ref_kdi_find_closest(void* h, const double* q, size_t K, double maxdist2, int32_t* idx,
                          int nthreads)
{
  RefTree* t = static_cast<RefTree*>(h);
  if (nthreads < 1) nthreads = 1;
  if (nthreads > MAX_OPENMP_NUM_THREADS) nthreads = MAX_OPENMP_NUM_THREADS;
#pragma omp parallel for schedule(static) num_threads(nthreads)
  for (long i = 0; i < (long)K; i++) {
    double p[3] = { q[3 * i], q[3 * i + 1], q[3 * i + 2] };
    size_t r = t->tree->FindClosest(p, maxdist2, omp_get_thread_num());
    idx[i] = (r == std::numeric_limits<size_t>::max()) ? -1 : (int32_t)r;
  }
}

void ref_kdi_find_closest_along_dir(void* h, const double* q, const double* dir, size_t K,
                                    double maxdist2, int32_t* idx)
{
  RefTree* t = static_cast<RefTree*>(h);
  for (size_t i = 0; i < K; i++) {
    double p[3] = { q[3 * i], q[3 * i + 1], q[3 * i + 2] };
    double d[3] = { dir[3 * i], dir[3 * i + 1], dir[3 * i + 2] };
    size_t r = t->tree->FindClosestAlongDir(p, d, maxdist2, 0);
    idx[i] = (r == std::numeric_limits<size_t>::max()) ? -1 : (int32_t)r;
  }
}

/* serial Align of minimizer `algo` (1 QUAT, 2 SVD, 6 APX, 10 NAPX: the -a ids of
 * src/slam6d/slam6D.cc:696-727) on explicit pair lists                         */
double ref_align(int algo, size_t n, const double* p1, const double* p2, const double* nrm,
                 const double* cm, const double* cd, double* alignxf)
{
  std::vector<PtPair> pairs;
  fill_pairs(pairs, n, p1, p2, nrm);
  for (int i = 0; i < 16; i++) alignxf[i] = (i % 5 == 0) ? 1.0 : 0.0;
  switch (algo) {
    case 1: { icp6D_QUAT m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 2: { icp6D_SVD m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 6: { icp6D_APX m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 10: { icp6D_NAPX m(true); return m.Align(pairs, alignxf, cm, cd); }
  }
  return -2.0;
}

/* the serial-only minimizers, by their -a id (slam6D.cc:703-723): 3 ORTHO, 4 DUAL, 5 HELIX, 7 LUMEULER,
 * 8 LUMQUAT, 9 QUAT_SCALE.  alignxf is in/out: icp6D::match hands LUMEULER / LUMQUAT the current scan's
 * transMat in it (icp6D.cc:237-241).                                                                  */
double ref_align_inout(int algo, size_t n, const double* p1, const double* p2, const double* cm,
                       const double* cd, double* alignxf)
{
  std::vector<PtPair> pairs;
  fill_pairs(pairs, n, p1, p2, nullptr);
  switch (algo) {
    case 3: { icp6D_ORTHO m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 4: { icp6D_DUAL m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 5: { icp6D_HELIX m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 7: { icp6D_LUMEULER m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 8: { icp6D_LUMQUAT m(true); return m.Align(pairs, alignxf, cm, cd); }
    case 9: { icp6D_QUAT_SCALE m(true); return m.Align(pairs, alignxf, cm, cd); }
  }
  return -2.0;
}

/* Align_Parallel of QUAT (1) / SVD (2): arrays are [OPENMP_NUM_THREADS] long     */
double ref_align_parallel(int algo, const unsigned int* n, const double* sum, const double* cm,
                          const double* cd, const double* Si, double* alignxf)
{
  unsigned int nn[OPENMP_NUM_THREADS];
  double ss[OPENMP_NUM_THREADS], m[OPENMP_NUM_THREADS][3], d[OPENMP_NUM_THREADS][3],
      S[OPENMP_NUM_THREADS][9];
  for (int t = 0; t < OPENMP_NUM_THREADS; t++) {
    nn[t] = n[t]; ss[t] = sum[t];
    for (int k = 0; k < 3; k++) { m[t][k] = cm[3 * t + k]; d[t][k] = cd[3 * t + k]; }
    for (int k = 0; k < 9; k++) S[t][k] = Si[9 * t + k];
  }
  if (algo == 1) { icp6D_QUAT q(true); return q.Align_Parallel(OPENMP_NUM_THREADS, nn, ss, m, d, S, alignxf); }
  if (algo == 2) { icp6D_SVD q(true); return q.Align_Parallel(OPENMP_NUM_THREADS, nn, ss, m, d, S, alignxf); }
  return -2.0;
}

/* APX Align_Parallel: pairs chunked per thread by counts n[t] (icp6Dapx.cc:136-307) */
double ref_apx_align_parallel(const unsigned int* n, const double* sum, const double* cm,
                              const double* cd, const double* p1, const double* p2,
                              double* alignxf)
{
  unsigned int nn[OPENMP_NUM_THREADS];
  double ss[OPENMP_NUM_THREADS], m[OPENMP_NUM_THREADS][3], d[OPENMP_NUM_THREADS][3];
  std::vector<PtPair> pairs[OPENMP_NUM_THREADS];
  size_t off = 0;
  for (int t = 0; t < OPENMP_NUM_THREADS; t++) {
    nn[t] = n[t]; ss[t] = sum[t];
    for (int k = 0; k < 3; k++) { m[t][k] = cm[3 * t + k]; d[t][k] = cd[3 * t + k]; }
    fill_pairs(pairs[t], n[t], p1 + 3 * off, p2 + 3 * off, nullptr);
    off += n[t];
  }
  omp_set_num_threads(OPENMP_NUM_THREADS);
  icp6D_APX q(true);
  return q.Align_Parallel(OPENMP_NUM_THREADS, nn, ss, m, d, pairs, alignxf);
}

}  // extern "C"
