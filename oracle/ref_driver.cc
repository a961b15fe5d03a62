/*
 * ref_driver.cc -- TEST INFRASTRUCTURE ONLY (our code, not reference code).
 *
 * A thin extern "C" face over the *reference's own* translation units
 * (src/slam6d/kdIndexed.cc, icp6Dquat.cc, icp6Dsvd.cc + vendored newmat,
 * icp6Dapx.cc, icp6Dnapx.cc, icp6Dortho.cc, icp6Ddual.cc, icp6Dhelix.cc,
 * icp6Dlumeuler.cc, icp6Dlumquat.cc, icp6Dquatscale.cc, pointfilter.cc), compiled where they lie under $REF by
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
#include "slam6d/globals.icc"
#include "slam6d/pairingMode.h"
#include "slam6d/pointfilter.h"
#include "newmat/newmatap.h"

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

/* Full ICP iterations as an OpenMP build of the reference runs them (icp6D::match, icp6D.cc:129-222 with
 * Scan::getPtPairsParallel scan.cc:1285-1353 and SearchTree::getPtPairs searchTree.cc:92-189 for the closest-point
 * mode): the loop scaffolding is restated here (those three TUs need Boost), every piece of arithmetic inside it is
 * the reference's own compiled code -- KDtreeIndexed::FindClosest, transform3, M4inv, Dist2, PtPair, icp6D_QUAT::
 * Align_Parallel -- including the `omp critical` around every push_back (searchTree.cc:179-180), the second pass
 * over the 208-byte pairs for Si (icp6D.cc:170-191) and the serial transformReduced (scan.cc:851-875).  T = the
 * value OPENMP_NUM_THREADS would have been compiled with.  xyz [N][3] is the data scan's "xyz reduced", moved in
 * place; trace [iters][18] = {pairs, rms, alignxf}.  This is bench.py's full-iteration cpu_baseline.          */
int ref_icp_iterations(void* h, const double* model_dalignxf, double* xyz, size_t N, double maxdist2, int T, int iters,
                       double* trace)
{
  RefTree* t = static_cast<RefTree*>(h);
  if (T < 1) T = 1;
  if (T > MAX_OPENMP_NUM_THREADS) T = MAX_OPENMP_NUM_THREADS;
  std::vector<unsigned int> n(T);
  std::vector<double> sum(T), cm(3 * (size_t)T), cd(3 * (size_t)T), Si(9 * (size_t)T);
  icp6D_QUAT quat(true);
  for (int it = 0; it < iters; it++) {
    std::vector<std::vector<PtPair> > pairs(T);
    const int step = (int)ceil((double)N / (double)T);
    for (int i = 0; i < T; i++) {
      sum[i] = 0.0; n[i] = 0;
      for (int k = 0; k < 3; k++) cm[3 * i + k] = cd[3 * i + k] = 0.0;
      for (int k = 0; k < 9; k++) Si[9 * i + k] = 0.0;
    }
#pragma omp parallel num_threads(T)
    {
      const int tn = omp_get_thread_num();
      double* centroid_m = &cm[3 * tn];
      double* centroid_d = &cd[3 * tn];
      size_t start = (size_t)tn * (size_t)step;
      size_t end = (tn == T - 1) ? N : start + (size_t)step;
      if (start > N) start = N;
      if (end > N) end = N;
      double local_alignxf_inverse[16];
      M4inv(model_dalignxf, local_alignxf_inverse);                              // searchTree.cc:110
      for (size_t i = start; i < end; i++) {
        double tp[3] = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
        double sp[3];
        transform3(local_alignxf_inverse, tp, sp);                               // :122
        const size_t r = t->tree->FindClosest(sp, maxdist2, tn);                  // :143
        if (r != std::numeric_limits<size_t>::max()) {
          transform3(model_dalignxf, t->ptrs[r], sp);                              // :147
          centroid_m[0] += sp[0]; centroid_m[1] += sp[1]; centroid_m[2] += sp[2];
          centroid_d[0] += tp[0]; centroid_d[1] += tp[1]; centroid_d[2] += tp[2];
          PtPair myPair(sp, tp);
          double p12[3] = { myPair.p1.x - myPair.p2.x, myPair.p1.y - myPair.p2.y, myPair.p1.z - myPair.p2.z };
          sum[tn] += Len2(p12);                                                   // :172-177
#pragma omp critical
          pairs[tn].push_back(myPair);                                            // :179-180
        }
      }
      const size_t size = pairs[tn].size();
      if (size != 0)
        for (int k = 0; k < 3; k++) { centroid_m[k] /= size; centroid_d[k] /= size; }   // scan.cc:1346-1352
      n[tn] = (unsigned int)size;
      double* S = &Si[9 * tn];
      for (unsigned int i = 0; i < n[tn]; i++) {                                  // icp6D.cc:170-191
        const double pp[3] = { pairs[tn][i].p1.x - centroid_m[0], pairs[tn][i].p1.y - centroid_m[1], pairs[tn][i].p1.z - centroid_m[2] };
        const double qq[3] = { pairs[tn][i].p2.x - centroid_d[0], pairs[tn][i].p2.y - centroid_d[1], pairs[tn][i].p2.z - centroid_d[2] };
        S[0] += pp[0] * qq[0]; S[1] += pp[0] * qq[1]; S[2] += pp[0] * qq[2];
        S[3] += pp[1] * qq[0]; S[4] += pp[1] * qq[1]; S[5] += pp[1] * qq[2];
        S[6] += pp[2] * qq[0]; S[7] += pp[2] * qq[1]; S[8] += pp[2] * qq[2];
      }
    }
    unsigned int pairssize = 0;
    for (int i = 0; i < T; i++) pairssize += n[i];
    double alignxf[16];
    M4identity(alignxf);
    double ret = 0.0;
    if (pairssize > 3)
      ret = quat.Align_Parallel(T, n.data(), sum.data(), reinterpret_cast<const double(*)[3]>(cm.data()),
                                reinterpret_cast<const double(*)[3]>(cd.data()),
                                reinterpret_cast<const double(*)[9]>(Si.data()), alignxf);
    for (size_t i = 0; i < N; i++) transform3(alignxf, xyz + 3 * i);             // Scan::transformReduced, serial
    if (trace) {
      trace[18 * it] = (double)pairssize; trace[18 * it + 1] = ret;
      memcpy(trace + 18 * it + 2, alignxf, sizeof alignxf);
    }
  }
  return 0;
}

/* SearchTree::getPtPairs, DataXYZ overload (searchTree.cc:92-189), all three pairing modes: the loop is restated here
 * (searchTree.cc includes scan.h -> Boost), every operation in it is the reference's compiled code -- M4inv, transform3,
 * Normalize3, transform3normal, KDtreeIndexed::FindClosest / FindClosestAlongDir, sub3 / Dot / scal_mul3 / add3, PtPair,
 * Len2.  idx [n] (-1 = none); p1 / p2 / pn [pairs][3] compact in query order; sums = {sum, centroid_m[3], centroid_d[3]}
 * un-normalised as the reference accumulates them.  Returns the number of pairs.                                   */
size_t ref_get_pt_pairs(void* h, const double* source_alignxf, const double* xyz_r, const double* normal_r, size_t n,
                        int pairing_mode, double max_dist_match2, int32_t* idx, double* p1, double* p2, double* pn,
                        double* sums)
{
  RefTree* tr = static_cast<RefTree*>(h);
  double local_alignxf_inv[16];
  M4inv(source_alignxf, local_alignxf_inv);
  double sum = 0, centroid_m[3] = {0, 0, 0}, centroid_d[3] = {0, 0, 0};
  size_t np = 0;
  double t[3], s[3], normal[3] = {0, 0, 0};
  for (size_t i = 0; i < n; i++) {
    t[0] = xyz_r[3 * i]; t[1] = xyz_r[3 * i + 1]; t[2] = xyz_r[3 * i + 2];
    transform3(local_alignxf_inv, t, s);
    if (pairing_mode != CLOSEST_POINT) {
      normal[0] = normal_r[3 * i]; normal[1] = normal_r[3 * i + 1]; normal[2] = normal_r[3 * i + 2];
      Normalize3(normal);
    }
    size_t r;
    if (pairing_mode == CLOSEST_POINT_ALONG_NORMAL_SIMPLE) {
      transform3normal(local_alignxf_inv, normal);
      r = tr->tree->FindClosestAlongDir(s, normal, max_dist_match2, 0);
    } else {
      r = tr->tree->FindClosest(s, max_dist_match2, 0);
    }
    const bool found = r != std::numeric_limits<size_t>::max();
    if (idx) idx[i] = found ? (int32_t)r : -1;
    if (!found) continue;
    transform3(source_alignxf, tr->ptrs[r], s);
    if (pairing_mode == CLOSEST_PLANE_SIMPLE) {
      double tmp[3], s_[3];
      sub3(s, t, tmp);
      const double dot = Dot(normal, tmp);
      scal_mul3(normal, dot, tmp);
      add3(tmp, t, s_);
      s[0] = s_[0]; s[1] = s_[1]; s[2] = s_[2];
    }
    centroid_m[0] += s[0]; centroid_m[1] += s[1]; centroid_m[2] += s[2];
    centroid_d[0] += t[0]; centroid_d[1] += t[1]; centroid_d[2] += t[2];
    PtPair myPair(s, t, normal);
    double p12[3] = { myPair.p1.x - myPair.p2.x, myPair.p1.y - myPair.p2.y, myPair.p1.z - myPair.p2.z };
    sum += Len2(p12);
    if (p1) { p1[3 * np] = myPair.p1.x; p1[3 * np + 1] = myPair.p1.y; p1[3 * np + 2] = myPair.p1.z; }
    if (p2) { p2[3 * np] = myPair.p2.x; p2[3 * np + 1] = myPair.p2.y; p2[3 * np + 2] = myPair.p2.z; }
    if (pn) { pn[3 * np] = myPair.p2.nx; pn[3 * np + 1] = myPair.p2.ny; pn[3 * np + 2] = myPair.p2.nz; }
    np++;
  }
  if (sums) {
    sums[0] = sum;
    for (int k = 0; k < 3; k++) { sums[1 + k] = centroid_m[k]; sums[4 + k] = centroid_d[k]; }
  }
  return np;
}

/* ---- the 4x4 / pose primitives of include/slam6d/globals.icc, as the reference compiles them ------------
 * (A12: M4inv :762-785, MMult :298-328, transform3 :1454-1490, transform3normal :1465-1475,
 * EulerToMatrix4 :501-531, Matrix4ToEuler :540-576, QuatToMatrix4 :988-1022, Matrix4ToQuat :1032-1075)      */
int ref_M4inv(const double* in, double* out) { return M4inv(in, out); }
void ref_MMult(const double* a, const double* b, double* out) { MMult(a, b, out); }
void ref_transform3_inplace(const double* alignxf, double* pts, size_t n)
{
  for (size_t i = 0; i < n; i++) transform3(alignxf, pts + 3 * i);
}
void ref_transform3(const double* alignxf, const double* in, double* out, size_t n)
{
  for (size_t i = 0; i < n; i++) transform3(alignxf, in + 3 * i, out + 3 * i);
}
void ref_transform3normal(const double* alignxf, double* nrm, size_t n)
{
  for (size_t i = 0; i < n; i++) transform3normal(alignxf, nrm + 3 * i);
}
void ref_EulerToMatrix4(const double* rPos, const double* rPosTheta, double* alignxf) { EulerToMatrix4(rPos, rPosTheta, alignxf); }
void ref_Matrix4ToEuler(const double* alignxf, double* rPosTheta, double* rPos) { Matrix4ToEuler(alignxf, rPosTheta, rPos); }
void ref_QuatToMatrix4(const double* quat, const double* t, double* mat) { QuatToMatrix4(quat, t, mat); }
void ref_Matrix4ToQuat(const double* mat, double* quat, double* t) { Matrix4ToQuat(mat, quat, t); }
double ref_Dist2(const double* a, const double* b) { return Dist2(a, b); }

/* newmat's `.i()` as lum6DEuler::covarianceEuler uses it (lum6Deuler.cc:194: D = MM.i() * MZ): A is n x n row-major.
 * Ainv (nullable) = A.i(); x (nullable) = A.i() * b.                                                          */
int ref_newmat_inverse_solve(int n, const double* A, const double* b, double* Ainv, double* x)
{
  try {
    NEWMAT::Matrix M(n, n);
    for (int r = 0; r < n; r++)
      for (int c = 0; c < n; c++) M(r + 1, c + 1) = A[r * n + c];
    NEWMAT::Matrix Mi = M.i();
    if (Ainv)
      for (int r = 0; r < n; r++)
        for (int c = 0; c < n; c++) Ainv[r * n + c] = Mi(r + 1, c + 1);
    if (x && b) {
      NEWMAT::ColumnVector B(n);
      for (int r = 0; r < n; r++) B(r + 1) = b[r];
      NEWMAT::ColumnVector X = M.i() * B;
      for (int r = 0; r < n; r++) x[r] = X(r + 1);
    }
  } catch (...) {
    return -1;
  }
  return 0;
}

/* lum6DEuler::covarianceEuler's arithmetic after the pair search (lum6Deuler.cc:143-232) on an explicit pair
 * list, written with the same newmat objects and expressions the reference TU uses (the TU itself needs Boost
 * through scan.h); p1 = ak (first scan's points), p2 = bk.  Returns m; C 6x6 row-major, CD 6.                */
int ref_lum_covariance_euler(size_t m, const double* p1, const double* p2, double* C, double* CD, double* ss_out,
                             double* D_out)
{
  using namespace NEWMAT;
  Matrix MM(6, 6);
  ColumnVector MZ(6), D(6);
  MM = 0.0; MZ = 0.0;
  double sum[3] = {0, 0, 0}, xpy = 0, xpz = 0, ypz = 0, xy = 0, yz = 0, xz = 0;
  for (size_t j = 0; j < m; j++) {
    const double* a = p1 + 3 * j;
    const double* b = p2 + 3 * j;
    const double x = (a[0] + b[0]) / 2.0, y = (a[1] + b[1]) / 2.0, z = (a[2] + b[2]) / 2.0;
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    sum[0] += x; sum[1] += y; sum[2] += z;
    xpy += x * x + y * y; xpz += x * x + z * z; ypz += y * y + z * z;
    xy += x * y; xz += x * z; yz += y * z;
    MZ(1) += dx; MZ(2) += dy; MZ(3) += dz;
    MZ(4) += -z * dy + y * dz;
    MZ(5) += -y * dx + x * dy;
    MZ(6) += z * dx - x * dz;
  }
  MM(1, 1) = MM(2, 2) = MM(3, 3) = (double)m;
  MM(4, 4) = ypz; MM(5, 5) = xpy; MM(6, 6) = xpz;
  MM(1, 5) = MM(5, 1) = -sum[1]; MM(1, 6) = MM(6, 1) = sum[2];
  MM(2, 4) = MM(4, 2) = -sum[2]; MM(2, 5) = MM(5, 2) = sum[0];
  MM(3, 4) = MM(4, 3) = sum[1];  MM(3, 6) = MM(6, 3) = -sum[0];
  MM(4, 5) = MM(5, 4) = -xz; MM(4, 6) = MM(6, 4) = -xy; MM(5, 6) = MM(6, 5) = -yz;
  D = MM.i() * MZ;
  double ss = 0.0;
  for (size_t j = 0; j < m; j++) {
    const double* a = p1 + 3 * j;
    const double* b = p2 + 3 * j;
    const double x = (a[0] + b[0]) / 2.0, y = (a[1] + b[1]) / 2.0, z = (a[2] + b[2]) / 2.0;
    ss += sqr((a[0] - b[0]) - (D(1) - y * D(5) + z * D(6))) +
          sqr((a[1] - b[1]) - (D(2) - z * D(4) + x * D(5))) +
          sqr((a[2] - b[2]) - (D(3) + y * D(4) - x * D(6)));
  }
  ss = ss / (2 * m - 3);
  if (ss_out) *ss_out = ss;
  if (D_out) for (int k = 0; k < 6; k++) D_out[k] = D(k + 1);
  for (int k = 0; k < 36; k++) C[k] = 0.0;
  for (int k = 0; k < 6; k++) CD[k] = 0.0;
  if (ss < 0.0000000000001) return (int)m;
  ss = 1.0 / ss;
  Matrix Cm = MM * ss;
  ColumnVector CDv = MZ * ss;
  for (int r = 0; r < 6; r++) {
    for (int c = 0; c < 6; c++) C[r * 6 + c] = Cm(r + 1, c + 1);
    CD[r] = CDv(r + 1);
  }
  return (int)m;
}

/* PointFilter with the -m / -M range (pointfilter.cc:63-70 setRange -> parameter strings -> CheckerRangeMax / CheckerRangeMin,
 * :162-188), applied to n points exactly as BasicScan does (basicScan.cc:160: filter.setRange(max, min); check(point) per
 * point): keep[i] = 1 if the reference's own compiled filter accepts point i.  Returns the number kept. */
size_t ref_point_filter_range(const double* xyz, size_t n, double maxDist, double minDist, unsigned char* keep)
{
  PointFilter filter;
  filter.setRange(maxDist, minDist);
  size_t kept = 0;
  for (size_t i = 0; i < n; i++) {
    double p[3] = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
    keep[i] = filter.check(p) ? 1 : 0;
    kept += keep[i];
  }
  return kept;
}

}  // extern "C"
