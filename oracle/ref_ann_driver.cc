/*
 * ref_ann_driver.cc -- TEST INFRASTRUCTURE ONLY (our code, not reference code).
 *
 * extern "C" face over the vendored ANN 1.1.1 (3rdparty/ann/ann_1.1.1_modified) and newmat's
 * EigenValues, the two libraries Scan::calcNormals stands on (scan.cc:419 ->
 * calculateNormalsApxKNN, normals.cc:35-111), compiled where they lie under $REF by
 * oracle/build_ref.sh into oracle/_ref/libref3dtk.so.
 *
 * normals.cc itself does not compile in this image (normals.h -> scan.h -> Boost), so
 * ref_normals_apx_knn() below restates its ~40 lines of glue (neighbour mean, covariance,
 * smallest eigenvector, flip towards the sensor) around the REAL annkSearch and the REAL
 * newmat EigenValues.  What this pins: the ANN tree, the approximate k-NN result lists (indices,
 * order, distances) and the eigen solver; the glue alone is "parity unpinned".
 */
#include <cmath>
#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include <ANN/ANN.h>
#include <ANN/ANNperf.h>
#include "newmat/newmat.h"
#include "newmat/newmatap.h"

using namespace NEWMAT;

struct RefAnn {
  ANNpointArray pa;
  ANNkd_tree* tree;
  int n;
};

extern "C" {

/* ANNkd_tree(pa, n, 3): bucket size 1, ANN_KD_SUGGEST = sliding midpoint (normals.cc:47-54) */
void* ref_ann_create(const double* xyz, int n)
{
  RefAnn* h = new RefAnn;
  h->n = n;
  h->pa = annAllocPts(n, 3);
  for (int i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) h->pa[i][d] = xyz[3 * i + d];
  h->tree = new ANNkd_tree(h->pa, n, 3);
  return h;
}

void ref_ann_destroy(void* hv)
{
  RefAnn* h = (RefAnn*)hv;
  delete h->tree;
  annDeallocPts(h->pa);
  delete h;
}

/* annkSearch for a batch of queries, one after the other (the library keeps its search state in globals) */
void ref_ann_ksearch(void* hv, const double* q, int nq, int k, double eps, int32_t* idx, double* dist)
{
  RefAnn* h = (RefAnn*)hv;
  std::vector<ANNidx> ni(k);
  std::vector<ANNdist> dd(k);
  double p[3];
  for (int i = 0; i < nq; i++) {
    p[0] = q[3 * i]; p[1] = q[3 * i + 1]; p[2] = q[3 * i + 2];
    h->tree->annkSearch(p, k, ni.data(), dd.data(), eps);
    for (int j = 0; j < k; j++) { idx[(size_t)i * k + j] = ni[j]; dist[(size_t)i * k + j] = dd[j]; }
  }
}

/* out[0..5] = depth, leaves, trivial leaves, splitting nodes, shrinking nodes, points */
void ref_ann_stats(void* hv, long* out)
{
  RefAnn* h = (RefAnn*)hv;
  ANNkdStats st;
  h->tree->getStats(st);
  out[0] = st.depth; out[1] = st.n_lf; out[2] = st.n_tl; out[3] = st.n_spl; out[4] = st.n_shr; out[5] = st.n_pts;
}

/* The tree in the library's own pre-order dump format, digested: for every splitting node its cutting
 * dimension (cut_dim[]), for every leaf its point index (leaf_pt[]), both in dump order.  Returns the
 * number of splitting nodes, or -1 if a leaf holds more or less than one point.  (Cut values are printed
 * with 15 digits only: cut_val[] is for approximate comparison, the searches pin them exactly.) */
long ref_ann_structure(void* hv, int32_t* cut_dim, double* cut_val, int32_t* leaf_pt, long* n_leaves)
{
  RefAnn* h = (RefAnn*)hv;
  std::ostringstream os;
  h->tree->Dump(ANNfalse, os);
  std::istringstream is(os.str());
  std::string tok;
  long ns = 0, nl = 0;
  while (is >> tok) {
    if (tok == "split") {
      int cd; double cv, lo, hi;
      is >> cd >> cv >> lo >> hi;
      if (cut_val) cut_val[ns] = cv;
      cut_dim[ns++] = cd;
    } else if (tok == "leaf") {
      int cnt, id;
      is >> cnt;
      if (cnt != 1) return -1;
      is >> id;
      leaf_pt[nl++] = id;
    }
  }
  *n_leaves = nl;
  return ns;
}

/* newmat's EigenValues(SymmetricMatrix, D, U) on a 3x3 (lower triangle of a[9], row-major):
 * eigenvalues ascending in d[3], eigenvectors in the columns of u[9] (row-major). */
void ref_eigen3(const double* a, double* d, double* u)
{
  SymmetricMatrix A(3);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c <= r; c++) A(r + 1, c + 1) = a[3 * r + c];
  DiagonalMatrix D(3);
  Matrix U(3, 3);
  EigenValues(A, D, U);
  for (int r = 0; r < 3; r++) {
    d[r] = D(r + 1);
    for (int c = 0; c < 3; c++) u[3 * r + c] = U(r + 1, c + 1);
  }
}

/* calculateNormalsApxKNN(normals, points, k, rPos, eps), normals.cc:35-111, glue restated (see header) */
void ref_normals_apx_knn(const double* xyz, int n, int k, const double* rPos, double eps, double* normals)
{
  RefAnn* h = (RefAnn*)ref_ann_create(xyz, n);
  std::vector<ANNidx> nidx(k);
  std::vector<ANNdist> d(k);
  for (int i = 0; i < n; i++) {
    ANNpoint p = h->pa[i];
    h->tree->annkSearch(p, k, nidx.data(), d.data(), eps);
    double mean[3] = { 0.0, 0.0, 0.0 };
    Matrix X(k, 3);
    SymmetricMatrix A(3);
    Matrix U(3, 3);
    DiagonalMatrix D(3);
    for (int j = 0; j < k; j++)
      for (int c = 0; c < 3; c++) mean[c] += xyz[3 * nidx[j] + c];
    for (int c = 0; c < 3; c++) mean[c] /= k;
    for (int j = 0; j < k; j++)
      for (int c = 0; c < 3; c++) X(j + 1, c + 1) = xyz[3 * nidx[j] + c] - mean[c];
    A << 1.0 / k * X.t() * X;
    EigenValues(A, D, U);
    ColumnVector nv(3);
    nv(1) = U(1, 1); nv(2) = U(2, 1); nv(3) = U(3, 1);
    ColumnVector pv(3);
    pv(1) = p[0] - rPos[0]; pv(2) = p[1] - rPos[1]; pv(3) = p[2] - rPos[2];
    pv = pv / pv.NormFrobenius();
    Real angle = (nv.t() * pv).AsScalar();
    if (angle < 0) nv *= -1.0;
    nv = nv / nv.NormFrobenius();
    normals[3 * i] = nv(1); normals[3 * i + 1] = nv(2); normals[3 * i + 2] = nv(3);
  }
  ref_ann_destroy(h);
}

}  // extern "C"
