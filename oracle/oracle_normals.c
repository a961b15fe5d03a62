/*
 * oracle_normals.c -- TEST INFRASTRUCTURE ONLY (part of liboracle.so, see oracle.c for the rules).
 *
 * CPU restatement of Scan::calcNormals (src/slam6d/scan.cc:398-427) =
 * calculateNormalsApxKNN(normals, points, 10, rPos, 1.0) (src/slam6d/normals.cc:35-111):
 *   - the ANN 1.1.1 kd-tree the reference builds per scan (3rdparty/ann/ann_1.1.1_modified:
 *     bucket size 1, sliding-midpoint rule; src/kd_tree.cpp:319-404, src/kd_split.cpp:146-213,
 *     src/kd_util.cpp:75-92,225-319),
 *   - its (1+eps)-approximate k-nearest-neighbour search (src/kd_search.cpp:89-210,
 *     src/pr_queue_k.h:66-115),
 *   - the per-point PCA: neighbour mean, covariance, newmat EigenValues (tred2 / tql2 / SortSV,
 *     3rdparty/newmat/newmat-10/newmat/evalue.cpp:24-176,283-284, sort.cpp:190-222), smallest
 *     eigenvector flipped towards the sensor.
 *
 * Parity status: PINNED for the tree, the k-NN lists and the eigen solver -- tests/test_oracle_vs_ref.py
 * checks them against the vendored ANN and newmat compiled by oracle/build_ref.sh
 * (oracle/ref_ann_driver.cc).  The ~40 lines of glue in normals.cc cannot be compiled here
 * (normals.h -> scan.h -> Boost): for them alone "parity unpinned" -- ref_ann_driver.cc restates
 * them around the real libraries, and this file is checked against that.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int cut_dim;
  double cut_val, lo, hi; /* cd_bnds[ANN_LO], cd_bnds[ANN_HI] */
  int child[2];           /* >= 0: splitting node, < 0: leaf holding point ~child */
} ann_split;

typedef struct orc_ann {
  int n;
  double *pts; /* private copy, [n][3] */
  int *pidx;
  ann_split *sp;
  long nsp, cap;
  int root;
  int depth;
  double bb_lo[3], bb_hi[3];
} orc_ann;

#define PA(i, d) (t->pts[3 * (size_t)pidx[(i)] + (d)])
#define ANN_ERR 0.001 /* kd_split.cpp:34 */

/* kd_util.cpp:291-319 */
static void plane_split(const orc_ann *t, int *pidx, int n, int d, double cv, int *br1, int *br2)
{
  int l = 0, r = n - 1;
  for (;;) {
    while (l < n && PA(l, d) < cv) l++;
    while (r >= 0 && PA(r, d) >= cv) r--;
    if (l > r) break;
    int tmp = pidx[l]; pidx[l] = pidx[r]; pidx[r] = tmp;
    l++; r--;
  }
  *br1 = l;
  r = n - 1;
  for (;;) {
    while (l < n && PA(l, d) <= cv) l++;
    while (r >= *br1 && PA(r, d) > cv) r--;
    if (l > r) break;
    int tmp = pidx[l]; pidx[l] = pidx[r]; pidx[r] = tmp;
    l++; r--;
  }
  *br2 = l;
}

/* kd_util.cpp:225-243 (annSpread) and :245-262 (annMinMax) */
static void min_max(const orc_ann *t, const int *pidx, int n, int d, double *mn, double *mx)
{
  *mn = PA(0, d);
  *mx = PA(0, d);
  for (int i = 1; i < n; i++) {
    const double c = PA(i, d);
    if (c < *mn) *mn = c;
    else if (c > *mx) *mx = c;
  }
}

/* kd_split.cpp:146-213 */
static void sl_midpt_split(const orc_ann *t, int *pidx, const double *blo, const double *bhi, int n,
                           int *cut_dim, double *cut_val, int *n_lo)
{
  double max_length = bhi[0] - blo[0];
  for (int d = 1; d < 3; d++) {
    const double length = bhi[d] - blo[d];
    if (length > max_length) max_length = length;
  }
  double max_spread = -1;
  for (int d = 0; d < 3; d++) {
    if ((bhi[d] - blo[d]) >= (1 - ANN_ERR) * max_length) {
      double mn, mx;
      min_max(t, pidx, n, d, &mn, &mx);
      const double spr = mx - mn;
      if (spr > max_spread) { max_spread = spr; *cut_dim = d; }
    }
  }
  const double ideal = (blo[*cut_dim] + bhi[*cut_dim]) / 2;
  double mn, mx;
  min_max(t, pidx, n, *cut_dim, &mn, &mx);
  if (ideal < mn) *cut_val = mn;
  else if (ideal > mx) *cut_val = mx;
  else *cut_val = ideal;
  int br1, br2;
  plane_split(t, pidx, n, *cut_dim, *cut_val, &br1, &br2);
  if (ideal < mn) *n_lo = 1;
  else if (ideal > mx) *n_lo = n - 1;
  else if (br1 > n / 2) *n_lo = br1;
  else if (br2 < n / 2) *n_lo = br2;
  else *n_lo = n / 2;
}

/* kd_tree.cpp:319-361 (rkd_tree, bucket size 1); splitting nodes numbered in pre-order */
static int rkd_tree(orc_ann *t, int *pidx, int n, double *blo, double *bhi, int depth)
{
  if (depth > t->depth) t->depth = depth;
  if (n <= 1) return ~pidx[0];
  if (t->nsp == t->cap) {
    t->cap = t->cap ? 2 * t->cap : 1024;
    t->sp = (ann_split *)realloc(t->sp, (size_t)t->cap * sizeof(ann_split));
  }
  const int me = (int)t->nsp++;
  int cd = 0, n_lo = 0;
  double cv = 0;
  sl_midpt_split(t, pidx, blo, bhi, n, &cd, &cv, &n_lo);
  const double lv = blo[cd], hv = bhi[cd];
  bhi[cd] = cv;
  const int lo = rkd_tree(t, pidx, n_lo, blo, bhi, depth + 1);
  bhi[cd] = hv;
  blo[cd] = cv;
  const int hi = rkd_tree(t, pidx + n_lo, n - n_lo, blo, bhi, depth + 1);
  blo[cd] = lv;
  ann_split *s = &t->sp[me];
  s->cut_dim = cd; s->cut_val = cv; s->lo = lv; s->hi = hv;
  s->child[0] = lo; s->child[1] = hi;
  return me;
}

/* kd_tree.cpp:370-404 with annEnclRect (kd_util.cpp:75-92) */
orc_ann *orc_ann_create(const double *xyz, int n)
{
  if (n <= 0) return NULL;
  orc_ann *t = (orc_ann *)calloc(1, sizeof(orc_ann));
  t->n = n;
  t->pts = (double *)malloc((size_t)n * 3 * sizeof(double));
  memcpy(t->pts, xyz, (size_t)n * 3 * sizeof(double));
  t->pidx = (int *)malloc((size_t)n * sizeof(int));
  for (int i = 0; i < n; i++) t->pidx[i] = i;
  for (int d = 0; d < 3; d++) min_max(t, t->pidx, n, d, &t->bb_lo[d], &t->bb_hi[d]);
  double blo[3], bhi[3];
  memcpy(blo, t->bb_lo, sizeof blo);
  memcpy(bhi, t->bb_hi, sizeof bhi);
  t->root = rkd_tree(t, t->pidx, n, blo, bhi, 0);
  return t;
}

void orc_ann_destroy(orc_ann *t)
{
  if (!t) return;
  free(t->pts); free(t->pidx); free(t->sp); free(t);
}

/* out[0..2] = depth (ANNkdStats convention: a lone leaf has depth 0), leaves, splitting nodes */
void orc_ann_stats(const orc_ann *t, long *out)
{
  out[0] = t->depth; out[1] = t->n; out[2] = t->nsp;
}

/* pre-order listing as ANNkd_tree::Dump prints it (kd_dump.cpp:136-158) */
static void dump_rec(const orc_ann *t, int ref, int32_t *cut_dim, double *cut_val, int32_t *leaf_pt, long *ns, long *nl)
{
  if (ref < 0) { leaf_pt[(*nl)++] = ~ref; return; }
  const ann_split *s = &t->sp[ref];
  if (cut_val) cut_val[*ns] = s->cut_val;
  cut_dim[(*ns)++] = s->cut_dim;
  dump_rec(t, s->child[0], cut_dim, cut_val, leaf_pt, ns, nl);
  dump_rec(t, s->child[1], cut_dim, cut_val, leaf_pt, ns, nl);
}

long orc_ann_structure(const orc_ann *t, int32_t *cut_dim, double *cut_val, int32_t *leaf_pt, long *n_leaves)
{
  long ns = 0, nl = 0;
  dump_rec(t, t->root, cut_dim, cut_val, leaf_pt, &ns, &nl);
  *n_leaves = nl;
  return ns;
}

/* flat copy of the splitting nodes for the host-logic tests: per node cut_dim, child[2] and 3 doubles */
void orc_ann_nodes(const orc_ann *t, int32_t *cut_dim, int32_t *child, double *cv_lo_hi, int32_t *root)
{
  for (long i = 0; i < t->nsp; i++) {
    cut_dim[i] = t->sp[i].cut_dim;
    child[2 * i] = t->sp[i].child[0]; child[2 * i + 1] = t->sp[i].child[1];
    cv_lo_hi[3 * i] = t->sp[i].cut_val; cv_lo_hi[3 * i + 1] = t->sp[i].lo; cv_lo_hi[3 * i + 2] = t->sp[i].hi;
  }
  *root = t->root;
}

/* ---- search ---------------------------------------------------------------------------------- */
typedef struct {
  const orc_ann *t;
  const double *q;
  double max_err;
  int k, n_act;
  double *key; /* k + 1 entries (pr_queue_k.h:80-82) */
  int *info;
  long visited[2]; /* splitting nodes, leaves */
} ann_search;

#define PQ_NULL_KEY DBL_MAX /* ANN_DIST_INF, ANN.h:196 */

static double mk_max_key(const ann_search *S) { return S->n_act == S->k ? S->key[S->k - 1] : PQ_NULL_KEY; }

/* pr_queue_k.h:100-114 */
static void mk_insert(ann_search *S, double kv, int inf)
{
  int i;
  for (i = S->n_act; i > 0; i--) {
    if (S->key[i - 1] > kv) { S->key[i] = S->key[i - 1]; S->info[i] = S->info[i - 1]; }
    else break;
  }
  S->key[i] = kv;
  S->info[i] = inf;
  if (S->n_act < S->k) S->n_act++;
}

/* kd_search.cpp:128-170 (splitting node) and :177-210 (leaf of one point) */
static void search_rec(ann_search *S, int ref, double box_dist)
{
  if (ref < 0) {
    const int pi = ~ref;
    const double *pp = S->t->pts + 3 * (size_t)pi;
    double min_dist = mk_max_key(S), dist = 0;
    int d;
    for (d = 0; d < 3; d++) {
      const double tt = S->q[d] - pp[d];
      if ((dist = dist + tt * tt) > min_dist) break;
    }
    if (d >= 3) mk_insert(S, dist, pi); /* ANN_ALLOW_SELF_MATCH is true (ANN.h:232) */
    S->visited[1]++;
    return;
  }
  const ann_split *s = &S->t->sp[ref];
  S->visited[0]++;
  const double cut_diff = S->q[s->cut_dim] - s->cut_val;
  if (cut_diff < 0) {
    search_rec(S, s->child[0], box_dist);
    double box_diff = s->lo - S->q[s->cut_dim];
    if (box_diff < 0) box_diff = 0;
    box_dist = box_dist + (cut_diff * cut_diff - box_diff * box_diff);
    if (box_dist * S->max_err < mk_max_key(S)) search_rec(S, s->child[1], box_dist);
  } else {
    search_rec(S, s->child[1], box_dist);
    double box_diff = S->q[s->cut_dim] - s->hi;
    if (box_diff < 0) box_diff = 0;
    box_dist = box_dist + (cut_diff * cut_diff - box_diff * box_diff);
    if (box_dist * S->max_err < mk_max_key(S)) search_rec(S, s->child[0], box_dist);
  }
}

/* kd_util.cpp:127-150 (annBoxDistance) */
static double box_distance(const double *q, const double *lo, const double *hi)
{
  double dist = 0.0;
  for (int d = 0; d < 3; d++) {
    if (q[d] < lo[d]) { const double t = lo[d] - q[d]; dist = dist + t * t; }
    else if (q[d] > hi[d]) { const double t = q[d] - hi[d]; dist = dist + t * t; }
  }
  return dist;
}

/* kd_search.cpp:89-121; idx/dist [nq][k]; visits (nullable) += splitting nodes, leaves.  Returns -1 if k > n. */
int orc_ann_ksearch(const orc_ann *t, const double *q, int nq, int k, double eps, int32_t *idx, double *dist,
                    long *visits)
{
  if (k > t->n || k < 1) return -1;
  ann_search S;
  S.t = t; S.k = k;
  S.max_err = (1.0 + eps) * (1.0 + eps);
  S.key = (double *)malloc((size_t)(k + 1) * sizeof(double));
  S.info = (int *)malloc((size_t)(k + 1) * sizeof(int));
  S.visited[0] = S.visited[1] = 0;
  for (int i = 0; i < nq; i++) {
    S.q = q + 3 * (size_t)i;
    S.n_act = 0;
    search_rec(&S, t->root, box_distance(S.q, t->bb_lo, t->bb_hi));
    for (int j = 0; j < k; j++) {
      idx[(size_t)i * k + j] = j < S.n_act ? S.info[j] : -1;
      dist[(size_t)i * k + j] = j < S.n_act ? S.key[j] : PQ_NULL_KEY;
    }
  }
  if (visits) { visits[0] += S.visited[0]; visits[1] += S.visited[1]; }
  free(S.key); free(S.info);
  return 0;
}

/* ---- newmat EigenValues for a symmetric 3x3 ----------------------------------------------------- */
static double nm_sign(double x, double y) { return (y >= 0) ? x : -x; } /* newmatrm.h:106-107 */

/* evalue.cpp:24-96; z[n*n] row-major holds A on entry */
static void tred2(int n, double *z, double *D, double *E)
{
  const double tol = DBL_MIN / DBL_EPSILON;
  for (int i = n - 1; i > 0; i--) {
    double f = z[i * n + i - 1], g = 0.0;
    for (int k = 0; k < i - 1; k++) g += z[i * n + k] * z[i * n + k];
    double h = g + f * f;
    if (g <= tol) { E[i] = f; h = 0.0; }
    else {
      g = nm_sign(-sqrt(h), f); E[i] = g; h -= f * g;
      z[i * n + i - 1] = f - g; f = 0.0;
      for (int j = 0; j < i; j++) {
        z[j * n + i] = z[i * n + j] / h; g = 0.0;
        for (int k = 0; k < j; k++) g += z[j * n + k] * z[i * n + k];
        for (int k = j; k < i; k++) g += z[k * n + j] * z[i * n + k];
        E[j] = g / h; f += g * z[j * n + i];
      }
      const double hh = f / (h + h);
      for (int j = 0; j < i; j++) {
        f = z[i * n + j]; g = E[j] - hh * f; E[j] = g;
        for (int k = 0; k <= j; k++) z[j * n + k] -= (f * E[k] + g * z[i * n + k]);
      }
    }
    D[i] = h;
  }
  D[0] = 0.0; E[0] = 0.0;
  for (int i = 0; i < n; i++) {
    if (D[i] != 0.0) {
      for (int j = 0; j < i; j++) {
        double g = 0.0;
        for (int k = 0; k < i; k++) g += z[i * n + k] * z[k * n + j];
        for (int k = 0; k < i; k++) z[k * n + j] -= g * z[k * n + i];
      }
    }
    for (int j = 0; j < i; j++) { z[i * n + j] = 0.0; z[j * n + i] = 0.0; }
    D[i] = z[i * n + i]; z[i * n + i] = 1.0;
  }
}

/* evalue.cpp:98-156; returns -1 where the reference throws ConvergenceException */
static int tql2(int n, double *D, double *E, double *z)
{
  const double eps = DBL_EPSILON;
  for (int l = 1; l < n; l++) E[l - 1] = E[l];
  double b = 0.0, f = 0.0;
  E[n - 1] = 0.0;
  for (int l = 0; l < n; l++) {
    double h = eps * (fabs(D[l]) + fabs(E[l]));
    if (b < h) b = h;
    int m;
    for (m = l; m < n; m++) if (fabs(E[m]) <= b) break;
    int test = 0;
    for (int j = 0; j < 30; j++) {
      if (m == l) { test = 1; break; }
      double g = D[l], p = (D[l + 1] - g) / (2.0 * E[l]), r = sqrt(p * p + 1.0);
      D[l] = E[l] / (p < 0.0 ? p - r : p + r);
      const double hh = g - D[l];
      f += hh;
      for (int i = l + 1; i < n; i++) D[i] -= hh;
      p = D[m];
      double c = 1.0, s = 0.0;
      for (int i = m - 1; i >= l; i--) {
        const double ei = E[i], di = D[i];
        g = c * ei; h = c * p;
        if (fabs(p) >= fabs(ei)) {
          c = ei / p; r = sqrt(c * c + 1.0);
          E[i + 1] = s * p * r; s = c / r; c = 1.0 / r;
        } else {
          c = p / ei; r = sqrt(c * c + 1.0);
          E[i + 1] = s * ei * r; s = 1.0 / r; c /= r;
        }
        p = c * di - s * g; D[i + 1] = h + s * (c * g + s * di);
        for (int k = 0; k < n; k++) {
          h = z[k * n + i + 1];
          z[k * n + i + 1] = s * z[k * n + i] + c * h;
          z[k * n + i] = c * z[k * n + i] - s * h;
        }
      }
      E[l] = s * p; D[l] = c * p;
      if (fabs(E[l]) <= b) { test = 1; break; }
    }
    if (!test) return -1;
    D[l] += f;
  }
  return 0;
}

/* EigenValues(A, D, U), evalue.cpp:283-284: tred2, tql2, SortSV ascending (sort.cpp:190-222).
 * a[9] row-major, lower triangle used; d[3] ascending; u[9] row-major, eigenvectors in columns. */
int orc_eigen3(const double *a, double *d, double *u)
{
  double E[3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) u[3 * r + c] = (c <= r) ? a[3 * r + c] : a[3 * c + r];
  tred2(3, u, d, E);
  if (tql2(3, d, E, u)) return -1;
  for (int i = 0; i < 3; i++) {
    int k = i;
    double p = d[i];
    for (int j = i + 1; j < 3; j++) if (d[j] < p) { k = j; p = d[j]; }
    if (k != i) {
      d[k] = d[i]; d[i] = p;
      for (int j = 0; j < 3; j++) { const double tmp = u[3 * j + i]; u[3 * j + i] = u[3 * j + k]; u[3 * j + k] = tmp; }
    }
  }
  return 0;
}

/* normals.cc:64-105 for one point p with its neighbour list */
static void normal_from_neighbours(const double *xyz, const int32_t *nidx, int k, const double *p, const double *rPos,
                                   double *nrm)
{
  double mean[3] = {0.0, 0.0, 0.0};
  for (int j = 0; j < k; j++)
    for (int c = 0; c < 3; c++) mean[c] += xyz[3 * (size_t)nidx[j] + c];
  for (int c = 0; c < 3; c++) mean[c] /= k;
  /* A << 1.0 / k * X.t() * X : newmat evaluates (s * X^T) * X, summing over the neighbours in list order
   * (newmat7.cpp mmMult), and the SymmetricMatrix keeps element (c, r), c <= r, of that product */
  const double s = 1.0 / k;
  double A[9] = {0};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c <= r; c++) {
      double acc = 0.0;
      for (int j = 0; j < k; j++) {
        const double xr = xyz[3 * (size_t)nidx[j] + r] - mean[r], xc = xyz[3 * (size_t)nidx[j] + c] - mean[c];
        acc += (s * xc) * xr;
      }
      A[3 * r + c] = acc;
    }
  double D[3], U[9];
  orc_eigen3(A, D, U);
  double n[3] = {U[0], U[3], U[6]};
  double pv[3] = {p[0] - rPos[0], p[1] - rPos[1], p[2] - rPos[2]};
  /* "v / norm" is v * (1.0 / norm) in newmat (newmat6.cpp:477-478) */
  const double pl = 1.0 / sqrt(pv[0] * pv[0] + pv[1] * pv[1] + pv[2] * pv[2]);
  for (int c = 0; c < 3; c++) pv[c] = pv[c] * pl;
  const double angle = n[0] * pv[0] + n[1] * pv[1] + n[2] * pv[2];
  if (angle < 0) for (int c = 0; c < 3; c++) n[c] *= -1.0;
  const double nl = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
  for (int c = 0; c < 3; c++) nrm[c] = n[c] * nl;
}

/* calculateNormalsApxKNN, normals.cc:35-111.  knn_out (nullable) receives the [n][k] neighbour lists. */
int orc_normals_apx_knn(const double *xyz, int n, int k, const double *rPos, double eps, double *normals,
                        int32_t *knn_out)
{
  orc_ann *t = orc_ann_create(xyz, n);
  if (!t || k > n) { orc_ann_destroy(t); return -1; }
  int32_t *nidx = (int32_t *)malloc((size_t)k * sizeof(int32_t));
  double *dd = (double *)malloc((size_t)k * sizeof(double));
  for (int i = 0; i < n; i++) {
    orc_ann_ksearch(t, xyz + 3 * (size_t)i, 1, k, eps, nidx, dd, NULL);
    if (knn_out) memcpy(knn_out + (size_t)i * k, nidx, (size_t)k * sizeof(int32_t));
    normal_from_neighbours(xyz, nidx, k, xyz + 3 * (size_t)i, rPos, normals + 3 * (size_t)i);
  }
  free(nidx); free(dd);
  orc_ann_destroy(t);
  return 0;
}

/* the PCA alone on given neighbour lists (lets the tests separate list parity from eigenvector conditioning) */
void orc_normals_from_knn(const double *xyz, int n, int k, const int32_t *knn, const double *rPos, double *normals)
{
  for (int i = 0; i < n; i++)
    normal_from_neighbours(xyz, knn + (size_t)i * k, k, xyz + 3 * (size_t)i, rPos, normals + 3 * (size_t)i);
}
