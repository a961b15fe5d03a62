/*
 * oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, CPU restatement of the slam6D ICP correspondence hot path of
 * JMUWRobotics/3DTK, used as the parity checker for the HIP implementation in
 * 3dtk_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (lib3dtk_hip.so) never
 * links, loads or calls anything in this directory.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  All arithmetic is fp64 with the reference's
 * left-to-right association; build with -ffp-contract=off (the reference is
 * compiled -O3 without -march=native, i.e. no FMA contraction on x86-64).
 *
 * Parity status: PINNED for the tree, the searches, getPtPairs and the 4x4
 * helpers: tests/test_oracle_vs_ref.py checks this file against the reference's
 * own translation units (kdIndexed.cc and the minimizer TUs) built by
 * oracle/build_ref.sh into oracle/_ref/, against the known-answer tests of
 * testing/kdtree/, and against the committed fixtures under tests/golden/.
 * PARITY UNPINNED for orc_octree_center (Boctree.h cannot be compiled here; see
 * the comment at that function).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ */
/* 4x4 column-major helpers (include/slam6d/globals.icc)               */
/* ------------------------------------------------------------------ */

/* globals.icc:388-395 */
static double m3det(const double *M)
{
  return (M[0] * (M[4] * M[8] - M[7] * M[5])
        - M[1] * (M[3] * M[8] - M[6] * M[5])
        + M[2] * (M[3] * M[7] - M[6] * M[4]));
}

/* globals.icc:715-730 (M4_submat) */
static void m4_submat(const double *Min, double *Mout, int i, int j)
{
  for (int di = 0; di < 3; di++)
    for (int dj = 0; dj < 3; dj++) {
      int si = di + ((di >= i) ? 1 : 0);
      int sj = dj + ((dj >= j) ? 1 : 0);
      Mout[di * 3 + dj] = Min[si * 4 + sj];
    }
}

/* globals.icc:738-751 (M4det) */
static double m4det(const double *M)
{
  double det, result = 0, i = 1.0;
  double sub[9];
  for (int n = 0; n < 4; n++, i *= -1.0) {
    m4_submat(M, sub, 0, n);
    det = m3det(sub);
    result += M[n] * det * i;
  }
  return result;
}

/* globals.icc:762-785 (M4inv): singular -> identity, returns 0 */
int orc_m4inv(const double *Min, double *Mout)
{
  double mdet = m4det(Min);
  if (fabs(mdet) < 0.00000000000005) {
    for (int k = 0; k < 16; k++) Mout[k] = (k % 5 == 0) ? 1.0 : 0.0;
    return 0;
  }
  double tmp[9];
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      int sign = 1 - ((i + j) % 2) * 2;
      m4_submat(Min, tmp, i, j);
      Mout[i + j * 4] = (m3det(tmp) * sign) / mdet;
    }
  return 1;
}

/* globals.icc:298-328 (MMult) */
void orc_mmult(const double *M1, const double *M2, double *Mout)
{
  double r[16];
  for (int c = 0; c < 4; c++)
    for (int rr = 0; rr < 4; rr++)
      r[c * 4 + rr] = M1[rr] * M2[c * 4] + M1[4 + rr] * M2[c * 4 + 1]
                    + M1[8 + rr] * M2[c * 4 + 2] + M1[12 + rr] * M2[c * 4 + 3];
  memcpy(Mout, r, sizeof r);
}

/* globals.icc:1477-1490 (transform3, out-of-place) */
static inline void xf3(const double *a, const double *p, double *o)
{
  o[0] = p[0] * a[0] + p[1] * a[4] + p[2] * a[8] + a[12];
  o[1] = p[0] * a[1] + p[1] * a[5] + p[2] * a[9] + a[13];
  o[2] = p[0] * a[2] + p[1] * a[6] + p[2] * a[10] + a[14];
}

/* globals.icc:1454-1463 (transform3, in place: (x*a0+y*a4+z*a8) then +a12) */
static inline void xf3_inplace(const double *a, double *p)
{
  double x = p[0] * a[0] + p[1] * a[4] + p[2] * a[8];
  double y = p[0] * a[1] + p[1] * a[5] + p[2] * a[9];
  double z = p[0] * a[2] + p[1] * a[6] + p[2] * a[10];
  p[0] = x + a[12];
  p[1] = y + a[13];
  p[2] = z + a[14];
}

/* globals.icc:1465-1475 (transform3normal) */
static inline void xf3normal(const double *a, double *n)
{
  double x = n[0] * a[0] + n[1] * a[1] + n[2] * a[2];
  double y = n[0] * a[4] + n[1] * a[5] + n[2] * a[6];
  double z = n[0] * a[8] + n[1] * a[9] + n[2] * a[10];
  n[0] = x; n[1] = y; n[2] = z;
}

/* scan.cc:851-875 (Scan::transformReduced): serial in-place pass */
void orc_transform_points(const double *alignxf, double *xyz, size_t n)
{
  for (size_t i = 0; i < n; i++) xf3_inplace(alignxf, xyz + 3 * i);
}

/* scan.cc:866-870: normals */
void orc_transform_normals(const double *alignxf, double *nrm, size_t n)
{
  for (size_t i = 0; i < n; i++) xf3normal(alignxf, nrm + 3 * i);
}

/* globals.icc:237-245 (Dist2) */
static inline double dist2(const double *x1, const double *x2)
{
  double dx = x2[0] - x1[0];
  double dy = x2[1] - x1[1];
  double dz = x2[2] - x1[2];
  return dx * dx + dy * dy + dz * dz;
}

/* ------------------------------------------------------------------ */
/* kd-tree: pointer tree exactly as KDTreeImpl (kdTreeImpl.h)          */
/* ------------------------------------------------------------------ */

typedef struct orc_node {
  int npts;                 /* kdTreeImpl.h:221-224 */
  int isleaf;
  double center[3], dx, dy, dz, r;
  int splitaxis;
  double splitval;
  struct orc_node *child1, *child2;
  int *p;                   /* leaf: copy of the index run */
} orc_node;

typedef struct orc_tree {
  const double *xyz;        /* borrowed [M][3] */
  size_t M;
  int *indices;             /* permuted in place by create */
  orc_node *root;
  long n_internal, n_leaves, max_depth;
} orc_tree;

typedef struct {
  const double *p;          /* query (tree frame) */
  const double *dir;
  int closest;              /* -1 = none */
  double closest_d2;
  long n_int, n_leaf, n_pts;
} orc_params;               /* kdparams.h:22-98, the fields the 1-NN uses */

/* kdTreeImpl.h:82-201 (KDTreeImpl::create) */
static orc_node *create(orc_tree *t, int *indices, size_t n, unsigned bucket, long depth)
{
  const double *pts = t->xyz;
  orc_node *nd = (orc_node *)calloc(1, sizeof *nd);
  if (depth > t->max_depth) t->max_depth = depth;

  double mins[3], maxs[3], centroid[3];
  for (int i = 0; i < 3; i++) {
    mins[i] = maxs[i] = centroid[i] = pts[3 * (size_t)indices[0] + i];
  }
  for (size_t i = 1; i < n; i++)
    for (int j = 0; j < 3; j++) {
      double v = pts[3 * (size_t)indices[i] + j];
      mins[j] = (v < mins[j]) ? v : mins[j];   /* std::min(mins, v) */
      maxs[j] = (maxs[j] < v) ? v : maxs[j];   /* std::max(maxs, v) */
      centroid[j] += v;
    }
  for (int i = 0; i < 3; i++) centroid[i] /= n;

  if (n > 0 && n <= bucket) {                  /* :114-123 */
    nd->npts = (int)n; nd->isleaf = 1;
    nd->p = (int *)malloc(n * sizeof(int));
    memcpy(nd->p, indices, n * sizeof(int));
    t->n_leaves++;
    return nd;
  }
  nd->npts = 0; nd->isleaf = 0;
  for (int i = 0; i < 3; i++) nd->center[i] = 0.5 * (mins[i] + maxs[i]);
  nd->dx = 0.5 * (maxs[0] - mins[0]);
  nd->dy = 0.5 * (maxs[1] - mins[1]);
  nd->dz = 0.5 * (maxs[2] - mins[2]);
  nd->r = sqrt(nd->dx * nd->dx + nd->dy * nd->dy + nd->dz * nd->dz);

  if (nd->dx > nd->dy) {                       /* :138-150 */
    nd->splitaxis = (nd->dx > nd->dz) ? 0 : 2;
  } else {
    nd->splitaxis = (nd->dy > nd->dz) ? 1 : 2;
  }
  double mx = nd->dx > nd->dy ? nd->dx : nd->dy;   /* std::max(std::max(dx,dy),dz) */
  mx = mx > nd->dz ? mx : nd->dz;
  if (fabs(mx) < 0.01) {                       /* :153-162 */
    nd->npts = (int)n; nd->isleaf = 1;
    nd->p = (int *)malloc(n * sizeof(int));
    memcpy(nd->p, indices, n * sizeof(int));
    t->n_leaves++;
    return nd;
  }
  nd->splitval = centroid[nd->splitaxis];      /* :170 */
  int ax = nd->splitaxis;
  int *left = indices, *right = indices + n - 1;
  while (1) {                                  /* :172-182 */
    while (pts[3 * (size_t)*left + ax] < nd->splitval) left++;
    while (pts[3 * (size_t)*right + ax] >= nd->splitval) right--;
    if (right < left) break;
    int tmp = *left; *left = *right; *right = tmp;
  }
  t->n_internal++;
  nd->child1 = create(t, indices, (size_t)(left - indices), bucket, depth + 1);
  nd->child2 = create(t, left, n - (size_t)(left - indices), bucket, depth + 1);
  return nd;
}

static void destroy(orc_node *nd)
{
  if (!nd) return;
  if (nd->isleaf) { free(nd->p); }
  else { destroy(nd->child1); destroy(nd->child2); }
  free(nd);
}

/* kd.cc:46-49 / kdIndexed.cc ctor; xyz is borrowed and must outlive the tree */
orc_tree *orc_tree_create(const double *xyz, size_t M, int bucket)
{
  if (M == 0) return NULL;                     /* kdTreeImpl.h:86-88 throws */
  orc_tree *t = (orc_tree *)calloc(1, sizeof *t);
  t->xyz = xyz; t->M = M;
  t->indices = (int *)malloc(M * sizeof(int));
  for (size_t i = 0; i < M; i++) t->indices[i] = (int)i;
  t->root = create(t, t->indices, M, (unsigned)bucket, 1);
  return t;
}

void orc_tree_destroy(orc_tree *t)
{
  if (!t) return;
  destroy(t->root); free(t->indices); free(t);
}

void orc_tree_stats(const orc_tree *t, long *out3)
{
  out3[0] = t->n_internal; out3[1] = t->n_leaves; out3[2] = t->max_depth;
}

/* post-build permutation == concatenation of the leaves left to right */
void orc_tree_perm(const orc_tree *t, int *out) { memcpy(out, t->indices, t->M * sizeof(int)); }

/* kdTreeImpl.h:345-383 (_FindClosest) */
static void find_closest(const orc_tree *t, const orc_node *nd, orc_params *pa)
{
  if (nd->isleaf) {
    pa->n_leaf++;
    for (int i = 0; i < nd->npts; i++) {
      double d2 = dist2(pa->p, t->xyz + 3 * (size_t)nd->p[i]);
      pa->n_pts++;
      if (d2 < pa->closest_d2) { pa->closest_d2 = d2; pa->closest = nd->p[i]; }
    }
    return;
  }
  pa->n_int++;
  double a = fabs(pa->p[0] - nd->center[0]) - nd->dx;
  double b = fabs(pa->p[1] - nd->center[1]) - nd->dy;
  double c = fabs(pa->p[2] - nd->center[2]) - nd->dz;
  double ab = (a < b) ? b : a;                  /* std::max */
  double approx = (ab < c) ? c : ab;
  if (approx >= 0 && approx * approx >= pa->closest_d2) return;

  double myd = nd->splitval - pa->p[nd->splitaxis];
  if (myd >= 0.0) {
    find_closest(t, nd->child1, pa);
    if (myd * myd < pa->closest_d2) find_closest(t, nd->child2, pa);
  } else {
    find_closest(t, nd->child2, pa);
    if (myd * myd < pa->closest_d2) find_closest(t, nd->child1, pa);
  }
}

/* kdTreeImpl.h:390-425 (_FindClosestAlongDir) */
static void find_closest_dir(const orc_tree *t, const orc_node *nd, orc_params *pa)
{
  if (nd->isleaf) {
    pa->n_leaf++;
    for (int i = 0; i < nd->npts; i++) {
      const double *x = t->xyz + 3 * (size_t)nd->p[i];
      double v[3] = { pa->p[0] - x[0], pa->p[1] - x[1], pa->p[2] - x[2] };
      double len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
      double dot = v[0] * pa->dir[0] + v[1] * pa->dir[1] + v[2] * pa->dir[2];
      double d2 = len2 - dot * dot;
      pa->n_pts++;
      if (d2 < pa->closest_d2) { pa->closest_d2 = d2; pa->closest = nd->p[i]; }
    }
    return;
  }
  pa->n_int++;
  double v[3] = { pa->p[0] - nd->center[0], pa->p[1] - nd->center[1], pa->p[2] - nd->center[2] };
  double len2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
  double dot = v[0] * pa->dir[0] + v[1] * pa->dir[1] + v[2] * pa->dir[2];
  double d2c = len2 - dot * dot;
  double lim = nd->r + sqrt(pa->closest_d2);
  if (d2c > lim * lim) return;
  if (pa->p[nd->splitaxis] < nd->splitval) {
    find_closest_dir(t, nd->child1, pa);
    find_closest_dir(t, nd->child2, pa);
  } else {
    find_closest_dir(t, nd->child2, pa);
    find_closest_dir(t, nd->child1, pa);
  }
}

/* kd.cc:78-87 (KDtree::FindClosest), batched.  idx = -1 when none.
 * counters (nullable) accumulates {internal nodes, leaves, leaf points}.   */
void orc_find_closest(const orc_tree *t, const double *q, size_t K, double maxdist2,
                      int32_t *idx, double *d2, long *counters, int nthreads)
{
  long c0 = 0, c1 = 0, c2 = 0;
  (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) reduction(+:c0,c1,c2)
#endif
  for (long i = 0; i < (long)K; i++) {
    orc_params pa; memset(&pa, 0, sizeof pa);
    pa.p = q + 3 * i; pa.closest = -1; pa.closest_d2 = maxdist2;
    find_closest(t, t->root, &pa);
    idx[i] = pa.closest;
    if (d2) d2[i] = pa.closest_d2;
    c0 += pa.n_int; c1 += pa.n_leaf; c2 += pa.n_pts;
  }
  if (counters) { counters[0] += c0; counters[1] += c1; counters[2] += c2; }
}

/* ---- the GPU kernel's "quick check deferred", stated on the reference's tree (test infrastructure for the ARGUMENT) ----------
 * 3dtk_amd/csrc/kernels.hip (search_refill_body<.., DEFER>) lets a query that starts from a previous hit walk WITHOUT the quick
 * check of kdTreeImpl.h:360-368 and searches it again, with every check, if it ever accepted a point that improved closest_d2
 * by `tie` or less (api.cpp: search_tie).  This is that rule on the pointer tree above, check skipped at EVERY node (the kernel
 * skips it at some): tests/test_oracle_vs_ref.py runs it over clouds built to sit on the roundings the argument is about --
 * lattices with axis-aligned queries, twins 1e-12 apart, coordinates far from the origin -- and asks for orc_find_closest's
 * answer, index and d2, every time.  Nothing in the product calls it. */
static void find_closest_nocheck(const orc_tree *t, const orc_node *nd, orc_params *pa, double tie, int *thin)
{
  if (nd->isleaf) {
    for (int i = 0; i < nd->npts; i++) {
      double d2 = dist2(pa->p, t->xyz + 3 * (size_t)nd->p[i]);
      if (d2 < pa->closest_d2) {
        if (pa->closest_d2 - d2 <= tie) *thin = 1;
        pa->closest_d2 = d2; pa->closest = nd->p[i];
      }
    }
    return;
  }
  double myd = nd->splitval - pa->p[nd->splitaxis];
  if (myd >= 0.0) {
    find_closest_nocheck(t, nd->child1, pa, tie, thin);
    if (myd * myd < pa->closest_d2) find_closest_nocheck(t, nd->child2, pa, tie, thin);
  } else {
    find_closest_nocheck(t, nd->child2, pa, tie, thin);
    if (myd * myd < pa->closest_d2) find_closest_nocheck(t, nd->child1, pa, tie, thin);
  }
}

double orc_search_tie(double absmax, double maxdist2)
{
  const double R = sqrt(maxdist2), E = ldexp(3.0 * absmax + R, -50);
  return 4.0 * (2.0 * E * R + E * E);
}

/* warm[i]: index of a point of the tree (the "previous hit") or -1.  redo (nullable) counts the second searches. */
void orc_find_closest_deferred(const orc_tree *t, const double *q, size_t K, double maxdist2, const int32_t *warm,
                               double absmax, int32_t *idx, double *d2, long *redo)
{
  const double tie = orc_search_tie(absmax, maxdist2);
  long nredo = 0;
  for (size_t i = 0; i < K; i++) {
    orc_params pa; memset(&pa, 0, sizeof pa);
    pa.p = q + 3 * i; pa.closest = -1; pa.closest_d2 = maxdist2;
    int deferred = 0, thin = 0;
    if (warm && warm[i] >= 0) {
      const double d = dist2(pa.p, t->xyz + 3 * (size_t)warm[i]);
      double up = nextafter(d, INFINITY);
      const double um = d + 2.0 * tie;
      if (um > up) up = um;
      if (up < maxdist2) { pa.closest_d2 = up; deferred = 1; }
    }
    if (deferred) {
      find_closest_nocheck(t, t->root, &pa, tie, &thin);
      if (thin) {                      /* again, as the reference does it */
        nredo++;
        pa.closest = -1; pa.closest_d2 = maxdist2;
        find_closest(t, t->root, &pa);
      }
    } else {
      find_closest(t, t->root, &pa);
    }
    idx[i] = pa.closest;
    if (d2) d2[i] = pa.closest_d2;
  }
  if (redo) *redo += nredo;
}

/* kd.cc:89-100 (KDtree::FindClosestAlongDir), batched, one dir per query */
void orc_find_closest_along_dir(const orc_tree *t, const double *q, const double *dir, size_t K,
                                double maxdist2, int32_t *idx, double *d2)
{
  for (size_t i = 0; i < K; i++) {
    orc_params pa; memset(&pa, 0, sizeof pa);
    pa.p = q + 3 * i; pa.dir = dir + 3 * i; pa.closest = -1; pa.closest_d2 = maxdist2;
    find_closest_dir(t, t->root, &pa);
    idx[i] = pa.closest;
    if (d2) d2[i] = pa.closest_d2;
  }
}

/* ------------------------------------------------------------------ */
/* SearchTree::getPtPairs, DataXYZ overload (searchTree.cc:92-189)     */
/* ------------------------------------------------------------------ */
/*
 * pairing_mode: 0 CLOSEST_POINT, 1 CLOSEST_POINT_ALONG_NORMAL_SIMPLE,
 *               2 CLOSEST_PLANE_SIMPLE        (include/slam6d/pairingMode.h:4-8)
 * rnd > 1 is not restated (std::rand() from threads is unreproducible,
 * SURVEY N-d); callers pass rnd <= 1.
 * Outputs (all caller-allocated, sized end-start): idx (model index per query
 * or -1), and compact pair lists p1 (model point in world frame), p2 (data
 * point), pn (normalised data normal) in query order.  sum / centroid_* are
 * accumulated into, not zeroed (searchTree.cc:165-177).  Returns pair count.
 */
size_t orc_get_pt_pairs(const orc_tree *t, const double *source_alignxf,
                        const double *xyz_r, const double *normal_r,
                        size_t start, size_t end, int pairing_mode, double maxdist2,
                        int32_t *idx, double *p1, double *p2, double *pn,
                        double *sum, double *centroid_m, double *centroid_d)
{
  double inv[16];
  orc_m4inv(source_alignxf, inv);              /* :110 */
  size_t np = 0;
  for (size_t i = start; i < end; i++) {
    double tt[3] = { xyz_r[3 * i], xyz_r[3 * i + 1], xyz_r[3 * i + 2] };
    double s[3], normal[3] = { 0, 0, 0 };
    xf3(inv, tt, s);                           /* :122 */
    if (pairing_mode != 0) {                   /* :126-131 */
      normal[0] = normal_r[3 * i]; normal[1] = normal_r[3 * i + 1]; normal[2] = normal_r[3 * i + 2];
      double len = sqrt(normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2]);
      normal[0] /= len; normal[1] /= len; normal[2] /= len;   /* Normalize3 globals.icc:262-270 */
    }
    orc_params pa; memset(&pa, 0, sizeof pa);
    pa.p = s; pa.closest = -1; pa.closest_d2 = maxdist2;
    if (pairing_mode == 1) {                   /* :133-141 */
      double nd[3] = { normal[0], normal[1], normal[2] };
      xf3normal(inv, nd);
      /* the reference rotates `normal` itself in place, so the PtPair keeps the rotated one */
      normal[0] = nd[0]; normal[1] = nd[1]; normal[2] = nd[2];
      pa.dir = normal;
      find_closest_dir(t, t->root, &pa);
    } else {
      find_closest(t, t->root, &pa);           /* :143 */
    }
    if (idx) idx[i - start] = pa.closest;
    if (pa.closest < 0) continue;
    xf3(source_alignxf, t->xyz + 3 * (size_t)pa.closest, s);   /* :147 */
    if (pairing_mode == 2) {                   /* :149-162 */
      double tmp[3] = { s[0] - tt[0], s[1] - tt[1], s[2] - tt[2] };
      double dot = normal[0] * tmp[0] + normal[1] * tmp[1] + normal[2] * tmp[2];
      s[0] = normal[0] * dot + tt[0];
      s[1] = normal[1] * dot + tt[1];
      s[2] = normal[2] * dot + tt[2];
    }
    centroid_m[0] += s[0]; centroid_m[1] += s[1]; centroid_m[2] += s[2];
    centroid_d[0] += tt[0]; centroid_d[1] += tt[1]; centroid_d[2] += tt[2];
    double d0 = s[0] - tt[0], d1 = s[1] - tt[1], d2 = s[2] - tt[2];
    *sum += d0 * d0 + d1 * d1 + d2 * d2;        /* Len2, :172-177 */
    if (p1) { p1[3 * np] = s[0]; p1[3 * np + 1] = s[1]; p1[3 * np + 2] = s[2]; }
    if (p2) { p2[3 * np] = tt[0]; p2[3 * np + 1] = tt[1]; p2[3 * np + 2] = tt[2]; }
    if (pn) { pn[3 * np] = normal[0]; pn[3 * np + 1] = normal[1]; pn[3 * np + 2] = normal[2]; }
    np++;
  }
  return np;
}

/* ------------------------------------------------------------------ */
/* timing helper for bench.py's cpu_baseline leg ("port" kind)          */
/* ------------------------------------------------------------------ */
int orc_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* input recipes for the golden vectors             */
/* ------------------------------------------------------------------ */
/* The 64-bit Mersenne Twister (Matsumoto & Nishimura, public algorithm ==
 * std::mt19937_64) followed by std::uniform_real_distribution<double>(lo,hi)
 * exactly as libstdc++ evaluates it for a 64-bit engine: one draw u per value,
 * c = (double)u / 2^64 (clamped below 1), value = c*(hi-lo) + lo.            */
void orc_gen_mt64_uniform(uint64_t seed, size_t n, double lo, double hi, double *out)
{
  enum { NN = 312, MM = 156 };
  static const uint64_t MAT = 0xB5026F5AA96619E9ULL, UM = 0xFFFFFFFF80000000ULL, LM = 0x7FFFFFFFULL;
  uint64_t mt[NN];
  int mti;
  mt[0] = seed;
  for (mti = 1; mti < NN; mti++)
    mt[mti] = 6364136223846793005ULL * (mt[mti - 1] ^ (mt[mti - 1] >> 62)) + (uint64_t)mti;
  for (size_t k = 0; k < n; k++) {
    if (mti >= NN) {
      int i;
      for (i = 0; i < NN - MM; i++) {
        uint64_t x = (mt[i] & UM) | (mt[i + 1] & LM);
        mt[i] = mt[i + MM] ^ (x >> 1) ^ ((x & 1ULL) ? MAT : 0ULL);
      }
      for (; i < NN - 1; i++) {
        uint64_t x = (mt[i] & UM) | (mt[i + 1] & LM);
        mt[i] = mt[i + (MM - NN)] ^ (x >> 1) ^ ((x & 1ULL) ? MAT : 0ULL);
      }
      uint64_t x = (mt[NN - 1] & UM) | (mt[0] & LM);
      mt[NN - 1] = mt[MM - 1] ^ (x >> 1) ^ ((x & 1ULL) ? MAT : 0ULL);
      mti = 0;
    }
    uint64_t x = mt[mti++];
    x ^= (x >> 29) & 0x5555555555555555ULL;
    x ^= (x << 17) & 0x71D67FFFEDA60000ULL;
    x ^= (x << 37) & 0xFFF7EEE000000000ULL;
    x ^= (x >> 43);
    double c = (double)x / 18446744073709551616.0;
    if (c >= 1.0) c = nextafter(1.0, 0.0);
    out[k] = c * (hi - lo) + lo;
  }
}

/* XOR hash of SURVEY 8(c) K5 over the FOUND correspondences: h ^= idx*1315423911 + i
 * (queries without a partner are skipped; reproduces 0x5cdee3d50429c / 0x34a9be6b29435) */
uint64_t orc_k5_hash(const int32_t *idx, size_t n)
{
  uint64_t h = 0;
  for (size_t i = 0; i < n; i++) {
    if (idx[i] < 0) continue;
    h ^= (uint64_t)idx[i] * 1315423911ULL + (uint64_t)i;
  }
  return h;
}

/* ---- octree reduction, centre mode ("-r <voxelSize>") -------------------------------------
 * PARITY UNPINNED: include/slam6d/Boctree.h needs boost/interprocess/offset_ptr.hpp, which this
 * image does not have, so the reference octree cannot be compiled into oracle/_ref and the
 * reference holds no golden vector for it.  This is a restatement by reading:
 *   BOctTree(P* const* pts, int n, T voxelSize)   Boctree.h:222-270  (root cube: bbox centre,
 *       half size = largest half extent + 1.0; the root is always split)
 *   countPointsAndQueueFast / branch              Boctree.h:1163-1195, 1268-1300 (occupied octants in
 *       index order; a child is a leaf once its half size <= voxelSize)
 *   fullsort / sort                               Boctree.h:1737-1816: Scan::calcReducedPoints hands the constructor a
 *       double**, i.e. the ARRAY overload, whose octants are cut with `p < centre` (left) / `p >= centre` (right); a point
 *       exactly on a centre plane therefore goes UP (childIndex, :1353-1355, strict >, is what the vector overload and
 *       the searches use -- round 3 corrected this restatement, which had followed childIndex)
 *   childcenter                                   Boctree.h:612-657  (centre -/+ size/2.0)
 *   GetOctTreeCenter                              Boctree.h:928-948  (DFS, child index order, emits the
 *       leaf cell's centre)
 * as used by Scan::calcReducedPoints, src/slam6d/scan.cc:577-603 (reduction_nrpts == 0).
 * Recursive with explicit index lists, i.e. structured like the reference, unlike the sort-based
 * device path it checks. */
typedef struct {
  const double *xyz;
  double voxel;
  double *out;
  size_t n_out;
} oct_ctx;

static void oct_childcenter(const double *pc, double *cc, double size, int i)
{
  cc[0] = (i & 1) ? pc[0] + size / 2.0 : pc[0] - size / 2.0;
  cc[1] = (i & 2) ? pc[1] + size / 2.0 : pc[1] - size / 2.0;
  cc[2] = (i & 4) ? pc[2] + size / 2.0 : pc[2] - size / 2.0;
}

/* split the points of one cell (centre c, half size `size`) over its octants */
static void oct_split(oct_ctx *C, uint32_t *idx, size_t n, const double *c, double size)
{
  size_t cnt[8] = {0}, off[9];
  unsigned char *ci = (unsigned char *)malloc(n ? n : 1);
  for (size_t k = 0; k < n; k++) {
    const double *p = C->xyz + 3 * (size_t)idx[k];
    ci[k] = (unsigned char)((!(p[0] < c[0])) | ((!(p[1] < c[1])) << 1) | ((!(p[2] < c[2])) << 2));   /* sort(): `< splitval` stays left */
    cnt[ci[k]]++;
  }
  off[0] = 0;
  for (int j = 0; j < 8; j++) off[j + 1] = off[j] + cnt[j];
  uint32_t *tmp = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  size_t pos[8];
  for (int j = 0; j < 8; j++) pos[j] = off[j];
  for (size_t k = 0; k < n; k++) tmp[pos[ci[k]]++] = idx[k];
  memcpy(idx, tmp, n * sizeof(uint32_t));
  free(tmp);
  free(ci);
  const double size_new = size / 2.0;
  for (int j = 0; j < 8; j++) {
    if (!cnt[j]) continue;
    double cc[3];
    oct_childcenter(c, cc, size, j);
    if (size_new <= C->voxel) {                 /* branch(): leaf */
      double *o = C->out + 3 * C->n_out++;
      o[0] = cc[0]; o[1] = cc[1]; o[2] = cc[2];
    } else {
      oct_split(C, idx + off[j], cnt[j], cc, size_new);
    }
  }
}

/* out has room for n points; returns the number of leaf cells */
size_t orc_octree_center(const double *xyz, size_t n, double voxel, double *out)
{
  if (n == 0) return 0;
  double mins[3], maxs[3], center[3];
  for (int a = 0; a < 3; a++) {
    mins[a] = maxs[a] = xyz[a];
    for (size_t j = 1; j < n; j++) {
      const double v = xyz[3 * j + a];
      mins[a] = v < mins[a] ? v : mins[a];
      maxs[a] = maxs[a] < v ? v : maxs[a];
    }
    center[a] = 0.5 * (mins[a] + maxs[a]);
  }
  double size = 0.5 * (maxs[0] - mins[0]);
  if (size < 0.5 * (maxs[1] - mins[1])) size = 0.5 * (maxs[1] - mins[1]);
  if (size < 0.5 * (maxs[2] - mins[2])) size = 0.5 * (maxs[2] - mins[2]);
  size += 1.0;
  uint32_t *idx = (uint32_t *)malloc(n * sizeof(uint32_t));
  for (size_t k = 0; k < n; k++) idx[k] = (uint32_t)k;
  oct_ctx C = {xyz, voxel, out, 0};
  oct_split(&C, idx, n, center, size);
  free(idx);
  return C.n_out;
}

/* ---- octree reduction, random modes ("-r <voxelSize> -O <nrpts>", nrpts >= 1) ------------------------------------
 * PARITY UNPINNED (Boctree.h cannot be compiled here, see above).  Restatement by reading, structured like the
 * reference: an array of point pointers (here: indices) partitioned IN PLACE,
 *   fullsort                Boctree.h:1737-1780: z, then y inside each z half, then x inside each quarter
 *   sort                    Boctree.h:1784-1816: the two-pointer partition as written (`< splitval` | `>= splitval`)
 *   countPointsAndQueueFast Boctree.h:1268-1300, branch :1163-1195: a leaf copies its points in the order the
 *                           partitions left them
 *   GetOctTreeRandom        Boctree.h:985-1018 (nrpts == 1): per leaf in child-index order one point,
 *                           index rand(length) = (int)(length * std::rand() / (RAND_MAX + 1.0))  (globals.icc:607-610)
 *   GetOctTreeRandom        Boctree.h:1020-1062 (nrpts > 1, rm_scatter == false): all points of a leaf with at most
 *                           nrpts points; otherwise a std::set of nrpts distinct rand(length - 1) draws, in ascending order
 * as used by Scan::calcReducedPoints, src/slam6d/scan.cc:586-596.  rand() is the C library's: callers seed it. */
typedef struct {
  const double *xyz;
  double voxel;
  int nrpts;
  double *out;
  size_t n_out;
} octr_ctx;

static uint32_t *octr_sort(const double *xyz, uint32_t *points, uint32_t n, double splitval, int index)
{
  if (n == 0) return points;
  if (n == 1) return (xyz[3 * (size_t)points[0] + index] < splitval) ? points + 1 : points;
  uint32_t *left = points, *right = points + n - 1;
  for (;;) {
    while (xyz[3 * (size_t)*left + index] < splitval) { left++; if (right < left) break; }
    while (xyz[3 * (size_t)*right + index] >= splitval) { right--; if (right < left) break; }
    if (right < left) break;
    const uint32_t t = *left; *left = *right; *right = t;
  }
  return left;
}

static int octr_rand(int rnd) { return (int)((double)rnd * (double)rand() / (RAND_MAX + 1.0)); }

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

static void octr_leaf(octr_ctx *C, const uint32_t *pts, uint32_t n)
{
  if (C->nrpts == 1) {
    const uint32_t k = pts[octr_rand((int)n)];
    memcpy(C->out + 3 * C->n_out++, C->xyz + 3 * (size_t)k, 3 * sizeof(double));
    return;
  }
  if ((uint32_t)C->nrpts >= n) {
    for (uint32_t j = 0; j < n; j++) memcpy(C->out + 3 * C->n_out++, C->xyz + 3 * (size_t)pts[j], 3 * sizeof(double));
    return;
  }
  int *idx = (int *)malloc(sizeof(int) * (size_t)C->nrpts);
  int have = 0;
  while (have < C->nrpts) {                       /* std::set<int>::insert */
    const int t = octr_rand((int)n - 1);
    int dup = 0;
    for (int k = 0; k < have; k++) dup |= idx[k] == t;
    if (!dup) idx[have++] = t;
  }
  qsort(idx, (size_t)have, sizeof(int), cmp_int);  /* iteration order of the set */
  for (int k = 0; k < have; k++) memcpy(C->out + 3 * C->n_out++, C->xyz + 3 * (size_t)pts[idx[k]], 3 * sizeof(double));
  free(idx);
}

static void octr_split(octr_ctx *C, uint32_t *points, uint32_t n, const double *c, double size)
{
  uint32_t *blocks[9];
  blocks[0] = points; blocks[8] = points + n;
  {   /* fullsort */
    uint32_t *L0 = octr_sort(C->xyz, points, n, c[2], 2);
    const uint32_t n0L = (uint32_t)(L0 - points);
    uint32_t *L1 = octr_sort(C->xyz, points, n0L, c[1], 1);
    uint32_t n1L = (uint32_t)(L1 - points);
    uint32_t *L2 = octr_sort(C->xyz, points, n1L, c[0], 0);
    blocks[1] = L2;
    uint32_t n1R = n0L - n1L;
    L2 = octr_sort(C->xyz, L1, n1R, c[0], 0);
    blocks[2] = L1; blocks[3] = L2;
    const uint32_t n0R = n - n0L;
    L1 = octr_sort(C->xyz, L0, n0R, c[1], 1);
    n1L = (uint32_t)(L1 - L0);
    L2 = octr_sort(C->xyz, L0, n1L, c[0], 0);
    blocks[4] = L0; blocks[5] = L2;
    n1R = n0R - n1L;
    L2 = octr_sort(C->xyz, L1, n1R, c[0], 0);
    blocks[6] = L1; blocks[7] = L2;
  }
  const double size_new = size / 2.0;
  for (int j = 0; j < 8; j++) {
    const uint32_t cnt = (uint32_t)(blocks[j + 1] - blocks[j]);
    if (!cnt) continue;
    double cc[3];
    oct_childcenter(c, cc, size, j);
    if (size_new <= C->voxel) octr_leaf(C, blocks[j], cnt);
    else octr_split(C, blocks[j], cnt, cc, size_new);
  }
}

/* out has room for n points; returns the number of points kept.  perm_out (nullable, [n]) receives the order the
 * partitions leave the points in (leaf by leaf in child-index order): what the device path must reproduce. */
size_t orc_octree_random(const double *xyz, size_t n, double voxel, int nrpts, double *out, uint32_t *perm_out)
{
  if (n == 0 || nrpts < 1) return 0;
  double mins[3], maxs[3], center[3];
  for (int a = 0; a < 3; a++) {
    mins[a] = maxs[a] = xyz[a];
    for (size_t j = 1; j < n; j++) {
      const double v = xyz[3 * j + a];
      mins[a] = v < mins[a] ? v : mins[a];
      maxs[a] = maxs[a] < v ? v : maxs[a];
    }
    center[a] = 0.5 * (mins[a] + maxs[a]);
  }
  double size = 0.5 * (maxs[0] - mins[0]);
  if (size < 0.5 * (maxs[1] - mins[1])) size = 0.5 * (maxs[1] - mins[1]);
  if (size < 0.5 * (maxs[2] - mins[2])) size = 0.5 * (maxs[2] - mins[2]);
  size += 1.0;
  uint32_t *idx = (uint32_t *)malloc(n * sizeof(uint32_t));
  for (size_t k = 0; k < n; k++) idx[k] = (uint32_t)k;
  octr_ctx C = {xyz, voxel, nrpts, out, 0};
  octr_split(&C, idx, (uint32_t)n, center, size);
  if (perm_out) memcpy(perm_out, idx, n * sizeof(uint32_t));
  free(idx);
  return C.n_out;
}

/* ------------------------------------------------------------------ */
/* Packet traversal study (analysis / test infrastructure only): G consecutive queries walk the tree TOGETHER, a    */
/* node is visited if any of them needs it, every query prunes with its own current best.  The reference's answer   */
/* for a query is argmin over points with d2 < maxdist2 of (d2, position in the query's own near-first depth-first  */
/* order); the walk order of the packet differs from that order, so (a) pruning is relaxed to strict '>' (a subtree  */
/* at exactly the current best distance may hold the tie that comes first) and (b) ties are broken by the path key: */
/* bit (63 - level) = 1 where the path takes the query's FAR child, then the position inside the bucket.            */
/* Returns the same indices as orc_find_closest (checked in tests); counts what a wave-wide walk would touch.       */
/* ------------------------------------------------------------------ */
typedef struct {
  const orc_tree *t;
  const double *q;      /* G queries */
  int G;
  double maxd2;
  double best[64];
  int bk[64];
  unsigned long long key[64], bkey[64];
  int bpos[64];
  long n_int, n_leaf, n_pts, lane_pts;
} pk_state;

static void pk_visit(pk_state *S, const orc_node *nd, unsigned long long mask, int depth)
{
  if (nd->isleaf) {
    S->n_leaf++;
    S->n_pts += nd->npts;
    for (int i = 0; i < nd->npts; i++)
      for (int l = 0; l < S->G; l++) {
        if (!((mask >> l) & 1ull)) continue;
        S->lane_pts++;
        const double d2 = dist2(S->q + 3 * l, S->t->xyz + 3 * (size_t)nd->p[i]);
        int take = 0;
        if (d2 < S->best[l]) take = 1;
        else if (d2 == S->best[l] && S->bk[l] >= 0 &&
                 (S->key[l] < S->bkey[l] || (S->key[l] == S->bkey[l] && i < S->bpos[l]))) take = 1;
        if (take) { S->best[l] = d2; S->bk[l] = nd->p[i]; S->bkey[l] = S->key[l]; S->bpos[l] = i; }
      }
    return;
  }
  S->n_int++;
  unsigned long long m = 0, near1 = 0;
  double m2[64];
  for (int l = 0; l < S->G; l++) {
    if (!((mask >> l) & 1ull)) continue;
    const double *p = S->q + 3 * l;
    double a = fabs(p[0] - nd->center[0]) - nd->dx;
    double b = fabs(p[1] - nd->center[1]) - nd->dy;
    double c = fabs(p[2] - nd->center[2]) - nd->dz;
    double ab = (a < b) ? b : a;
    double approx = (ab < c) ? c : ab;
    if (approx >= 0 && approx * approx > S->best[l]) continue;        /* relaxed: '>' where the reference has '>=' */
    m |= 1ull << l;
    const double myd = nd->splitval - p[nd->splitaxis];
    m2[l] = myd * myd;
    if (myd >= 0.0) near1 |= 1ull << l;
  }
  if (!m) return;
  const int first_is_1 = __builtin_popcountll(near1 & m) * 2 >= __builtin_popcountll(m);
  for (int pass = 0; pass < 2; pass++) {
    const int c1 = (pass == 0) ? first_is_1 : !first_is_1;            /* this pass walks child1? */
    unsigned long long sub = 0;
    for (int l = 0; l < S->G; l++) {
      if (!((m >> l) & 1ull)) continue;
      const int is_near = (((near1 >> l) & 1ull) != 0) == (c1 != 0);
      if (!is_near && m2[l] > S->best[l]) continue;                   /* relaxed plane test, with the CURRENT best */
      sub |= 1ull << l;
      const unsigned long long bit = 1ull << (63 - depth);
      S->key[l] = (S->key[l] & ~((bit << 1) - 1ull)) | (is_near ? 0ull : bit);
    }
    if (sub) pk_visit(S, c1 ? nd->child1 : nd->child2, sub, depth + 1);
  }
}

/* q: K queries in the order in which they are grouped (G <= 64 consecutive ones form a packet);
 * counters[0..3] += nodes, buckets, bucket points the packets touch, and point tests summed over lanes */
int orc_packet_find_closest(const orc_tree *t, const double *q, size_t K, int G, double maxdist2, int32_t *idx,
                            double *d2, long *counters)
{
  if (G < 1 || G > 64 || t->max_depth > 63) return -1;
  pk_state S;
  S.t = t; S.maxd2 = maxdist2;
  S.n_int = S.n_leaf = S.n_pts = S.lane_pts = 0;
  for (size_t base = 0; base < K; base += (size_t)G) {
    S.G = (int)((K - base < (size_t)G) ? (K - base) : (size_t)G);
    S.q = q + 3 * base;
    for (int l = 0; l < S.G; l++) { S.best[l] = maxdist2; S.bk[l] = -1; S.key[l] = 0; S.bkey[l] = 0; S.bpos[l] = 0; }
    pk_visit(&S, t->root, (S.G == 64) ? ~0ull : ((1ull << S.G) - 1ull), 0);
    for (int l = 0; l < S.G; l++) { idx[base + l] = S.bk[l]; if (d2) d2[base + l] = S.best[l]; }
  }
  if (counters) { counters[0] += S.n_int; counters[1] += S.n_leaf; counters[2] += S.n_pts; counters[3] += S.lane_pts; }
  return 0;
}
