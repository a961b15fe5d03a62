// Scan::calcNormals on the GPU: calculateNormalsApxKNN(normals, points, k = 10, rPos, eps = 1.0)
// (src/slam6d/scan.cc:398-427, src/slam6d/normals.cc:35-111).
//
// The reference answers "the k approximate nearest neighbours of every point of the scan" with the ANN 1.1.1
// library: a kd-tree with ONE point per leaf built by the sliding-midpoint rule, searched with a (1+eps) error
// bound.  With eps = 1 the neighbour lists are far from the exact ones and depend on the shape of that very tree
// and on the order in which it is walked, so the same tree is built here and walked in the same order:
//
//   build (3rdparty/ann/ann_1.1.1_modified/src/kd_tree.cpp:319-404, kd_split.cpp:146-213, kd_util.cpp:225-319)
//     level by level for every cell of the level at once while cells are large -- point min/max per cell by one
//     wavefront, the library's two in-place Hoare passes (annPlaneSplit: "< cv | >= cv", then "== cv | > cv" on the
//     right part) each as "the k-th misplaced element from the left swaps with the k-th misplaced from the right
//     end" (prefix sums + one swap kernel), which reproduces the permutation and therefore which of several points
//     ON the cutting plane goes to which side -- and, once a cell holds <= 64 points, the rest of its subtree by
//     ONE wavefront (a lane per point, ballots instead of prefix sums) without leaving the registers.
//     Bucket size 1 means a cell of n points always yields n-1 splitting nodes: the node that separates positions
//     p and p+1 of the final point order gets index p, so no node allocation or compaction is needed.
//   search + PCA (kd_search.cpp:89-210, pr_queue_k.h:66-115, normals.cc:64-105)
//     one thread per scan point, in leaf order so that neighbouring lanes walk neighbouring paths; the k best are a
//     sorted list in registers; the far-child stack lives in LDS with a spill area in global memory; then the
//     neighbour mean, covariance, newmat's tred2/tql2 (evalue.cpp:24-156) on the 3x3, the flip towards the sensor.
//
// All fp64, no contraction: neighbour lists and normals are bit-identical to the library's.
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"
#include "part_scan.h"

namespace tdtk {

#define WAVE 64
#define ANN_SMALL 64u         // cells up to this size are finished by one wavefront
#ifndef ANN_MID
#define ANN_MID 2048u         // round 5: cells up to this size are taken down to ANN_SMALL-point cells by one workgroup in LDS (k_ann_mid)
#endif
#define ANN_ERR 0.001         // kd_split.cpp:34
#define A_LEAF 0x20000000u    // child reference: leaf flag | position (29 bits); c0 bits 30..31 = cutting dimension
#define A_VAL 0x1FFFFFFFu
#define NOSEG 0xFFFFFFFFu
#define ANN_MAX_LEVELS 4096

struct ASeg {
  uint32_t start, n;
  int32_t parent;   // node index, -1 for the root
  uint32_t side;    // 0 -> low child, 1 -> high child
  uint32_t depth, pad;
  double blo[3], bhi[3];   // the cell (kd_tree.cpp:346-357), not the points' own bounding box
};
struct AMeas { double mn[3], mx[3]; };
struct AMeasU { unsigned long long mn[3], mx[3]; };   // order-preserving integer images of the doubles (enc_f64)
#define ENC_PINF 0xFFF0000000000000ull   // enc(+inf)
#define ENC_NINF 0x000FFFFFFFFFFFFFull   // enc(-inf)
struct ADec { double cv; uint32_t cd, mode, n_lo, slot0, slot1, pad; };   // mode 0 midpoint, 1 slid to min, 2 slid to max

static __device__ __forceinline__ double coord_of(const double* __restrict__ cx, const double* __restrict__ cy,
                                                  const double* __restrict__ cz, uint32_t ax, uint32_t p)
{
  return (ax == 0) ? cx[p] : ((ax == 1) ? cy[p] : cz[p]);
}

// cutting dimension / value of sl_midpt_split (kd_split.cpp:158-209) from the cell and the points' min/max
static __device__ __forceinline__ void sl_midpt_rule(const double* blo, const double* bhi, const double* mn,
                                                     const double* mx, uint32_t& cd, double& cv, uint32_t& mode)
{
  double max_length = bhi[0] - blo[0];
#pragma unroll
  for (int d = 1; d < 3; d++) {
    const double length = bhi[d] - blo[d];
    if (length > max_length) max_length = length;
  }
  double max_spread = -1;
  cd = 0;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    if ((bhi[d] - blo[d]) >= (1 - ANN_ERR) * max_length) {
      const double spr = mx[d] - mn[d];
      if (spr > max_spread) { max_spread = spr; cd = (uint32_t)d; }
    }
  }
  const double lo = (cd == 0) ? blo[0] : ((cd == 1) ? blo[1] : blo[2]);
  const double hi = (cd == 0) ? bhi[0] : ((cd == 1) ? bhi[1] : bhi[2]);
  const double pmn = (cd == 0) ? mn[0] : ((cd == 1) ? mn[1] : mn[2]);
  const double pmx = (cd == 0) ? mx[0] : ((cd == 1) ? mx[1] : mx[2]);
  const double ideal = (lo + hi) / 2;
  if (ideal < pmn) { cv = pmn; mode = 1; }
  else if (ideal > pmx) { cv = pmx; mode = 2; }
  else { cv = ideal; mode = 0; }
}
static __device__ __forceinline__ uint32_t sl_midpt_nlo(uint32_t mode, uint32_t n, uint32_t br1, uint32_t br2)
{
  if (mode == 1) return 1u;           // kd_split.cpp:208-212
  if (mode == 2) return n - 1u;
  if (br1 > n / 2) return br1;
  if (br2 < n / 2) return br2;
  return n / 2;
}

// ---- level-parallel part -----------------------------------------------------------------------
__global__ void k_ann_init(const double* __restrict__ xyz, uint32_t M, uint32_t* __restrict__ perm,
                           uint32_t* __restrict__ seg_of, double* __restrict__ cx, double* __restrict__ cy,
                           double* __restrict__ cz, uint32_t* __restrict__ bad, uint32_t mid_cap)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const double x = xyz[3 * (size_t)p], y = xyz[3 * (size_t)p + 1], z = xyz[3 * (size_t)p + 2];
  perm[p] = p; seg_of[p] = (M > mid_cap) ? 0u : NOSEG;
  cx[p] = x; cy[p] = y; cz[p] = z;
  if (!(isfinite(x) && isfinite(y) && isfinite(z))) atomicOr(bad, 1u);
}

// the root cell = annEnclRect of all points (kd_tree.cpp:381-385); small[4] counts the small cells
__global__ void k_ann_root(const double* __restrict__ box, uint32_t M, ASeg* __restrict__ segs,
                           ASeg* __restrict__ small_list, uint32_t* __restrict__ small, double* __restrict__ bb,
                           AMeasU* __restrict__ meas, unsigned long long* __restrict__ cnt, uint32_t* __restrict__ lvl,
                           ASeg* __restrict__ mid_list, uint32_t mid_cap)
{
  lvl[0] = (M > mid_cap) ? 1u : 0u;
  for (int d = 0; d < 3; d++) { meas[0].mn[d] = ENC_PINF; meas[0].mx[d] = ENC_NINF; }
  cnt[0] = 0ull;
  ASeg r;
  r.start = 0; r.n = M; r.parent = -1; r.side = 0; r.depth = 0; r.pad = 0;
  for (int d = 0; d < 3; d++) { r.blo[d] = box[d]; r.bhi[d] = box[3 + d]; bb[d] = box[d]; bb[3 + d] = box[3 + d]; }
  if (M > mid_cap) segs[0] = r;
  else if (M > ANN_SMALL) { mid_list[0] = r; small[5] = 1; }
  else if (M > 1) { small_list[0] = r; small[4] = 1; }
  else small[0] = A_LEAF | 0u;     // a single point: the root is its leaf
}

// Point min / max of every cell of the level.  A wavefront takes 64 consecutive positions; cells are contiguous
// runs of positions (at least ANN_SMALL + 1 long), so the wave sees at most a few of them: one reduction and six
// atomics per run.  min / max do not depend on the order, unlike the centroid sum of the search tree's builder.
static __device__ __forceinline__ unsigned long long enc_f64(double v)
{
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static __device__ __forceinline__ double dec_f64(unsigned long long e)
{
  return __longlong_as_double((long long)((e >> 63) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e));
}

#define MEAS_ITERS 16   // a wavefront covers 64 * MEAS_ITERS consecutive positions

static __device__ __forceinline__ void meas_flush(AMeasU* __restrict__ out, uint32_t sg, const double* mn, const double* mx)
{
#pragma unroll
  for (int d = 0; d < 3; d++) {
    atomicMin(&out[sg].mn[d], enc_f64(mn[d]));
    atomicMax(&out[sg].mx[d], enc_f64(mx[d]));
  }
}

__global__ void __launch_bounds__(256) k_ann_measure(uint32_t* __restrict__ seg_of, uint32_t M,
                                                     const double* __restrict__ cx, const double* __restrict__ cy,
                                                     const double* __restrict__ cz, AMeasU* __restrict__ out,
                                                     const ASeg* __restrict__ prev_segs, const ADec* __restrict__ prev_dec)
{
  __shared__ uint32_t s_seg[4];
  __shared__ double s_mn[4][3], s_mx[4][3];
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const uint32_t base = (blockIdx.x * 4u + wv) * (WAVE * MEAS_ITERS);
  // all loads of the wave's range are issued up front (positions outside any cell cost a label only)
  uint32_t sgs[MEAS_ITERS];
  double vs[MEAS_ITERS][3];
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    const uint32_t p = base + (uint32_t)it * WAVE + lane;
    sgs[it] = (p < M) ? seg_of[p] : NOSEG;
  }
  if (prev_segs) {
    // the labels still name the cells of the previous level: move every position to the child cell it fell into
    // (k_ann_children left the children's slots in dec) -- what a separate relabel pass over all positions did
#pragma unroll
    for (int it = 0; it < MEAS_ITERS; it++) {
      const uint32_t p = base + (uint32_t)it * WAVE + lane;
      if (sgs[it] != NOSEG) {
        const uint32_t sg = sgs[it];
        sgs[it] = ((p - prev_segs[sg].start) < prev_dec[sg].n_lo) ? prev_dec[sg].slot0 : prev_dec[sg].slot1;
        seg_of[p] = sgs[it];
      }
    }
  }
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    const uint32_t p = base + (uint32_t)it * WAVE + lane;
    vs[it][0] = vs[it][1] = vs[it][2] = 0;
    if (sgs[it] != NOSEG) { vs[it][0] = cx[p]; vs[it][1] = cy[p]; vs[it][2] = cz[p]; }
  }
  // the run the wave is in the middle of (`pend`, wave-uniform) is kept as PER-LANE partial min / max: while whole
  // rows of 64 positions stay inside it nothing crosses lanes; the shuffle reduction runs once, when the run ends
  uint32_t pend = NOSEG;
  double amn[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, amx[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
  auto reduce_lanes = [&](double* mn, double* mx) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        double t;
        t = __shfl_xor(mn[d], off, WAVE); mn[d] = (t < mn[d]) ? t : mn[d];
        t = __shfl_xor(mx[d], off, WAVE); mx[d] = (mx[d] < t) ? t : mx[d];
      }
  };
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    const uint32_t sg = sgs[it];
    const double v[3] = {vs[it][0], vs[it][1], vs[it][2]};
    if (pend != NOSEG && __ballot(sg == pend) == ~0ull) {
#pragma unroll
      for (int d = 0; d < 3; d++) { amn[d] = (v[d] < amn[d]) ? v[d] : amn[d]; amx[d] = (amx[d] < v[d]) ? v[d] : amx[d]; }
      continue;
    }
    unsigned long long todo = __ballot(sg != NOSEG);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t cur = __shfl(sg, leader, WAVE);
      const bool mine = (sg == cur);
      if (cur != pend) {
        if (pend != NOSEG) {
          reduce_lanes(amn, amx);
          if (lane == 0) meas_flush(out, pend, amn, amx);
        }
        pend = cur;
#pragma unroll
        for (int d = 0; d < 3; d++) { amn[d] = HUGE_VAL; amx[d] = -HUGE_VAL; }
      }
      if (mine) {
#pragma unroll
        for (int d = 0; d < 3; d++) { amn[d] = (v[d] < amn[d]) ? v[d] : amn[d]; amx[d] = (amx[d] < v[d]) ? v[d] : amx[d]; }
      }
      todo &= ~__ballot(mine);
    }
  }
  double pmn[3] = {amn[0], amn[1], amn[2]}, pmx[3] = {amx[0], amx[1], amx[2]};
  if (pend != NOSEG) reduce_lanes(pmn, pmx);
  // the four waves of the block cover adjacent ranges: merge equal cells before touching memory
  if (lane == 0) {
    s_seg[wv] = pend;
    for (int d = 0; d < 3; d++) { s_mn[wv][d] = pmn[d]; s_mx[wv][d] = pmx[d]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t cur = NOSEG;
    double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int w = 0; w < 4; w++) {
      const uint32_t sg = s_seg[w];
      if (sg == NOSEG) continue;
      if (sg != cur) {
        if (cur != NOSEG) meas_flush(out, cur, mn, mx);
        cur = sg;
        for (int d = 0; d < 3; d++) { mn[d] = s_mn[w][d]; mx[d] = s_mx[w][d]; }
      } else {
        for (int d = 0; d < 3; d++) { mn[d] = (s_mn[w][d] < mn[d]) ? s_mn[w][d] : mn[d]; mx[d] = (mx[d] < s_mx[w][d]) ? s_mx[w][d] : mx[d]; }
      }
    }
    if (cur != NOSEG) meas_flush(out, cur, mn, mx);
  }
}

__global__ void k_ann_decide(const ASeg* __restrict__ segs, const uint32_t* __restrict__ nseg_ptr,
                             const AMeasU* __restrict__ meas, ADec* __restrict__ dec)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *nseg_ptr) return;       // the number of cells of this level is known on the device only
  const ASeg sg = segs[i];
  AMeas m;
  for (int d = 0; d < 3; d++) { m.mn[d] = dec_f64(meas[i].mn[d]); m.mx[d] = dec_f64(meas[i].mx[d]); }
  ADec d;
  sl_midpt_rule(sg.blo, sg.bhi, m.mn, m.mx, d.cd, d.cv, d.mode);
  d.n_lo = 0; d.slot0 = d.slot1 = NOSEG; d.pad = 0;
  dec[i] = d;
}

// Per cell: how many of its points lie below the cutting plane (-> br1) and how many on it (br2 = br1 + that):
// all annPlaneSplit's breaks need, and independent of the order of the points.  Same run-wise reduction as
// k_ann_measure; one 64-bit atomicAdd per run (low word "< cv", high word "== cv").
__global__ void __launch_bounds__(256) k_ann_count(const uint32_t* __restrict__ seg_of, uint32_t M,
                                                   const ADec* __restrict__ dec, const double* __restrict__ cx,
                                                   const double* __restrict__ cy, const double* __restrict__ cz,
                                                   unsigned long long* __restrict__ cnt)
{
  __shared__ uint32_t s_seg[4];
  __shared__ unsigned long long s_cnt[4];
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const uint32_t base = (blockIdx.x * 4u + wv) * (WAVE * MEAS_ITERS);
  uint32_t sgs[MEAS_ITERS];
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    const uint32_t p = base + (uint32_t)it * WAVE + lane;
    sgs[it] = (p < M) ? seg_of[p] : NOSEG;
  }
  // phase by phase, MEAS_ITERS independent loads deep: the cell's decision, then the coordinate it asks for
  uint32_t cds[MEAS_ITERS];
  double cvs[MEAS_ITERS], cs[MEAS_ITERS];
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    cds[it] = 0; cvs[it] = 0.0;
    if (sgs[it] != NOSEG) { cds[it] = dec[sgs[it]].cd; cvs[it] = dec[sgs[it]].cv; }
  }
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    const uint32_t p = base + (uint32_t)it * WAVE + lane;
    cs[it] = 0.0;
    if (sgs[it] != NOSEG) cs[it] = coord_of(cx, cy, cz, cds[it], p);
  }
  uint32_t pend = NOSEG;
  unsigned long long pcnt = 0;
#pragma unroll
  for (int it = 0; it < MEAS_ITERS; it++) {
    const uint32_t sg = sgs[it];
    const bool lt = (sg != NOSEG) && (cs[it] < cvs[it]);
    const bool eq = (sg != NOSEG) && (cs[it] <= cvs[it]) && !lt;
    unsigned long long todo = __ballot(sg != NOSEG);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t cur = __shfl(sg, leader, WAVE);
      const bool mine = (sg == cur);
      const unsigned long long add = (unsigned long long)__popcll(__ballot(mine && lt)) |
                                     ((unsigned long long)__popcll(__ballot(mine && eq)) << 32);
      if (cur != pend) {
        if (pend != NOSEG && lane == 0) atomicAdd(&cnt[pend], pcnt);
        pend = cur; pcnt = add;
      } else pcnt += add;
      todo &= ~__ballot(mine);
    }
  }
  if (lane == 0) { s_seg[wv] = pend; s_cnt[wv] = pcnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t cur = NOSEG;
    unsigned long long acc = 0;
    for (int w = 0; w < 4; w++) {
      if (s_seg[w] == NOSEG) continue;
      if (s_seg[w] != cur) {
        if (cur != NOSEG) atomicAdd(&cnt[cur], acc);
        cur = s_seg[w]; acc = s_cnt[w];
      } else acc += s_cnt[w];
    }
    if (cur != NOSEG) atomicAdd(&cnt[cur], acc);
  }
}

// pass 1: "< cv" belongs left of br1.  pass 2 (positions >= br1 only): "<= cv", i.e. "== cv", belongs left of br2
template <int PASS>
__global__ void k_ann_misplaced(const uint32_t* __restrict__ seg_of, const ASeg* __restrict__ segs,
                                const ADec* __restrict__ dec, const unsigned long long* __restrict__ cnt,
                                const double* __restrict__ cx, const double* __restrict__ cy,
                                const double* __restrict__ cz, uint32_t M, unsigned long long* __restrict__ LR,
                                uint32_t* __restrict__ eq_flag)
{
  // The second Hoare pass (annPlaneSplit's: what lies ON the cutting plane to the front of the right part) has work only where
  // a cell has points on its plane -- a slide onto a point, or coordinates that repeat.  Pass 1 raises the level's flag if any
  // cell has such points; pass 2 and the scan, list and swap kernels behind it return at once when it is down (round 6: on a
  // cloud without repeated coordinates that is four of a level's eleven passes over the points).
  if (PASS == 2 && eq_flag != nullptr && *eq_flag == 0u) return;
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p > M) return;
  uint32_t l = 0, r = 0;
  if (p < M) {
    const uint32_t sg = seg_of[p];
    if (sg != NOSEG) {
      const uint32_t rel = p - segs[sg].start;
      const unsigned long long cn = cnt[sg];
      if (PASS == 1 && eq_flag != nullptr && (uint32_t)(cn >> 32) != 0u) *eq_flag = 1u;
      const uint32_t br1 = (uint32_t)cn, br2 = br1 + (uint32_t)(cn >> 32);
      const double c = coord_of(cx, cy, cz, dec[sg].cd, p), cv = dec[sg].cv;
      if (PASS == 1) {
        const bool left_region = rel < br1, f = c < cv;
        l = (left_region && !f) ? 1u : 0u;
        r = (!left_region && f) ? 1u : 0u;
      } else if (rel >= br1) {
        const bool left_region = rel < br2, g = c <= cv;
        l = (left_region && !g) ? 1u : 0u;
        r = (!left_region && g) ? 1u : 0u;
      }
    }
  }
  LR[p] = (unsigned long long)l | ((unsigned long long)r << 32);   // index M is a zero terminator (totals)
}

// k-th misplaced from the left pairs with the k-th misplaced from the right END of the cell
__global__ void k_ann_swaplist(const uint32_t* __restrict__ seg_of, const ASeg* __restrict__ segs,
                               const unsigned long long* __restrict__ LR, const unsigned long long* __restrict__ AB,
                               uint32_t M, uint32_t* __restrict__ posL, uint32_t* __restrict__ posR, const uint32_t* __restrict__ gate)
{
  if (gate != nullptr && *gate == 0u) return;        // (a second pass with nothing to do: see k_ann_misplaced)
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const unsigned long long lr = LR[p];
  if ((uint32_t)lr) posL[(uint32_t)AB[p]] = p;
  if ((uint32_t)(lr >> 32)) {
    const uint32_t sg = seg_of[p];
    const uint32_t s = segs[sg].start, n = segs[sg].n;
    const uint32_t Bs = (uint32_t)(AB[s] >> 32);
    const uint32_t total = (uint32_t)(AB[s + n] >> 32) - Bs;
    const uint32_t kfwd = (uint32_t)(AB[p] >> 32) - Bs;
    posR[Bs + (total - 1u - kfwd)] = p;
  }
}
__global__ void k_ann_swap(const uint32_t* __restrict__ posL, const uint32_t* __restrict__ posR,
                           const unsigned long long* __restrict__ nswap_ptr, uint32_t* __restrict__ perm,
                           double* __restrict__ cx, double* __restrict__ cy, double* __restrict__ cz, const uint32_t* __restrict__ gate)
{
  if (gate != nullptr && *gate == 0u) return;
  const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= (uint32_t)*nswap_ptr) return;
  const uint32_t a = posL[c], b = posR[c];
  const uint32_t pa = perm[a], pb = perm[b];
  perm[a] = pb; perm[b] = pa;
  double t;
  t = cx[a]; cx[a] = cx[b]; cx[b] = t;
  t = cy[a]; cy[a] = cy[b]; cy[b] = t;
  t = cz[a]; cz[a] = cz[b]; cz[b] = t;
}

// ---- a Hoare pass in TWO passes over the points (round 6; build.hip, k_part_scan, has the argument) -----------------------
// Pass 1 of annPlaneSplit works on the whole cell with "below the cutting value", pass 2 on what lies right of br1 with
// "not above it"; both counts are known (k_ann_count), so a pass is: one segmented scan of the "stays right" flags that
// writes the run's index list (those from the front in order, the others from the back), then the swaps driven from the
// list's front part.  misplaced + scan + swap list + swap were four passes over the points and two 8-byte words per point.
template <int PASS>
__global__ void __launch_bounds__(PS_THREADS) k_ann_part_scan(const uint32_t* __restrict__ seg_of, const ASeg* __restrict__ segs,
                                                              const ADec* __restrict__ dec, unsigned long long* __restrict__ cnt,
                                                              const double* __restrict__ cx, const double* __restrict__ cy,
                                                              const double* __restrict__ cz, uint32_t M, uint32_t* __restrict__ list,
                                                              unsigned long long* __restrict__ status, uint32_t* __restrict__ counter,
                                                              uint32_t epoch, uint32_t ntiles, uint32_t* __restrict__ err,
                                                              uint32_t* __restrict__ eq_flag)
{
  if (PASS == 2 && eq_flag != nullptr && *eq_flag == 0u) return;     // (no cell of the level has points on its plane: see k_ann_misplaced)
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  const uint32_t tile = ps_draw_tile(counter, ntiles);
  const uint32_t wbase = tile * PS_TILE + wv * (WAVE * PS_ROWS) + lane;
  uint32_t sg[PS_ROWS], geb[PS_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PS_ROWS; r++) { const uint32_t p = wbase + r * WAVE; sg[r] = (p < M) ? seg_of[p] : NOSEG; }
  uint32_t gebits = 0u, headbits = 0u, actbits = 0u;
  {
    uint32_t cds[PS_ROWS], sts[PS_ROWS];
    unsigned long long cns[PS_ROWS];
    double c[PS_ROWS], cvs[PS_ROWS];
    bool any_eq = false;
#pragma unroll
    for (uint32_t r = 0; r < PS_ROWS; r++) {
      cds[r] = 0u; cvs[r] = 0.0; sts[r] = 0u; cns[r] = 0ull;
      if (sg[r] != NOSEG) { cds[r] = dec[sg[r]].cd; cvs[r] = dec[sg[r]].cv; sts[r] = segs[sg[r]].start; if (PASS == 2) cns[r] = cnt[sg[r]]; }
    }
#pragma unroll
    for (uint32_t r = 0; r < PS_ROWS; r++) {
      c[r] = 0.0;
      if (sg[r] != NOSEG) c[r] = coord_of(cx, cy, cz, cds[r], wbase + r * WAVE);
    }
#pragma unroll
    for (uint32_t r = 0; r < PS_ROWS; r++) {
      if (sg[r] == NOSEG) continue;
      const uint32_t rel = wbase + r * WAVE - sts[r], br1 = (uint32_t)cns[r];
      if (PASS == 1) {
        if (c[r] == cvs[r]) any_eq = true;
        actbits |= 1u << r;
        if (!(c[r] < cvs[r])) gebits |= 1u << r;
        if (rel == 0u) headbits |= 1u << r;
      } else if (rel >= br1) {
        actbits |= 1u << r;
        if (!(c[r] <= cvs[r])) gebits |= 1u << r;
        if (rel == br1) headbits |= 1u << r;
      }
    }
    if (PASS == 1 && eq_flag != nullptr && any_eq) *eq_flag = 1u;
    if (PASS == 1) {
      // the cell's counts on the way (what k_ann_count's own pass over the points did): points below the cutting value |
      // points ON it << 32, one atomic per run a wave sees, the workgroup's waves merged first
      __shared__ uint32_t s_cseg[PS_THREADS / WAVE];
      __shared__ unsigned long long s_ccnt[PS_THREADS / WAVE];
      uint32_t pend = NOSEG;
      unsigned long long pcnt = 0;
#pragma unroll
      for (uint32_t r = 0; r < PS_ROWS; r++) {
        const uint32_t g = sg[r];
        const bool lt = (g != NOSEG) && (c[r] < cvs[r]);
        const bool eq = (g != NOSEG) && (c[r] == cvs[r]);
        unsigned long long todo = __ballot(g != NOSEG);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t cur = __shfl(g, leader, WAVE);
          const bool mine = (g == cur);
          const unsigned long long add = (unsigned long long)__popcll(__ballot(mine && lt)) | ((unsigned long long)__popcll(__ballot(mine && eq)) << 32);
          if (cur != pend) {
            if (pend != NOSEG && lane == 0) atomicAdd(&cnt[pend], pcnt);
            pend = cur; pcnt = add;
          } else pcnt += add;
          todo &= ~__ballot(mine);
        }
      }
      if (lane == 0) { s_cseg[wv] = pend; s_ccnt[wv] = pcnt; }
      __syncthreads();
      if (threadIdx.x == 0) {
        uint32_t cur = NOSEG;
        unsigned long long acc = 0;
        for (uint32_t w = 0; w < PS_THREADS / WAVE; w++) {
          if (s_cseg[w] == NOSEG) continue;
          if (s_cseg[w] != cur) { if (cur != NOSEG) atomicAdd(&cnt[cur], acc); cur = s_cseg[w]; acc = s_ccnt[w]; }
          else acc += s_ccnt[w];
        }
        if (cur != NOSEG) atomicAdd(&cnt[cur], acc);
      }
    }
  }
  uint32_t ext = 0u, win = 0u;
  ps_scan_core(gebits, headbits, tile, status, epoch, err, geb, ext, win);
#pragma unroll
  for (uint32_t r = 0; r < PS_ROWS; r++) {
    if (!((actbits >> r) & 1u)) continue;
    const uint32_t p = wbase + r * WAVE;
    const uint32_t g = geb[r] + (((ext >> r) & 1u) ? win : 0u);
    const uint32_t st = segs[sg[r]].start, n = segs[sg[r]].n;
    const uint32_t base = st + ((PASS == 1) ? 0u : (uint32_t)cnt[sg[r]]), j = p - base;
    const bool ge = (gebits >> r) & 1u;
    list[ge ? (base + g) : (st + n - 1u - (j - g))] = p;
  }
}
#define APW_ROWS 4u
template <int PASS>
__global__ void __launch_bounds__(256) k_ann_part_swap(const uint32_t* __restrict__ list, const uint32_t* __restrict__ seg_of,
                                                       const ASeg* __restrict__ segs, const unsigned long long* __restrict__ cnt,
                                                       uint32_t M, uint32_t* __restrict__ perm, double* __restrict__ cx,
                                                       double* __restrict__ cy, double* __restrict__ cz, const uint32_t* __restrict__ gate)
{
  if (gate != nullptr && *gate == 0u) return;
  const uint32_t base0 = blockIdx.x * (256u * APW_ROWS) + threadIdx.x;
  uint32_t sg[APW_ROWS], a[APW_ROWS], b[APW_ROWS];
  bool sw[APW_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < APW_ROWS; r++) {
    const uint32_t p = base0 + r * 256u;
    sg[r] = (p < M) ? seg_of[p] : NOSEG;
    a[r] = (p < M) ? list[p] : 0u;
  }
#pragma unroll
  for (uint32_t r = 0; r < APW_ROWS; r++) {
    const uint32_t p = base0 + r * 256u;
    sw[r] = false; b[r] = 0u;
    if (sg[r] == NOSEG) continue;
    const uint32_t st = segs[sg[r]].start, n = segs[sg[r]].n;
    const unsigned long long cn = cnt[sg[r]];
    const uint32_t br1 = (uint32_t)cn, br2 = br1 + (uint32_t)(cn >> 32);
    const uint32_t base = st + ((PASS == 1) ? 0u : br1), bound = st + ((PASS == 1) ? br1 : br2);
    if (p < base) continue;
    const uint32_t nge = st + n - bound, j = p - base;          // those that stay right of the boundary
    if (j < nge && a[r] < bound) { sw[r] = true; b[r] = list[p + nge]; }
  }
  uint32_t pa[APW_ROWS], pb[APW_ROWS];
  double xa[APW_ROWS], xb[APW_ROWS], ya[APW_ROWS], yb[APW_ROWS], za[APW_ROWS], zb[APW_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < APW_ROWS; r++)
    if (sw[r]) {
      pa[r] = perm[a[r]]; pb[r] = perm[b[r]];
      xa[r] = cx[a[r]]; xb[r] = cx[b[r]]; ya[r] = cy[a[r]]; yb[r] = cy[b[r]]; za[r] = cz[a[r]]; zb[r] = cz[b[r]];
    }
#pragma unroll
  for (uint32_t r = 0; r < APW_ROWS; r++)
    if (sw[r]) {
      perm[a[r]] = pb[r]; perm[b[r]] = pa[r];
      cx[a[r]] = xb[r]; cx[b[r]] = xa[r]; cy[a[r]] = yb[r]; cy[b[r]] = ya[r]; cz[a[r]] = zb[r]; cz[b[r]] = za[r];
    }
}

static __device__ __forceinline__ void ann_hook(AnnNode* __restrict__ nodes, uint32_t* __restrict__ root_ref,
                                                int32_t parent, uint32_t side, uint32_t ref)
{
  if (parent < 0) *root_ref = ref;
  else if (side) nodes[parent].c1 = ref;
  else nodes[parent].c0 = (nodes[parent].c0 & 0xC0000000u) | ref;
}

// per cell: n_lo, the splitting node, the two child cells (leaf / small cell / cell of the next level).
// small: [0] root_ref [1] max depth [2] err [3] cells of the next level [4] small cells [5] mid cells (k_ann_mid)
__global__ void k_ann_children(const ASeg* __restrict__ segs, const uint32_t* __restrict__ nseg_ptr, ADec* __restrict__ dec,
                               const unsigned long long* __restrict__ cnt, AnnNode* __restrict__ nodes,
                               ASeg* __restrict__ next, ASeg* __restrict__ small_list, uint32_t* __restrict__ small,
                               AMeasU* __restrict__ meas_next, unsigned long long* __restrict__ cnt_next,
                               ASeg* __restrict__ mid_list, uint32_t mid_cap)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nseg_ptr[0]) return;     // nseg_ptr[1] counts the cells of the next level
  const ASeg sg = segs[i];
  ADec d = dec[i];
  const uint32_t br1 = (uint32_t)cnt[i], br2 = br1 + (uint32_t)(cnt[i] >> 32);
  const uint32_t n_lo = sl_midpt_nlo(d.mode, sg.n, br1, br2);
  if (n_lo == 0 || n_lo >= sg.n) { atomicExch(small + 2, 1u); return; }   // cannot happen for finite input
  const uint32_t me = sg.start + n_lo - 1u;
  AnnNode nd;
  nd.cut_val = d.cv;
  nd.lo = (d.cd == 0) ? sg.blo[0] : ((d.cd == 1) ? sg.blo[1] : sg.blo[2]);
  nd.hi = (d.cd == 0) ? sg.bhi[0] : ((d.cd == 1) ? sg.bhi[1] : sg.bhi[2]);
  nd.c0 = d.cd << 30; nd.c1 = 0;
  uint32_t slot[2] = {NOSEG, NOSEG};
  for (uint32_t side = 0; side < 2; side++) {
    ASeg ch = sg;
    ch.start = side ? sg.start + n_lo : sg.start;
    ch.n = side ? sg.n - n_lo : n_lo;
    ch.parent = (int32_t)me; ch.side = side; ch.depth = sg.depth + 1;
    if (side) { if (d.cd == 0) ch.blo[0] = d.cv; else if (d.cd == 1) ch.blo[1] = d.cv; else ch.blo[2] = d.cv; }
    else { if (d.cd == 0) ch.bhi[0] = d.cv; else if (d.cd == 1) ch.bhi[1] = d.cv; else ch.bhi[2] = d.cv; }
    if (ch.n == 1) {
      if (side) nd.c1 = A_LEAF | ch.start; else nd.c0 |= A_LEAF | ch.start;
      if (ch.depth > *(volatile uint32_t*)(small + 1)) atomicMax(small + 1, ch.depth);
    } else if (ch.n <= ANN_SMALL) {
      small_list[atomicAdd(small + 4, 1u)] = ch;
    } else if (ch.n <= mid_cap) {
      mid_list[atomicAdd(small + 5, 1u)] = ch;
    } else {
      slot[side] = atomicAdd(const_cast<uint32_t*>(nseg_ptr) + 1, 1u);
      next[slot[side]] = ch;
      for (int a = 0; a < 3; a++) { meas_next[slot[side]].mn[a] = ENC_PINF; meas_next[slot[side]].mx[a] = ENC_NINF; }
      cnt_next[slot[side]] = 0ull;
    }
  }
  nodes[me] = nd;     // children that are cells hook themselves in later (they run after this kernel)
  ann_hook(nodes, small + 0, sg.parent, sg.side, me);
  d.n_lo = n_lo; d.slot0 = slot[0]; d.slot1 = slot[1];
  dec[i] = d;
}
// ---- small cells: one wavefront per cell, one lane per point --------------------------------------
// Once a cell holds <= 64 points its whole subtree is built by one wavefront without leaving the registers: every
// lane carries one point and the description of the sub-cell it currently belongs to (a contiguous range of lanes,
// its box, its parent), and all sub-cells of a level are split at once -- point min / max by a segmented shuffle
// reduction, the two Hoare passes of annPlaneSplit by ballots (rank among the misplaced of the own sub-cell) and
// one exchange through LDS to find the partner lane.  Same permutation, same nodes as the recursion.
static __device__ __forceinline__ double sel3(const double* a, uint32_t d) { return (d == 0) ? a[0] : ((d == 1) ? a[1] : a[2]); }
static __device__ __forceinline__ void put3(double* a, uint32_t d, double v) { if (d == 0) a[0] = v; else if (d == 1) a[1] = v; else a[2] = v; }

// the k-th misplaced lane from the left end of a sub-cell swaps with the k-th misplaced lane from its right end
static __device__ __forceinline__ uint32_t hoare_partner(bool ML, bool MR, unsigned long long mask, uint32_t cs, uint32_t lane,
                                                        volatile uint32_t* slotL, volatile uint32_t* slotR)
{
  const unsigned long long bML = __ballot(ML) & mask, bMR = __ballot(MR) & mask;
  const unsigned long long below = (1ull << lane) - 1ull, above = (~0ull << lane) << 1;
  const uint32_t kL = (uint32_t)__popcll(bML & below), kR = (uint32_t)__popcll(bMR & above);
  if (ML) slotL[cs + kL] = lane;
  if (MR) slotR[cs + kR] = lane;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  uint32_t partner = lane;
  if (ML) partner = slotR[cs + kL];
  if (MR) partner = slotL[cs + kR];
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
  return partner;
}

__global__ void __launch_bounds__(256) k_ann_small(const ASeg* __restrict__ small_list, uint32_t nsmall,
                                                   uint32_t* __restrict__ perm, double* __restrict__ cx,
                                                   double* __restrict__ cy, double* __restrict__ cz,
                                                   AnnNode* __restrict__ nodes, uint32_t* __restrict__ small)
{
  __shared__ uint32_t s_slot[4][2][WAVE];
  const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / WAVE);
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  if (w >= nsmall) return;
  volatile uint32_t* slotL = s_slot[threadIdx.x / WAVE][0];
  volatile uint32_t* slotR = s_slot[threadIdx.x / WAVE][1];
  const ASeg sg = small_list[w];
  const uint32_t S = sg.start, N = sg.n;
  if (N == 0) return;        // an unused slot of a stretch k_ann_mid reserved
  bool active = lane < N;
  double x = 0, y = 0, z = 0;
  uint32_t pm = 0;
  if (active) { x = cx[S + lane]; y = cy[S + lane]; z = cz[S + lane]; pm = perm[S + lane]; }
  // the sub-cell this lane's point is in
  uint32_t cs = 0, cn = N, side = sg.side, depth = sg.depth, pcd = 0;
  int32_t parent = sg.parent;
  bool top = true;
  double blo[3] = {sg.blo[0], sg.blo[1], sg.blo[2]}, bhi[3] = {sg.bhi[0], sg.bhi[1], sg.bhi[2]};
  uint32_t maxdepth = 0;
  for (;;) {
    if (active && cn == 1) {                    // a leaf: hook it into its splitting node
      const uint32_t ref = A_LEAF | (S + lane);
      if (top) ann_hook(nodes, small + 0, parent, side, ref);
      else if (side) nodes[parent].c1 = ref;
      else nodes[parent].c0 = (pcd << 30) | ref;
      maxdepth = (depth > maxdepth) ? depth : maxdepth;
      active = false;
    }
    if (!__ballot(active)) break;
    const unsigned long long mask = ((cn >= 64u) ? ~0ull : ((1ull << cn) - 1ull)) << cs;
    const uint32_t rel = lane - cs, seg_end = cs + cn;
    // annSpread / annMinMax of every sub-cell (kd_util.cpp:225-262): segmented reduction, then the head's value.
    // sl_midpt_split looks at the spreads of the dimensions whose cell side is within 0.1 % of the longest only
    // (kd_split.cpp:171-182) -- almost always ONE dimension, known from the cell's box before any point is looked at, and
    // then its choice does not depend on the spreads: the reduction carries that dimension's min / max alone (two doubles
    // through the LDS crossbar per step instead of six; the kernel is bound by those cross-lane moves).  An iteration in
    // which some sub-cell has two candidate dimensions reduces all three as before.
    uint32_t cd, mode;
    double cv;
    {
      double max_length = bhi[0] - blo[0];
#pragma unroll
      for (int d = 1; d < 3; d++) { const double length = bhi[d] - blo[d]; if (length > max_length) max_length = length; }
      uint32_t nq = 0, dq = 0;
#pragma unroll
      for (int d = 0; d < 3; d++)
        if ((bhi[d] - blo[d]) >= (1 - ANN_ERR) * max_length) { if (!nq) dq = (uint32_t)d; nq++; }
      if (!__ballot(active && nq != 1u)) {
        const double c1 = (dq == 0) ? x : ((dq == 1) ? y : z);
        double m1 = c1, M1 = c1;
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1) {
          const double a = __shfl_down(m1, off, WAVE), b = __shfl_down(M1, off, WAVE);
          if (lane + off < seg_end) { m1 = (a < m1) ? a : m1; M1 = (M1 < b) ? b : M1; }
        }
        m1 = __shfl(m1, cs, WAVE); M1 = __shfl(M1, cs, WAVE);
        // sl_midpt_rule with the one candidate: cd = dq, the cut at the cell's middle unless the points lie to one side of it
        cd = dq;
        const double lo = sel3(blo, dq), hi = sel3(bhi, dq);
        const double ideal = (lo + hi) / 2;
        if (ideal < m1) { cv = m1; mode = 1; }
        else if (ideal > M1) { cv = M1; mode = 2; }
        else { cv = ideal; mode = 0; }
      } else {
        double mn[3] = {x, y, z}, mx[3] = {x, y, z};
#pragma unroll
        for (int off = 1; off < WAVE; off <<= 1)
#pragma unroll
          for (int d = 0; d < 3; d++) {
            const double a = __shfl_down(mn[d], off, WAVE), b = __shfl_down(mx[d], off, WAVE);
            if (lane + off < seg_end) { mn[d] = (a < mn[d]) ? a : mn[d]; mx[d] = (mx[d] < b) ? b : mx[d]; }
          }
#pragma unroll
        for (int d = 0; d < 3; d++) { mn[d] = __shfl(mn[d], cs, WAVE); mx[d] = __shfl(mx[d], cs, WAVE); }
        sl_midpt_rule(blo, bhi, mn, mx, cd, cv, mode);
      }
    }
    // annPlaneSplit (kd_util.cpp:291-319), first pass: "< cv" | ">= cv"
    double c = (cd == 0) ? x : ((cd == 1) ? y : z);
    const bool f = active && (c < cv);
    const uint32_t br1 = (uint32_t)__popcll(__ballot(f) & mask);
    {
      const bool left_region = rel < br1;
      const bool ML = active && left_region && !f, MR = active && !left_region && f;
      if (__ballot(ML || MR)) {                 // (a pass in which no point of the wave is misplaced moves nothing)
        const uint32_t partner = hoare_partner(ML, MR, mask, cs, lane, slotL, slotR);
        x = __shfl(x, partner, WAVE); y = __shfl(y, partner, WAVE); z = __shfl(z, partner, WAVE); pm = __shfl(pm, partner, WAVE);
      }
    }
    // second pass on [br1, n): "== cv" | "> cv"
    c = (cd == 0) ? x : ((cd == 1) ? y : z);
    const bool g = active && (rel >= br1) && (c <= cv);
    const uint32_t br2 = br1 + (uint32_t)__popcll(__ballot(g) & mask);
    {
      const bool in2 = active && (rel >= br1);
      const bool left_region = rel < br2;
      const bool ML = in2 && left_region && !g, MR = in2 && !left_region && g;
      if (__ballot(ML || MR)) {                 // (points ON the plane that are out of place: rare)
        const uint32_t partner = hoare_partner(ML, MR, mask, cs, lane, slotL, slotR);
        x = __shfl(x, partner, WAVE); y = __shfl(y, partner, WAVE); z = __shfl(z, partner, WAVE); pm = __shfl(pm, partner, WAVE);
      }
    }
    const uint32_t n_lo = sl_midpt_nlo(mode, cn, br1, br2);
    if (active && (n_lo == 0 || n_lo >= cn)) atomicExch(small + 2, 1u);    // cannot happen for finite input
    if (__ballot(active && (n_lo == 0 || n_lo >= cn))) return;
    const uint32_t me = S + cs + n_lo - 1u;
    if (active && lane == cs) {                 // the sub-cell's first lane writes its splitting node ...
      AnnNode* nd = nodes + me;
      nd->cut_val = cv; nd->lo = sel3(blo, cd); nd->hi = sel3(bhi, cd);
      if (top) { nd->c0 = cd << 30; nd->c1 = 0; ann_hook(nodes, small + 0, parent, side, me); }
      else if (side) nodes[parent].c1 = me;     // ... and hooks it in (the child words are written by the children only)
      else nodes[parent].c0 = (pcd << 30) | me;
    }
    if (active) {
      const bool go_hi = rel >= n_lo;           // kd_tree.cpp:346-357
      if (go_hi) { put3(blo, cd, cv); cs += n_lo; cn -= n_lo; }
      else { put3(bhi, cd, cv); cn = n_lo; }
      parent = (int32_t)me; side = go_hi ? 1u : 0u; pcd = cd; depth++;
      top = false;
    }
  }
  if (lane < N) { cx[S + lane] = x; cy[S + lane] = y; cz[S + lane] = z; perm[S + lane] = pm; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const uint32_t t = __shfl_xor(maxdepth, off, WAVE); maxdepth = (t > maxdepth) ? t : maxdepth; }
  // (one atomic per wave on one word were 0.3 ms of a 1M-point build: only who can raise the maximum tries)
  if (lane == 0 && maxdepth > *(volatile uint32_t*)(small + 1)) atomicMax(small + 1, maxdepth);
}

// ---- mid cells (round 5): one workgroup per cell of 65 .. ANN_MID points, the cell's points in LDS -----------------
// The level loop above costs thirteen launches per level whatever the level holds, and a 1M-point scan has a dozen levels
// between 2048-point and 64-point cells (the sliding-midpoint tree is not balanced): 69 % of a calcNormals call were those
// launches.  Here ONE workgroup of four wavefronts takes a cell of at most ANN_MID points -- coordinates in LDS, the
// original indices as 16-bit positions into the cell's own stretch of `perm` -- down to cells of at most ANN_SMALL points
// (which k_ann_small finishes): per cell the points' min / max (kd_util.cpp:225-262), the sliding-midpoint rule, the
// breaks, and annPlaneSplit's two in-place Hoare passes (kd_util.cpp:291-319) each as "the k-th misplaced element from the
// left swaps with the k-th misplaced from the right end" -- the formulation of k_ann_misplaced / k_ann_swaplist /
// k_ann_swap and of k_ann_small, with the ranks from ballots and running counts instead of a prefix sum.  Same
// permutation, same nodes, same hooks as the level loop.  Cells above ANN_MID_WAVE points are split by the four
// wavefronts together (depth first, a stack of at most seven); what falls to ANN_MID_WAVE points or below goes onto a
// list from which each wavefront then draws and finishes whole subtrees on its own, without a workgroup barrier.
// (First form, one wavefront per 2048-point cell and everything serial: 404 us of a 1M-point build, two cells per CU.)
#define ANN_MID_BLOCK 256
#define ANN_MID_NW (ANN_MID_BLOCK / WAVE)
#define ANN_MID_WAVE 256u
struct AMidCell { uint32_t cs, cn; int32_t parent; uint32_t side, depth, pad; double blo[3], bhi[3]; };
struct AMidShared {
  double X[ANN_MID], Y[ANN_MID], Z[ANN_MID];
  unsigned short PM[ANN_MID];                         // position (in the cell as it was loaded) of the point now at o
  unsigned short slotL[ANN_MID / 2], slotR[ANN_MID / 2];
  double red[ANN_MID_NW][6];
  uint32_t cnt[ANN_MID_NW][4];
  AMidCell cstack[ANN_MID / (ANN_MID_WAVE + 1) + 2];
  AMidCell wl[ANN_MID / (ANN_SMALL + 1) + 2];         // cells of 65 .. ANN_MID_WAVE points: one wavefront each
  AMidCell wstack[ANN_MID_NW][ANN_MID_WAVE / (ANN_SMALL + 1) + 2];
  AMidCell next;
  AMidCell tmp[ANN_MID_NW][2];                        // the mid-cell children of the split just made (ann_mid_emit)
  uint32_t wl_n, wl_take, next_ok, sbase, csp;
};

// One split of the cell [cs, cs + cn) by NW wavefronts (NW == 1: the calling wavefront alone, no workgroup barrier).
// Returns cd / cv / n_lo (identical in every participating thread); the points of the cell are permuted in LDS.
template <int NW>
static __device__ __forceinline__ void ann_mid_split(AMidShared& sh, const uint32_t wv, const uint32_t lane, const uint32_t cs, const uint32_t cn,
                                                     const double* blo, const double* bhi, uint32_t& cd, double& cv, uint32_t& n_lo)
{
  auto sync = [&]() {
    if (NW > 1) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  };
  const uint32_t w = (NW > 1) ? wv : 0u, T = (uint32_t)NW * WAVE, tid = w * WAVE + lane;
  const unsigned long long below = (1ull << lane) - 1ull;
  // ---- point min / max of the cell (annMinMax per dimension: independent of the order)
  double mn[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, mx[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL};
  for (uint32_t o = tid; o < cn; o += T) {
    const double v[3] = {sh.X[cs + o], sh.Y[cs + o], sh.Z[cs + o]};
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = (v[d] < mn[d]) ? v[d] : mn[d]; mx[d] = (mx[d] < v[d]) ? v[d] : mx[d]; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1)
#pragma unroll
    for (int d = 0; d < 3; d++) {
      double t;
      t = __shfl_xor(mn[d], off, WAVE); mn[d] = (t < mn[d]) ? t : mn[d];
      t = __shfl_xor(mx[d], off, WAVE); mx[d] = (mx[d] < t) ? t : mx[d];
    }
  if (NW > 1) {
    if (lane == 0) { for (int d = 0; d < 3; d++) { sh.red[w][d] = mn[d]; sh.red[w][3 + d] = mx[d]; } }
    __syncthreads();
#pragma unroll
    for (int d = 0; d < 3; d++) { mn[d] = sh.red[0][d]; mx[d] = sh.red[0][3 + d]; }
    for (int v = 1; v < NW; v++)
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const double a = sh.red[v][d], b = sh.red[v][3 + d];
        mn[d] = (a < mn[d]) ? a : mn[d]; mx[d] = (mx[d] < b) ? b : mx[d];
      }
  }
  uint32_t mode;
  sl_midpt_rule(blo, bhi, mn, mx, cd, cv, mode);
  const double* C = (cd == 0) ? sh.X : ((cd == 1) ? sh.Y : sh.Z);       // (the passes below swap through X / Y / Z: no restrict)
  // ---- the breaks: br1 = points below the plane, br2 - br1 = points on it
  uint32_t br1 = 0, neq = 0;
  for (uint32_t o0 = w * WAVE; o0 < cn; o0 += T) {
    const uint32_t o = o0 + lane;
    const double c = (o < cn) ? C[cs + o] : HUGE_VAL;
    br1 += (uint32_t)__popcll(__ballot(o < cn && c < cv));
    neq += (uint32_t)__popcll(__ballot(o < cn && c == cv));
  }
  if (NW > 1) {
    if (lane == 0) { sh.cnt[w][0] = br1; sh.cnt[w][1] = neq; }
    __syncthreads();
    br1 = 0; neq = 0;
    for (int v = 0; v < NW; v++) { br1 += sh.cnt[v][0]; neq += sh.cnt[v][1]; }
    __syncthreads();                     // (cnt is written again below)
  }
  const uint32_t br2 = br1 + neq;
  // ---- annPlaneSplit: pass 1 on [0, cn) around br1 with "< cv", pass 2 on [br1, cn) around br2 with "<= cv"
  for (int pass = 1; pass <= 2; pass++) {
    if (pass == 2 && neq == 0u) break;      // no point ON the plane: the second pass has nothing to move (and four barriers less)
    const uint32_t lo = (pass == 1) ? 0u : br1, brk = (pass == 1) ? br1 : br2;
    // every wavefront takes a contiguous share of the left part (in position order) and of the right part (from the END)
    const uint32_t lenL = brk - lo, lenR = cn - brk;
    const uint32_t shL = (NW > 1) ? (((lenL + NW - 1) / NW + WAVE - 1) / WAVE) * WAVE : lenL;
    const uint32_t shR = (NW > 1) ? (((lenR + NW - 1) / NW + WAVE - 1) / WAVE) * WAVE : lenR;
    const uint32_t l0 = min(w * shL, lenL), l1 = min(l0 + shL, lenL), r0 = min(w * shR, lenR), r1 = min(r0 + shR, lenR);
    uint32_t baseL = 0, baseR = 0;
    if (NW > 1) {
      uint32_t cL = 0, cR = 0;
      for (uint32_t o0 = l0; o0 < l1; o0 += WAVE) {
        const uint32_t o = lo + o0 + lane;
        bool mis = false;
        if (o0 + lane < l1) { const double c = C[cs + o]; mis = (pass == 1) ? !(c < cv) : !(c <= cv); }
        cL += (uint32_t)__popcll(__ballot(mis));
      }
      for (uint32_t t0 = r0; t0 < r1; t0 += WAVE) {
        const uint32_t t = t0 + lane;
        bool mis = false;
        if (t < r1) { const double c = C[cs + cn - 1u - t]; mis = (pass == 1) ? (c < cv) : (c <= cv); }
        cR += (uint32_t)__popcll(__ballot(mis));
      }
      if (lane == 0) { sh.cnt[w][2] = cL; sh.cnt[w][3] = cR; }
      __syncthreads();
      for (uint32_t v = 0; v < w; v++) { baseL += sh.cnt[v][2]; baseR += sh.cnt[v][3]; }
    }
    // misplaced on the left of the break, in position order
    uint32_t nL = baseL;
    for (uint32_t o0 = l0; o0 < l1; o0 += WAVE) {
      const uint32_t o = lo + o0 + lane;
      bool mis = false;
      if (o0 + lane < l1) { const double c = C[cs + o]; mis = (pass == 1) ? !(c < cv) : !(c <= cv); }
      const unsigned long long b = __ballot(mis);
      if (mis) sh.slotL[(cs >> 1) + nL + (uint32_t)__popcll(b & below)] = (unsigned short)o;
      nL += (uint32_t)__popcll(b);
    }
    // misplaced on the right of the break, counted from the right END of the cell
    uint32_t nR = baseR;
    for (uint32_t t0 = r0; t0 < r1; t0 += WAVE) {
      const uint32_t t = t0 + lane;                // distance from the last position
      bool mis = false;
      uint32_t o = 0;
      if (t < r1) { o = cn - 1u - t; const double c = C[cs + o]; mis = (pass == 1) ? (c < cv) : (c <= cv); }
      const unsigned long long b = __ballot(mis);
      if (mis) sh.slotR[(cs >> 1) + nR + (uint32_t)__popcll(b & below)] = (unsigned short)o;
      nR += (uint32_t)__popcll(b);
    }
    uint32_t nswap = nL;                 // (as many points of the left part are on the wrong side as of the right part)
    if (NW > 1) {
      __syncthreads();                   // lists complete (and every wave has read the counts)
      nswap = 0;
      for (int v = 0; v < NW; v++) nswap += sh.cnt[v][2];
    } else sync();
    for (uint32_t k = tid; k < nswap; k += T) {
      const uint32_t a = cs + sh.slotL[(cs >> 1) + k], b = cs + sh.slotR[(cs >> 1) + k];
      double t;
      t = sh.X[a]; sh.X[a] = sh.X[b]; sh.X[b] = t;
      t = sh.Y[a]; sh.Y[a] = sh.Y[b]; sh.Y[b] = t;
      t = sh.Z[a]; sh.Z[a] = sh.Z[b]; sh.Z[b] = t;
      const unsigned short u = sh.PM[a]; sh.PM[a] = sh.PM[b]; sh.PM[b] = u;
    }
    sync();
  }
  n_lo = sl_midpt_nlo(mode, cn, br1, br2);
}

// what ONE thread does behind a split: the splitting node, its hook, the leaf children, the children that are small cells;
// a child that is still a mid cell is written to out0 (low child) / out1 (high child) -- LDS -- and its size returned in
// n0 / n1 (0: that child needs nothing more here).  No local arrays, no run-time indexed structs: the first form kept the
// children in a private array, which the compiler put into scratch memory -- 304 bytes per lane, a hundred dependent trips
// to it per mid cell on the one thread everybody waits for, and most of the kernel's 230 us.
static __device__ __forceinline__ void ann_mid_emit(const AMidCell& c, const uint32_t S, const uint32_t cd, const double cv, const uint32_t n_lo,
                                                    AnnNode* __restrict__ nodes, ASeg* __restrict__ small_list, const uint32_t sbase,
                                                    uint32_t* __restrict__ small, AMidCell* out0, AMidCell* out1, uint32_t& n0, uint32_t& n1,
                                                    uint32_t& maxdepth)
{
  const uint32_t me = S + c.cs + n_lo - 1u;
  uint32_t c0 = cd << 30, c1 = 0;
  n0 = 0; n1 = 0;
#pragma unroll
  for (uint32_t sd = 0; sd < 2; sd++) {
    const uint32_t ccs = sd ? c.cs + n_lo : c.cs, ccn = sd ? c.cn - n_lo : n_lo;
    if (ccn == 1) {
      if (sd) c1 = A_LEAF | (S + ccs); else c0 |= A_LEAF | (S + ccs);
      maxdepth = (c.depth + 1u > maxdepth) ? c.depth + 1u : maxdepth;
      continue;
    }
    // the child's cell (kd_tree.cpp:346-357): the parent's with one face moved to the cutting plane
    const double b0 = (sd && cd == 0) ? cv : c.blo[0], b1 = (sd && cd == 1) ? cv : c.blo[1], b2 = (sd && cd == 2) ? cv : c.blo[2];
    const double h0 = (!sd && cd == 0) ? cv : c.bhi[0], h1 = (!sd && cd == 1) ? cv : c.bhi[1], h2 = (!sd && cd == 2) ? cv : c.bhi[2];
    if (ccn <= ANN_SMALL) {
      // (a small cell has at least two points, so (start within the mid cell) / 2 is a slot of its own in the stretch of
      // the list this workgroup reserved: no atomic per small cell -- a quarter of a million of them on ONE counter were
      // 3 ms of a 10M-point build, whatever the rest of the kernel did)
      ASeg* o = small_list + sbase + (ccs >> 1);
      o->start = S + ccs; o->n = ccn; o->parent = (int32_t)me; o->side = sd; o->depth = c.depth + 1u; o->pad = 0;
      o->blo[0] = b0; o->blo[1] = b1; o->blo[2] = b2; o->bhi[0] = h0; o->bhi[1] = h1; o->bhi[2] = h2;
    } else {
      AMidCell* o = sd ? out1 : out0;
      o->cs = ccs; o->cn = ccn; o->parent = (int32_t)me; o->side = sd; o->depth = c.depth + 1u; o->pad = cd | 4u;   // pad: the parent's cutting dimension (| 4: known)
      o->blo[0] = b0; o->blo[1] = b1; o->blo[2] = b2; o->bhi[0] = h0; o->bhi[1] = h1; o->bhi[2] = h2;
      if (sd) n1 = ccn; else n0 = ccn;
    }
  }
  AnnNode* nd = nodes + me;      // children that are cells hook themselves in later (this workgroup further down, or k_ann_small)
  nd->cut_val = cv; nd->lo = sel3(c.blo, cd); nd->hi = sel3(c.bhi, cd); nd->c0 = c0; nd->c1 = c1;
  // the hook into the parent: a plain store where the parent's cutting dimension came down with the cell (ann_hook reads the
  // parent's word back first)
  if (c.pad & 4u) { if (c.side) nodes[c.parent].c1 = me; else nodes[c.parent].c0 = ((c.pad & 3u) << 30) | me; }
  else ann_hook(nodes, small + 0, c.parent, c.side, me);
}

__global__ void __launch_bounds__(ANN_MID_BLOCK) k_ann_mid(const ASeg* __restrict__ mid_list, uint32_t nmid, uint32_t* __restrict__ perm,
                                                           double* __restrict__ cx, double* __restrict__ cy, double* __restrict__ cz,
                                                           AnnNode* __restrict__ nodes, ASeg* __restrict__ small_list, uint32_t* __restrict__ small)
{
  __shared__ AMidShared sh;
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  if (blockIdx.x >= nmid) return;
  const ASeg sg = mid_list[blockIdx.x];
  const uint32_t S = sg.start, N = sg.n;
  for (uint32_t o = tid; o < N; o += ANN_MID_BLOCK) { sh.X[o] = cx[S + o]; sh.Y[o] = cy[S + o]; sh.Z[o] = cz[S + o]; sh.PM[o] = (unsigned short)o; }
  // this cell's stretch of the small-cell list: N / 2 slots (see ann_mid_emit), reserved with ONE atomic, emptied (n = 0: nothing
  // here) before anything is put into it
  if (tid == 0) {
    sh.wl_n = 0; sh.wl_take = 0; sh.next_ok = 1; sh.csp = 0; sh.sbase = atomicAdd(small + 4, N >> 1);
    AMidCell* r = (N <= ANN_MID_WAVE) ? &sh.wl[0] : &sh.next;
    r->cs = 0; r->cn = N; r->parent = sg.parent; r->side = sg.side; r->depth = sg.depth; r->pad = 0;
    r->blo[0] = sg.blo[0]; r->blo[1] = sg.blo[1]; r->blo[2] = sg.blo[2]; r->bhi[0] = sg.bhi[0]; r->bhi[1] = sg.bhi[1]; r->bhi[2] = sg.bhi[2];
    if (N <= ANN_MID_WAVE) { sh.wl_n = 1; sh.next_ok = 0; }
  }
  __syncthreads();
  const uint32_t sbase = sh.sbase;
  for (uint32_t k = tid; k < (N >> 1); k += ANN_MID_BLOCK) small_list[sbase + k].n = 0u;
  __threadfence_block();
  uint32_t maxdepth = 0;
  bool failed = false;
  // ---- phase 1: cells above ANN_MID_WAVE points, all four wavefronts on one cell, depth first (sh.next: the cell in hand)
  while (sh.next_ok != 0) {
    const AMidCell cur = sh.next;
    __syncthreads();          // (everybody has the cell: thread 0 rewrites sh.next below)
    uint32_t cd, n_lo;
    double cv;
    ann_mid_split<ANN_MID_NW>(sh, wv, lane, cur.cs, cur.cn, cur.blo, cur.bhi, cd, cv, n_lo);
    if (n_lo == 0 || n_lo >= cur.cn) { failed = true; break; }     // cannot happen for finite input (uniform: every thread leaves)
    if (tid == 0) {
      uint32_t n0, n1;
      ann_mid_emit(cur, S, cd, cv, n_lo, nodes, small_list, sbase, small, &sh.tmp[0][0], &sh.tmp[0][1], n0, n1, maxdepth);
      // children of at most ANN_MID_WAVE points wait for phase 2; of the others the low one is next, the high one is stacked
      const bool coop0 = n0 > ANN_MID_WAVE, coop1 = n1 > ANN_MID_WAVE;
      if (n0 && !coop0) sh.wl[sh.wl_n++] = sh.tmp[0][0];
      if (n1 && !coop1) sh.wl[sh.wl_n++] = sh.tmp[0][1];
      if (coop0 && coop1) sh.cstack[sh.csp++] = sh.tmp[0][1];
      if (coop0) sh.next = sh.tmp[0][0];
      else if (coop1) sh.next = sh.tmp[0][1];
      else if (sh.csp > 0) sh.next = sh.cstack[--sh.csp];
      else sh.next_ok = 0;
    }
    __syncthreads();
  }
  __syncthreads();
  // ---- phase 2: every wavefront draws cells of 65 .. ANN_MID_WAVE points and finishes their subtrees on its own
  const uint32_t wl_n = sh.wl_n;
  while (!failed) {
    uint32_t take = 0;
    if (lane == 0) take = atomicAdd(&sh.wl_take, 1u);
    take = (uint32_t)__builtin_amdgcn_readfirstlane((int)take);
    if (take >= wl_n) break;
    AMidCell c = sh.wl[take];
    uint32_t sp = 0;
    for (;;) {
      uint32_t cd, n_lo;
      double cv;
      ann_mid_split<1>(sh, wv, lane, c.cs, c.cn, c.blo, c.bhi, cd, cv, n_lo);
      if (n_lo == 0 || n_lo >= c.cn) { failed = true; break; }
      uint32_t n0 = 0, n1 = 0;
      if (lane == 0) ann_mid_emit(c, S, cd, cv, n_lo, nodes, small_list, sbase, small, &sh.tmp[wv][0], &sh.tmp[wv][1], n0, n1, maxdepth);
      n0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n0); n1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)n1);
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // the low child next, the high one stacked (at most ANN_MID_WAVE / 65 of them wait at any time)
      if (n0 && n1) { if (lane == 0) sh.wstack[wv][sp] = sh.tmp[wv][1]; ++sp; c = sh.tmp[wv][0]; }
      else if (n0) c = sh.tmp[wv][0];
      else if (n1) c = sh.tmp[wv][1];
      else { if (sp == 0) break; --sp; c = sh.wstack[wv][sp]; }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (failed && lane == 0) atomicExch(small + 2, 1u);
  __syncthreads();
  // the points in their final order within the cell's range (the small cells below continue from global memory); the
  // original indices are gathered from the cell's own stretch of perm -- all of it read before any of it is written
  uint32_t pv[ANN_MID / ANN_MID_BLOCK];
#pragma unroll
  for (uint32_t k = 0; k < ANN_MID / ANN_MID_BLOCK; k++) { const uint32_t o = tid + k * ANN_MID_BLOCK; pv[k] = (o < N) ? perm[S + sh.PM[o]] : 0u; }
  __syncthreads();
#pragma unroll
  for (uint32_t k = 0; k < ANN_MID / ANN_MID_BLOCK; k++) {
    const uint32_t o = tid + k * ANN_MID_BLOCK;
    if (o < N) { cx[S + o] = sh.X[o]; cy[S + o] = sh.Y[o]; cz[S + o] = sh.Z[o]; perm[S + o] = pv[k]; }
  }
  if (lane == 0 && maxdepth > *(volatile uint32_t*)(small + 1)) atomicMax(small + 1, maxdepth);
}

__global__ void k_ann_points(const uint32_t* __restrict__ perm, const double* __restrict__ cx,
                             const double* __restrict__ cy, const double* __restrict__ cz, uint32_t M,
                             KdPoint* __restrict__ pts)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  KdPoint q;
  q.x = cx[p]; q.y = cy[p]; q.z = cz[p]; q.orig = (int32_t)perm[p]; q.pad = 0;
  pts[p] = q;
}

static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }

#define ACHK(expr)                                       \
  do {                                                   \
    hipError_t _e = (expr);                              \
    if (_e != hipSuccess) { res.err = _e; return res; }  \
  } while (0)

// (lab library: 256 guard bytes behind every region of the arena, as in build.hip)
#ifdef TDTK_LAB
constexpr size_t ANN_GUARD = 256;
static thread_local std::vector<size_t>* g_ann_guard_sink = nullptr;
struct AnnGuardList { uint32_t n; uint32_t pad; size_t off[60]; };
__global__ void __launch_bounds__(64) k_ann_guard_fill(char* arena, AnnGuardList G)
{
  if (blockIdx.x < G.n) reinterpret_cast<uint32_t*>(arena + G.off[blockIdx.x])[threadIdx.x] = 0x5AC35AC3u ^ blockIdx.x;
}
__global__ void __launch_bounds__(64) k_ann_guard_check(const char* arena, AnnGuardList G, uint32_t* bad)
{
  if (blockIdx.x < G.n && reinterpret_cast<const uint32_t*>(arena + G.off[blockIdx.x])[threadIdx.x] != (0x5AC35AC3u ^ blockIdx.x))
    atomicMax(bad, blockIdx.x + 1u);
}
#else
constexpr size_t ANN_GUARD = 0;
#endif
static size_t ann_layout(size_t M, size_t* O, size_t* scan_tmp_out)
{
  size_t scan_tmp = 0;
  {
    unsigned long long* z8 = nullptr;
    (void)rocprim::exclusive_scan(nullptr, scan_tmp, z8, z8, 0ull, M + 1, rocprim::plus<unsigned long long>(), (hipStream_t)0);
  }
  const size_t n1 = M + 1, nlarge = M / (ANN_SMALL + 1) + 2, nsmall = M / 2 + 2;
  size_t off = 0;
  int k = 0;
  auto take = [&](size_t bytes) {
    size_t o = off; off += (bytes + 255) & ~(size_t)255;
#ifdef TDTK_LAB
    if (g_ann_guard_sink) g_ann_guard_sink->push_back(off);
#endif
    off += ANN_GUARD;
    if (O) O[k] = o; k++; return o; };
  take(4 * n1); take(4 * n1); take(8 * n1); take(8 * n1); take(8 * n1);      // 0 perm 1 segof 2 cx 3 cy 4 cz
  take(256); take(256); take(8 * n1); take(8 * n1);                          // 5, 6 spare 7 LR 8 AB
  take(4 * n1); take(4 * n1);                                                // 9 posL 10 posR
  take(sizeof(ASeg) * nlarge); take(sizeof(ASeg) * nlarge); take(sizeof(ASeg) * nsmall);   // 11 segA 12 segB 13 small cells
  take(sizeof(AMeasU) * nlarge); take(sizeof(ADec) * nlarge); take(8 * nlarge); take(8 * nlarge);   // 14 meas 15 dec 16 counts 17 counts of the next level
  take(scan_tmp + 256); take(256);                                           // 18 tmp 19 small
  take(bbox_temp_bytes() + 256); take(256);                                  // 20 bbox partials 21 box
  take(sizeof(AMeasU) * nlarge);                                             // 22 meas of the next level
  take(8 * (ANN_MAX_LEVELS + 2));                                            // 23 cells per level; behind them the levels' "points on a plane" flags
  take(std::max(scan_pair27_state_bytes(n1), part_state_bytes(n1)));         // 24 state of the one-launch scans (sort.hip's, k_ann_part_scan's)
  take(sizeof(ASeg) * nlarge);                                               // 25 mid cells (k_ann_mid)
  if (scan_tmp_out) *scan_tmp_out = scan_tmp;
  return off;
}
size_t ann_build_arena_bytes(size_t M) { return ann_layout(M, nullptr, nullptr); }

// Builds on `s` inside the caller's scratch.  nodes (M-1 records, may be null for M == 1), pts (M records) and
// bb (6 doubles) are caller-owned device memory and are final when this returns.
AnnBuildResult ann_build_tree(const double* d_xyz, size_t M_, void* arena_, AnnNode* nodes, KdPoint* pts,
                              double* bb, hipStream_t s)
{
  AnnBuildResult res{};
  const uint32_t M = (uint32_t)M_;
  char* arena = static_cast<char*>(arena_);
  size_t scan_tmp = 0;
  size_t O[32];
  (void)ann_layout(M_, O, &scan_tmp);
#ifdef TDTK_LAB
  AnnGuardList guards{};
  {
    std::vector<size_t> g;
    g_ann_guard_sink = &g;
    (void)ann_layout(M_, nullptr, nullptr);
    g_ann_guard_sink = nullptr;
    guards.n = (uint32_t)std::min<size_t>(g.size(), 60);
    for (uint32_t k = 0; k < guards.n; k++) guards.off[k] = g[k];
    if (guards.n) hipLaunchKernelGGL(k_ann_guard_fill, dim3(guards.n), dim3(64), 0, s, arena, guards);
  }
#endif
  const size_t n1 = (size_t)M + 1;
  uint32_t* perm = (uint32_t*)(arena + O[0]); uint32_t* seg_of = (uint32_t*)(arena + O[1]);
  double *cx = (double*)(arena + O[2]), *cy = (double*)(arena + O[3]), *cz = (double*)(arena + O[4]);
  unsigned long long *LR = (unsigned long long*)(arena + O[7]), *AB = (unsigned long long*)(arena + O[8]);
  uint32_t *posL = (uint32_t*)(arena + O[9]), *posR = (uint32_t*)(arena + O[10]);
  ASeg *segs = (ASeg*)(arena + O[11]), *next = (ASeg*)(arena + O[12]), *small_list = (ASeg*)(arena + O[13]);
  AMeasU* meas = (AMeasU*)(arena + O[14]); AMeasU* meas_next = (AMeasU*)(arena + O[22]); ADec* dec = (ADec*)(arena + O[15]);
  unsigned long long *cnt = (unsigned long long*)(arena + O[16]), *cnt_next = (unsigned long long*)(arena + O[17]);
  void* tmp = arena + O[18];
  uint32_t* small = (uint32_t*)(arena + O[19]);
  double* partial = (double*)(arena + O[20]); double* box = (double*)(arena + O[21]);

  uint32_t* lvl = (uint32_t*)(arena + O[23]);
  ASeg* mid_list = (ASeg*)(arena + O[25]);
  // cells of at most this many points leave the level loop for k_ann_mid (TDTK_ANN_MID=0: the level loop down to 64-point cells,
  // the round-2 form)
  static const uint32_t mid_cap = [] { const char* e = getenv("TDTK_ANN_MID"); return (e && e[0] == '0') ? ANN_SMALL : ANN_MID; }();
  ACHK(hipMemsetAsync(small, 0, 256, s));
  ACHK(hipMemsetAsync(lvl, 0, 8 * (ANN_MAX_LEVELS + 2), s));
  uint32_t* const eqf = lvl + (ANN_MAX_LEVELS + 2);
  // the partition's scans in one launch each while the positions fit their 27-bit counters (TDTK_OWN_SCAN=0: rocPRIM's two)
  static const bool own_scan_env = [] { const char* e = getenv("TDTK_OWN_SCAN"); return !(e && e[0] == '0'); }();
  const bool own_scan = own_scan_env && n1 < ((size_t)1 << 27);
  if (own_scan) ACHK(hipMemsetAsync(arena + O[24], 0, std::max(scan_pair27_state_bytes(n1), part_state_bytes(n1)), s));
  hipLaunchKernelGGL(k_ann_init, dim3(cdiv(M, 256)), dim3(256), 0, s, d_xyz, M, perm, seg_of, cx, cy, cz, small + 2, mid_cap);
  ACHK(launch_bbox(d_xyz, M, partial, box, s));
  hipLaunchKernelGGL(k_ann_root, dim3(1), dim3(1), 0, s, box, M, segs, small_list, small, bb, meas, cnt, lvl, mid_list, mid_cap);
  // Levels are enqueued without waiting for their cell counts (those stay on the device: lvl[]); the host looks
  // at them only after a batch -- the first batch is as deep as a balanced tree gets down to 64-point cells, then
  // two levels at a time.  A level enqueued past the last one finds no cell and costs a few empty launches.
  uint32_t level = 0, more = (M > mid_cap) ? 1u : 0u;
  uint32_t batch = 1;
  for (size_t c = mid_cap; c < M_; c <<= 1) batch++;
  const size_t nlarge = M_ / (ANN_SMALL + 1) + 2;
  while (more) {
    for (uint32_t b = 0; b < batch && level < ANN_MAX_LEVELS; b++, level++) {
      const size_t bound = (level < 31 && ((size_t)1 << level) < nlarge) ? ((size_t)1 << level) : nlarge;
      hipLaunchKernelGGL(k_ann_measure, dim3(cdiv(M, 256 * MEAS_ITERS)), dim3(256), 0, s, seg_of, M, cx, cy, cz, meas,
                         level ? (const ASeg*)next : (const ASeg*)nullptr, level ? (const ADec*)dec : (const ADec*)nullptr);
      hipLaunchKernelGGL(k_ann_decide, dim3(cdiv(bound, 256)), dim3(256), 0, s, segs, lvl + level, meas, dec);
      // (the cells' counts: inside the first pass's scan when that is the one-launch one -- k_ann_part_scan<1> --, else a pass of their own)
      const bool part2_level = [] { const char* e = lab_env("TDTK_ANN_PART"); return !(e && e[0] == '0'); }() && own_scan && 2u * level + 2u < 255u;
      if (!part2_level) hipLaunchKernelGGL(k_ann_count, dim3(cdiv(M, 256 * MEAS_ITERS)), dim3(256), 0, s, seg_of, M, dec, cx, cy, cz, cnt);
      // the library's first Hoare pass, then its second one on what lies right of br1
      for (int pass = 1; pass <= 2; pass++) {
        // (the gate needs every kernel of the pass to honour it: with rocPRIM's scan, which does not, the pass runs ungated)
        const bool one_launch = own_scan && 2u * level + (uint32_t)pass < 255u;
        if (part2_level) {      // (lab, TDTK_ANN_PART=0: round 3's four launches)
          // a Hoare pass = a scan that writes the index list + the swaps
          const uint32_t ntiles = cdiv(M, PS_TILE), nbw = cdiv(M, 256u * APW_ROWS);
          uint32_t* counter = reinterpret_cast<uint32_t*>(arena + O[24]);
          unsigned long long* status = reinterpret_cast<unsigned long long*>(arena + O[24] + 64);
          uint32_t* const flag = eqf + level;
          if (pass == 1) {
            hipLaunchKernelGGL(k_ann_part_scan<1>, dim3(ntiles), dim3(PS_THREADS), 0, s, seg_of, segs, dec, cnt, cx, cy, cz, M, posL, status, counter,
                               2u * level + 1u, ntiles, small + 2, (2u * level + 2u < 255u) ? flag : (uint32_t*)nullptr);
            hipLaunchKernelGGL(k_ann_part_swap<1>, dim3(nbw), dim3(256), 0, s, posL, seg_of, segs, cnt, M, perm, cx, cy, cz, (const uint32_t*)nullptr);
          } else {
            hipLaunchKernelGGL(k_ann_part_scan<2>, dim3(ntiles), dim3(PS_THREADS), 0, s, seg_of, segs, dec, cnt, cx, cy, cz, M, posL, status, counter,
                               2u * level + 2u, ntiles, small + 2, flag);
            hipLaunchKernelGGL(k_ann_part_swap<2>, dim3(nbw), dim3(256), 0, s, posL, seg_of, segs, cnt, M, perm, cx, cy, cz, (const uint32_t*)flag);
          }
          continue;
        }
        uint32_t* const flag = one_launch ? eqf + level : nullptr;
        const uint32_t* const gate = (pass == 2) ? flag : nullptr;
        if (pass == 1)
          hipLaunchKernelGGL(k_ann_misplaced<1>, dim3(cdiv(n1, 256)), dim3(256), 0, s, seg_of, segs, dec, cnt, cx, cy, cz, M, LR,
                             (own_scan && 2u * level + 2u < 255u) ? eqf + level : (uint32_t*)nullptr);
        else
          hipLaunchKernelGGL(k_ann_misplaced<2>, dim3(cdiv(n1, 256)), dim3(256), 0, s, seg_of, segs, dec, cnt, cx, cy, cz, M, LR, flag);
        if (one_launch) {
          ACHK(launch_scan_pair27(LR, AB, n1, arena + O[24], 2u * level + (uint32_t)pass, small + 2, s, gate));
        } else {
          size_t st = scan_tmp;
          ACHK(rocprim::exclusive_scan(tmp, st, LR, AB, 0ull, n1, rocprim::plus<unsigned long long>(), s));
        }
        hipLaunchKernelGGL(k_ann_swaplist, dim3(cdiv(M, 256)), dim3(256), 0, s, seg_of, segs, LR, AB, M, posL, posR, gate);
        hipLaunchKernelGGL(k_ann_swap, dim3(cdiv((size_t)M / 2 + 1, 256)), dim3(256), 0, s, posL, posR, AB + M, perm, cx, cy, cz, gate);
      }
      hipLaunchKernelGGL(k_ann_children, dim3(cdiv(bound, 256)), dim3(256), 0, s, segs, lvl + level, dec, cnt, nodes, next,
                         small_list, small, meas_next, cnt_next, mid_list, mid_cap);
      ASeg* t = segs; segs = next; next = t;
      AMeasU* tm = meas; meas = meas_next; meas_next = tm;
      unsigned long long* tc = cnt; cnt = cnt_next; cnt_next = tc;
    }
    uint32_t bad = 0;
    ACHK(hipMemcpyAsync(&bad, small + 2, 4, hipMemcpyDeviceToHost, s));
    ACHK(hipMemcpyAsync(&more, lvl + level, 4, hipMemcpyDeviceToHost, s));
    ACHK(hipStreamSynchronize(s));
    if (bad & 0x10000u) { res.err = hipErrorLaunchTimeOut; return res; }     // the one-launch scan gave up waiting (sort.hip): not an input problem
    if (bad || (more && level >= ANN_MAX_LEVELS)) { res.err = hipErrorInvalidValue; res.degenerate = true; return res; }
    batch = 2;
  }
  if (M <= mid_cap) {       // no level ran: the finiteness flag of k_ann_init has not been looked at yet
    uint32_t bad = 0;
    ACHK(hipMemcpyAsync(&bad, small + 2, 4, hipMemcpyDeviceToHost, s));
    ACHK(hipStreamSynchronize(s));
    if (bad) { res.err = hipErrorInvalidValue; res.degenerate = true; return res; }
  }
  uint32_t h_small[6];
  ACHK(hipMemcpyAsync(h_small, small, sizeof h_small, hipMemcpyDeviceToHost, s));
  ACHK(hipStreamSynchronize(s));
  if (h_small[5]) {
    // the mid cells first: they add to the list of small cells (whose length the next launch needs on the host)
    hipLaunchKernelGGL(k_ann_mid, dim3(h_small[5]), dim3(ANN_MID_BLOCK), 0, s, mid_list, h_small[5], perm, cx, cy, cz, nodes, small_list, small);
    ACHK(hipMemcpyAsync(h_small, small, sizeof h_small, hipMemcpyDeviceToHost, s));
    ACHK(hipStreamSynchronize(s));
    if (h_small[2]) { res.err = hipErrorInvalidValue; res.degenerate = true; return res; }
  }
  if (h_small[4])
    hipLaunchKernelGGL(k_ann_small, dim3(cdiv((size_t)h_small[4] * WAVE, 256)), dim3(256), 0, s, small_list, h_small[4], perm, cx, cy, cz,
                       nodes, small);
  hipLaunchKernelGGL(k_ann_points, dim3(cdiv(M, 256)), dim3(256), 0, s, perm, cx, cy, cz, M, pts);
  ACHK(hipMemcpyAsync(h_small, small, sizeof h_small, hipMemcpyDeviceToHost, s));
  ACHK(hipStreamSynchronize(s));
  ACHK(hipGetLastError());
  if (h_small[2]) { res.err = hipErrorInvalidValue; res.degenerate = true; return res; }
#ifdef TDTK_LAB
  if (guards.n) {
    uint32_t* bad = small + 40;      // (a word of `small` nothing else uses)
    uint32_t h_bad = 0;
    (void)hipMemsetAsync(bad, 0, 4, s);
    hipLaunchKernelGGL(k_ann_guard_check, dim3(guards.n), dim3(64), 0, s, arena, guards, bad);
    (void)hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    if (h_bad) {
      fprintf(stderr, "ANN tree build: arena guard %u of %u overwritten (M = %u)\n", h_bad - 1u, guards.n, M);
      res.err = hipErrorAssert;
      return res;
    }
  }
#endif
  res.root_ref = h_small[0];
  res.max_depth = h_small[1];
  res.levels = level;
  return res;
}

// ---- search + PCA -------------------------------------------------------------------------------
// newmat's EigenValues on a symmetric 3x3 (evalue.cpp:24-156, 283-284; sort.cpp:190-222): tred2, tql2, ascending
// sort.  z holds the matrix (lower triangle used) on entry and the eigenvectors (columns) on exit.
static __device__ __forceinline__ double nm_sign(double x, double y) { return (y >= 0) ? x : -x; }

static __device__ void eigen3_newmat(double z[3][3], double D[3])
{
  double E[3];
  const double tol = DBL_MIN / DBL_EPSILON;
  // tred2, n = 3
#pragma unroll
  for (int i = 2; i > 0; i--) {
    double f = z[i][i - 1], g = 0.0;
#pragma unroll
    for (int k = 0; k < i - 1; k++) g += z[i][k] * z[i][k];
    double h = g + f * f;
    if (g <= tol) { E[i] = f; h = 0.0; }
    else {
      g = nm_sign(-__dsqrt_rn(h), f); E[i] = g; h -= f * g;
      z[i][i - 1] = f - g; f = 0.0;
#pragma unroll
      for (int j = 0; j < i; j++) {
        z[j][i] = z[i][j] / h; g = 0.0;
#pragma unroll
        for (int k = 0; k < j; k++) g += z[j][k] * z[i][k];
#pragma unroll
        for (int k = j; k < i; k++) g += z[k][j] * z[i][k];
        E[j] = g / h; f += g * z[j][i];
      }
      const double hh = f / (h + h);
#pragma unroll
      for (int j = 0; j < i; j++) {
        f = z[i][j]; g = E[j] - hh * f; E[j] = g;
#pragma unroll
        for (int k = 0; k <= j; k++) z[j][k] -= (f * E[k] + g * z[i][k]);
      }
    }
    D[i] = h;
  }
  D[0] = 0.0; E[0] = 0.0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (D[i] != 0.0) {
#pragma unroll
      for (int j = 0; j < i; j++) {
        double g = 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) g += z[i][k] * z[k][j];
#pragma unroll
        for (int k = 0; k < i; k++) z[k][j] -= g * z[k][i];
      }
    }
#pragma unroll
    for (int j = 0; j < i; j++) { z[i][j] = 0.0; z[j][i] = 0.0; }
    D[i] = z[i][i]; z[i][i] = 1.0;
  }
  // tql2, n = 3
  const double eps = DBL_EPSILON;
  E[0] = E[1]; E[1] = E[2];
  double b = 0.0, f = 0.0;
  E[2] = 0.0;
#pragma unroll
  for (int l = 0; l < 3; l++) {
    double h = eps * (fabs(D[l]) + fabs(E[l]));
    if (b < h) b = h;
    int m = 3;
#pragma unroll
    for (int mm = 2; mm >= 0; mm--)
      if (mm >= l && fabs(E[mm]) <= b) m = mm;     // first m >= l with |E[m]| <= b (E[2] == 0 ends it)
    for (int j = 0; j < 30; j++) {
      if (m == l) break;
      double g = D[l];
      const double dl1 = (l == 0) ? D[1] : D[2];   // l < m <= 2 here
      double p = (dl1 - g) / (2.0 * E[l]), r = __dsqrt_rn(p * p + 1.0);
      D[l] = E[l] / (p < 0.0 ? p - r : p + r);
      const double hh = g - D[l];
      f += hh;
#pragma unroll
      for (int i = 1; i < 3; i++) if (i > l) D[i] -= hh;
      p = (m == 1) ? D[1] : D[2];
      double c = 1.0, s = 0.0;
#pragma unroll
      for (int i = 1; i >= 0; i--) {
        if (i <= m - 1 && i >= l) {
          const double ei = E[i], di = D[i];
          g = c * ei; h = c * p;
          if (fabs(p) >= fabs(ei)) {
            c = ei / p; r = __dsqrt_rn(c * c + 1.0);
            E[i + 1] = s * p * r; s = c / r; c = 1.0 / r;
          } else {
            c = p / ei; r = __dsqrt_rn(c * c + 1.0);
            E[i + 1] = s * ei * r; s = 1.0 / r; c /= r;
          }
          p = c * di - s * g; D[i + 1] = h + s * (c * g + s * di);
#pragma unroll
          for (int k = 0; k < 3; k++) {
            h = z[k][i + 1];
            z[k][i + 1] = s * z[k][i] + c * h;
            z[k][i] = c * z[k][i] - s * h;
          }
        }
      }
      E[l] = s * p; D[l] = c * p;
      if (fabs(E[l]) <= b) break;
    }
    D[l] += f;     // (30 sweeps without convergence throw in the reference; unreachable for a 3x3)
  }
  // SortSV ascending: selection sort, columns follow
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int k = i;
    double p = D[i];
#pragma unroll
    for (int j = i + 1; j < 3; j++) if (D[j] < p) { k = j; p = D[j]; }
    if (k != i) {
#pragma unroll
      for (int kk = 1; kk < 3; kk++)
        if (kk == k) {
          D[kk] = D[i]; D[i] = p;
#pragma unroll
          for (int j = 0; j < 3; j++) { const double t = z[j][i]; z[j][i] = z[j][kk]; z[j][kk] = t; }
        }
    }
  }
}

#define ANN_LDS_STACK 12   // far-child entries per thread kept in LDS; deeper ones spill to global memory

// One thread per scan point (grid-stride over leaf positions).  KMAX >= k; the list keeps KMAX - k sentinels
// (key -1) in front so that every register index is static.
// COUNT: also tally the splitting nodes and leaf points every query visits (cnt[0], cnt[1]) -- the instrumented
// instantiation bench.py takes the algorithmic bytes of the launch from.
template <int KMAX, bool COUNT>
__global__ void __launch_bounds__(256) k_ann_normals(const AnnNode* __restrict__ nodes, uint32_t root_ref,
                                                     const KdPoint* __restrict__ pts, uint32_t n, int k,
                                                     double max_err, const double* __restrict__ bb, double rx,
                                                     double ry, double rz, uint32_t* __restrict__ spill_ref,
                                                     double* __restrict__ spill_bd, uint32_t spill_depth,
                                                     double* __restrict__ normals, int32_t* __restrict__ knn_out,
                                                     unsigned long long* __restrict__ cnt)
{
  unsigned c_split = 0, c_leaf = 0;
  __shared__ uint32_t s_ref[ANN_LDS_STACK][256];
  __shared__ double s_bd[ANN_LDS_STACK][256];
  const uint32_t T = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x, tx = threadIdx.x;
  (void)spill_depth;
  const double blo[3] = {bb[0], bb[1], bb[2]}, bhi[3] = {bb[3], bb[4], bb[5]};
  for (uint32_t qi = tid; qi < n; qi += T) {
    const KdPoint qp = pts[qi];
    const double q[3] = {qp.x, qp.y, qp.z};
    double key[KMAX];
    uint32_t info[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; j++) { key[j] = (j < KMAX - k) ? -1.0 : DBL_MAX; info[j] = 0xFFFFFFFFu; }
    // annBoxDistance (kd_util.cpp:127-150)
    double bd = 0.0;
#pragma unroll
    for (int d = 0; d < 3; d++) {
      if (q[d] < blo[d]) { const double t = blo[d] - q[d]; bd = bd + t * t; }
      else if (q[d] > bhi[d]) { const double t = q[d] - bhi[d]; bd = bd + t * t; }
    }
    uint32_t cur = root_ref;
    int sp = 0;
    for (;;) {
      while (!(cur & A_LEAF)) {                      // ANNkd_split::ann_search (kd_search.cpp:128-170)
        if (COUNT) ++c_split;
        const AnnNode nd = nodes[cur & A_VAL];
        const uint32_t cd = nd.c0 >> 30;
        const double qd = (cd == 0) ? q[0] : ((cd == 1) ? q[1] : q[2]);
        const double cut_diff = qd - nd.cut_val;
        const bool low = cut_diff < 0;
        double box_diff = low ? (nd.lo - qd) : (qd - nd.hi);
        if (box_diff < 0) box_diff = 0;
        const double fbd = bd + (cut_diff * cut_diff - box_diff * box_diff);
        const uint32_t c0 = nd.c0 & (A_LEAF | A_VAL), c1 = nd.c1;
        const uint32_t far = low ? c1 : c0;
        // the reference tests "box_dist * max_err < max_key()" after the near subtree; the k-th key only ever
        // shrinks, so a far child failing now fails then too and need not be remembered
        if (fbd * max_err < key[KMAX - 1]) {
          if (sp < ANN_LDS_STACK) { s_ref[sp][tx] = far; s_bd[sp][tx] = fbd; }
          else { spill_ref[(size_t)(sp - ANN_LDS_STACK) * T + tid] = far; spill_bd[(size_t)(sp - ANN_LDS_STACK) * T + tid] = fbd; }
          sp++;
        }
        cur = low ? c0 : c1;
      }
      {                                              // ANNkd_leaf::ann_search, one point (kd_search.cpp:177-210)
        const uint32_t pos = cur & A_VAL;
        if (COUNT) ++c_leaf;
        const KdPoint p = pts[pos];
        const double t0 = q[0] - p.x, t1 = q[1] - p.y, t2 = q[2] - p.z;
        const double dist = (t0 * t0 + t1 * t1) + t2 * t2;
        if (!(dist > key[KMAX - 1])) {               // no partial sum exceeded the k-th key
          // ANNmin_k::insert (pr_queue_k.h:100-114): behind every key <= dist, the last one drops out
          double ck = dist;
          uint32_t ci = pos;
          bool shifting = false;
#pragma unroll
          for (int j = 0; j < KMAX; j++) {
            shifting = shifting || (key[j] > ck);
            if (shifting) {
              const double tk = key[j]; key[j] = ck; ck = tk;
              const uint32_t ti = info[j]; info[j] = ci; ci = ti;
            }
          }
        }
      }
      bool done = false;
      for (;;) {
        if (sp == 0) { done = true; break; }
        sp--;
        if (sp < ANN_LDS_STACK) { cur = s_ref[sp][tx]; bd = s_bd[sp][tx]; }
        else { cur = spill_ref[(size_t)(sp - ANN_LDS_STACK) * T + tid]; bd = spill_bd[(size_t)(sp - ANN_LDS_STACK) * T + tid]; }
        if (bd * max_err < key[KMAX - 1]) break;
      }
      if (done) break;
    }
    // ---- normals.cc:64-105 ----
    double mean[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < KMAX; j++)
      if (j >= KMAX - k) {
        const KdPoint p = pts[info[j]];
        mean[0] += p.x; mean[1] += p.y; mean[2] += p.z;
        if (knn_out) knn_out[(size_t)qp.orig * k + (j - (KMAX - k))] = p.orig;
      }
    mean[0] /= k; mean[1] /= k; mean[2] /= k;
    // A << 1.0 / k * X.t() * X: (s * X^T) * X summed in list order, element (c, r), c <= r, kept
    const double sc = 1.0 / k;
    double z[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
    for (int j = 0; j < KMAX; j++)
      if (j >= KMAX - k) {
        const KdPoint p = pts[info[j]];
        const double x[3] = {p.x - mean[0], p.y - mean[1], p.z - mean[2]};
#pragma unroll
        for (int r = 0; r < 3; r++)
#pragma unroll
          for (int c = 0; c <= r; c++) z[r][c] += (sc * x[c]) * x[r];
      }
    z[0][1] = z[1][0]; z[0][2] = z[2][0]; z[1][2] = z[2][1];
    double D[3];
    eigen3_newmat(z, D);
    double nv[3] = {z[0][0], z[1][0], z[2][0]};
    double pv[3] = {q[0] - rx, q[1] - ry, q[2] - rz};
    const double pl = 1.0 / __dsqrt_rn((pv[0] * pv[0] + pv[1] * pv[1]) + pv[2] * pv[2]);   // "v / norm" is v * (1 / norm) in newmat
    pv[0] *= pl; pv[1] *= pl; pv[2] *= pl;
    const double angle = (nv[0] * pv[0] + nv[1] * pv[1]) + nv[2] * pv[2];
    if (angle < 0) { nv[0] *= -1.0; nv[1] *= -1.0; nv[2] *= -1.0; }
    const double nl = 1.0 / __dsqrt_rn((nv[0] * nv[0] + nv[1] * nv[1]) + nv[2] * nv[2]);
    normals[3 * (size_t)qp.orig] = nv[0] * nl;
    normals[3 * (size_t)qp.orig + 1] = nv[1] * nl;
    normals[3 * (size_t)qp.orig + 2] = nv[2] * nl;
  }
  if (COUNT && cnt) {
    unsigned long long a = c_split, b = c_leaf;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&cnt[0], a); atomicAdd(&cnt[1], b); }
  }
}

uint32_t ann_search_threads(size_t n)
{
  const size_t blocks = (n + 255) / 256;
  return (uint32_t)((blocks < 1024 ? blocks : 1024) * 256);
}
size_t ann_spill_entries(size_t n, uint32_t max_depth)
{
  const size_t d = (max_depth + 1 > ANN_LDS_STACK) ? (max_depth + 1 - ANN_LDS_STACK) : 0;
  return d * (size_t)ann_search_threads(n);
}

hipError_t launch_ann_normals(const AnnNode* nodes, uint32_t root_ref, const KdPoint* pts, size_t n, int k, double eps,
                              const double* d_bb, const double rPos[3], uint32_t* spill_ref, double* spill_bd,
                              uint32_t max_depth, double* d_normals, int32_t* d_knn, unsigned long long* d_cnt, hipStream_t s)
{
  const uint32_t T = ann_search_threads(n);
  const double max_err = (1.0 + eps) * (1.0 + eps);    // ANN_POW(1.0 + eps), kd_search.cpp:108
  const dim3 grid(T / 256), block(256);
#define ANN_LAUNCH(KM)                                                                                                \
  do {                                                                                                                \
    if (d_cnt)                                                                                                        \
      hipLaunchKernelGGL((k_ann_normals<KM, true>), grid, block, 0, s, nodes, root_ref, pts, (uint32_t)n, k, max_err, d_bb, \
                         rPos[0], rPos[1], rPos[2], spill_ref, spill_bd, max_depth, d_normals, d_knn, d_cnt);          \
    else                                                                                                              \
      hipLaunchKernelGGL((k_ann_normals<KM, false>), grid, block, 0, s, nodes, root_ref, pts, (uint32_t)n, k, max_err, d_bb, \
                         rPos[0], rPos[1], rPos[2], spill_ref, spill_bd, max_depth, d_normals, d_knn, d_cnt);          \
  } while (0)
  if (k <= 10) ANN_LAUNCH(10);
  else if (k <= 16) ANN_LAUNCH(16);
  else if (k <= 32) ANN_LAUNCH(32);
  else return hipErrorInvalidValue;
#undef ANN_LAUNCH
  return hipGetLastError();
}

}  // namespace tdtk
