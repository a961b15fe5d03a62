// lib3dtk_hip.so -- gfx950 kernels of the slam6D correspondence + accumulation hot path.
//
// Written for CDNA4 directly (wave64, 256 CUs in 8 XCDs, 160 KB LDS/CU); no CUDA shims, no
// dual paths.  Everything is fp64 with FMA contraction off (-ffp-contract=off) and the
// reference's association, because the deliverable is the reference's *indices*, bit for bit.
//
// Kernel inventory (DESIGN.md has the roofline of each):
//   k_search      one lane = one query: near-first DFS over the BFS-flattened tree with an
//                 explicit per-lane stack in LDS (+ HBM overflow), leaf buckets scanned as
//                 32-byte records.  Optionally applies the pending ICP transform in place
//                 first and maps the query into the tree frame.      [hot: ~all the time]
//   k_search_dir  FindClosestAlongDir variant (bounding-sphere pruning only).
//   k_accum       streams (query, hit) pairs and reduces the pair sums with wave64
//                 shuffles -> LDS -> one partial row per workgroup.
//   k_final       fixed-order reduction of the partial rows (deterministic).
//   k_transform   Scan::transformReduced on the resident scan.
//   k_bin_*       counting sort of an unsorted query batch into spatial order.
//   k_scatter_idx sorted-position hits -> caller-order model indices.
#include <atomic>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.h"
#ifdef TDTK_LAB
#include "loop_dev.h"
#endif

namespace tdtk {

#define WAVE 64
#define LAZY_MAX 8      // longest chain of queued scan moves a search launch applies itself (SearchArgs::moves)

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------
// transform3 (globals.icc:1477-1490): ((x*a0 + y*a4) + z*a8) + a12
__device__ __forceinline__ void dev_xf3(const Mat4& A, double x, double y, double z, double& ox,
                                        double& oy, double& oz)
{
  ox = x * A.m[0] + y * A.m[4] + z * A.m[8] + A.m[12];
  oy = x * A.m[1] + y * A.m[5] + z * A.m[9] + A.m[13];
  oz = x * A.m[2] + y * A.m[6] + z * A.m[10] + A.m[14];
}
// in-place transform3 (globals.icc:1454-1463): (x*a0 + y*a4 + z*a8), then + a12
__device__ __forceinline__ void dev_xf3_inplace(const Mat4& A, double& x, double& y, double& z)
{
  const double xn = x * A.m[0] + y * A.m[4] + z * A.m[8];
  const double yn = x * A.m[1] + y * A.m[5] + z * A.m[9];
  const double zn = x * A.m[2] + y * A.m[6] + z * A.m[10];
  x = xn + A.m[12];
  y = yn + A.m[13];
  z = zn + A.m[14];
}
// transform3normal (globals.icc:1465-1475): multiplies by the transposed rotation block
__device__ __forceinline__ void dev_xf3normal(const Mat4& A, double& x, double& y, double& z)
{
  const double xn = x * A.m[0] + y * A.m[1] + z * A.m[2];
  const double yn = x * A.m[4] + y * A.m[5] + z * A.m[6];
  const double zn = x * A.m[8] + y * A.m[9] + z * A.m[10];
  x = xn; y = yn; z = zn;
}

typedef const double __attribute__((address_space(4))) * const_d_ptr_fwd;  // constant address space -> s_load

// One of the in-place moves queued on a resident scan (scan.cc:851-875 for each): the matrix comes through the scalar
// cache (the chain is read-only for the whole launch), the arithmetic is dev_xf3_inplace / dev_xf3normal.
__device__ __forceinline__ void load_move(const Mat4* mats, const int k, Mat4& A)
{
  const_d_ptr_fwd sm = (const_d_ptr_fwd)(mats + k);
#pragma unroll
  for (int j = 0; j < 16; j++) A.m[j] = sm[j];
}

// XCD-aware work assignment: workgroup b runs on XCD b % 8 (observed dispatch order, used for
// speed only).  Giving XCD x the x-th contiguous eighth of the spatially sorted queries keeps
// each XCD's private 4 MB L2 on one eighth of the leaves instead of all of them.
__device__ __forceinline__ uint32_t xcd_chunk(uint32_t b, uint32_t nb)
{
  const uint32_t per = nb >> 3;  // nb is a multiple of 8 (launcher guarantees)
  return (b & 7u) * per + (b >> 3);
}

// A kernel's by-value argument block, read through the kernarg segment pointer instead of through the parameter: a
// by-value parameter is known dereferenceable and loop-invariant, so the compiler hoists every field (three 4x4 fp64
// matrices among them) into SGPRs up front and then spills them into VGPR lanes (80 in round 2's timed kernel).  Behind an
// opaque pointer of the constant address space the fields are s_load'ed where they are used.  The block must be the
// kernel's FIRST parameter.
template <class A>
__device__ __forceinline__ const A& kernarg_block()
{
  typedef const A __attribute__((address_space(4))) * kernarg_ptr;
  kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const A*)p;
}

// ------------------------------------------------------------------------------------------
// per-lane DFS stack: first SD entries in LDS ([level][lane] so a wave's pushes at one level
// are bank-conflict free), deeper entries in an HBM overflow area ([level][global lane]).
// An entry is (far child ref, myd^2); it is only pushed if myd^2 < closest_d2 at push time --
// closest_d2 never grows, so the reference's test after the near child returns
// (kdTreeImpl.h:373,378) would fail for every entry we skip.
// ------------------------------------------------------------------------------------------
// The overflow path lives in its own (rarely called) functions.  They name their address space (global) in every access,
// so they could be inlined without the compiler merging LDS and global accesses into flat ones -- round 4 measured both
// inlined forms (a per-lane branch; a wave-uniform branch around it): the 32 bytes of scratch per lane that the call frame
// costs every search kernel are gone then, and k_search of the 1M-vs-1M loop takes 0.2008-0.2028 ms instead of
// 0.1937-0.1946 (driver's arguments, gpurun_out/r4h, r4i).  Out of line it stays.
#define TDTK_OVF_INLINE __noinline__
__device__ TDTK_OVF_INLINE void stack_spill(double* g_m2, uint32_t* g_ref, size_t off, uint32_t ref, double m2)
{
  typedef double __attribute__((address_space(1))) * gd;
  typedef uint32_t __attribute__((address_space(1))) * gu;
  ((gd)g_m2)[off] = m2;
  ((gu)g_ref)[off] = ref;
}
__device__ TDTK_OVF_INLINE void stack_fill(const double* g_m2, const uint32_t* g_ref, size_t off, uint32_t& ref,
                                           double& m2)
{
  typedef const double __attribute__((address_space(1))) * gd;
  typedef const uint32_t __attribute__((address_space(1))) * gu;
  m2 = ((gd)g_m2)[off];
  ref = ((gu)g_ref)[off];
}

template <int BLOCK, int SD>
struct LaneStack {
  double* l_m2;    // &lds_m2[0][lane]
  uint32_t* l_ref; // &lds_ref[0][lane]
  double* g_m2;    // overflow, may be null when the tree is shallow
  uint32_t* g_ref;
  size_t gstride;
  int sp;
  __device__ __forceinline__ void push(uint32_t ref, double m2)
  {
    if (__builtin_expect(sp < SD, 1)) {
      l_m2[sp * BLOCK] = m2;
      l_ref[sp * BLOCK] = ref;
    } else {
      stack_spill(g_m2, g_ref, (size_t)(sp - SD) * gstride, ref, m2);
    }
    ++sp;
  }
  __device__ __forceinline__ void top(uint32_t& ref, double& m2) const
  {
    const int s = sp;
    if (__builtin_expect(s < SD, 1)) {
      m2 = l_m2[s * BLOCK];
      ref = l_ref[s * BLOCK];
    } else {
      stack_fill(g_m2, g_ref, (size_t)(s - SD) * gstride, ref, m2);
    }
  }
};

// The same stack with 16-byte entries { myd^2, far child, - }: one ds_write_b128 / ds_read_b128 and one address per push /
// pop instead of two of each (the persistent-lane kernel, round 3).  8 KB of LDS per 128-thread workgroup at SD = 4.
template <int BLOCK, int SD>
struct LaneStackQ {
  uint4* l_e;      // &lds_e[0][lane]
  double* g_m2;    // overflow area (wave-uniform base), may be null when the tree is shallow ...
  uint32_t* g_ref;
  size_t gcol;     // ... and this lane's column in it (round 6: kept apart -- as two per-lane pointers they were four vector
                   // registers live across the whole kernel for a path hardly ever taken; a column is one, or none)
  size_t gstride;
  int sp;
  __device__ __forceinline__ void push(uint32_t ref, double m2)
  {
    // (the overflow path behind a wave-uniform branch: the common case -- no lane of the wave beyond the LDS levels -- is a
    // scalar jump over it, not code every lane steps through with an empty mask)
    if (__builtin_expect(__ballot(sp >= SD) == 0ull, 1)) {
      l_e[sp * BLOCK] = make_uint4((uint32_t)__double2loint(m2), (uint32_t)__double2hiint(m2), ref, 0u);
    } else {
      if (sp < SD) l_e[sp * BLOCK] = make_uint4((uint32_t)__double2loint(m2), (uint32_t)__double2hiint(m2), ref, 0u);
      else stack_spill(g_m2, g_ref, (size_t)(sp - SD) * gstride + gcol, ref, m2);
    }
    ++sp;
  }
  __device__ __forceinline__ void top(uint32_t& ref, double& m2) const
  {
    const int s = sp;
    if (__builtin_expect(__ballot(s >= SD) == 0ull, 1)) {
      const uint4 e = l_e[s * BLOCK];
      m2 = __hiloint2double((int)e.y, (int)e.x);
      ref = e.z;
    } else if (s < SD) {
      const uint4 e = l_e[s * BLOCK];
      m2 = __hiloint2double((int)e.y, (int)e.x);
      ref = e.z;
    } else {
      stack_fill(g_m2, g_ref, (size_t)(s - SD) * gstride + gcol, ref, m2);
    }
  }
};

// descend() for the persistent-lane kernel: the axis comes with the record (KdHot::axis) and the child references keep
// their axis bit -- whoever turns a reference into an address or a leaf value masks it anyway (24-bit multiply, REF_VAL)
template <class STK>
__device__ __forceinline__ uint32_t descend_ax(const double splitval, const uint32_t c1, const uint32_t c2, const uint32_t axis,
                                               const double qx, const double qy, const double qz, const double best, STK& st)
{
  const double qa = (axis == 0) ? qx : ((axis == 1) ? qy : qz);
  const double myd = splitval - qa;                 // kdTreeImpl.h:371
  const bool first = (myd >= 0.0);
  const uint32_t far = first ? c2 : c1;
  const double m2 = myd * myd;
  if (m2 < best) st.push(far, m2);
  return first ? c1 : c2;
}

// ------------------------------------------------------------------------------------------
// 1-NN within radius: exact replay of KDTreeImpl::_FindClosest (kdTreeImpl.h:345-383)
// ------------------------------------------------------------------------------------------
typedef const double __attribute__((address_space(4))) * const_d_ptr;  // constant AS -> s_load
typedef const uint32_t __attribute__((address_space(4))) * const_u_ptr;

// one internal node of _FindClosest (kdTreeImpl.h:360-382).  Returns the near child, or sets
// need_pop when the box test prunes the node.  Inlined twice: once with the node in SGPRs
// (wave-uniform visit) and once with per-lane VGPR values.
template <int BLOCK, int SD>
__device__ __forceinline__ uint32_t visit_node(const double cx, const double cy, const double cz,
                                               const double hx, const double hy, const double hz,
                                               const double splitval, const uint32_t c1,
                                               const uint32_t c2, const double qx, const double qy,
                                               const double qz, const double best,
                                               LaneStack<BLOCK, SD>& st, bool& need_pop)
{
  const double ax = fabs(qx - cx) - hx;
  const double ay = fabs(qy - cy) - hy;
  const double az = fabs(qz - cz) - hz;
  const double ab = (ax < ay) ? ay : ax;  // std::max(ax, ay)
  const double ap = (ab < az) ? az : ab;
  if (ap >= 0.0 && ap * ap >= best) {  // kdTreeImpl.h:362-368
    need_pop = true;
    return REF_DONE;
  }
  const uint32_t axis = ((c1 >> 30) & 1u) | (((c2 >> 30) & 1u) << 1);
  const double qa = (axis == 0) ? qx : ((axis == 1) ? qy : qz);
  const double myd = splitval - qa;  // kdTreeImpl.h:371
  const uint32_t r1 = c1 & ~REF_AXIS, r2 = c2 & ~REF_AXIS;
  const bool first = (myd >= 0.0);
  const uint32_t far = first ? r2 : r1;
  const double m2 = myd * myd;
  if (m2 < best) st.push(far, m2);
  return first ? r1 : r2;
}

// Loads through an explicitly GLOBAL pointer.  The several-links-per-launch kernel finds its array pointers in a table in
// device memory, where nothing tells the compiler which address space they point to: it emitted flat_load (two counters,
// aperture check) for every record of the tree.  The arrays of a tree / scan are always global memory.
template <class V>
__device__ __forceinline__ V gload(const char* base, const uint32_t off)
{
  // (HIP's vector structs cannot be copied out of an address-space-qualified reference: load raw dwords, then bit-copy)
  typedef uint32_t raw __attribute__((ext_vector_type(sizeof(V) / 4)));
  typedef const raw __attribute__((address_space(1))) * gp;
  const raw r = *(gp)(base + off);
  V v;
  __builtin_memcpy(&v, &r, sizeof v);
  return v;
}
template <class V>
__device__ __forceinline__ void gstore(char* base, const uint32_t off, const V v)
{
  typedef uint32_t raw __attribute__((ext_vector_type(sizeof(V) / 4)));
  typedef raw __attribute__((address_space(1))) * gp;
  raw r;
  __builtin_memcpy(&r, &v, sizeof v);
  *(gp)(base + off) = r;
}
// One empty asm over one register of EVERY load of a batch: all of them must have been issued, and have arrived, here --
// so the compiler can neither sink one of them behind a branch nor (what it did in the several-links kernel, round 3) put
// a wait between them: 84 link passes 10.8 -> 12.6 ms from one such wait per node visit.
#define TDTK_PIN_BATCH3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))

// The compiler sinks the load of a node's last 16 bytes (split value + child references) behind the box test, because
// only the not-pruned path uses them -- which turns every node visit into TWO dependent memory round trips.  Naming the
// values in an empty asm right behind the loads keeps all four quarters of the record in one batch.
#define TDTK_PIN_V64(x) asm volatile("" : "+v"(x))
#define TDTK_PIN_S64(x) asm volatile("" : "+s"(x))
#define TDTK_PIN_S32(x) asm volatile("" : "+s"(x))
#define TDTK_PIN_SF(x) asm volatile("" : "+s"(x))
#define TDTK_PIN_VF(x) asm volatile("" : "+v"(x))

// ---- the box test at fp32 rate, without changing a single decision ---------------------------------------------
// The reference prunes a node iff  a >= 0 && a*a >= closest_d2,  a = max_i(|q_i - c_i| - h_i)  in fp64
// (kdTreeImpl.h:360-368).  With the node's box and the query rounded to fp32 the same expression a32 differs from the
// exact value by at most  3.01 eps32 (|q|_inf + |c|_inf + |h|_inf) <= delta = 3e-7 (|q|_inf + 2 absmax)  (eps32 = 2^-24;
// three conversions, two subtractions).  With r32 = sqrtf((float)closest_d2), off by at most 1.3 * 2^-23 relative:
//   a32 >= r32 (1 + 1e-6) + delta   =>  a >= sqrt(closest_d2) (1 + 2^-22) > 0   =>  the reference prunes;
//   a32 <  r32 (1 - 1e-6) - delta   =>  a <  sqrt(closest_d2) (1 - 2^-22)       =>  the reference does not;
// (the fp64 roundings of the reference's own a and a*a are 2^-53 relative, eleven orders below these margins).
// Anything in between -- a band of relative width ~1e-6 -- takes the exact fp64 test on the full record, and so does
// everything fp32 cannot hold: a query with an inf / NaN / > 3e38 component, or closest_d2 > FLT_MAX (the thresholds
// are NaN then, so both comparisons fail; tests/test_gpu_parity.py::test_box_shortcut_gives_way_outside_float_range).  Visits therefore stay the reference's node for node (the instrumented instantiations count the same
// numbers as the reference), while the common case costs 9 fp32 operations instead of 15 at fp64 rate and the hot
// record is 48 bytes instead of 64.
struct BoxF32 {
  float qx, qy, qz;   // the query in fp32
  float delta;        // error bound of a32 for this query; NaN when the query does not fit fp32 (inf / NaN component)
  float thi, tlo;     // decision thresholds for the current closest_d2; NaN: take the exact test
  float ec;           // bucket groups: sqrt(3) x the error bound of one fp32 coordinate difference (NaN like delta)
  float pthr;         // bucket groups: a shadow squared distance >= pthr proves myd2 >= closest_d2 (NaN: proves nothing)
  // 16-bit bucket shadows (bucket_scan_q16): qs = cells per unit, rounded up (0: the tree has no such shadow); pthr16 = the
  // squared grid distance from which a slot provably fails the reference's `<` against the current closest_d2 (inf / NaN: none)
  float qs, pthr16;
  __device__ __forceinline__ void set_query(const double x, const double y, const double z, const float absmax)
  {
    qx = (float)x; qy = (float)y; qz = (float)z;
    qs = 0.f;           // (the persistent-lane kernel sets it behind this call when the tree has a 16-bit shadow)
    const float qm = fmaxf(fmaxf(fabsf(qx), fabsf(qy)), fabsf(qz));
    delta = 3.0e-7f * (qm + 2.0f * absmax) + 1.0e-30f;
    // fmaxf drops a NaN operand and (float)1e39 is +inf: a query with such a component must never be decided in fp32.
    // The sum of the three is NaN or +-inf exactly when one of them is not finite (or when they are within a factor of
    // three of FLT_MAX, which may take the exact test as well).
    // GroupF32 (below): |fl32(fl32(p_i) - fl32(q_i)) - (p_i - q_i)| <= 2^-24 (|p_i| + |q_i|) (1 + 2^-23) + 2^-24 |p_i - q_i|;
    // ec covers sqrt(3) times the first term with 1.7 % to spare
    ec = 1.05e-7f * (qm + absmax) + 1.0e-30f;
    const float probe = fabsf(qx) + fabsf(qy) + fabsf(qz);
    if (!(probe <= 3.0e38f)) { delta = __builtin_nanf(""); ec = delta; }
  }
  // smallest shadow squared distance that PROVES a true distance > R (R an fp32 upper bound of the radius in question):
  // sqrt(s) (1 - 3.4 * 2^-24) - ec <= true distance, so s >= ((R + ec) (1 + 1e-6))^2 (1 + 1e-6) leaves 1.6e-6 R of margin
  // (the 1e-6 factors are 17 ulp each: they also cover the roundings of this very expression).  Not finite or beyond
  // 1e37: NaN, which no comparison passes.
  __device__ __forceinline__ float reject_from(const float R) const
  {
    const float w = (R * 1.000001f + ec) * 1.000001f;
    const float t = w * w * 1.000001f;
    return (t <= 1.0e37f) ? t : __builtin_nanf("");
  }
  __device__ __forceinline__ void set_radius(const double best)
  {
    const float r32 = __builtin_amdgcn_sqrtf((float)best);   // v_sqrt_f32: 1 ulp
    thi = r32 * 1.000001f + delta;
    tlo = r32 * 0.999999f - delta;
    // closest_d2 beyond float range ((float)best = +inf), or a NaN delta: both comparisons of the fast path must fail
    // (a32 >= NaN and a32 < NaN are false), which sends the visit to the exact fp64 test
    if (!(thi <= 3.0e38f)) { thi = __builtin_nanf(""); tlo = thi; }
    pthr = reject_from(r32);     // r32 >= sqrt(closest_d2) (1 - 2.5e-7): inside reject_from's margin
    // grid distance t of a slot (in cells) >= sqrt(s) - sqrt(3) (1 + 2e-9) (both ends rounded to the nearest cell); so
    // sqrt(s) >= r qs (1 + 1e-6) + 1.74  =>  t / scale > sqrt(closest_d2) (1 + 5e-7)  =>  the reference's fp64 d >= closest_d2.
    // (r32 qs is within 4e-7 relative of the exact product; the 1e-6 factors cover that and the roundings here.)
    const float w16 = r32 * qs * 1.000001f + 1.74f;
    pthr16 = w16 * w16 * 1.000001f;
  }
};

// exact box test of the reference on the full record (kdTreeImpl.h:360-368)
__device__ __forceinline__ bool box_prunes_exact(const double cx, const double cy, const double cz, const double hx,
                                                 const double hy, const double hz, const double qx, const double qy,
                                                 const double qz, const double best)
{
  const double ax = fabs(qx - cx) - hx;
  const double ay = fabs(qy - cy) - hy;
  const double az = fabs(qz - cz) - hz;
  const double ab = (ax < ay) ? ay : ax;
  const double ap = (ab < az) ? az : ab;
  return ap >= 0.0 && ap * ap >= best;
}

// the split-plane half of visit_node (kdTreeImpl.h:371-382): near child, far child pushed if it can still matter
template <int BLOCK, int SD, class STK = LaneStack<BLOCK, SD>>
__device__ __forceinline__ uint32_t descend(const double splitval, const uint32_t c1, const uint32_t c2, const double qx,
                                            const double qy, const double qz, const double best, STK& st)
{
  const uint32_t axis = ((c1 >> 30) & 1u) | (((c2 >> 30) & 1u) << 1);
  const double qa = (axis == 0) ? qx : ((axis == 1) ? qy : qz);
  const double myd = splitval - qa;
  const uint32_t r1 = c1 & ~REF_AXIS, r2 = c2 & ~REF_AXIS;
  const bool first = (myd >= 0.0);
  const uint32_t far = first ? r2 : r1;
  const double m2 = myd * myd;
  if (m2 < best) st.push(far, m2);
  return first ? r1 : r2;
}

// descend(), also telling whether the near child is child 1 (the two-level walk then knows which half of the fat record
// describes it)
template <int BLOCK, int SD, class STK = LaneStack<BLOCK, SD>>
__device__ __forceinline__ uint32_t descend_which(const double splitval, const uint32_t c1, const uint32_t c2, const double qx,
                                                  const double qy, const double qz, const double best, STK& st,
                                                  bool& near_is_c1)
{
  const uint32_t axis = ((c1 >> 30) & 1u) | (((c2 >> 30) & 1u) << 1);
  const double qa = (axis == 0) ? qx : ((axis == 1) ? qy : qz);
  const double myd = splitval - qa;
  const uint32_t r1 = c1 & ~REF_AXIS, r2 = c2 & ~REF_AXIS;
  const bool first = (myd >= 0.0);
  near_is_c1 = first;
  const uint32_t far = first ? r2 : r1;
  const double m2 = myd * myd;
  if (m2 < best) st.push(far, m2);
  return first ? r1 : r2;
}

// Warm start of a repeated pass (ICP iteration i+1 over the scan iteration i has just searched): the previous hit is a
// real point of the tree, so the nearest point is at most that far away, and starting with closest_d2 one ulp ABOVE
// its squared distance (instead of maxdist2) cannot lose it: every test of the traversal prunes only what lies at
// or beyond closest_d2, the walk order is unchanged, and whatever the reference would have visited before reaching a
// point at that distance it still visits.  Same index, same d2 -- from a radius of a few units instead of 25.
// "Prunes only what lies at or beyond closest_d2" holds up to the ROUNDING of the tests themselves (the quick check's
// a = max|q - c| - h carries ~2^-50 (3 absmax + R)): at a radius one ulp above the hit's own distance a rounding could cut off
// the very node that holds the hit when q - p is almost axis-aligned, where the reference, walking with maxdist2, visits it.
// The radius is therefore d2 + 2 * SearchArgs::margin (api.cpp: search_margin bounds that rounding, times four).
static __device__ __forceinline__ double warm_radius_kp(const SearchArgs& a, const int kp, const double qx, const double qy,
                                                        const double qz)
{
  double best = a.maxd2;
  if (a.warm) {
    if (kp >= 0) {
      const uint32_t po = (uint32_t)kp << 5;
      const double2 pxy = gload<double2>(reinterpret_cast<const char*>(a.T.pts), po);
      const double pzz = gload<double>(reinterpret_cast<const char*>(a.T.pts), po + 16);
      const double dx = pxy.x - qx, dy = pxy.y - qy, dz = pzz - qz;
      const double d = dx * dx + dy * dy + dz * dz;
      double up = __longlong_as_double(__double_as_longlong(d) + 1);   // next double above d (d >= 0, finite)
      const double um = d + 2.0 * a.margin;                               // (SearchArgs::margin; 0: one ulp)
      if (um > up) up = um;
      if (up < best) best = up;
    }
  }
  return best;
}
static __device__ __forceinline__ double warm_radius(const SearchArgs& a, const size_t i, const double qx, const double qy,
                                                     const double qz)
{
  return warm_radius_kp(a, a.warm ? a.kpos[i] : -1, qx, qy, qz);
}

// bytes of a hot record the persistent-lane kernel's divergent visit loads: 0 = all 48 (16 + 16 + 16), 2 = 40 (16 + 8 + 16, the
// split axis from the child references' bits).  Measured (round 4): 40 bytes 0.1957-0.1970 ms per k_search of the 1M-vs-1M
// loop against 0.1933-0.1959 with 48, the 84-link round 10.59 ms against 10.45-10.52 -- nothing, like the 64-byte-aligned
// record of round 3: the visit's cost is not its bytes.
constexpr int HOT_NARROW = 0;

// groups of four points a lane takes in per round trip of the bucket filter: 5 = a whole default bucket (-b 20)
constexpr int GRP_TRIP = 5;
// which shadow the persistent-lane kernels filter buckets with: the 16-bit grid (bucket_scan_q16, eight loads per bucket) or the
// fp32 groups (bucket_scan_groups, fifteen); -DTDTK_BUCKET_FP32 builds the latter (the A/B of round 5)
#ifdef TDTK_BUCKET_FP32
constexpr bool BUCKET_Q16 = false;
#else
constexpr bool BUCKET_Q16 = true;
#endif

// One bucket of up to 4 * GRP_TRIP points, scanned through its fp32 shadow groups (tree_pad_buckets in api.cpp: buckets are
// padded to whole groups of four slots, `start` is a multiple of 4).  pb = the fp64 point array, o0 = byte offset of the
// bucket in it.  On return best / bk are what the reference's leaf loop (kdTreeImpl.h:351-357) leaves behind.
__device__ __forceinline__ void bucket_scan_groups(const char* __restrict__ t_grp, const char* __restrict__ pb, const int start,
                                             const int count, const uint32_t o0, const BoxF32& bx, const double qx,
                                             const double qy, const double qz, double& best, int& bk)
{
  // ---- bucket groups: the whole bucket in ONE round trip of fp32 shadow records, then only the points that can
  // still matter from the fp64 array.  The reference's leaf loop (kdTreeImpl.h:351-357) leaves behind
  //   closest_d2 = min(closest_d2, min_j d_j),  closest = the FIRST j that attains a smaller value,
  // and nothing else of it is observable.  A point is dropped only on proof that it is not that j:
  //   (A) s_j >= pthr            =>  d_j >= closest_d2 (BoxF32::reject_from): it fails the reference's '<';
  //   (B) s_j >= reject_from(R)  with R >= the true distance of the point k with the smallest shadow distance
  //                              =>  d_j > d_k: somebody else is strictly closer.
  // The survivors -- the nearest point, exact duplicates of it, anything within ~1e-5 relative -- are tested in
  // fp64 in bucket order with the strict '<', which is the reference's loop restricted to the points that can win.
  typedef float v2f __attribute__((ext_vector_type(2)));
  const uint32_t go = (uint32_t)(start >> 2) * 48u;     // the bucket's first group (start is a multiple of 4)
  const uint32_t glast = go + (uint32_t)((count - 1) >> 2) * 48u;
  float4 X[GRP_TRIP], Y[GRP_TRIP], Z[GRP_TRIP];
#pragma unroll
  for (int k = 0; k < GRP_TRIP; k++) {
    // groups past the bucket's last re-read the last one (an L1 hit, no branch); their bits are masked out below.
    // (Loading each group under its own lane mask instead -- no access for a group the bucket does not have -- costs
    // the register allocation 188 VGPRs instead of 122, i.e. half the resident waves; masking only the fifth group
    // keeps 122 and changes nothing: 0.1987 / 0.1990 ms against 0.1974 / 0.1970, gpurun_out/r3f.)
    // (Round 4: a group the bucket does not have read from ONE place instead -- a group of four points at +inf behind the
    // last bucket, the same address for every lane that lacks the group: L1 accesses per launch 83.3 M -> 80.3 M, time
    // 0.1955-0.1966 ms against 0.1933-0.1945.  The vector L1 spends its cycles per instruction, not per distinct line.)
    const uint32_t gk = min(go + 48u * (uint32_t)k, glast);
    X[k] = gload<float4>(t_grp, gk);
    Y[k] = gload<float4>(t_grp, gk + 16);
    Z[k] = gload<float4>(t_grp, gk + 32);
  }
  const v2f qxx = {bx.qx, bx.qx}, qyy = {bx.qy, bx.qy}, qzz = {bx.qz, bx.qz};
  float sv[4 * GRP_TRIP];
#pragma unroll
  for (int k = 0; k < GRP_TRIP; k++) {
    const v2f dxa = (v2f){X[k].x, X[k].y} - qxx, dxb = (v2f){X[k].z, X[k].w} - qxx;
    const v2f dya = (v2f){Y[k].x, Y[k].y} - qyy, dyb = (v2f){Y[k].z, Y[k].w} - qyy;
    const v2f dza = (v2f){Z[k].x, Z[k].y} - qzz, dzb = (v2f){Z[k].z, Z[k].w} - qzz;
    const v2f sa = __builtin_elementwise_fma(dza, dza, __builtin_elementwise_fma(dya, dya, dxa * dxa));
    const v2f sb = __builtin_elementwise_fma(dzb, dzb, __builtin_elementwise_fma(dyb, dyb, dxb * dxb));
    sv[4 * k] = sa.x; sv[4 * k + 1] = sa.y; sv[4 * k + 2] = sb.x; sv[4 * k + 3] = sb.y;
  }
  float smin = sv[0];
#pragma unroll
  for (int j = 1; j < 4 * GRP_TRIP; j++) smin = fminf(smin, sv[j]);   // pad slots and re-read groups repeat real points
  unsigned surv = 0u;
  if (!(smin >= bx.pthr)) {               // else (A) drops every point of the bucket: the usual case after the first
    // (B): sqrt(smin) (1 + 4 * 2^-24) + ec >= the true distance of that point
    const float rub = __builtin_amdgcn_sqrtf(smin) * 1.000001f + bx.ec;
    const float thr = fminf(bx.pthr, bx.reject_from(rub));   // fminf skips a NaN operand: either proof alone is valid
    // bit j = [sv[j] < thr] = the sign of sv[j] - thr (an exact zero and +inf reject, as `>=` does); two differences
    // per packed instruction, one v_alignbit per point shifts the sign in -- half the instructions of compare / select / or
    const v2f th2 = {thr, thr};
#pragma unroll
    for (int j = 4 * GRP_TRIP - 2; j >= 0; j -= 2) {
      const v2f dd2 = (v2f){sv[j], sv[j + 1]} - th2;
      surv = __builtin_amdgcn_alignbit(surv, __float_as_uint(dd2.y), 31);
      surv = __builtin_amdgcn_alignbit(surv, __float_as_uint(dd2.x), 31);
    }
    if (!(thr == thr)) surv = 0xFFFFFFFFu;    // no proof available (NaN threshold): every point is tested exactly
    surv &= (count >= 32) ? 0xFFFFFFFFu : ((1u << count) - 1u);
  }
  while (surv) {                          // in bucket order: lowest set bit first
    const uint32_t j = (uint32_t)__builtin_ctz(surv);
    surv &= surv - 1u;
    const uint32_t oj = o0 + (j << 5);
    const double2 pxy = gload<double2>(pb, oj);
    const double pz = gload<double>(pb, oj + 16);
    const double dx = pxy.x - qx, dy = pxy.y - qy, dz = pz - qz;
    const double dj = dx * dx + dy * dy + dz * dz;
    if (dj < best) { best = dj; bk = (int)(oj >> 5); }
  }
}

// grid index - 32768 of one coordinate on the 16-bit grid of a tree (TreeDev::q16): round to nearest, so a coordinate inside
// the root box is off by at most half a cell (+ 1e-10 of fp64 rounding); `out` is raised for a coordinate outside the grid
// (a query beyond the box; NaN), which then sits on the nearest face -- see bucket_scan_q16 for what that still proves
__device__ __forceinline__ int q16_index(const double v, const double lo, const double scale, bool& out)
{
  const double u = (v - lo) * scale + 0.5;
  if (!(u >= 0.0)) { out = true; return -32768; }
  if (!(u < 65536.0)) { out = true; return 32767; }
  return (int)u - 32768;                     // u >= 0: truncation is floor
}

// ---- bucket_scan_q16 (round 5): the same filter on a 16-bit shadow -- eight 16-byte loads per bucket instead of fifteen.
// Every slot's coordinates sit on ONE grid over the tree's root box (TreeDev::q16: 65536 cells per axis, cell = largest
// extent / 65535, int16 = index - 32768, two slots per 12 bytes { (x0,y0), (x1,y1), (z0,z1) }), the query is put on the same
// grid once (qxy, qzz; q_in: it lies inside the grid on all three axes), and a slot's squared grid distance
//   s = sum_a sat16(P_a - Q_a)^2          (v_pk_sub_i16 clamp, v_dot2_i32_i16 clamp: 4.5 instructions per slot)
// brackets its true distance t (in cells):  | sqrt(s) - t | <= sqrt(3) (1 + 2e-9)  when nothing saturated and q_in -- both
// ends are rounded to the nearest cell --, and  t >= sqrt(s) - sqrt(3) (1 + 2e-9)  ALWAYS: a saturated difference is
// smaller than the true one, and a query outside the box is further from every point of the box than its projection
// onto it, which is what it was quantised as (exactly, on those axes).  The two proofs of bucket_scan_groups:
//   (A) s_j >= pthr16 (BoxF32::set_radius)                   =>  d_j >= closest_d2: fails the reference's '<';
//   (B) sqrt(s_j) > sqrt(s_k) + 2 sqrt(3) + margin, k = argmin s, q_in, s_k < 1e9 (no difference of k saturated)
//                                                             =>  t_j > t_k: somebody else is strictly closer.
// Survivors are tested in fp64, in bucket order, with the strict '<' (kdTreeImpl.h:351-357 restricted to who can win).
// No proof available (radius beyond the grid, thresholds not finite): every slot of the bucket is a survivor.
__device__ __forceinline__ void bucket_scan_q16(const char* __restrict__ t_q16, const char* __restrict__ pb, const int start, const int count,
                                                const uint32_t o0, const BoxF32& bx, const uint32_t qxy, const uint32_t qzz, const bool q_in,
                                                const double qx, const double qy, const double qz, double& best, int& bk, const double tie, bool& thin)
{
  typedef short v2s __attribute__((ext_vector_type(2)));
  const uint32_t go = (uint32_t)start * 6u;      // 6 bytes per slot; start is a multiple of 4: 8-byte aligned
  uint4 W[8];
#pragma unroll
  for (int k = 0; k < 8; k++) W[k] = gload<uint4>(t_q16, go + 16u * (uint32_t)k);     // 128 bytes: the bucket's <= 20 slots (+ whatever follows: masked)
  uint32_t w[32];
#pragma unroll
  for (int k = 0; k < 8; k++) { w[4 * k] = W[k].x; w[4 * k + 1] = W[k].y; w[4 * k + 2] = W[k].z; w[4 * k + 3] = W[k].w; }
  v2s q_xy, q_zz;
  __builtin_memcpy(&q_xy, &qxy, 4); __builtin_memcpy(&q_zz, &qzz, 4);
  int sv[20];
#pragma unroll
  for (int p = 0; p < 10; p++) {
    v2s a0, a1, az;
    __builtin_memcpy(&a0, &w[3 * p], 4); __builtin_memcpy(&a1, &w[3 * p + 1], 4); __builtin_memcpy(&az, &w[3 * p + 2], 4);
    const v2s d0 = __builtin_elementwise_sub_sat(a0, q_xy), d1 = __builtin_elementwise_sub_sat(a1, q_xy), dz = __builtin_elementwise_sub_sat(az, q_zz);
    uint32_t dzu;
    __builtin_memcpy(&dzu, &dz, 4);
    const uint32_t zlo = dzu & 0xFFFFu, zhi = dzu & 0xFFFF0000u;
    v2s dz0, dz1;
    __builtin_memcpy(&dz0, &zlo, 4); __builtin_memcpy(&dz1, &zhi, 4);
    sv[2 * p] = __builtin_amdgcn_sdot2(dz, dz0, __builtin_amdgcn_sdot2(d0, d0, 0, true), true);         // saturates at INT_MAX
    sv[2 * p + 1] = __builtin_amdgcn_sdot2(dz, dz1, __builtin_amdgcn_sdot2(d1, d1, 0, true), true);
  }
  // (slots past the bucket's last belong to the next bucket or to the slack behind the array: masked out of the minimum and
  // of the survivors; pad slots of this bucket repeat its last point)
  const uint32_t cmask20 = (count >= 32) ? 0xFFFFFFFFu : ((1u << count) - 1u);
  int smin = 0x7FFFFFFF;
#pragma unroll
  for (int j = 0; j < 20; j++) smin = min(smin, (j < count) ? sv[j] : 0x7FFFFFFF);
  unsigned surv = 0u;
  const float sminf = (float)smin;                         // (round to nearest: 6e-8 relative, inside the 1e-6 margins)
  if (!(sminf * 0.999999f >= bx.pthr16)) {                 // else (A) drops every slot of the bucket (a NaN / inf threshold proves nothing)
    float thr = bx.pthr16;
    if (q_in && smin < 1000000000) {
      // (B): t_k <= sqrt(s_k) + 1.7321, t_j >= sqrt(s_j) - 1.7321
      const float rub = __builtin_amdgcn_sqrtf(sminf) * 1.000001f + 3.47f;
      thr = fminf(thr, rub * rub * 1.000001f);               // fminf skips a NaN operand: either proof alone is valid
    }
    if (thr < 2.0e9f) {
      const int thr_i = (int)thr + 1;                        // s_j >= thr_i  =>  s_j >= thr: rejected on proof
#pragma unroll
      for (int j = 19; j >= 0; j--) surv = __builtin_amdgcn_alignbit(surv, (uint32_t)(sv[j] - thr_i), 31);   // both in [0, 2^31): the sign is [s_j < thr_i]
    } else surv = 0xFFFFFu;                                  // no proof available: every slot is tested exactly
    surv &= cmask20;
  }
  while (surv) {                          // in bucket order: lowest set bit first
    const uint32_t j = (uint32_t)__builtin_ctz(surv);
    surv &= surv - 1u;
    const uint32_t oj = o0 + (j << 5);
    const double2 pxy = gload<double2>(pb, oj);
    const double pz = gload<double>(pb, oj + 16);
    const double dx = pxy.x - qx, dy = pxy.y - qy, dz = pz - qz;
    const double dj = dx * dx + dy * dy + dz * dz;
    if (dj < best) { thin = thin || (best - dj <= tie); best = dj; bk = (int)(oj >> 5); }
  }
}

template <int BLOCK, int SD, bool COUNT, bool UNI, int PTS = 4>
__device__ __forceinline__ void kd_search(const TreeDev& T, const double qx, const double qy,
                                          const double qz, double& best, int& bk,
                                          LaneStack<BLOCK, SD>& st, unsigned long long* cnt)
{
  uint32_t cur = T.root_ref;
  st.sp = 0;
  unsigned c_int = 0, c_leaf = 0, c_pts = 0;
  const double4* __restrict__ nodes = reinterpret_cast<const double4*>(T.nodes);
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(T.pts);
  const char* __restrict__ hotb = reinterpret_cast<const char*>(T.hot);
  BoxF32 bx;
  bx.set_query(qx, qy, qz, T.absmax);
  bx.set_radius(best);

  for (;;) {
    // ---- phase 1: walk internal nodes until this lane holds a leaf (or is finished) ----
    while (!(cur & REF_LEAF)) {
      if (COUNT) ++c_int;
      bool need_pop = false;
      uint32_t next = REF_DONE;
      bool uniform = false;
      uint32_t ucur = 0;
      if (UNI) {
        // spatially sorted queries walk the same upper nodes: when every active lane of the
        // wave holds the same node, fetch its hot record once through the scalar cache (operands stay in
        // SGPRs) instead of 64 lanes x 3 vector loads
        ucur = __builtin_amdgcn_readfirstlane(cur);
        uniform = __all(cur == ucur);
      }
      if (UNI && uniform) {
        typedef const float __attribute__((address_space(4))) * const_f_ptr;
        const_f_ptr sf = (const_f_ptr)(hotb + (size_t)ucur * sizeof(KdHot));
        const_d_ptr sd = (const_d_ptr)(sf + 8);
        const_u_ptr su = (const_u_ptr)(sf + 10);
        double s_split = sd[0];
        uint32_t s_c1 = su[0], s_c2 = su[1];
        TDTK_PIN_S64(s_split); TDTK_PIN_S32(s_c1); TDTK_PIN_S32(s_c2);
        const float a32 = fmaxf(fmaxf(fabsf(bx.qx - sf[0]) - sf[3], fabsf(bx.qy - sf[1]) - sf[4]), fabsf(bx.qz - sf[2]) - sf[5]);
        bool prune = a32 >= bx.thi;
        if (__builtin_expect(!prune && !(a32 < bx.tlo), 0)) {
          const_d_ptr sn = (const_d_ptr)(T.nodes) + (size_t)ucur * 8;
          prune = box_prunes_exact(sn[0], sn[1], sn[2], sn[3], sn[4], sn[5], qx, qy, qz, best);
        }
        if (prune) need_pop = true;
        else next = descend<BLOCK, SD>(s_split, s_c1, s_c2, qx, qy, qz, best, st);
      } else {
        const char* hp = hotb + (size_t)cur * sizeof(KdHot);
        const float4 b0 = *reinterpret_cast<const float4*>(hp);          // cx cy cz hx
        const float2 b1 = *reinterpret_cast<const float2*>(hp + 16);     // hy hz
        double2 sc = *reinterpret_cast<const double2*>(hp + 32);         // splitval {c1, c2}
        TDTK_PIN_V64(sc.x); TDTK_PIN_V64(sc.y);
        const float a32 = fmaxf(fmaxf(fabsf(bx.qx - b0.x) - b0.w, fabsf(bx.qy - b0.y) - b1.x), fabsf(bx.qz - b0.z) - b1.y);
        bool prune = a32 >= bx.thi;
        if (__builtin_expect(!prune && !(a32 < bx.tlo), 0)) {
          const double4 n0 = nodes[(size_t)cur * 2];
          const double4 n1 = nodes[(size_t)cur * 2 + 1];
          prune = box_prunes_exact(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, qx, qy, qz, best);
        }
        if (prune) need_pop = true;
        else next = descend<BLOCK, SD>(sc.x, (uint32_t)__double2loint(sc.y), (uint32_t)__double2hiint(sc.y), qx, qy, qz, best, st);
      }
      if (need_pop) {
        next = REF_DONE;
        while (st.sp > 0) {
          --st.sp;
          uint32_t r; double m2;
          st.top(r, m2);
          if (m2 < best) { next = r; break; }
        }
      }
      cur = next;
    }
    if (cur == REF_DONE) break;

    // ---- phase 2: scan the leaf bucket in stored order, strict '<' (kdTreeImpl.h:351-357).
    // Four points per round trip; the last group re-reads the final point instead of running a
    // scalar tail (a repeated point can never pass the strict '<' a second time).
    {
      const uint32_t v = cur & REF_VAL;
      int start, count;
      if (T.leaf_tab) {
        const LeafEntry le = T.leaf_tab[v];
        start = le.start; count = le.count;
      } else {
        start = (int)(v >> T.cb);
        count = (int)(v & T.cmask);
      }
      if (COUNT) { ++c_leaf; c_pts += (unsigned)count; }
      if (T.grp != nullptr && count <= 4 * GRP_TRIP) {     // the whole bucket in one round trip (bucket_scan_groups)
        bucket_scan_groups(reinterpret_cast<const char*>(T.grp), reinterpret_cast<const char*>(pts), start, count,
                           (uint32_t)start << 5, bx, qx, qy, qz, best, bk);
      } else {
      const double4* __restrict__ P = pts + start;
      const int last = count - 1;
      // PTS points per round trip (small batches run one wave per SIMD: nothing else hides the latency,
      // so they keep more loads in flight)
      for (int i = 0; i < count; i += PTS) {
        int id[PTS];
        double4 p[PTS];
        double d[PTS];
#pragma unroll
        for (int j = 0; j < PTS; j++) { id[j] = min(i + j, last); p[j] = P[id[j]]; }
#pragma unroll
        for (int j = 0; j < PTS; j++) {
          const double dx = p[j].x - qx, dy = p[j].y - qy, dz = p[j].z - qz;
          d[j] = dx * dx + dy * dy + dz * dz;
        }
#pragma unroll
        for (int j = 0; j < PTS; j++)
          if (d[j] < best) { best = d[j]; bk = start + id[j]; }
      }
      }
      bx.set_radius(best);
    }
    // pop the next pending far child that still passes sqr(myd) < closest_d2
    cur = REF_DONE;
    while (st.sp > 0) {
      --st.sp;
      uint32_t r; double m2;
      st.top(r, m2);
      if (m2 < best) { cur = r; break; }
    }
    if (cur == REF_DONE) break;
  }
  if (COUNT && cnt) {
    atomicAdd(&cnt[0], (unsigned long long)c_int);
    atomicAdd(&cnt[1], (unsigned long long)c_leaf);
    atomicAdd(&cnt[2], (unsigned long long)c_pts);
  }
}

#ifdef TDTK_LAB
#include "lab_coop.inc"   // the wave-cooperative kernel (k_search_coop)
#endif

// ------------------------------------------------------------------------------------------
// nearest point to a line: exact replay of _FindClosestAlongDir (kdTreeImpl.h:390-425).
// No split-plane pruning in the reference: both children are always visited, near side first.
// ------------------------------------------------------------------------------------------
template <int BLOCK, int SD>
__device__ __forceinline__ void kd_search_dir(const TreeDev& T, const double qx, const double qy,
                                              const double qz, const double ux, const double uy,
                                              const double uz, double& best, int& bk,
                                              LaneStack<BLOCK, SD>& st)
{
  uint32_t cur = T.root_ref;
  st.sp = 0;
  const double4* __restrict__ nodes = reinterpret_cast<const double4*>(T.nodes);
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(T.pts);
  for (;;) {
    while (!(cur & REF_LEAF)) {
      const double4 n0 = nodes[(size_t)cur * 2];
      const double4 n1 = nodes[(size_t)cur * 2 + 1];
      const double r = T.node_r[cur];
      const double vx = qx - n0.x, vy = qy - n0.y, vz = qz - n0.z;
      const double len2 = vx * vx + vy * vy + vz * vz;
      const double dot = vx * ux + vy * uy + vz * uz;
      const double d2c = len2 - dot * dot;
      const double lim = r + __dsqrt_rn(best);
      uint32_t next = REF_DONE;
      if (d2c > lim * lim) {
        if (st.sp > 0) { --st.sp; double m2; st.top(next, m2); }
      } else {
        const uint32_t c1 = (uint32_t)__double2loint(n1.w);
        const uint32_t c2 = (uint32_t)__double2hiint(n1.w);
        const uint32_t axis = ((c1 >> 30) & 1u) | (((c2 >> 30) & 1u) << 1);
        const double qa = (axis == 0) ? qx : ((axis == 1) ? qy : qz);
        const uint32_t r1 = c1 & ~REF_AXIS, r2 = c2 & ~REF_AXIS;
        const bool first = (qa < n1.z);  // p[axis] < splitval -> child1 first
        next = first ? r1 : r2;
        st.push(first ? r2 : r1, 0.0);
      }
      cur = next;
    }
    if (cur == REF_DONE) break;
    {
      const uint32_t v = cur & REF_VAL;
      int start, count;
      if (T.leaf_tab) {
        const LeafEntry le = T.leaf_tab[v];
        start = le.start; count = le.count;
      } else {
        start = (int)(v >> T.cb);
        count = (int)(v & T.cmask);
      }
      const double4* __restrict__ P = pts + start;
      for (int i = 0; i < count; i++) {
        const double4 p = P[i];
        const double vx = qx - p.x, vy = qy - p.y, vz = qz - p.z;
        const double len2 = vx * vx + vy * vy + vz * vz;
        const double dot = vx * ux + vy * uy + vz * uz;
        const double d = len2 - dot * dot;
        if (d < best) { best = d; bk = start + i; }
      }
    }
    cur = REF_DONE;
    if (st.sp > 0) { --st.sp; double m2; st.top(cur, m2); }
    if (cur == REF_DONE) break;
  }
}

#ifdef TDTK_LAB
#include "lab_part_1.inc"
#endif

// FUSE (k_search, k_search_g8: the batches too small for the persistent-lane kernel): once the workgroup is through with
// its chunk of the queries it sums the base pair block of that chunk itself (defined behind wave_sum below), so an
// ICP iteration on a small scan is two launches instead of three -- at that size nothing is short of issue slots and
// the separate k_accum is a fifth of the iteration
template <int BLOCK>
__device__ __forceinline__ void chunk_pair_sums(const SearchArgs& a, size_t lo, size_t hi, uint32_t row);

// ------------------------------------------------------------------------------------------
// k_search: the hot kernel
// ------------------------------------------------------------------------------------------
template <int BLOCK, int SD, bool COUNT, int DIRMODE, bool UNI, int WPS, int PTS = 4, bool FUSE = false>
__device__ __forceinline__ void search_plain_body(const SearchArgs& a, const uint32_t bid, const uint32_t nb)
{
  __shared__ double lds_m2[SD][BLOCK];
  __shared__ uint32_t lds_ref[SD][BLOCK];

  const uint32_t chunk = xcd_chunk(bid, nb);
  const size_t gl = (size_t)bid * BLOCK + threadIdx.x;  // for the overflow slot only

  LaneStack<BLOCK, SD> st;
  st.l_m2 = &lds_m2[0][threadIdx.x];
  st.l_ref = &lds_ref[0][threadIdx.x];
  st.g_m2 = a.ovf_m2 ? a.ovf_m2 + gl : nullptr;
  st.g_ref = a.ovf_ref ? a.ovf_ref + gl : nullptr;
  st.gstride = (size_t)nb * BLOCK;
  st.sp = 0;

  // each workgroup owns a contiguous slab of the sorted queries and walks it BLOCK at a time
  const size_t per = (a.n + nb - 1) / nb;
  const size_t lo = (size_t)chunk * per;
  size_t hi = lo + per;
  if (hi > a.n) hi = a.n;

  for (size_t base = lo; base < hi; base += BLOCK) {
    const size_t i = base + threadIdx.x;
    if (i >= hi) continue;
    double tx = a.x[i], ty = a.y[i], tz = a.z[i];
    double ux = 0, uy = 0, uz = 0;
    if (a.has_pending) {  // Scan::transformReduced fused in (scan.cc:851-875)
      dev_xf3_inplace(a.pending, tx, ty, tz);
      a.x[i] = tx; a.y[i] = ty; a.z[i] = tz;
      if (a.nx) {
        double px = a.nx[i], py = a.ny[i], pz = a.nz[i];
        dev_xf3normal(a.pending, px, py, pz);
        a.nx[i] = px; a.ny[i] = py; a.nz[i] = pz;
      }
    }
    if (a.skip && a.skip[i]) {      // -R: not drawn this pass (searchTree.cc:118) -- moved above, not searched
      a.kpos[i] = -1;
      if (a.d2) a.d2[i] = a.maxd2;
      continue;
    }
    double sx = tx, sy = ty, sz = tz;
    if (a.has_inv) dev_xf3(a.inv, tx, ty, tz, sx, sy, sz);  // searchTree.cc:122
    if (DIRMODE) {
      ux = a.nx[i]; uy = a.ny[i]; uz = a.nz[i];
      if (DIRMODE == 2) {  // raw directions supplied by the caller (FindClosestAlongDir API)
      } else {             // searchTree.cc:126-135: Normalize3 then rotate into the tree frame
        const double len = __dsqrt_rn(ux * ux + uy * uy + uz * uz);
        ux /= len; uy /= len; uz /= len;
        if (a.has_inv) dev_xf3normal(a.inv, ux, uy, uz);
      }
    }
    double best = DIRMODE ? a.maxd2 : warm_radius(a, i, sx, sy, sz);
    int bk = -1;
    if (DIRMODE) kd_search_dir<BLOCK, SD>(a.T, sx, sy, sz, ux, uy, uz, best, bk, st);
    else kd_search<BLOCK, SD, COUNT, UNI, PTS>(a.T, sx, sy, sz, best, bk, st, a.counters);
    a.kpos[i] = bk;
    if (a.d2) a.d2[i] = best;
  }
  if constexpr (FUSE) chunk_pair_sums<BLOCK>(a, lo, hi, bid);
}

template <int BLOCK, int SD, bool COUNT, int DIRMODE, bool UNI, int WPS, int PTS = 4, bool FUSE = false>
__global__ void __launch_bounds__(BLOCK, WPS) k_search(const SearchArgs a_by_value)
{
  (void)a_by_value;
  search_plain_body<BLOCK, SD, COUNT, DIRMODE, UNI, WPS, PTS, FUSE>(kernarg_block<SearchArgs>(), blockIdx.x, gridDim.x);
}
// several batches in one launch (see k_search_refill_multi)
template <int BLOCK, int SD, bool COUNT, bool UNI, int WPS>
__global__ void __launch_bounds__(BLOCK, WPS) k_search_multi(const SearchArgs* __restrict__ args, const uint32_t* __restrict__ base,
                                                             int nbatch)
{
  int l = 0;
  while (l + 1 < nbatch && blockIdx.x >= base[l + 1]) ++l;
  l = __builtin_amdgcn_readfirstlane(l);
  const uint32_t b0 = base[l], b1 = base[l + 1];
  search_plain_body<BLOCK, SD, COUNT, 0, UNI, WPS>(args[l], blockIdx.x - b0, b1 - b0);
}

// ------------------------------------------------------------------------------------------
// k_search_g8: eight lanes per query, for batches too small to fill the machine with one lane per
// query (a bundled 81K-point scan gives 1.2 waves per SIMD; the kernel then runs as long as the
// dependent-load chain of its slowest lane, ~0.7 us a step under 64-way address divergence).
// The eight lanes of a group hold the same query and walk the same nodes (their loads hit one line,
// so a wave touches 8 lines per instruction instead of 64), keep one stack per group in LDS, and
// scan a bucket eight points at a time: lane j takes point i+j and the group takes the minimum,
// lowest index first among equals -- which is what the serial strict '<' scan in stored order
// returns.  Same visiting order as k_search, hence the same indices.
// ------------------------------------------------------------------------------------------
template <int BLOCK, int SD, int GS = 8, bool FUSE = false, bool LOOP = false>
__device__ __forceinline__ void search_g8_body(const SearchArgs& a, const uint32_t bid, const uint32_t nb)
{
  constexpr int NG = BLOCK / GS;
  __shared__ double lds_m2[SD][NG];
  __shared__ uint32_t lds_ref[SD][NG];

  const uint32_t chunk = xcd_chunk(bid, nb);
  const int grp = threadIdx.x / GS, sub = threadIdx.x & (GS - 1);
  const size_t gg = (size_t)bid * NG + grp;     // overflow slot of the group

  LaneStack<NG, SD> st;
  st.l_m2 = &lds_m2[0][grp];
  st.l_ref = &lds_ref[0][grp];
  st.g_m2 = a.ovf_m2 ? a.ovf_m2 + gg : nullptr;
  st.g_ref = a.ovf_ref ? a.ovf_ref + gg : nullptr;
  st.gstride = (size_t)nb * NG;
  st.sp = 0;

  const TreeDev& T = a.T;
  const double4* __restrict__ nodes = reinterpret_cast<const double4*>(T.nodes);
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(T.pts);
  // the argument block sits behind an opaque pointer (kernarg_block): what the loops need is fetched once, here
  const char* const t_hot = reinterpret_cast<const char*>(T.hot);
  const char* const t_grp = reinterpret_cast<const char*>(T.grp);
  // null unless TDTK_FAT_SMALL=1 (the launcher clears the pointer)
  const char* const t_fat = reinterpret_cast<const char*>(T.fat);
  const LeafEntry* const t_leaf_tab = T.leaf_tab;
  const uint32_t t_cb = T.cb, t_cmask = T.cmask, t_root = T.root_ref;
  const float t_absmax = T.absmax;

  const size_t per = (a.n + nb - 1) / nb;
  const size_t lo = (size_t)chunk * per;
  size_t hi = lo + per;
  if (hi > a.n) hi = a.n;

  Mat4 pend;
  bool has_pending = a.has_pending != 0;
#ifdef TDTK_LAB
  if constexpr (LOOP) {
    has_pending = a.loop_iter > 0;
    if (has_pending && !loop_prologue<BLOCK>(a, bid, pend)) return;      // (the loop has ended)
  } else
#endif
  {
    if (has_pending) pend = a.pending;
  }
  for (size_t base = lo; base < hi; base += NG) {
    const size_t i = base + grp;
    if (i >= hi) continue;
    double tx = a.x[i], ty = a.y[i], tz = a.z[i];
    if (has_pending) {  // Scan::transformReduced fused in (scan.cc:851-875); every lane computes, one writes
      dev_xf3_inplace(pend, tx, ty, tz);
      if (sub == 0) { a.x[i] = tx; a.y[i] = ty; a.z[i] = tz; }
      if (a.nx && sub == 0) {
        double px = a.nx[i], py = a.ny[i], pz = a.nz[i];
        dev_xf3normal(pend, px, py, pz);
        a.nx[i] = px; a.ny[i] = py; a.nz[i] = pz;
      }
    }
    if (a.skip && a.skip[i]) {      // -R: not drawn this pass (searchTree.cc:118); the whole lane group leaves together
      if (sub == 0) { a.kpos[i] = -1; if (a.d2) a.d2[i] = a.maxd2; }
      continue;
    }
    double qx = tx, qy = ty, qz = tz;
    if (a.has_inv) dev_xf3(a.inv, tx, ty, tz, qx, qy, qz);  // searchTree.cc:122
    double best = warm_radius(a, i, qx, qy, qz);
    int bk = -1;
    uint32_t cur = t_root;
    st.sp = 0;
    BoxF32 bx;
    bx.set_query(qx, qy, qz, t_absmax);
    bx.set_radius(best);
    for (;;) {
#ifdef TDTK_LAB
#include "lab_fat_small.inc"   // if (t_fat != nullptr): two tree levels per round trip (a measured negative)
#endif
      while (!(cur & REF_LEAF)) {
        bool need_pop = false;
        uint32_t next = REF_DONE;
        {
          const char* hp = t_hot + (size_t)cur * sizeof(KdHot);
          const float4 b0 = *reinterpret_cast<const float4*>(hp);
          const float2 b1 = *reinterpret_cast<const float2*>(hp + 16);
          double2 sc = *reinterpret_cast<const double2*>(hp + 32);
          TDTK_PIN_V64(sc.x); TDTK_PIN_V64(sc.y);
          const float a32 = fmaxf(fmaxf(fabsf(bx.qx - b0.x) - b0.w, fabsf(bx.qy - b0.y) - b1.x), fabsf(bx.qz - b0.z) - b1.y);
          bool prune = a32 >= bx.thi;
          if (__builtin_expect(!prune && !(a32 < bx.tlo), 0)) {
            const double4 n0 = nodes[(size_t)cur * 2];
            const double4 n1 = nodes[(size_t)cur * 2 + 1];
            prune = box_prunes_exact(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, qx, qy, qz, best);
          }
          if (prune) need_pop = true;
          else next = descend<NG, SD>(sc.x, (uint32_t)__double2loint(sc.y), (uint32_t)__double2hiint(sc.y), qx, qy, qz, best, st);
        }
        if (need_pop) {
          next = REF_DONE;
          while (st.sp > 0) {
            --st.sp;
            uint32_t r; double m2;
            st.top(r, m2);
            if (m2 < best) { next = r; break; }
          }
        }
        cur = next;
      }
      if (cur == REF_DONE) break;
      {
        const uint32_t v = cur & REF_VAL;
        int start, count;
        if (t_leaf_tab) {
          const LeafEntry le = t_leaf_tab[v];
          start = le.start; count = le.count;
        } else {
          start = (int)(v >> t_cb);
          count = (int)(v & t_cmask);
        }
        const double4* __restrict__ P = pts + start;
        const double inf = __longlong_as_double(0x7ff0000000000000ll);
        if (GS == 4 && t_grp != nullptr && count <= 8 * GS) {
          // Bucket groups (see search_refill_body): the four lanes of the query take two shadow groups each, so a bucket
          // of up to 32 points is filtered in ONE round trip; what can still win is tested in fp64, four candidates per
          // trip, in bucket order (group minimum with the lowest index among equals = the serial strict '<').  A small
          // scan's search lasts as long as the dependent chain of its slowest query; this takes the 4 .. 8 trips of a
          // bucket scan out of that chain.
          typedef float v2f __attribute__((ext_vector_type(2)));
          const char* gb = t_grp;
          const uint32_t go = (uint32_t)(start >> 2) * 48u;
          const uint32_t glast = go + (uint32_t)((count - 1) >> 2) * 48u;
          float sv[8];
          const v2f qxx = {bx.qx, bx.qx}, qyy = {bx.qy, bx.qy}, qzz = {bx.qz, bx.qz};
          float4 X[2], Y[2], Z[2];
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const uint32_t gk = min(go + 48u * (uint32_t)(sub + GS * k), glast);
            X[k] = *reinterpret_cast<const float4*>(gb + gk);
            Y[k] = *reinterpret_cast<const float4*>(gb + gk + 16);
            Z[k] = *reinterpret_cast<const float4*>(gb + gk + 32);
          }
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const v2f dxa = (v2f){X[k].x, X[k].y} - qxx, dxb = (v2f){X[k].z, X[k].w} - qxx;
            const v2f dya = (v2f){Y[k].x, Y[k].y} - qyy, dyb = (v2f){Y[k].z, Y[k].w} - qyy;
            const v2f dza = (v2f){Z[k].x, Z[k].y} - qzz, dzb = (v2f){Z[k].z, Z[k].w} - qzz;
            const v2f sa = __builtin_elementwise_fma(dza, dza, __builtin_elementwise_fma(dya, dya, dxa * dxa));
            const v2f sb = __builtin_elementwise_fma(dzb, dzb, __builtin_elementwise_fma(dyb, dyb, dxb * dxb));
            sv[4 * k] = sa.x; sv[4 * k + 1] = sa.y; sv[4 * k + 2] = sb.x; sv[4 * k + 3] = sb.y;
          }
          // which of this lane's eight slots are points of the bucket (the rest repeat the last group)
          unsigned valid = 0u;
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int p0 = 4 * (sub + GS * k);
            const int nv = count - p0;                       // slots p0 .. p0 + 3
            valid |= ((nv >= 4) ? 15u : (nv > 0 ? ((1u << nv) - 1u) : 0u)) << (4 * k);
          }
          float smin = __builtin_inff();
#pragma unroll
          for (int j = 0; j < 8; j++) smin = fminf(smin, ((valid >> j) & 1u) ? sv[j] : __builtin_inff());
#pragma unroll
          for (int o = 1; o < GS; o <<= 1) smin = fminf(smin, __shfl_xor(smin, o, GS));
          unsigned surv = 0u;
          if (!(smin >= bx.pthr)) {
            const float rub = __builtin_amdgcn_sqrtf(smin) * 1.000001f + bx.ec;
            const float thr = fminf(bx.pthr, bx.reject_from(rub));
            unsigned m = 0u;
#pragma unroll
            for (int j = 0; j < 8; j++) m |= ((sv[j] >= thr) ? 0u : 1u) << j;
            m &= valid;
            unsigned mine = ((m & 15u) << (4 * sub)) | ((m >> 4) << (4 * (sub + GS)));
#pragma unroll
            for (int o = 1; o < GS; o <<= 1) mine |= (unsigned)__shfl_xor((int)mine, o, GS);
            surv = mine;
          }
          while (surv) {
            unsigned t = surv;                               // lane `sub` takes the sub-th lowest candidate
#pragma unroll
            for (int k = 0; k < GS - 1; k++) if (k < sub) t &= t - 1u;
            double d = inf;
            int jj = 0x7fffffff;
            if (t) {
              jj = __builtin_ctz(t);
              const double4 p = P[jj];
              const double dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
              d = dx * dx + dy * dy + dz * dz;
            }
#pragma unroll
            for (int o = 1; o < GS; o <<= 1) {               // group minimum, lowest index among equals
              const double od = __shfl_xor(d, o, GS);
              const int oj = __shfl_xor(jj, o, GS);
              if (od < d || (od == d && oj < jj)) { d = od; jj = oj; }
            }
            if (d < best) { best = d; bk = start + jj; }
#pragma unroll
            for (int k = 0; k < GS; k++) surv &= surv - 1u;  // (0 & anything stays 0)
          }
        } else
        for (int j0 = 0; j0 < count; j0 += GS) {
          const int j = j0 + sub;
          double d = inf;
          if (j < count) {
            const double4 p = P[j];
            const double dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
            d = dx * dx + dy * dy + dz * dz;
          }
          int jj = j;
#pragma unroll
          for (int o = 1; o < GS; o <<= 1) {           // group minimum, lowest index among equals
            const double od = __shfl_xor(d, o, GS);
            const int oj = __shfl_xor(jj, o, GS);
            if (od < d || (od == d && oj < jj)) { d = od; jj = oj; }
          }
          if (d < best) { best = d; bk = start + jj; }
        }
        bx.set_radius(best);
      }
      cur = REF_DONE;
      while (st.sp > 0) {
        --st.sp;
        uint32_t r; double m2;
        st.top(r, m2);
        if (m2 < best) { cur = r; break; }
      }
      if (cur == REF_DONE) break;
    }
    if (sub == 0) {
      a.kpos[i] = bk;
      if (a.d2) a.d2[i] = best;
    }
  }
  if constexpr (FUSE) chunk_pair_sums<BLOCK>(a, lo, hi, bid);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

template <int BLOCK>
__device__ __forceinline__ void chunk_pair_sums(const SearchArgs& a, size_t lo, size_t hi, uint32_t row)
{
  constexpr int NW = BLOCK / WAVE;
  __shared__ double fred[NW][ACC_DD];
  // hits and moved coordinates of the chunk were written by threads of this workgroup: the barrier (with its
  // workgroup-scope fence) makes them visible to the others -- nothing wider is needed, or affordable
  __syncthreads();
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(a.T.pts);
  double acc[ACC_DD];
#pragma unroll
  for (int k = 0; k < ACC_DD; k++) acc[k] = 0.0;
  for (size_t i0 = lo + threadIdx.x; i0 < hi; i0 += 2 * (size_t)BLOCK) {
    const size_t i1 = i0 + BLOCK;
    const int k0 = a.kpos[i0];
    const int k1 = (i1 < hi) ? a.kpos[i1] : -1;
    double4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    double t0x = 0, t0y = 0, t0z = 0, t1x = 0, t1y = 0, t1z = 0;
    if (k0 >= 0) { c0 = pts[k0]; t0x = a.x[i0]; t0y = a.y[i0]; t0z = a.z[i0]; }
    if (k1 >= 0) { c1 = pts[k1]; t1x = a.x[i1]; t1y = a.y[i1]; t1z = a.z[i1]; }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if ((u ? k1 : k0) < 0) continue;
      const double4 c = u ? c1 : c0;
      const double tx = u ? t1x : t0x, ty = u ? t1y : t0y, tz = u ? t1z : t0z;
      double mx, my, mz;
      dev_xf3(a.A, c.x, c.y, c.z, mx, my, mz);  // searchTree.cc:147
      const double px = mx - tx, py = my - ty, pz = mz - tz;
      acc[ACC_N] += 1.0;
      acc[ACC_SUM] += px * px + py * py + pz * pz;
      const double m0 = mx - a.shift[0], m1 = my - a.shift[1], m2 = mz - a.shift[2];
      const double d0 = tx - a.shift[0], d1 = ty - a.shift[1], d2 = tz - a.shift[2];
      acc[ACC_SM + 0] += m0; acc[ACC_SM + 1] += m1; acc[ACC_SM + 2] += m2;
      acc[ACC_SD + 0] += d0; acc[ACC_SD + 1] += d1; acc[ACC_SD + 2] += d2;
      acc[ACC_P + 0] += m0 * d0; acc[ACC_P + 1] += m0 * d1; acc[ACC_P + 2] += m0 * d2;
      acc[ACC_P + 3] += m1 * d0; acc[ACC_P + 4] += m1 * d1; acc[ACC_P + 5] += m1 * d2;
      acc[ACC_P + 6] += m2 * d0; acc[ACC_P + 7] += m2 * d1; acc[ACC_P + 8] += m2 * d2;
    }
  }
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if (hi - lo <= (size_t)WAVE) {
    // a chunk of one wave's worth of queries (the four-lanes-per-query kernel on a small scan): the first wave alone
    // holds everything -- butterfly sums, lane k stores column k, no second barrier and no LDS pass
    if (wv != 0) return;
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < ACC_DD; k++) {
      double v = acc[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
      if (lane == k) mine = v;
    }
#ifdef TDTK_LAB
    if (a.loop != nullptr) {          // inside the host-free loop: [column][row], what the next launch's prologue reads coalesced
      if (lane < ACC_DD) a.partials[(size_t)lane * a.loop_rows + row] = mine;
      return;
    }
#endif
    a.partials[(size_t)row * ACC_TOTAL + lane] = mine;                       // columns 0 .. 63 (zero from ACC_DD on)
    if (lane + WAVE < ACC_TOTAL) a.partials[(size_t)row * ACC_TOTAL + WAVE + lane] = 0.0;
    return;
  }
#pragma unroll
  for (int k = 0; k < ACC_DD; k++) {
    const double sk = wave_sum(acc[k]);
    if (lane == 0) fred[wv][k] = sk;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ACC_TOTAL; k += BLOCK) {
    double sk = 0.0;
    if (k < ACC_DD)
      for (int w = 0; w < NW; w++) sk += fred[w][k];
#ifdef TDTK_LAB
    if (a.loop != nullptr) { if (k < ACC_DD) a.partials[(size_t)k * a.loop_rows + row] = sk; continue; }     // (see above)
#endif
    a.partials[(size_t)row * ACC_TOTAL + k] = sk;
  }
}
__device__ __forceinline__ unsigned long long wave_sum_u(unsigned v)
{
  unsigned long long t = v;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, WAVE);
  return t;
}

template <int BLOCK, int SD, int GS = 8, bool FUSE = false, bool LOOP = false>
__global__ void __launch_bounds__(BLOCK) k_search_g8(const SearchArgs a_by_value)
{
  (void)a_by_value;
  search_g8_body<BLOCK, SD, GS, FUSE, LOOP>(kernarg_block<SearchArgs>(), blockIdx.x, gridDim.x);
}
template <int BLOCK, int SD, int GS>
__global__ void __launch_bounds__(BLOCK) k_search_g8_multi(const SearchArgs* __restrict__ args, const uint32_t* __restrict__ base,
                                                           int nbatch)
{
  int l = 0;
  while (l + 1 < nbatch && blockIdx.x >= base[l + 1]) ++l;
  l = __builtin_amdgcn_readfirstlane(l);
  const uint32_t b0 = base[l], b1 = base[l + 1];
  search_g8_body<BLOCK, SD, GS>(args[l], blockIdx.x - b0, b1 - b0);
}

// ------------------------------------------------------------------------------------------
// k_search_refill: same per-lane traversal, but a wave owns a.qpw consecutive (sorted) queries and a
// lane that finishes its query immediately takes the next one of the wave's slab ("persistent
// lanes").  Motivation (PMC, profiles/r01_pmc_bench.json): k_search issues ~6800 VALU
// instructions per wave for ~1400 per lane-query -- the SIMDs are ~85 % busy issuing mostly
// masked-off fp64 instructions, because queries of one wave need different numbers of buckets
// (1..8) and the wave runs as long as its slowest lane.  Refilling keeps the lanes occupied.
// Results are written by query index, so the processing order is irrelevant.
// ------------------------------------------------------------------------------------------
// COUNT: also tally the internal nodes / buckets / bucket points each query visits (the traversal is the same one,
// so the counts are those of exactly this launch: warm radius, fused transform and all).
// FUSE: retire-time accumulation of the base pair sums (n, sum|p1-p2|^2, sum p1, sum p2, sum p1 p2^T about
// `shift`): a lane that finishes a query has the hit in hand and its bucket still in L1/L2, so the separate
// k_accum pass over (x, y, z, kpos, pts[kpos]) disappears.  Per-lane fp64 accumulators in registers (their order
// is the lane's retire order, which is a function of the traversal alone -> run-to-run bit-identical), one wave
// reduction at the end, one row per workgroup for k_final.  Pairing mode 0 / base block only.
// DYN (TDTK_SEARCH_VARIANT=30, a measured negative): the wave does not own a fixed slab; it draws `a.slab` consecutive
// sorted queries at a time from a work queue (one atomic counter per eighth of the scan, starting with the eighth of
// the XCD the wave really runs on and moving on to the others when that is empty), so that it drains only once, at
// the end of the launch, instead of at the end of every slab.  Two things defeat it: (a) a device-scope atomic on a
// contended counter costs ~0.6 us and they serialise per counter (1M queries, 64-query draws: 0.93 ms against 0.27 ms
// static; 256-query draws: 0.58 ms), and (b) the premise is wrong at this size -- a chip full of resident waves
// (7168) leaves 1M queries only ~140 per wave, i.e. MORE drains per query than the static 224..256-query slabs.
// diagnostics (TDTK_WAVE_TRACE=<launch index>): start / end time (100 MHz) and XCD of every wave of one launch
#define WTRACE_MAX 32768u
#ifdef TDTK_LAB
__device__ unsigned long long g_wtrace[3 * WTRACE_MAX];
#else
__device__ unsigned long long g_wtrace[3];     // (never touched: a.trace is a lab switch)
#endif

// the body of k_search_refill for workgroup `bid` of the `nb` that search one batch of queries (the kernel proper and
// the several-batches-in-one-launch kernel below share it)
template <int BLOCK, int SD, int THRESH, int WPS, bool COUNT, int FUSE, bool DYN, bool ORDER = !DYN, int PTS = 4, int PROBE = 0, bool FAT = false,
          int ORD_MAX = 256, bool LAZY = false, int TOP = 0, bool SHARE = false, bool PIPE = false, bool DEFER = false>
__device__ __forceinline__ void search_refill_body(const SearchArgs& a, const uint32_t bid, const uint32_t nb)
{
  __shared__ uint4 lds_stk[SD][BLOCK];
  // TOP > 0 (round 4): the first TOP hot records -- the tree's upper levels, the array is breadth-first -- are staged in
  // LDS once per workgroup, and a visit of one of them reads LDS instead of the vector L1.  A query's first descent is 17
  // dependent visits of which the upper ten are the same few hundred records for everybody: they were 47 % of the node
  // visits' L1 tag look-ups (the resource this kernel is closest to saturating) and of their round trips.  Same records,
  // same tests, same order.  One copy serves all waves of the workgroup, hence the big workgroups of these instantiations.
  __shared__ uint4 lds_top[TOP > 0 ? 3 * TOP : 1];
  uint32_t top_n = 0;
  if constexpr (TOP > 0) {
    top_n = min((uint32_t)TOP, a.T.n_hot);
    const uint4* __restrict__ h16 = reinterpret_cast<const uint4*>(a.T.hot);
    for (uint32_t k = threadIdx.x; k < 3u * top_n; k += BLOCK) lds_top[k] = h16[k];
    __syncthreads();
  }

  const size_t gl = (size_t)bid * BLOCK + threadIdx.x;
  const unsigned lane = threadIdx.x & (WAVE - 1);
  // diagnostics (TDTK_WAVE_TRACE): the start time goes to memory right away -- kept in registers it is live across the
  // whole kernel, and the compiler spilled it
  if (kLab && a.trace && lane == 0) {
    const uint32_t wid0 = bid * (BLOCK / WAVE) + threadIdx.x / WAVE;
    if (wid0 < WTRACE_MAX) g_wtrace[3 * wid0] = wall_clock64();
  }
  // lazy scan moves: the chain (at most LAZY_MAX matrices, the host sees to that) is staged in LDS once per workgroup
  __shared__ double lds_mv[LAZY ? LAZY_MAX * 16 : 1];
  if constexpr (LAZY) {
    if (a.nmoves) {
      const double* gm = reinterpret_cast<const double*>(a.moves);
      for (int k = threadIdx.x; k < a.nmoves * 16; k += BLOCK) lds_mv[k] = gm[k];
      __syncthreads();
    }
  }
  LaneStackQ<BLOCK, SD> st;
  st.l_e = &lds_stk[0][threadIdx.x];
  st.g_m2 = a.ovf_m2; st.g_ref = a.ovf_ref; st.gcol = gl;
  st.gstride = (size_t)nb * BLOCK;
  st.sp = 0;

  const TreeDev& T = a.T;
  const double4* __restrict__ nodes = reinterpret_cast<const double4*>(T.nodes);
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(T.pts);
  // The argument block is read through an opaque pointer (see k_search_refill): what the inner loops and every retire
  // need is fetched once, here, and stays in SGPRs; the rest (matrices, query arrays) is read where a lane takes a query.
  const LeafEntry* const t_leaf_tab = T.leaf_tab;
  const uint32_t t_cb = T.cb, t_cmask = T.cmask;
  int* const a_kpos = a.kpos;
  double* const a_d2 = a.d2;
  unsigned char* const a_cost = a.cost;
  const char* const t_grp = reinterpret_cast<const char*>(T.grp);
  const char* const t_fat = reinterpret_cast<const char*>(T.fat);
  // "expensive queries first": the order in which a piece of the slab is handed out (offsets within the piece), by the
  // number of buckets each query visited in the previous pass.  Lanes that work on queries of similar length at the same
  // time waste fewer of each other's issue slots, and the drain at the end of the piece is over cheap queries
  // (tools/sim/wave_sched.c: -7 % wave instructions on a converged pair, -14 % mid-ICP with the true costs as the key).
  // (ORD_MAX: the longest piece that can be ordered -- 256 for the single pass, 320 for the link passes' half slabs)
  typedef typename std::conditional<(ORD_MAX > 256), unsigned short, unsigned char>::type ord_t;
  __shared__ ord_t lds_order[ORDER ? BLOCK / WAVE : 1][ORDER ? ORD_MAX : 4];
  ord_t* const my_order = lds_order[ORDER ? threadIdx.x / WAVE : 0];
  bool ordered = false;
  size_t piece0 = 0;
  auto order_piece = [&](size_t p0, size_t p1) {
    ordered = false;
    piece0 = p0;
    if (!ORDER || !a.use_cost || !a.cost || p1 <= p0 || p1 - p0 > (size_t)ORD_MAX) return;
    const uint32_t cntp = (uint32_t)(p1 - p0);
    // eight classes between the cheapest and the most expensive query of the piece (the key is node visits + 4 per
    // bucket of the previous pass; its range depends on the depth of the tree and on how far the poses still are)
    int cls[ORD_MAX / WAVE];
    unsigned cvv[ORD_MAX / WAVE];
    unsigned mn = 255u, mx = 0u;
#pragma unroll
    for (int r = 0; r < ORD_MAX / WAVE; r++) {
      const uint32_t o = (uint32_t)r * WAVE + lane;
      cvv[r] = (o < cntp) ? (unsigned)a.cost[p0 + o] : 0u;
      if (o < cntp) { mn = min(mn, cvv[r]); mx = max(mx, cvv[r]); }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn = min(mn, (unsigned)__shfl_xor((int)mn, off, WAVE));
      mx = max(mx, (unsigned)__shfl_xor((int)mx, off, WAVE));
    }
    const unsigned span = (mx > mn) ? mx - mn + 1u : 1u;
#pragma unroll
    for (int r = 0; r < ORD_MAX / WAVE; r++) {
      const uint32_t o = (uint32_t)r * WAVE + lane;
      cls[r] = (o < cntp) ? (int)min(((cvv[r] - mn) * 8u) / span, 7u) : -1;
    }
    uint32_t base = 0;
    for (int k = 7; k >= 0; k--) {
#pragma unroll
      for (int r = 0; r < ORD_MAX / WAVE; r++) {
        if ((uint32_t)r * WAVE >= cntp) continue;
        const unsigned long long m = __ballot(cls[r] == k);
        if (cls[r] == k) my_order[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (ord_t)(r * WAVE + lane);
        base += (uint32_t)__popcll(m);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    ordered = true;
  };
  size_t next_q, end_q;  // wave-uniform: the queries this wave may still hand to its lanes
  // SHARE (round 4): the waves of a workgroup hand out their slabs TOGETHER -- one cursor in LDS over the workgroup's
  // stretch of the sorted scan (the waves' slabs are consecutive), piece after piece, each piece in the order its owner
  // sorted it.  A wave that runs ahead of its neighbours takes more of the stretch; the launch no longer waits for the
  // unluckiest of 4096 independent slabs (sigma = 9 % of a wave's life, TDTK_WAVE_TRACE) but for the unluckiest of the
  // workgroups.  Who searches a query does not change its result; the sums pass (FUSE 3) still covers each wave's own slab.
  __shared__ uint32_t lds_cursor;
  size_t wg0 = 0;
  uint32_t wg_len = 0;
  bool ord_on = false;
  size_t sub = 0, reg0 = 0, pstride = 0, slab_end = a.n;
  int nph = 1;       // pieces of this wave's slab
  bool exhausted = false;
  int phase = 0;
  uint32_t xq = 0, tried = 0, nslab = 0, per_x = 0;
  if (DYN) {
    next_q = end_q = 0;
    nslab = (uint32_t)((a.n + (size_t)a.slab - 1) / (size_t)a.slab);
    per_x = (nslab + 7u) >> 3;
    xq = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;   // HW_REG_XCC_ID[3:0]: speed only
    // the counters of the NEXT launch on this stream (the two sets alternate): nobody reads them during this one
    if (bid == 0 && threadIdx.x < 8) a.q_ctr_next[threadIdx.x] = 0u;
  } else {
    // a wave owns qpw consecutive sorted queries (256 unless the batch is so large that the grid is capped)
    // ... handed out in `phases` pieces: the waves of one XCD (workgroup b runs on XCD b % 8) first cover the first
    // 1/phases of the XCD's eighth of the scan together, then the next, so that the leaves the XCD's 4 MB L2 has to
    // hold at any one time are those under 1/phases of the eighth (the 32-byte points of an eighth of a 1M-point
    // model alone are 4 MB)
    const uint32_t wpx = (nb >> 3) * (BLOCK / WAVE);                      // waves per XCD
    // (wave-uniform, and said so: the slab geometry derived from it then lives in scalar registers)
    const uint32_t wx = (uint32_t)__builtin_amdgcn_readfirstlane((int)((bid >> 3) * (BLOCK / WAVE) + threadIdx.x / WAVE));   // this wave among them
    if (kLab && a.pool_slab) {
      // Static slab + pool: the waves of a launch do not finish together -- at 1M queries the first is done after 60 %
      // of the launch, the median after 77 % (TDTK_WAVE_TRACE) -- so only part of an XCD's region is dealt out in
      // advance and the rest is drawn in small pieces by whichever wave runs dry.  One counter per XCD, touched first by
      // the waves of that XCD (the real XCC_ID) and by the others only once their own pool is dry.  (Round 2 used
      // workgroup-scope atomics here on the assumption that no other XCD ever touches a counter, which made completeness
      // depend on XCC_ID; agent scope since round 3 -- a measured negative either way.)
      xq = (uint32_t)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u;
      const size_t r0 = (size_t)(bid & 7u) * a.region;
      const size_t rend = (r0 + a.region < a.n) ? r0 + a.region : a.n;
      next_q = r0 + (size_t)wx * (size_t)a.qpw;
      end_q = next_q + (size_t)a.qpw;
      if (next_q > rend) next_q = rend;
      if (end_q > rend) end_q = rend;
      if (bid == 0 && threadIdx.x < 8) a.q_ctr_next[threadIdx.x] = 0u;
    } else {
      if (kLab && !LAZY && a.bounds) {
        // Slabs of equal COST instead of equal length (SearchArgs::bounds, k_slab_bounds below): wave w of the launch --
        // numbered XCD by XCD, so that an XCD's waves still share a contiguous stretch of the sorted scan -- owns the
        // queries [bounds[w], bounds[w + 1]), handed out in pieces of ORD_MAX (each ordered by cost as before)
        const uint32_t w = (bid & 7u) * wpx + wx;
        reg0 = (size_t)a.bounds[w];
        slab_end = (size_t)a.bounds[w + 1];
        if (slab_end > a.n) slab_end = a.n;
        if (reg0 > slab_end) reg0 = slab_end;
        sub = (size_t)ORD_MAX;
        pstride = sub;
        nph = (int)((slab_end - reg0 + sub - 1) / sub);
        if (nph < 1) nph = 1;
      } else {
        sub = (size_t)(a.qpw / a.phases);
        reg0 = (size_t)(bid & 7u) * wpx * (size_t)a.qpw + (size_t)wx * sub;
        pstride = (size_t)wpx * sub;
        slab_end = a.n;
        nph = a.phases;
      }
      next_q = reg0;
      end_q = next_q + sub;
      if (next_q > slab_end) next_q = slab_end;
      if (end_q > slab_end) end_q = slab_end;
      order_piece(next_q, end_q);
      if constexpr (SHARE) {
        // (one piece per wave -- the launcher sets phases = 1 --, so the workgroup's slabs are one stretch of the scan)
        wg0 = (size_t)(bid & 7u) * wpx * (size_t)a.qpw + (size_t)((bid >> 3) * (BLOCK / WAVE)) * sub;
        if (wg0 > a.n) wg0 = a.n;
        const size_t len = (size_t)(BLOCK / WAVE) * sub;
        wg_len = (uint32_t)((wg0 + len <= a.n) ? len : a.n - wg0);
        ord_on = ORDER && a.use_cost && a.cost && sub <= (size_t)ORD_MAX;
        if (threadIdx.x == 0) lds_cursor = 0u;
        __syncthreads();          // every wave's order (lds_order[w]) and the cursor are in place
      }
    }
  }

  if constexpr (LAZY && !DYN) {
    if (a.nmoves) {
      // Lazy scan moves (SearchArgs::moves): before it searches, the wave moves ITS slab -- the unmoved points from
      // sx / sy / sz, the queued transforms in order (the arithmetic of k_transform_chain_batch: dev_xf3_inplace per
      // matrix), the result into x / y / z, which are this link's own (the scan's spare arrays for the link that owns the
      // update, a scratch copy for any other link of the launch that reads the scan).  Consecutive lanes, consecutive
      // queries, every lane busy: ten trips for a slab of 640.  Everything below reads x / y / z, as without moves; a
      // slab is a whole number of 128-byte lines and nobody else touches it, so a fence of workgroup scope is all the
      // wave needs to see its own stores (like the sums pass further down).
      // (Applied where a lane takes a query instead -- matrices through the scalar cache at every hand-out, 8-byte
      // stores in cost order -- the launch of 84 links took 10.3 ms against 9.5 + 0.54 for the pass it replaces.)
      // (five trips' loads are issued before the first of them is used: two memory round trips per slab instead of ten --
      // one trip at a time the pre-pass cost the 84-link launch 0.6 ms, as much as the pass it replaces)
      const uint32_t slab0 = (uint32_t)a.qpw, sub0 = (uint32_t)sub;
      const bool nrm = a.nx != nullptr;          // (set for the owner of the update only: normals move in place)
      constexpr int MT = 5;
      for (uint32_t j0 = 0; j0 < slab0; j0 += MT * WAVE) {
        size_t q[MT];
        bool ok[MT];
        double px[MT], py[MT], pz[MT];
#pragma unroll
        for (int u = 0; u < MT; u++) {
          uint32_t j = j0 + (uint32_t)u * WAVE + lane, ph = 0;
          const bool in = j < slab0;
          while (j >= sub0 && in) { j -= sub0; ph++; }
          q[u] = reg0 + (size_t)ph * pstride + j;
          ok[u] = in && q[u] < a.n;
          px[u] = py[u] = pz[u] = 0.0;
          if (ok[u]) { px[u] = a.sx[q[u]]; py[u] = a.sy[q[u]]; pz[u] = a.sz[q[u]]; }
        }
        for (int k = 0; k < a.nmoves; k++) {
          const double* m = lds_mv + 16 * k;
          const double m0 = m[0], m1 = m[1], m2 = m[2], m4 = m[4], m5 = m[5], m6 = m[6], m8 = m[8], m9 = m[9], m10 = m[10],
                       m12 = m[12], m13 = m[13], m14 = m[14];
#pragma unroll
          for (int u = 0; u < MT; u++) {
            const double xn = px[u] * m0 + py[u] * m4 + pz[u] * m8;      // dev_xf3_inplace
            const double yn = px[u] * m1 + py[u] * m5 + pz[u] * m9;
            const double zn = px[u] * m2 + py[u] * m6 + pz[u] * m10;
            px[u] = xn + m12; py[u] = yn + m13; pz[u] = zn + m14;
          }
        }
#pragma unroll
        for (int u = 0; u < MT; u++)
          if (ok[u]) { a.x[q[u]] = px[u]; a.y[q[u]] = py[u]; a.z[q[u]] = pz[u]; }
        if (nrm) {
#pragma unroll
          for (int u = 0; u < MT; u++) {
            if (!ok[u]) continue;
            double ux = a.nx[q[u]], uy = a.ny[q[u]], uz = a.nz[q[u]];
            for (int k = 0; k < a.nmoves; k++) {
              const double* m = lds_mv + 16 * k;
              const double an = ux * m[0] + uy * m[1] + uz * m[2];       // dev_xf3normal
              const double bn = ux * m[4] + uy * m[5] + uz * m[6];
              const double cn = ux * m[8] + uy * m[9] + uz * m[10];
              ux = an; uy = bn; uz = cn;
            }
            a.nx[q[u]] = ux; a.ny[q[u]] = uy; a.nz[q[u]] = uz;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    }
  }
  uint32_t cur = REF_DONE;
  double best = 0.0, qx = 0, qy = 0, qz = 0;
  int bk = -1;
  size_t qi = 0;
  bool have = false;
  // PIPE (round 4): a lane that takes a query does not make the wave wait for it.  The loads of its coordinates and of
  // its previous hit's index are ISSUED at the hand-out and land in the registers the lane is not using (qx / qy / qz, bk;
  // cur = REF_STAGE1); the lane joins the node walk one trip later, and the two round trips -- the coordinates, then the
  // previous hit's point for the warm start -- are covered by the trip the other lanes make meanwhile (a hand-out used
  // to stall all 64 lanes for both, ten times per slab).
  // (Staged that way with registers of their own -- the stored point, the previous hit's point, a stage counter -- the
  // kernel needed 133 VGPRs, i.e. lost its fourth wave per SIMD, or spilled; the form below carries nothing extra.)
  BoxF32 bx;
  bx.qx = bx.qy = bx.qz = 0.f; bx.delta = 0.f; bx.thi = 0.f; bx.tlo = 0.f; bx.ec = 0.f; bx.pthr = 0.f; bx.qs = 0.f; bx.pthr16 = 0.f;
  // the query on the tree's 16-bit grid (bucket_scan_q16): (x, y) and (z, z) as packed int16, and whether it lies inside the grid
  // (the single-pass kernel only: the several-links launch -- LAZY, 118-120 VGPRs with either filter, four waves per SIMD, its
  // vector ALUs busier -- is SLOWER with it: 84 links 9.9 -> 10.8 ms, six links of 10M queries 11.6 -> 12.6; gpurun_out/r5g)
  constexpr bool USE_Q16 = BUCKET_Q16 && !LAZY;
  uint32_t q16xy = 0u, q16zz = 0u;
  bool q16in = false;
  const char* const t_q16 = USE_Q16 ? reinterpret_cast<const char*>(T.q16) : nullptr;
  auto q16_query = [&]() {
    if (USE_Q16 && t_q16) {
      const SearchArgs* ap = &a;
      asm volatile("" : "+s"(ap));      // (read where a lane takes a query, like the matrices: not hoisted, not kept live)
      bool out = false;
      const double sc = ap->T.q_scale;
      const uint32_t ix = (uint32_t)q16_index(qx, ap->T.q_lo[0], sc, out) & 0xFFFFu, iy = (uint32_t)q16_index(qy, ap->T.q_lo[1], sc, out) & 0xFFFFu,
                     iz = (uint32_t)q16_index(qz, ap->T.q_lo[2], sc, out) & 0xFFFFu;
      q16xy = ix | (iy << 16); q16zz = iz | (iz << 16); q16in = !out;
      bx.qs = (float)sc * 1.0000002f;    // cells per unit, rounded up
    }
  };
  const char* __restrict__ hotb = reinterpret_cast<const char*>(T.hot);
  // ---- the quick check deferred (round 5) ----
  // A query that starts from a previous hit (SearchArgs::warm, tie > 0) makes its DIVERGENT visits without the quick check of
  // kdTreeImpl.h:360-368: 16 bytes { splitval, children } instead of the 48-byte record, one load instead of three.  What it
  // walks is the reference's walk plus subtrees the reference would have cut off there; the order is the reference's, a point
  // of such a subtree lies at d2 >= closest_d2 - tie by the very test that was skipped (tie bounds 2 E R + E^2, E the rounding of
  // fabs(q - center) - half-width against the exact face distance, R the search radius: api.cpp, search_tie), and the strict
  // `<` never takes an equal one.  So as long as no accepted point improved closest_d2 by `tie` or less -- `thin` -- every
  // acceptance is one the reference makes, in its order: same index, same d2, same ties.  A query that did accept thinly is
  // searched again when it retires, cold and with every check: the reference's own walk.  (A deferring lane skips the check in
  // wave-uniform visits too; lanes that make it -- no previous hit -- make it everywhere.)  The warm radius is the previous hit's
  // d2 + 2 tie, so that finding that very point again is not thin.  Measured with six waves per SIMD, 1M-vs-1M at the driver's
  // arguments: k_search 0.1665 -> 0.157 ms; 21.4 -> 21.7 node visits, 2.77 -> 3.12 buckets per query (the checks compiled out
  // altogether, which is not exact: 0.145-0.151).  This is an instantiation of its own (DEFER): a pass with no warm queries
  // runs the kernel without any of it.
  constexpr bool DEFER_OK = DEFER && USE_Q16 && !FAT && !PIPE && TOP == 0 && PROBE == 0;
  const char* const splitb = DEFER_OK ? reinterpret_cast<const char*>(T.split) : nullptr;
  const uint32_t split_off = (DEFER_OK && splitb != nullptr) ? (uint32_t)(splitb - hotb) : 0u;
  const double a_tie = (DEFER_OK && splitb != nullptr && t_q16 != nullptr) ? a.tie : 0.0;
  // (the two per-lane flags ride in the top bits of nbk, the query's cost counter: as lane masks of their own they were two more
  // SGPR pairs in a kernel that has none to spare -- spilled into VGPR lanes, +5 % on every launch)
  constexpr uint32_t NBK_DEFER = 0x80000000u, NBK_THIN = 0x40000000u, NBK_COST = 0x3FFFFFFFu;
  unsigned c_int = 0, c_leaf = 0, c_pts = 0, c_redo = 0;
  unsigned c_t1 = 0, c_t2 = 0;   // lab, instrumented instantiations: trips of the wave through the node walk / the bucket scan
  unsigned nbk = 0;   // buckets this lane's query has visited (the next pass's ordering key)
  // FUSE 1: the base pair sums (ACC_N .. ACC_P) at retire time; FUSE 2: n, sum and the LUM block of a graph-SLAM link
  // (acc[0] = n, [1] = sum |delta|^2, [2 .. 16] = the 15 sums of lum6Deuler.cc:143-175, [17] = sum u.delta)
  constexpr int NACC = (FUSE == 2 || FUSE == 5) ? 18 : ACC_DD;   // FUSE 5: the block of FUSE 2, added up like FUSE 3
  double acc[NACC];   // live only when FUSE
  if (FUSE) {
#pragma unroll
    for (int k = 0; k < NACC; k++) acc[k] = 0.0;
  }

  for (;;) {
    // ---- retire finished queries, hand out new ones ----
    if (DEFER_OK && cur == REF_DONE && have && (nbk & (NBK_DEFER | NBK_THIN)) == (NBK_DEFER | NBK_THIN)) {
      // accepted a point within `tie` of its closest_d2 on a walk without quick checks: again, as the reference does it
      nbk &= NBK_COST;
      if (COUNT) ++c_redo;
      cur = T.root_ref; bk = -1; st.sp = 0;
      { const SearchArgs* ap = &a; asm volatile("" : "+s"(ap)); best = ap->maxd2; }
      bx.set_radius(best);
    }
    const bool idle = (cur == REF_DONE);
    if (idle && have) {
      gstore<int>(reinterpret_cast<char*>(a_kpos), (uint32_t)qi << 2, bk);
      if (ORDER && a_cost) a_cost[qi] = (unsigned char)min(nbk & NBK_COST, 255u);   // nbk: node visits + 4 per bucket
      if (a_d2) gstore<double>(reinterpret_cast<char*>(a_d2), (uint32_t)qi << 3, best);
      have = false;
      if constexpr (FUSE == 2) if (bk >= 0) {
        const double tx = a.x[qi], ty = a.y[qi], tz = a.z[qi];
        const double4 c = *reinterpret_cast<const double4*>(reinterpret_cast<const char*>(pts) + ((uint32_t)bk << 5));
        double mx, my, mz;
        dev_xf3(a.A, c.x, c.y, c.z, mx, my, mz);  // searchTree.cc:147
        const double dx = mx - tx, dy = my - ty, dz = mz - tz;
        const double x = (mx + tx) / 2.0, y = (my + ty) / 2.0, z = (mz + tz) / 2.0;
        acc[0] += 1.0;
        acc[1] += dx * dx + dy * dy + dz * dz;
        acc[2] += x; acc[3] += y; acc[4] += z;
        acc[5] += x * x + y * y;
        acc[6] += x * x + z * z;
        acc[7] += y * y + z * z;
        acc[8] += x * y; acc[9] += x * z; acc[10] += y * z;
        acc[11] += dx; acc[12] += dy; acc[13] += dz;
        acc[14] += -z * dy + y * dz;
        acc[15] += -y * dx + x * dy;
        acc[16] += z * dx - x * dz;
        acc[17] += x * dx + y * dy + z * dz;
      }
      if constexpr (FUSE == 1) if (bk >= 0) {
        const double tx = a.x[qi], ty = a.y[qi], tz = a.z[qi];   // the (already moved) data point, world frame
        const double4 c = *reinterpret_cast<const double4*>(reinterpret_cast<const char*>(pts) + ((uint32_t)bk << 5));
        double mx, my, mz;
        dev_xf3(a.A, c.x, c.y, c.z, mx, my, mz);  // searchTree.cc:147
        const double px = mx - tx, py = my - ty, pz = mz - tz;
        acc[ACC_N] += 1.0;
        acc[ACC_SUM] += px * px + py * py + pz * pz;
        const double m0 = mx - a.shift[0], m1 = my - a.shift[1], m2 = mz - a.shift[2];
        const double d0 = tx - a.shift[0], d1 = ty - a.shift[1], d2 = tz - a.shift[2];
        acc[ACC_SM + 0] += m0; acc[ACC_SM + 1] += m1; acc[ACC_SM + 2] += m2;
        acc[ACC_SD + 0] += d0; acc[ACC_SD + 1] += d1; acc[ACC_SD + 2] += d2;
        acc[ACC_P + 0] += m0 * d0; acc[ACC_P + 1] += m0 * d1; acc[ACC_P + 2] += m0 * d2;
        acc[ACC_P + 3] += m1 * d0; acc[ACC_P + 4] += m1 * d1; acc[ACC_P + 5] += m1 * d2;
        acc[ACC_P + 6] += m2 * d0; acc[ACC_P + 7] += m2 * d1; acc[ACC_P + 8] += m2 * d2;
      }
    }
    const unsigned long long idlem = __ballot(idle);
    const unsigned long long activem = __ballot(!idle);
    const bool fill = (activem == 0 || __popcll(idlem) >= THRESH);
#ifdef TDTK_LAB
#include "lab_draws.inc"   // DYN / pool_slab: draws from a work queue or the XCD's pool (measured negatives)
#endif
    if (!DYN && !a.pool_slab && fill && next_q >= end_q && phase + 1 < nph) {
      ++phase;
      next_q = reg0 + (size_t)phase * pstride;
      end_q = next_q + sub;
      if (next_q > slab_end) next_q = slab_end;
      if (end_q > slab_end) end_q = slab_end;
      order_piece(next_q, end_q);
    }
    uint32_t sh_base = 0;
    bool sh_draw = false;
    if constexpr (SHARE) {
      if (fill && !exhausted) {
        if (lane == 0) sh_base = atomicAdd(&lds_cursor, (uint32_t)__popcll(idlem));
        sh_base = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_base);
        if (sh_base >= wg_len) exhausted = true; else sh_draw = true;
      }
    }
    if (SHARE ? sh_draw : (next_q < end_q && fill)) {
      const unsigned rank = (unsigned)__popcll(idlem & ((1ull << lane) - 1ull));
      size_t slot, mine;
      bool got;
      if constexpr (SHARE) {
        const uint32_t pos = sh_base + rank;              // position in the hand-out order of the workgroup's stretch
        got = idle && pos < wg_len;
        const uint32_t pj = pos / (uint32_t)sub, po = pos - pj * (uint32_t)sub;     // piece (= its owner's wave), offset in it
        slot = wg0 + pos;
        mine = (ord_on && got) ? wg0 + (size_t)pj * sub + (size_t)lds_order[ORDER ? pj : 0][po] : slot;
      } else {
        slot = next_q + rank;                // position in the hand-out order of the piece
        got = idle && slot < end_q;
        mine = (ORDER && got && ordered) ? piece0 + (size_t)my_order[slot - piece0] : slot;
      }
      if (PIPE && got) {
        // (requested here, used in the node walk -- see there; the registers of a lane without a query hold them meanwhile)
        bk = a.warm ? gload<int>(reinterpret_cast<const char*>(a_kpos), (uint32_t)mine << 2) : -1;
        const uint32_t m8 = (uint32_t)mine << 3;
        qx = gload<double>(reinterpret_cast<const char*>(a.x), m8); qy = gload<double>(reinterpret_cast<const char*>(a.y), m8);
        qz = gload<double>(reinterpret_cast<const char*>(a.z), m8);
        qi = mine; have = true; nbk = 0; st.sp = 0;
        cur = REF_STAGE1;
      }
      if (!PIPE && got) {
        // the previous hit (warm start) is requested with the coordinates, not behind them: one round trip less
        // (global loads / stores at 32-bit byte offsets: a scan has < 2^27 points)
        const uint32_t m8 = (uint32_t)mine << 3;
        const int kp_prev = a.warm ? gload<int>(reinterpret_cast<const char*>(a_kpos), (uint32_t)mine << 2) : -1;
        double tx = gload<double>(reinterpret_cast<const char*>(a.x), m8), ty = gload<double>(reinterpret_cast<const char*>(a.y), m8),
               tz = gload<double>(reinterpret_cast<const char*>(a.z), m8);
        if (a.has_pending) {  // Scan::transformReduced fused in (scan.cc:851-875)
          dev_xf3_inplace(a.pending, tx, ty, tz);
          gstore<double>(reinterpret_cast<char*>(a.x), m8, tx); gstore<double>(reinterpret_cast<char*>(a.y), m8, ty);
          gstore<double>(reinterpret_cast<char*>(a.z), m8, tz);
          if (a.nx) {
            double px = a.nx[mine], py = a.ny[mine], pz = a.nz[mine];
            dev_xf3normal(a.pending, px, py, pz);
            a.nx[mine] = px; a.ny[mine] = py; a.nz[mine] = pz;
          }
        }
        if (a.skip && a.skip[mine]) {
          // -R: not drawn this pass (searchTree.cc:118): the point has moved, it is no candidate; the lane stays idle
          gstore<int>(reinterpret_cast<char*>(a_kpos), (uint32_t)mine << 2, -1);
          if (ORDER && a_cost) a_cost[mine] = 0;
          if (a_d2) gstore<double>(reinterpret_cast<char*>(a_d2), m8, a.maxd2);
        } else {
        qx = tx; qy = ty; qz = tz;
        if (a.has_inv) dev_xf3(a.inv, tx, ty, tz, qx, qy, qz);  // searchTree.cc:122
        qi = mine; have = true; nbk = 0;
        cur = T.root_ref; best = warm_radius_kp(a, kp_prev, qx, qy, qz); bk = -1; st.sp = 0;
        if (DEFER_OK && a_tie > 0.0 && kp_prev >= 0 && best < a.maxd2) nbk = NBK_DEFER;
        bx.set_query(qx, qy, qz, T.absmax);
        q16_query();
        bx.set_radius(best);
        }
      }
      next_q += (size_t)__popcll(idlem);
    }
    if (__ballot(cur != REF_DONE) == 0) {
      if constexpr (SHARE) { if (exhausted) break; continue; }
      if (next_q >= end_q && (DYN ? tried >= 8u : ((kLab && a.pool_slab) ? exhausted : phase + 1 >= nph))) break;
      continue;
    }

    // ---- phase 1: walk internal nodes until this lane holds a bucket (or is finished) ----
#ifdef TDTK_LAB
#include "lab_fat_walk.inc"   // if constexpr (FAT): two tree levels per round trip (a measured negative)
#endif
    if constexpr (!FAT) while (!(cur & REF_LEAF) || (PIPE && cur == REF_STAGE1)) {
      // PIPE: a lane that has just been handed a query (cur == REF_STAGE1: its coordinates and its previous hit's index were
      // requested at the hand-out and sit in qx / qy / qz / bk) spends this trip becoming a query: the stored point moved and
      // mapped into the tree's frame, the previous hit's point requested HERE and used at the bottom of the trip, behind
      // the wait for the node records the other lanes requested meanwhile.  Nothing of it is carried around the loop.
      double w_x = 0.0, w_y = 0.0, w_z = 0.0;
      bool fin = false;
#ifdef TDTK_LAB
#include "lab_pipe_stage.inc"   // if constexpr (PIPE): a lane just handed a query becomes one during this trip
#endif
      if (!PIPE || !(cur & REF_LEAF)) {
      if (COUNT) ++c_int;
      if (COUNT && kLab) { const unsigned long long on = __ballot(true); if ((int)lane == __ffsll((long long)on) - 1) ++c_t1; }
      if (ORDER) ++nbk;     // (a key of buckets alone saves this instruction and orders no better: 0.1934 / 0.1968 against 0.1946 / 0.1923 ms)
      bool need_pop = false;
      uint32_t next = REF_DONE;
      // `cur` may carry the axis bit of the reference it came from (REF_AXIS): the 24-bit multiply below ignores it, and
      // two lanes at the same node hold the same bits (a node has one parent), so the uniformity test is unaffected
      // (Testing for a wave-uniform node only while the previous test succeeded -- re-armed by a refill -- saves two
      // instructions per divergent trip, 42.1 M -> 40.0 M per launch, and LOSES: lanes do meet again after a bucket, the
      // scalar path is taken less often, vector loads 3.03 M -> 3.33 M, k_search 0.1952 -> 0.2106 ms; gpurun_out/r3_check3.)
      const uint32_t ucur = __builtin_amdgcn_readfirstlane(cur);
      if (__all(cur == ucur)) {
        // wave-uniform visit: the 48-byte hot record through the scalar cache
        typedef const float __attribute__((address_space(4))) * const_f_ptr;
        const_f_ptr sf = (const_f_ptr)(hotb + (size_t)(ucur & REF_VAL) * sizeof(KdHot));
        const_d_ptr sd = (const_d_ptr)(sf + 8);
        const_u_ptr su = (const_u_ptr)sf;
        double s_split = sd[0];
        uint32_t s_c1 = su[10], s_c2 = su[11], s_axis = su[6];
        TDTK_PIN_S64(s_split); TDTK_PIN_S32(s_c1); TDTK_PIN_S32(s_c2); TDTK_PIN_S32(s_axis);
        bool prune = false;
        // (the quick check only for the lanes that make it: a wave whose lanes all defer it jumps over the arithmetic)
        if (!(DEFER_OK && (nbk & NBK_DEFER))) {
        const float a32 = fmaxf(fmaxf(fabsf(bx.qx - sf[0]) - sf[3], fabsf(bx.qy - sf[1]) - sf[4]), fabsf(bx.qz - sf[2]) - sf[5]);
        prune = a32 >= bx.thi;
        if (__builtin_expect(!prune && !(a32 < bx.tlo), 0)) {      // undecidable in fp32 (or not finite): the exact test
          // (one visit in a million: per-lane loads of the same record -- twelve more live SGPRs here made the compiler
          // spill in this very branch)
          const uint32_t no = (uint32_t)((cur & REF_VAL) << 6);
          const double4 n0 = gload<double4>(reinterpret_cast<const char*>(nodes), no);
          const double2 n1 = gload<double2>(reinterpret_cast<const char*>(nodes), no + 32);
          prune = box_prunes_exact(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, qx, qy, qz, best);
        }
        }
        if (prune) need_pop = true;
        else next = descend_ax(s_split, s_c1, s_c2, s_axis, qx, qy, qz, best, st);
      } else {
        // 32-bit byte offset from a scalar base: global_load with SGPR base + VGPR offset, no 64-bit address arithmetic
        // on the vector ALU (the hot array is < 4 GB by construction); v_mul_u32_u24 is full rate and drops bit 30
        const uint32_t ho = __umul24(cur, (uint32_t)sizeof(KdHot));
        float4 b0, b1;
        double2 sc;
        if constexpr (DEFER_OK) {
          // One batch of loads whatever the lanes' modes (a branch per mode would be two round trips in a row for a wave that
          // holds both): the split half for everybody, the box for the lanes that make the quick check.
          const bool dfr = (nbk & NBK_DEFER) != 0u;
          // (the split halves sit behind the hot records in ONE allocation: one scalar base, the lane picks the offset -- a lane
          // that makes the check reads its 48 bytes from one record, as before)
          sc = gload<double2>(hotb, dfr ? split_off + __umul24(cur, 16u) : ho + 32u);
          bool prune = false;
          if (!dfr) {
            b0 = gload<float4>(hotb, ho); b1 = gload<float4>(hotb, ho + 16);
            asm volatile("" : "+v"(b0.x), "+v"(b1.x));        // (both requested behind sc's load, before anything waits)
            const float a32 = fmaxf(fmaxf(fabsf(bx.qx - b0.x) - b0.w, fabsf(bx.qy - b0.y) - b1.x), fabsf(bx.qz - b0.z) - b1.y);
            prune = a32 >= bx.thi;
            if (__builtin_expect(!prune && !(a32 < bx.tlo), 0)) {      // undecidable in fp32 (or not finite): the exact test
              const uint32_t no = (uint32_t)((cur & REF_VAL) << 6);
              const double4 n0 = gload<double4>(reinterpret_cast<const char*>(nodes), no);
              const double2 n1 = gload<double2>(reinterpret_cast<const char*>(nodes), no + 32);
              prune = box_prunes_exact(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, qx, qy, qz, best);
            }
          }
          const uint32_t c1b = (uint32_t)__double2loint(sc.y), c2b = (uint32_t)__double2hiint(sc.y);
          if (prune) need_pop = true;
          else next = descend_ax(sc.x, c1b, c2b, ((c1b >> 30) & 1u) | (((c2b >> 30) & 1u) << 1), qx, qy, qz, best, st);
        } else {
        if (TOP > 0 && ho < top_n * (uint32_t)sizeof(KdHot)) {
          const char* lp = reinterpret_cast<const char*>(lds_top) + ho;
          b0 = *reinterpret_cast<const float4*>(lp);
          b1 = *reinterpret_cast<const float4*>(lp + 16);
          sc = *reinterpret_cast<const double2*>(lp + 32);
        } else {
        b0 = gload<float4>(hotb, ho);                 // cx cy cz hx
        if constexpr (HOT_NARROW == 2) {              // hy hz only: the axis from the child references' bits
          const float2 h2 = gload<float2>(hotb, ho + 16);
          b1.x = h2.x; b1.y = h2.y;
        } else
        b1 = gload<float4>(hotb, ho + 16);            // hy hz axis -
        sc = gload<double2>(hotb, ho + 32);           // splitval {c1, c2}
        if constexpr (HOT_NARROW == 2) {
          const uint32_t c1b = (uint32_t)__double2loint(sc.y), c2b = (uint32_t)__double2hiint(sc.y);
          b1.z = __uint_as_float(((c1b >> 30) & 1u) | (((c2b >> 30) & 1u) << 1));
        }
        }
        if constexpr (PROBE == 3) {   // sensitivity probe (TDTK_BUCKET_PTS=43): one more 16-byte load per node visit, result unused
          float4 w = gload<float4>(hotb, ho + 8);
          asm volatile("" : "+v"(b0.x), "+v"(b1.z), "+v"(sc.x), "+v"(w.x));
        } else
        TDTK_PIN_BATCH3(b0.x, b1.x, sc.x);
        const float a32 = fmaxf(fmaxf(fabsf(bx.qx - b0.x) - b0.w, fabsf(bx.qy - b0.y) - b1.x), fabsf(bx.qz - b0.z) - b1.y);
        bool prune = a32 >= bx.thi;
        if (__builtin_expect(!prune && !(a32 < bx.tlo), 0)) {      // undecidable in fp32 (or not finite): the exact test
          const uint32_t no = (uint32_t)((cur & REF_VAL) << 6);
          const double4 n0 = gload<double4>(reinterpret_cast<const char*>(nodes), no);
          const double2 n1 = gload<double2>(reinterpret_cast<const char*>(nodes), no + 32);
          prune = box_prunes_exact(n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, qx, qy, qz, best);
        }
        if (prune) need_pop = true;
        else next = descend_ax(sc.x, (uint32_t)__double2loint(sc.y), (uint32_t)__double2hiint(sc.y), __float_as_uint(b1.z), qx, qy, qz, best, st);
        }
      }
      if (need_pop) {
        next = REF_DONE;
        while (st.sp > 0) {
          --st.sp;
          uint32_t r; double m2;
          st.top(r, m2);
          if (m2 < best) { next = r; break; }
        }
      }
      cur = next;
      }
#ifdef TDTK_LAB
#include "lab_pipe_fin.inc"   // if constexpr (PIPE): the warm radius of a lane that has just become a query
#endif
    }

    // ---- phase 2: scan the bucket, then pop ----
    if (cur != REF_DONE) {
      const uint32_t v = cur & REF_VAL;
      int start, count;
      if (t_leaf_tab) {
        const LeafEntry le = t_leaf_tab[v];
        start = le.start; count = le.count;
      } else {
        start = (int)(v >> t_cb);
        count = (int)(v & t_cmask);
      }
      if (COUNT) { ++c_leaf; c_pts += (unsigned)count; }
      if (COUNT && kLab) { const unsigned long long on = __ballot(true); if ((int)lane == __ffsll((long long)on) - 1) ++c_t2; }
      if (ORDER) nbk += 4u;
      const char* pb = reinterpret_cast<const char*>(pts);
      const uint32_t o0 = (uint32_t)start << 5;              // byte offset of the bucket (< 4 GB)
      const uint32_t olast = o0 + ((uint32_t)(count - 1) << 5);
      // (one filter per build: with both in the kernel the register allocation is the fp32 one plus the grid query -- 130 VGPRs,
      // three waves per SIMD; a tree without the filter this build uses -- degenerate box, no memory -- takes the fp64 loop)
      if (USE_Q16 && (PROBE == 0 || PROBE == 3) && t_q16 != nullptr && count <= 20) {
        bool thin = false;
        bucket_scan_q16(t_q16, pb, start, count, o0, bx, q16xy, q16zz, q16in, qx, qy, qz, best, bk, a_tie, thin);
        if (DEFER_OK && thin) nbk |= NBK_THIN;
      } else
      if (!USE_Q16 && (PROBE == 0 || PROBE == 3) && t_grp != nullptr && count <= 4 * GRP_TRIP) {
        bucket_scan_groups(t_grp, pb, start, count, o0, bx, qx, qy, qz, best, bk);
      } else
      // PTS points per round trip, all their loads issued before the first use; the last group re-reads the final point
      // instead of running a scalar tail (a repeated point can never pass the strict '<' a second time)
      for (uint32_t o = o0; o <= olast; o += 32u * PTS) {
        uint32_t oo[PTS];
        double px[PTS], py[PTS], pz[PTS];
#pragma unroll
        for (int j = 0; j < PTS; j++) {
          oo[j] = (j == 0) ? o : min(o + 32u * (uint32_t)j, olast);
          const double2 pxy = *reinterpret_cast<const double2*>(pb + oo[j]);      // x y
          px[j] = pxy.x; py[j] = pxy.y;
          pz[j] = *reinterpret_cast<const double*>(pb + oo[j] + 16);              // z (the caller's index is not needed here)
          if constexpr (PROBE == 1) {   // sensitivity probe (TDTK_BUCKET_PTS=41): one more load per point, result unused
            double w = *reinterpret_cast<const double*>(pb + oo[j] + 24);
            asm volatile("" :: "v"(w));
          }
        }
        double dd[PTS];
#pragma unroll
        for (int j = 0; j < PTS; j++) {
          const double dx = px[j] - qx, dy = py[j] - qy, dz = pz[j] - qz;
          dd[j] = dx * dx + dy * dy + dz * dz;
          if constexpr (PROBE == 2) {   // sensitivity probe (TDTK_BUCKET_PTS=42): eight more fp64 VALU instructions per point
            double w = dx;
#pragma unroll
            for (int r = 0; r < 8; r++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w) : "v"(dy));
            asm volatile("" :: "v"(w));
          }
        }
#pragma unroll
        for (int j = 0; j < PTS; j++)
          if (dd[j] < best) { if (DEFER_OK && best - dd[j] <= a_tie) nbk |= NBK_THIN; best = dd[j]; bk = (int)(oo[j] >> 5); }
      }
      bx.set_radius(best);
      cur = REF_DONE;
      while (st.sp > 0) {
        --st.sp;
        uint32_t r; double m2;
        st.top(r, m2);
        if (m2 < best) { cur = r; break; }
      }
    }
  }
  if (kLab && a.trace && lane == 0) {
    const uint32_t wid = bid * (BLOCK / WAVE) + threadIdx.x / WAVE;
    if (wid < WTRACE_MAX) {
      g_wtrace[3 * wid + 1] = wall_clock64();
      // XCC_ID in bits 0-2, HW_REG_HW_ID (register 4: wave, simd, cu, sh, se ...) above it
      g_wtrace[3 * wid + 2] = (unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u) |
                              ((unsigned long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) << 8);
    }
  }
  if (COUNT && a.counters) {
    const unsigned long long s_int = wave_sum_u(c_int), s_leaf = wave_sum_u(c_leaf), s_pts = wave_sum_u(c_pts);
    if (lane == 0) {
      atomicAdd(&a.counters[0], s_int);
      atomicAdd(&a.counters[1], s_leaf);
      atomicAdd(&a.counters[2], s_pts);
    }
    if (DEFER_OK) {   // queries searched a second time, with every check (a thin acceptance on the walk that defers them)
      const unsigned long long s_redo = wave_sum_u(c_redo);
      if (lane == 0 && s_redo) atomicAdd(&a.counters[7], s_redo);
    }
    if (kLab) {       // wave trips: (lane-visits of a phase) / (64 x its trips) = the share of lane-slots that phase keeps busy
      const unsigned long long s_t1 = wave_sum_u(c_t1), s_t2 = wave_sum_u(c_t2);
      if (lane == 0) { atomicAdd(&a.counters[8], s_t1); atomicAdd(&a.counters[9], s_t2); }
    }
  }
  if constexpr (FUSE == 3 || FUSE == 5) {
    // The base pair sums of this wave's own queries, AFTER its last query has retired: the accumulators are not live
    // during the search (no occupancy cost, unlike FUSE 1), and the waves that finish early -- the median wave is done
    // after 77 % of a launch -- do this work while the machine would otherwise wait for the slowest ones.  The hits and
    // the moved coordinates were written by other lanes of this wave: make them visible before reading them back.  A
    // fence of workgroup scope is enough (the CU's L1 is write-through and shared by the workgroup) and costs a wait;
    // one of agent scope writes back and invalidates the XCD's L2 once per wave: k_search 0.218 -> 0.322 ms.
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    if constexpr (SHARE) __syncthreads();     // this wave's slab was searched by all waves of the workgroup
    // four queries per lane and trip, all their loads issued before the first use: two memory round trips for a whole
    // slab of up to 256 queries instead of one pair per 64
    const bool balanced = kLab && !LAZY && a.bounds != nullptr;     // this wave's slab is [reg0, slab_end), in one piece
    const uint32_t slab = balanced ? (uint32_t)(slab_end - reg0) : (uint32_t)a.qpw, sub32 = balanced ? 0x7FFFFFFFu : (uint32_t)sub;
    for (uint32_t j0 = 0; j0 < slab; j0 += 4 * WAVE) {
      size_t qq[4];
      int kk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        uint32_t j = j0 + (uint32_t)u * WAVE + lane, ph = 0;
        const bool in = j < slab;
        while (j >= sub32 && in) { j -= sub32; ph++; }
        qq[u] = reg0 + (size_t)ph * pstride + j;
        kk[u] = (in && qq[u] < slab_end) ? a.kpos[qq[u]] : -1;
      }
#pragma unroll
      for (int h = 0; h < 4; h += 2) {
        double cx[2], cy[2], cz[2], tx[2], ty[2], tz[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          cx[u] = cy[u] = cz[u] = tx[u] = ty[u] = tz[u] = 0.0;
          if (kk[h + u] >= 0) {
            const double4 c = *reinterpret_cast<const double4*>(reinterpret_cast<const char*>(pts) + ((uint32_t)kk[h + u] << 5));
            cx[u] = c.x; cy[u] = c.y; cz[u] = c.z;
            tx[u] = a.x[qq[h + u]]; ty[u] = a.y[qq[h + u]]; tz[u] = a.z[qq[h + u]];
          }
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          if (kk[h + u] < 0) continue;
          double mx, my, mz;
          dev_xf3(a.A, cx[u], cy[u], cz[u], mx, my, mz);  // searchTree.cc:147
          if constexpr (FUSE == 5) {   // lum6Deuler.cc:143-175 (as FUSE 2 above)
            const double dx = mx - tx[u], dy = my - ty[u], dz = mz - tz[u];
            const double x = (mx + tx[u]) / 2.0, y = (my + ty[u]) / 2.0, z = (mz + tz[u]) / 2.0;
            acc[0] += 1.0;
            acc[1] += dx * dx + dy * dy + dz * dz;
            acc[2] += x; acc[3] += y; acc[4] += z;
            acc[5] += x * x + y * y;
            acc[6] += x * x + z * z;
            acc[7] += y * y + z * z;
            acc[8] += x * y; acc[9] += x * z; acc[10] += y * z;
            acc[11] += dx; acc[12] += dy; acc[13] += dz;
            acc[14] += -z * dy + y * dz;
            acc[15] += -y * dx + x * dy;
            acc[16] += z * dx - x * dz;
            acc[17] += x * dx + y * dy + z * dz;
            continue;
          }
          const double px = mx - tx[u], py = my - ty[u], pz = mz - tz[u];
          acc[ACC_N] += 1.0;
          acc[ACC_SUM] += px * px + py * py + pz * pz;
          const double m0 = mx - a.shift[0], m1 = my - a.shift[1], m2 = mz - a.shift[2];
          const double d0 = tx[u] - a.shift[0], d1 = ty[u] - a.shift[1], d2 = tz[u] - a.shift[2];
          acc[ACC_SM + 0] += m0; acc[ACC_SM + 1] += m1; acc[ACC_SM + 2] += m2;
          acc[ACC_SD + 0] += d0; acc[ACC_SD + 1] += d1; acc[ACC_SD + 2] += d2;
          acc[ACC_P + 0] += m0 * d0; acc[ACC_P + 1] += m0 * d1; acc[ACC_P + 2] += m0 * d2;
          acc[ACC_P + 3] += m1 * d0; acc[ACC_P + 4] += m1 * d1; acc[ACC_P + 5] += m1 * d2;
          acc[ACC_P + 6] += m2 * d0; acc[ACC_P + 7] += m2 * d1; acc[ACC_P + 8] += m2 * d2;
        }
      }
    }
  }
  if (FUSE) {
    // wave64 reduction, then across the workgroup's waves through LDS: one row of ACC_TOTAL per workgroup
    constexpr int NW = BLOCK / WAVE;
    __shared__ double red[NW][NACC];
    const int wv = threadIdx.x / WAVE;
#pragma unroll
    for (int k = 0; k < NACC; k++) {
      const double s = wave_sum(acc[k]);
      if (lane == 0) red[wv][k] = s;
    }
    __syncthreads();
    // (a workgroup of more than 128 threads writes one row per pair of waves, where the 128-thread workgroup that owns
    // those two slabs in the same grid of slabs would write it: the sums do not depend on the workgroup size)
    constexpr int RS = (BLOCK > 128) ? BLOCK / 128 : 1, WPR = NW / RS;
    for (int kk = threadIdx.x; kk < RS * ACC_TOTAL; kk += BLOCK) {
      const int sr = kk / ACC_TOTAL, k = kk - sr * ACC_TOTAL;
      // column k of the row <- which accumulator (FUSE 2: n, sum, ACC_L .. ACC_L + 14, ACC_LU)
      int src = -1;
      if (FUSE == 2 || FUSE == 5) src = (k == ACC_N) ? 0 : (k == ACC_SUM) ? 1 : (k >= ACC_L && k < ACC_L + 15) ? 2 + (k - ACC_L) : (k == ACC_LU) ? 17 : -1;
      else if (k < ACC_DD) src = k;
      double s = 0.0;
      if (src >= 0)
        for (int w = 0; w < WPR; w++) s += red[sr * WPR + w][src];
      const size_t row = (RS == 1) ? (size_t)bid : ((size_t)((bid >> 3) * RS + sr) * 8u + (bid & 7u));
      a.partials[row * ACC_TOTAL + k] = s;
    }
  }
}

template <int BLOCK, int SD, int THRESH, int WPS, bool COUNT, int FUSE, bool DYN, int PTS = 4, int PROBE = 0, bool FAT = false, int TOP = 0, bool SHARE = false, bool PIPE = false,
          bool DEFER = false>
__global__ void __launch_bounds__(BLOCK, WPS) k_search_refill(const SearchArgs a_by_value)
{
  // The argument block (three 4x4 fp64 matrices among its 700 bytes) is read through the kernarg segment pointer, not
  // through the by-value parameter: a by-value parameter is known dereferenceable and loop-invariant, so the compiler
  // hoists every field into SGPRs up front -- 106 SGPRs with 80 of them spilled into VGPR lanes in round 2, every use a
  // v_readlane in a kernel that is short of issue slots.  Behind an opaque pointer the fields are s_load'ed where they
  // are used (the matrices only when a lane takes a new query), like k_search_refill_multi reads its table entry.
  (void)a_by_value;
  search_refill_body<BLOCK, SD, THRESH, WPS, COUNT, FUSE, DYN, !DYN, PTS, PROBE, FAT, 256, false, TOP, SHARE, PIPE, DEFER>(kernarg_block<SearchArgs>(), blockIdx.x, gridDim.x);
}

#ifdef TDTK_LAB
#include "lab_two_per_lane.inc"   // two queries per lane (k_search_refill2)
#endif

// Several whole-scan passes (the links of a graph-SLAM round) in ONE launch: workgroups base[l] .. base[l+1]-1 search
// batch l with the arguments args[l] (device memory; every base[] a multiple of 8, so a workgroup's XCD is the one its
// batch-relative index says).  Workgroups are dispatched in order, so the tail of one batch is filled by the next --
// what several streams give, without depending on how the runtime maps streams to hardware queues.
template <int BLOCK, int SD, int THRESH, int WPS, bool COUNT, int FUSE, bool ORDER = false, bool PIPE = false>
__global__ void __launch_bounds__(BLOCK, WPS) k_search_refill_multi(const SearchArgs* __restrict__ args,
                                                                    const uint32_t* __restrict__ base, int nbatch)
{
  int l = 0;
  while (l + 1 < nbatch && blockIdx.x >= base[l + 1]) ++l;
  l = __builtin_amdgcn_readfirstlane(l);
  const uint32_t b0 = base[l], b1 = base[l + 1];
  search_refill_body<BLOCK, SD, THRESH, WPS, COUNT, FUSE, false, ORDER, 4, 0, false, 320, true, 0, false, PIPE>(args[l], blockIdx.x - b0, b1 - b0);
}

#ifdef TDTK_LAB
#include "lab_slab_bounds.inc"   // slabs of equal cost (k_slab_bounds) and the one-loop kernel (k_search_step)
#endif

// ------------------------------------------------------------------------------------------
// pair-sum accumulation
// ------------------------------------------------------------------------------------------

// acc layout (doubles): see kernels.h ACC_*
template <int BLOCK, unsigned WANT, int PMODE>
__device__ __forceinline__ void accum_body(const AccumArgs& a, const uint32_t bid, const uint32_t nb)
{
  constexpr int NW = BLOCK / WAVE;
  // ACC_WANT_NO_CROSS: the caller needs n, sum and its own block only (a lum6DEuler link: 17 columns instead of 34 to
  // accumulate and, what costs more with four pairs per lane, to reduce across the wave)
  constexpr bool CROSS = !(WANT & ACC_WANT_NO_CROSS);
  __shared__ double red[NW][ACC_TOTAL];
  // pairing mode 1 keeps the normal rotated into the tree frame: the nine rotation entries of `inv` wait in LDS (read at a
  // wave-uniform address where they are used) -- as eighteen more scalar registers beside A, the shift and the sums'
  // bookkeeping they were the ones the NAPX instantiations spilled
  __shared__ double s_rot[PMODE == 1 ? 9 : 1];
  if (PMODE == 1) {
    if (threadIdx.x < 9) s_rot[threadIdx.x] = a.inv.m[(threadIdx.x / 3) * 4 + (threadIdx.x % 3)];
    __syncthreads();
  }

  double acc[ACC_TOTAL];
#pragma unroll
  for (int k = 0; k < ACC_TOTAL; k++) acc[k] = 0.0;

  const double4* __restrict__ pts = reinterpret_cast<const double4*>(a.T.pts);
  // a workgroup walks 4 * BLOCK consecutive queries per step: the four hit positions, then the four gathers and the
  // query coordinates are all requested before the first use (the gather depends on the hit position; one query at a
  // time exposes that round trip four times)
  constexpr int U = 4;
  const size_t stride = (size_t)nb * BLOCK * U;
  for (size_t base = (size_t)bid * BLOCK * U + threadIdx.x; base < a.n; base += stride) {
    int kk[U];
    double4 cc[U];
    double qx_[U], qy_[U], qz_[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t i = base + (size_t)u * BLOCK;
      kk[u] = (i < a.n) ? a.kpos[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t i = base + (size_t)u * BLOCK;
      if (kk[u] >= 0) { cc[u] = pts[kk[u]]; qx_[u] = a.x[i]; qy_[u] = a.y[i]; qz_[u] = a.z[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
    const size_t i = base + (size_t)u * BLOCK;
    const int k = kk[u];
    if (k < 0) continue;
    const double tx = qx_[u], ty = qy_[u], tz = qz_[u];
    const double4 c = cc[u];
    double mx, my, mz;
    dev_xf3(a.A, c.x, c.y, c.z, mx, my, mz);  // searchTree.cc:147
    double nxv = 0, nyv = 0, nzv = 0;
    if (PMODE != 0 || (WANT & TDTK_WANT_NAPX)) {
      nxv = a.nx[i]; nyv = a.ny[i]; nzv = a.nz[i];
      const double len = __dsqrt_rn(nxv * nxv + nyv * nyv + nzv * nzv);
      nxv /= len; nyv /= len; nzv /= len;
      if (PMODE == 1) {   // the reference keeps the rotated one: dev_xf3normal(a.inv, ..) with the rotation from LDS
        const double xn = nxv * s_rot[0] + nyv * s_rot[1] + nzv * s_rot[2];
        const double yn = nxv * s_rot[3] + nyv * s_rot[4] + nzv * s_rot[5];
        const double zn = nxv * s_rot[6] + nyv * s_rot[7] + nzv * s_rot[8];
        nxv = xn; nyv = yn; nzv = zn;
      }
    }
    if (PMODE == 2) {  // searchTree.cc:149-162: project the hit onto the data point's plane
      const double ex = mx - tx, ey = my - ty, ez = mz - tz;
      const double dot = nxv * ex + nyv * ey + nzv * ez;
      mx = nxv * dot + tx; my = nyv * dot + ty; mz = nzv * dot + tz;
    }
    const double px = mx - tx, py = my - ty, pz = mz - tz;  // p1 - p2
    acc[ACC_N] += 1.0;
    acc[ACC_SUM] += px * px + py * py + pz * pz;
    const double m0 = mx - a.shift[0], m1 = my - a.shift[1], m2 = mz - a.shift[2];
    const double d0 = tx - a.shift[0], d1 = ty - a.shift[1], d2 = tz - a.shift[2];
    if (CROSS) {
      acc[ACC_SM + 0] += m0; acc[ACC_SM + 1] += m1; acc[ACC_SM + 2] += m2;
      acc[ACC_SD + 0] += d0; acc[ACC_SD + 1] += d1; acc[ACC_SD + 2] += d2;
      acc[ACC_P + 0] += m0 * d0; acc[ACC_P + 1] += m0 * d1; acc[ACC_P + 2] += m0 * d2;
      acc[ACC_P + 3] += m1 * d0; acc[ACC_P + 4] += m1 * d1; acc[ACC_P + 5] += m1 * d2;
      acc[ACC_P + 6] += m2 * d0; acc[ACC_P + 7] += m2 * d1; acc[ACC_P + 8] += m2 * d2;
    }
    if (WANT & TDTK_WANT_GAPX) {
      acc[ACC_MM + 0] += m0 * m0; acc[ACC_MM + 1] += m0 * m1; acc[ACC_MM + 2] += m0 * m2;
      acc[ACC_MM + 3] += m1 * m1; acc[ACC_MM + 4] += m1 * m2; acc[ACC_MM + 5] += m2 * m2;
    }
    if (WANT & (TDTK_WANT_APX | TDTK_WANT_GAPX)) {
      acc[ACC_DD + 0] += d0 * d0; acc[ACC_DD + 1] += d0 * d1; acc[ACC_DD + 2] += d0 * d2;
      acc[ACC_DD + 3] += d1 * d1; acc[ACC_DD + 4] += d1 * d2; acc[ACC_DD + 5] += d2 * d2;
    }
    if (WANT & TDTK_WANT_NAPX) {
      // v = [ (d - shift) x n ; n ],  A0 += v v^T (upper), B0 += v, sum += ((p1-p2).n)^2
      double v[6];
      v[0] = d1 * nzv - d2 * nyv;
      v[1] = d2 * nxv - d0 * nzv;
      v[2] = d0 * nyv - d1 * nxv;
      v[3] = nxv; v[4] = nyv; v[5] = nzv;
      int q = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int s = r; s < 6; s++) acc[ACC_NA + (q++)] += v[r] * v[s];
#pragma unroll
      for (int r = 0; r < 6; r++) acc[ACC_NB + r] += v[r];
      const double dd = px * nxv + py * nyv + pz * nzv;
      acc[ACC_NS] += dd * dd;
    }
    if (WANT & TDTK_WANT_LUM) {
      // lum6Deuler.cc:143-175, ak = p1 (model, world), bk = p2 (data)
      const double x = (mx + tx) / 2.0, y = (my + ty) / 2.0, z = (mz + tz) / 2.0;
      const double dx = mx - tx, dy = my - ty, dz = mz - tz;
      acc[ACC_L + 0] += x; acc[ACC_L + 1] += y; acc[ACC_L + 2] += z;
      acc[ACC_L + 3] += x * x + y * y;
      acc[ACC_L + 4] += x * x + z * z;
      acc[ACC_L + 5] += y * y + z * z;
      acc[ACC_L + 6] += x * y; acc[ACC_L + 7] += x * z; acc[ACC_L + 8] += y * z;
      acc[ACC_L + 9] += dx; acc[ACC_L + 10] += dy; acc[ACC_L + 11] += dz;
      acc[ACC_L + 12] += -z * dy + y * dz;
      acc[ACC_L + 13] += -y * dx + x * dy;
      acc[ACC_L + 14] += z * dx - x * dz;
      acc[ACC_LU] += x * dx + y * dy + z * dz;   // lum6Dquat.cc:163
      if (a.has_D) {  // second pass: residual against the solved D (lum6Deuler.cc:199-210)
        const double e0 = dx - (a.D[0] - y * a.D[4] + z * a.D[5]);
        const double e1 = dy - (a.D[1] - z * a.D[3] + x * a.D[4]);
        const double e2 = dz - (a.D[2] + y * a.D[3] - x * a.D[5]);
        acc[ACC_LSS] += e0 * e0 + e1 * e1 + e2 * e2;
      }
    }
    }
  }

  // wave64 shuffle reduction, then across the block's waves through LDS
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
#pragma unroll
  for (int k = 0; k < ACC_TOTAL; k++) {
    const bool used = (k < ACC_SM) || (CROSS && k < ACC_DD) || ((WANT & (TDTK_WANT_APX | TDTK_WANT_GAPX)) && k >= ACC_DD && k < ACC_NA) ||
                      ((WANT & TDTK_WANT_NAPX) && k >= ACC_NA && k < ACC_L) ||
                      ((WANT & TDTK_WANT_LUM) && ((k >= ACC_L && k < ACC_MM) || k == ACC_LU)) ||
                      ((WANT & TDTK_WANT_GAPX) && k >= ACC_MM && k < ACC_LU);
    if (!used) continue;
    const double s = wave_sum(acc[k]);
    if (lane == 0) red[wv][k] = s;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ACC_TOTAL; k += BLOCK) {
    const bool used = (k < ACC_SM) || (CROSS && k < ACC_DD) || ((WANT & (TDTK_WANT_APX | TDTK_WANT_GAPX)) && k >= ACC_DD && k < ACC_NA) ||
                      ((WANT & TDTK_WANT_NAPX) && k >= ACC_NA && k < ACC_L) ||
                      ((WANT & TDTK_WANT_LUM) && ((k >= ACC_L && k < ACC_MM) || k == ACC_LU)) ||
                      ((WANT & TDTK_WANT_GAPX) && k >= ACC_MM && k < ACC_LU);
    double s = 0.0;
    if (used)
      for (int w = 0; w < NW; w++) s += red[w][k];
    a.partials[(size_t)bid * ACC_TOTAL + k] = s;
  }
}

template <int BLOCK, unsigned WANT, int PMODE>
__global__ void __launch_bounds__(BLOCK) k_accum(const AccumArgs a_by_value)
{
  // (the argument block through the kernarg pointer, like k_search_refill: by value the compiler hoists its three matrices
  // into scalar registers up front and -- in the NAPX instantiations -- spills four of them)
  (void)a_by_value;
  accum_body<BLOCK, WANT, PMODE>(kernarg_block<AccumArgs>(), blockIdx.x, gridDim.x);
}
// the pair sums of several batches in one launch (see k_search_refill_multi)
template <int BLOCK, unsigned WANT, int PMODE>
__global__ void __launch_bounds__(BLOCK) k_accum_multi(const AccumArgs* __restrict__ args, const uint32_t* __restrict__ base, int nbatch)
{
  int l = 0;
  while (l + 1 < nbatch && blockIdx.x >= base[l + 1]) ++l;
  l = __builtin_amdgcn_readfirstlane(l);
  const uint32_t b0 = base[l], b1 = base[l + 1];
  accum_body<BLOCK, WANT, PMODE>(args[l], blockIdx.x - b0, b1 - b0);
}

// fixed-order reduction of the per-workgroup rows: one 256-thread workgroup per accumulator
// column; thread t adds rows t, t+256, ... then wave64 shuffles and a 4-entry LDS pass.  The
// association is fixed by (rows, 256), so repeated runs give bit-identical sums.
__global__ void __launch_bounds__(256) k_final(const double* __restrict__ partials, int rows,
                                               double* __restrict__ out, int ncols)
{
  __shared__ double red[4];
  const int k = blockIdx.x;
  // `out` may be pinned host memory the host is watching word by word (await_sums in api.cpp): each sum is published
  // with a store of system scope
  if (k >= ncols) {   // a column the pass did not fill (its rows hold +0.0): the same +0.0 without reading them
    if (threadIdx.x == 0) __hip_atomic_store(&out[k], 0.0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  double s = 0.0;
  for (int r = threadIdx.x; r < rows; r += 256) s += partials[(size_t)r * ACC_TOTAL + k];
  s = wave_sum(s);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&out[k], ((red[0] + red[1]) + red[2]) + red[3], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// k_final for several batches: blockIdx.y = batch (its rows, its output)
__global__ void __launch_bounds__(256) k_final_multi(const FinalDesc* __restrict__ desc)
{
  __shared__ double red[4];
  const FinalDesc d = desc[blockIdx.y];
  const int k = blockIdx.x;
  // pad == 1: the rows of a lum6DEuler link without the cross block -- n, sum, the LUM columns; the others hold +0.0
  if (d.pad == 1 && !(k < ACC_SM || (k >= ACC_L && k < ACC_MM) || k == ACC_LU)) {
    if (threadIdx.x == 0) d.out[k] = 0.0;
    return;
  }
  double s = 0.0;
  for (int r = threadIdx.x; r < d.rows; r += 256) s += d.partials[(size_t)r * ACC_TOTAL + k];
  s = wave_sum(s);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) d.out[k] = ((red[0] + red[1]) + red[2]) + red[3];
}

// ------------------------------------------------------------------------------------------
// resident-scan transform (Scan::transformReduced, scan.cc:851-875)
// ------------------------------------------------------------------------------------------
__global__ void k_transform(double* __restrict__ x, double* __restrict__ y, double* __restrict__ z,
                            double* __restrict__ nx, double* __restrict__ ny,
                            double* __restrict__ nz, size_t n, const Mat4 A)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double px = x[i], py = y[i], pz = z[i];
    dev_xf3_inplace(A, px, py, pz);
    x[i] = px; y[i] = py; z[i] = pz;
    if (nx) {
      double ax = nx[i], ay = ny[i], az = nz[i];
      dev_xf3normal(A, ax, ay, az);
      nx[i] = ax; ny[i] = ay; nz[i] = az;
    }
  }
}

// Scan::transformToEuler (scan.cc:1061-1083) for many resident scans in one launch, and whatever else was queued on
// them: the in-place transforms of a scan's chain (inverse of the old pose, the new pose, the same again for the next
// round ...) are applied one after the other per point -- the arithmetic of one k_transform pass per matrix, one trip
// of the points through HBM, one launch.  blockIdx.y = scan.
__global__ void __launch_bounds__(256) k_transform_chain_batch(const XfChainDesc* __restrict__ desc)
{
  const XfChainDesc& d = desc[blockIdx.y];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += stride) {
    double px = d.x[i], py = d.y[i], pz = d.z[i];
    double ax = 0, ay = 0, az = 0;
    if (d.nx) { ax = d.nx[i]; ay = d.ny[i]; az = d.nz[i]; }
    for (int k = 0; k < d.nm; k++) {
      Mat4 A;
      load_move(d.mats, k, A);
      dev_xf3_inplace(A, px, py, pz);
      if (d.nx) dev_xf3normal(A, ax, ay, az);
    }
    d.x[i] = px; d.y[i] = py; d.z[i] = pz;
    if (d.nx) { d.nx[i] = ax; d.ny[i] = ay; d.nz[i] = az; }
  }
}

// ------------------------------------------------------------------------------------------
// spatial binning of an unsorted device-resident query batch (counting sort on a 32^3 grid
// over the tree's root box, cells visited in Morton order).  Order inside a cell is arbitrary:
// every query is independent and results are written back by original index.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t spread5(uint32_t v)
{
  // 5 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}
__device__ __forceinline__ uint32_t cell_of(const BinArgs& b, double x, double y, double z)
{
  double fx = (x - b.lo[0]) * b.scale[0], fy = (y - b.lo[1]) * b.scale[1], fz = (z - b.lo[2]) * b.scale[2];
  fx = fmin(fmax(fx, 0.0), 31.0); fy = fmin(fmax(fy, 0.0), 31.0); fz = fmin(fmax(fz, 0.0), 31.0);
  return spread5((uint32_t)fx) | (spread5((uint32_t)fy) << 1) | (spread5((uint32_t)fz) << 2);
}
__global__ void k_bin_count(const BinArgs b)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += stride) {
    const uint32_t c = cell_of(b, b.q[3 * i], b.q[3 * i + 1], b.q[3 * i + 2]);
    b.cell[i] = c;
    atomicAdd(&b.hist[c], 1u);
  }
}
// exclusive scan of the 32768 counters by one 1024-thread workgroup (32 per thread)
__global__ void __launch_bounds__(1024) k_bin_scan(uint32_t* __restrict__ hist)
{
  __shared__ uint32_t part[1024];
  const int t = threadIdx.x;
  uint32_t loc[32];
  uint32_t s = 0;
  for (int k = 0; k < 32; k++) { loc[k] = s; s += hist[t * 32 + k]; }
  part[t] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    uint32_t v = (t >= off) ? part[t - off] : 0u;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const uint32_t base = (t == 0) ? 0u : part[t - 1];
  for (int k = 0; k < 32; k++) hist[t * 32 + k] = base + loc[k];
}
__global__ void k_bin_scatter(const BinArgs b)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < b.n; i += stride) {
    const uint32_t pos = atomicAdd(&b.hist[b.cell[i]], 1u);
    b.sx[pos] = b.q[3 * i]; b.sy[pos] = b.q[3 * i + 1]; b.sz[pos] = b.q[3 * i + 2];
    if (b.dir) { b.sdx[pos] = b.dir[3 * i]; b.sdy[pos] = b.dir[3 * i + 1]; b.sdz[pos] = b.dir[3 * i + 2]; }
    b.order[pos] = (int32_t)i;
  }
}
// AoS -> SoA without reordering (presorted batches)
__global__ void k_split_soa(const double* __restrict__ q, size_t n, double* __restrict__ x,
                            double* __restrict__ y, double* __restrict__ z)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    x[i] = q[3 * i]; y[i] = q[3 * i + 1]; z[i] = q[3 * i + 2];
  }
}

// sorted-position hit -> caller-order model index (+ distance)
__global__ void k_scatter_idx(const int* __restrict__ kpos, const double* __restrict__ d2s,
                              const int32_t* __restrict__ order, const KdPoint* __restrict__ pts,
                              size_t n, int32_t* __restrict__ idx_out, double* __restrict__ d2_out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const int k = kpos[j];
    const size_t dst = order ? (size_t)order[j] : j;
    if (idx_out) idx_out[dst] = (k >= 0) ? pts[k].orig : -1;
    if (d2_out) d2_out[dst] = d2s[j];
  }
}

// -R (rnd > 1): the host draws one std::rand() per query in the caller's index order (the reference's loop, searchTree.cc:116-118)
// and sends the keep-mask as bits; a search reads one byte per query at its sorted position.
__global__ void __launch_bounds__(256) k_skip_from_mask(const unsigned char* __restrict__ mask_bits, const int32_t* __restrict__ order, size_t n,
                                                        unsigned char* __restrict__ skip)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const size_t i = order ? (size_t)order[j] : j;
    skip[j] = ((mask_bits[i >> 3] >> (i & 7u)) & 1u) ? 0 : 1;
  }
}

// The K5 hash of SURVEY 8(c) over one pass's correspondences, computed where they are (tdtk_icp_index_hashes):
// h = XOR over the found queries of (model index x 1315423911 + query index), both in the CALLER's numbering -- the hash the tests
// compute on the host for the index array tdtk_find_closest would have returned.  XOR commutes: any order.
__global__ void __launch_bounds__(256) k_idx_hash(const int* __restrict__ kpos, const int32_t* __restrict__ order, const KdPoint* __restrict__ pts,
                                                  size_t n, unsigned long long* __restrict__ out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long h = 0ull;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const int k = kpos[j];
    if (k >= 0) h ^= (unsigned long long)(unsigned)pts[k].orig * 1315423911ull + (unsigned long long)(order ? (size_t)order[j] : j);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)h, off, WAVE), hi = (unsigned)__shfl_xor((int)(unsigned)(h >> 32), off, WAVE);
    h ^= ((unsigned long long)hi << 32) | lo;
  }
  if ((threadIdx.x & (WAVE - 1)) == 0 && h) atomicXor(out, h);
}

// ---- compact PtPair list (searchTree.cc:147-180) in the caller's query order ----------------------
// found flags in caller order -> exclusive scan (rocPRIM, sort.hip) -> this kernel writes
// p1 = transform3(dalignxf, closest) [projected onto the data point's plane in mode 2],
// p2 = data point, pn = the normal the reference stores in the pair.
__global__ void k_found_flags(const int* __restrict__ kpos, const int32_t* __restrict__ order, size_t n,
                              uint32_t* __restrict__ flags)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
    flags[order[j]] = (kpos[j] >= 0) ? 1u : 0u;
}

template <int PMODE>
__global__ void k_pair_list(const PairListArgs a)
{
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(a.T.pts);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < a.n; j += stride) {
    const int k = a.kpos[j];
    if (k < 0) continue;
    const size_t slot = a.slot[a.order[j]];
    const double tx = a.x[j], ty = a.y[j], tz = a.z[j];
    const double4 c = pts[k];
    double mx, my, mz;
    dev_xf3(a.A, c.x, c.y, c.z, mx, my, mz);
    double nxv = 0, nyv = 0, nzv = 0;
    if (a.nx) {
      nxv = a.nx[j]; nyv = a.ny[j]; nzv = a.nz[j];
      const double len = __dsqrt_rn(nxv * nxv + nyv * nyv + nzv * nzv);
      nxv /= len; nyv /= len; nzv /= len;
      if (PMODE == 1) dev_xf3normal(a.inv, nxv, nyv, nzv);
    }
    if (PMODE == 2) {
      const double ex = mx - tx, ey = my - ty, ez = mz - tz;
      const double dot = nxv * ex + nyv * ey + nzv * ez;
      mx = nxv * dot + tx; my = nyv * dot + ty; mz = nzv * dot + tz;
    }
    if (a.p1) { a.p1[3 * slot] = mx; a.p1[3 * slot + 1] = my; a.p1[3 * slot + 2] = mz; }
    if (a.p2) { a.p2[3 * slot] = tx; a.p2[3 * slot + 1] = ty; a.p2[3 * slot + 2] = tz; }
    if (a.pn) { a.pn[3 * slot] = nxv; a.pn[3 * slot + 1] = nyv; a.pn[3 * slot + 2] = nzv; }
  }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static int g_num_cu = 0;
static int num_cu()
{
  if (!g_num_cu) {
    hipDeviceProp_t p;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cu = p.multiProcessorCount;
    if (g_num_cu <= 0) g_num_cu = 256;
  }
  return g_num_cu;
}

constexpr int SEARCH_BLOCK = 256;
// Levels of the traversal stack the single-pass persistent-lane kernel keeps in LDS (16 bytes per lane and level; what a query
// stacks beyond them goes to the overflow area in HBM, 12 bytes a push and as many a pop).  Round 6: the "write amplification"
// of the 10M-query passes (WRITE_SIZE 1.03 GB for 0.05 GB of results) was not the results -- staged in LDS and written out in
// whole lines they left the counter where it was and cost 7 % (tools/patches/r6_stage_results_in_lds.patch, NEGATIVES.md) --
// but this overflow: 14.3 M write requests per pass, 12.5 M of them whole 64-byte lines = a wave's column of one stack level.
// Six levels instead of four (13 KB of LDS per workgroup: still twelve workgroups = six waves per SIMD on a CU) take the
// whole-scan pass over a 10M-point tree from 1.82-1.95 ms to 1.48 and the ICP iteration at that size from 1.24-1.30 to 1.17;
// five: 1.63 / 1.19; eight: 1.50 / 1.34 (nine workgroups per CU).  The instantiations that add up their pair sums inside the
// launch (FUSE 3: scans of up to 1.8M points) get SLOWER with six -- 1M-vs-1M cold pass 0.193 -> 0.244 ms, the timed loop
// 0.158 -> 0.198 -- while FUSE 0 over the same 1M tree does not care (0.181 -> 0.180): four stay four there.
template <int FUSE> constexpr int REFILL_SD = (FUSE == 0) ? 6 : 4;
// (the several-links launch, four waves per SIMD, does not care: eight levels against four, 84 links of 1M points 9.91 against
//  9.88 ms, three links of 10M points 3.66 against 3.68 -- it keeps four)
constexpr int MULTI_SD = 4;
constexpr int SEARCH_SD_MIN = 4;  // overflow area is sized for the shallowest LDS stack in use

// Variants of the hot instantiation (TDTK_SEARCH_VARIANT=<n>); the default picks by batch size.
// The full ladder that was measured is in DESIGN.md section 6.
//   0: the first working kernel (LDS stack 8 deep, per-lane node loads)
//   4: + 4-deep LDS stack, wave-uniform scalar node loads          (default for 96K..256K queries)
//   5: wave-cooperative LDS staging of distinct nodes / buckets    (kept as a measured negative)
//   8: persistent lanes, 256 queries per wave, 256-thread workgroups
//   9 / 10 / 11: eight / four / sixteen lanes per query (k_search_g8); four is the default below 96K queries
//  20: persistent lanes, 256 queries per wave, 128-thread workgroups
//  40: persistent lanes, one loop in which every lane takes one step per trip (k_search_step, "if-if")
//  30: persistent lanes fed from a work queue (k_search_refill<.., DYN>): resident waves draw 64-query slabs
// (the TDTK_* knobs are read on every launch: a getenv is nothing beside a launch, and tests / probes flip them
// inside one process)
static int search_variant()
{
  if (!kLab) return -2;              // the product build chooses by batch size: 20 / 4 / 10 below
  const char* e = lab_env("TDTK_SEARCH_VARIANT");
  int v = e ? atoi(e) : -2;          // -2: choose by batch size (see launch_search)
  if (v != 0 && v != 4 && v != 5 && v != 9 && v != 10 && v != 11 && v != 8 && v != 20 && v != 30 && v != 40 && v != 41) v = -2;
  return v;
}

uint32_t search_grid(size_t n)
{
  // slabs of the sorted queries: enough workgroups to fill 256 CUs several times over so the
  // tail is short, multiple of 8 for the XCD mapping, never more than one slab per 256 queries
  size_t want = (n + SEARCH_BLOCK - 1) / SEARCH_BLOCK;
  size_t cap = (size_t)num_cu() * 16;
  size_t nb = want < cap ? want : cap;
  nb = (nb + 7) & ~(size_t)7;
  if (nb < 8) nb = 8;
  return (uint32_t)nb;
}
// persistent-lane grids: one wave per `qpw` consecutive sorted queries (256 by default), never more lanes than
// search_grid() launches -- the stack overflow area is sized for that many -- so a batch beyond the cap gives
// every wave a longer slab instead of adding waves (qpw is a kernel argument)
// Slab length of the persistent-lane kernel.  Two opposing costs: every slab ends with a drain (its last lanes finish
// alone), so longer slabs waste fewer lane-slots -- but the chip holds 7 of these waves per SIMD and a batch of 1M
// queries is only ~15 queries per lane-slot, so long slabs leave SIMDs without enough waves to hide the dependent
// loads.  Measured on the ICP loop (gpurun_out/r2a, r2b): 1M queries: 128 -> 0.276 ms, 160 -> 0.280, 192 -> 0.258,
// 224 -> 0.255, 256 -> 0.273, 384 -> 0.341, 512 -> 0.386; 4M queries: 192 -> 0.998, 256 -> 0.965 (best), 384 -> 0.991,
// 512 -> 1.096.  Rule: 256, unless all waves fit on the chip at once anyway -- then ~4.5 waves per SIMD.
// `side_by_side` > 1: the launch is one of several whole-scan passes running on streams of their own (the link passes
// of a graph-SLAM round).  The chip is then filled by the passes together, and fewer, longer-lived waves per pass win:
// 84 link passes of 1M queries on 3 streams, slab 224 -> 13.6 ms, 256 -> 13.1, 320 -> 12.7, 384 -> 12.75, 448 -> 12.8,
// 512 -> 13.0, 640 -> 13.4 (tools/gs_knobs_probe.py) -- ~3 waves per SIMD and pass.
static int refill_waves_per_simd();      // (defined behind the kernel it asks about)
static int refill_qpw(size_t n, int side_by_side = 1)
{
  if (const char* e = lab_env("TDTK_REFILL_QPW")) {
    int v = atoi(e);
    if (v < 64) v = 64;
    return (v + 31) & ~31;
  }
  const size_t resident = (size_t)num_cu() * 4 * 7;
  if ((n + 255) / 256 >= resident) return 256;
  if (side_by_side > 1) {
    size_t q = (n / ((size_t)num_cu() * 12) + 16) & ~(size_t)31;           // 3 waves per SIMD
    if (q < 128) q = 128;
    if (q > 512) q = 512;
    return (int)q;
  }
  // One resident generation, no straggling second one: the launch is sized for exactly the waves the chip holds of THIS
  // kernel.  Round 3 (fp32 bucket groups, 122 VGPRs): four per SIMD.  Round 5 (16-bit bucket shadows, 94 VGPRs): five --
  // asked of the runtime, not assumed (refill_waves_per_simd) -- and the slab a multiple of 8, not of 32: 1M queries =
  // 5000 waves of 200 (k_search 0.1990 -> 0.1791 ms at the driver's arguments, 0.1777 -> 0.1624 over 100 iterations; with
  // 4465 waves of 224 or 5209 of 192 the fifth wave of some SIMDs or a second generation costs more than it brings:
  // 0.1876 / 0.2238; gpurun_out/r5f)
  const size_t slots = (size_t)num_cu() * 4 * (size_t)refill_waves_per_simd();
  size_t q = (n + slots - 1) / slots;
  q = (q + 7) & ~(size_t)7;
  if (q < 128) q = 128;
  if (q > 256) q = 256;
  return (int)q;
}
static uint32_t refill_grid_b(size_t n, int block, int* qpw_out, int side_by_side = 1)
{
  const size_t wpb = (size_t)block / WAVE;
  size_t qpw = (size_t)refill_qpw(n, side_by_side);
  size_t waves = (n + qpw - 1) / qpw;
  size_t nb = (waves + wpb - 1) / wpb;
  nb = (nb + 7) & ~(size_t)7;
  if (nb < 8) nb = 8;
  size_t cap = ((size_t)num_cu() * 16 * SEARCH_BLOCK / block) & ~(size_t)7;
  if (cap < 8) cap = 8;
  if (nb > cap) {
    nb = cap;
    waves = nb * wpb;
    qpw = (n + waves - 1) / waves;
    qpw = (qpw + 63) & ~(size_t)63;
  }
  *qpw_out = (int)qpw;
  return (uint32_t)nb;
}
#ifdef TDTK_LAB
static uint32_t refill_grid_b_fwd(size_t n, int* qpw_out) { return refill_grid_b(n, 128, qpw_out); }
#endif
// Upper tree levels in LDS (search_refill_body<.., TOP>): the workgroup size of the single-pass launch that stages them,
// 0 = the plain 128-thread kernel.  Only while the launch is ONE generation of resident waves: a big workgroup leaves
// its CU when its slowest wave does, which costs nothing when nobody is waiting for the CU.
// MEASURED NEGATIVE (lab only, TDTK_TOP_BLOCK=512|1024): 1M-vs-1M k_search 0.2041-0.2047 ms (1024) / 0.2071-0.2080 (512)
// against 0.1933-0.1945; the upper levels' visits coalesce in the vector L1 anyway -- a wave's sorted queries share their
// first ten nodes -- so LDS takes little off the tag pipeline and the mixed trips pay for two paths.
static int refill_top_block(size_t n, int side_by_side)
{
#ifdef TDTK_LAB
  const char* e = lab_env("TDTK_TOP_BLOCK");
  const int blk = e ? atoi(e) : 0;
  if (blk != 128 && blk != 512 && blk != 1024) return 0;      // (128: seven levels per 128-thread workgroup; TDTK_TOP_LEVELS=0 with 1024: no staging, the workgroup size alone)
  if (side_by_side > 1 || (n + 255) / 256 >= (size_t)num_cu() * 4 * 7) return 0;
  if (lab_env("TDTK_REFILL_POOL") || lab_env("TDTK_BUCKET_PTS") || lab_env("TDTK_FAT_NODES") || lab_env("TDTK_WAVE_TRACE") || lab_env("TDTK_REFILL_THRESH") ||
      lab_env("TDTK_TWO_PER_LANE") || lab_env("TDTK_FUSE_SUMS") || lab_env("TDTK_SEARCH_VARIANT") || lab_env("TDTK_REFILL_QPW"))
    return 0;
  return blk;
#else
  (void)n; (void)side_by_side;
  return 0;
#endif
}
// Slabs handed out by the workgroup's waves together (search_refill_body<.., SHARE>): the workgroup size of the single-pass
// launch, 0 = every wave for itself (128-thread workgroups).  While the launch is ONE generation of resident waves.
// MEASURED NEGATIVE (lab only, TDTK_SHARE_BLOCK=256|512|1024): 1M-vs-1M k_search 0.2075-0.2085 / 0.2171-0.2179 / 0.2121-0.2131 ms
// against 0.1945-0.1956: the waves of a big workgroup sit on ONE CU and share its vector L1, which is what the kernel is
// bound by (TCP busy 80-91 % of the launch, profiles/r04_tcp_diag.txt); 128-thread workgroups spread a CU's sixteen waves
// over eight distant stretches of the scan, and that averaging is worth more than what the shared cursor evens out.
static int refill_share_block(size_t n, int side_by_side)
{
#ifndef TDTK_LAB
  (void)n; (void)side_by_side;
  return 0;
#endif
  int blk = 0;
  if (const char* e = lab_env("TDTK_SHARE_BLOCK")) blk = atoi(e);
  if (blk != 256 && blk != 512 && blk != 1024) return 0;
  if (side_by_side > 1 || (n + 255) / 256 >= (size_t)num_cu() * 4 * 7) return 0;
  if (lab_env("TDTK_REFILL_POOL") || lab_env("TDTK_BUCKET_PTS") || lab_env("TDTK_FAT_NODES") || lab_env("TDTK_WAVE_TRACE") || lab_env("TDTK_REFILL_THRESH") ||
      lab_env("TDTK_TWO_PER_LANE") || lab_env("TDTK_FUSE_SUMS") || lab_env("TDTK_SEARCH_VARIANT") || lab_env("TDTK_REFILL_QPW") || lab_env("TDTK_TOP_BLOCK") ||
      lab_env("TDTK_REFILL_PHASES") || lab_env("TDTK_BALANCE"))
    return 0;
  return blk;
}
// lab (TDTK_SINGLE_BLOCK=64): the single-pass launch in workgroups of ONE wave (a CU's sixteen waves then come from sixteen
// distant stretches of the scan instead of eight)
static int refill_single64(size_t n, int side_by_side)
{
#ifdef TDTK_LAB
  const char* e = lab_env("TDTK_SINGLE_BLOCK");
  if (!(e && atoi(e) == 64)) return 0;
  if (side_by_side > 1 || (n + 255) / 256 >= (size_t)num_cu() * 4 * 7) return 0;
  if (lab_env("TDTK_REFILL_POOL") || lab_env("TDTK_BUCKET_PTS") || lab_env("TDTK_FAT_NODES") || lab_env("TDTK_WAVE_TRACE") || lab_env("TDTK_REFILL_THRESH") ||
      lab_env("TDTK_TWO_PER_LANE") || lab_env("TDTK_FUSE_SUMS") || lab_env("TDTK_SEARCH_VARIANT") || lab_env("TDTK_REFILL_QPW") || lab_env("TDTK_TOP_BLOCK") ||
      lab_env("TDTK_REFILL_PHASES") || lab_env("TDTK_BALANCE") || lab_env("TDTK_SHARE_BLOCK") || lab_env("TDTK_PIPE"))
    return 0;
  return 64;
#else
  (void)n; (void)side_by_side;
  return 0;
#endif
}
// the workgroup size of the single-pass persistent-lane launch for n queries (128 unless one of the two above applies)
static int refill_big_block(size_t n, int side_by_side)
{
  if (const int tb = refill_top_block(n, side_by_side)) return tb;
  return refill_share_block(n, side_by_side);
}
size_t search_max_lanes(size_t n)
{
  int q;
  {
    const int tb = refill_big_block(n, 1);
    if (tb) {
      const size_t t = (size_t)refill_grid_b(n, tb, &q) * (size_t)tb;
      const size_t a0 = (size_t)search_grid(n) * SEARCH_BLOCK, b0 = (size_t)refill_grid_b(n, 128, &q) * 128;
      return std::max(t, std::max(a0, b0));
    }
  }
  const size_t a = (size_t)search_grid(n) * SEARCH_BLOCK;
  const size_t b = std::max((size_t)refill_grid_b(n, 128, &q) * 128, (size_t)refill_grid_b(n, 64, &q) * 64);
  const size_t c = (size_t)refill_grid_b(n, SEARCH_BLOCK, &q) * SEARCH_BLOCK;
  const size_t d = (size_t)num_cu() * 4 * 8 * WAVE;     // the work-queue kernel: at most every wave slot of the chip
  const size_t m1 = a > b ? a : b, m2 = c > d ? c : d;
  return m1 > m2 ? m1 : m2;
}
static uint32_t g8_grid(size_t n)
{
  size_t nb = (n + 31) / 32;            // 32 queries (groups of 8 lanes) per 256-thread workgroup
  const size_t cap = (size_t)num_cu() * 32;
  if (nb > cap) nb = cap;
  nb = (nb + 7) & ~(size_t)7;
  return (uint32_t)(nb < 8 ? 8 : nb);
}
int search_lds_depth() { return SEARCH_SD_MIN; }
int search_block() { return SEARCH_BLOCK; }

// idle lanes a wave collects before it hands out new queries: 16, or 32 once the batch is several rounds of resident
// waves (4M queries: 0.926 -> 0.913 ms; 1M: 0.238 -> 0.241, gpurun_out/r2h/sweep.log)
static int refill_thresh(size_t n)
{
  if (const char* e = lab_env("TDTK_REFILL_THRESH")) {
    const int v = atoi(e);
    if ((kLab && v == 8) || v == 16 || v == 32) return v;
  }
  return ((n + 255) / 256 >= (size_t)num_cu() * 4 * 7) ? 32 : 16;
}
// which kernel a batch of n queries gets (TDTK_SEARCH_VARIANT overrides)
static int pick_variant(size_t n)
{
  int v = search_variant();
  // persistent lanes pay off once there are enough queries to keep every SIMD supplied with
  // several 256-query waves; small batches keep one query per lane
  if (v == -2) v = (n >= (size_t)262144) ? 20 : ((n >= (size_t)98304) ? 4 : 10);
  return v;
}
bool search_can_fuse(size_t n) { return pick_variant(n) == 20; }
// FUSE 3 (each wave adds up its own slab after its last query) pays while a launch is ONE generation of resident waves,
// whose early finishers would idle otherwise: 300K queries 79.4 -> 74.5 us per ICP iteration, 1M 0.2217 -> 0.2183 ms;
// with several generations (4M: 0.771 -> 0.788 ms) the sums only take issue slots from the next slabs
bool search_fuse_after_last_pays(size_t n) { return pick_variant(n) == 20 && (n + 255) / 256 < (size_t)num_cu() * 4 * 7; }
// how the base pair sums can come out of the search of a batch this size: 1 = the persistent-lane kernel's FUSE modes
// (measured negatives, on request only), 2 = the chunk epilogue of the small-batch kernels, 0 = not at all
int search_fuse_kind(size_t n)
{
  const int v = pick_variant(n);
  return v == 20 ? 1 : ((v == 10 || v == 4) ? 2 : 0);
}
static uint32_t g8_grid4(size_t n) { const uint32_t g = g8_grid(n) / 2; return g < 8 ? 8u : (g + 7) / 8 * 8; }
// share of an XCD's region that is not dealt out in advance but drawn from a pool (k_search_refill, pool_slab);
// 0 = off.  Only where all waves of the launch are resident at once and the launch has the chip to itself.
static int refill_pool_pct(size_t n, int side_by_side)
{
  if (!kLab) return 0;
  if (side_by_side > 1 || (n + 255) / 256 >= (size_t)num_cu() * 4 * 7) return 0;
  const char* e = lab_env("TDTK_REFILL_POOL");
  int v = e ? atoi(e) : 0;
  if (v < 0) v = 0;
  if (v > 90) v = 90;
  return v;
}
bool search_uses_queue(size_t n) { return pick_variant(n) == 30 || (pick_variant(n) == 20 && refill_pool_pct(n, 1) > 0); }
#ifdef TDTK_LAB
#include "lab_part_2.inc"
#endif
uint32_t search_fused_rows(size_t n, int side_by_side)
{
  const int v = pick_variant(n);
  if (v == 10) return g8_grid4(n);
  if (v == 4) return search_grid(n);
  int q;
#ifdef TDTK_LAB
  if (two_per_lane_for(n, side_by_side)) return refill2_grid(n, &q);
#endif
  if (const int tb = refill_big_block(n, side_by_side)) return refill_grid_b(n, tb, &q, side_by_side) * (uint32_t)(tb / 128);
  if (refill_single64(n, side_by_side)) return refill_grid_b(n, 64, &q, side_by_side);
  return refill_grid_b(n, 128, &q, side_by_side);
}

// pipelined hand-out (search_refill_body<.., PIPE>), lab only (TDTK_PIPE=1).  MEASURED NEGATIVE: parity-green, and the bucket
// scan -- where the register demand peaks -- finds 8-9 more registers live with it (phi copies of the query registers the
// staged loads land in): 134 VGPRs = three waves per SIMD, or at 128 nine spilled registers in hot code: 1M-vs-1M k_search
// 0.2348 ms against 0.1954-0.1959, one lum6DEuler round of 84 links 13.9 ms against 10.5.
#ifdef TDTK_LAB
static bool pipe_on()
{
  if (const char* e = lab_env("TDTK_PIPE")) return e[0] == '1';
  return false;
}
#endif
// waves of the single-pass persistent-lane kernel (the FUSE 3 instantiation the ICP loop runs) a SIMD holds at once, by the
// runtime's own occupancy calculation for this code object (registers, LDS); 4 if it cannot say
// (round 5, second step: SIX.  The compiler is asked for six waves per SIMD -- __launch_bounds__' second argument, 80 VGPRs --
// for the instantiations the product's loops run (REFILL_WPS); what it gives up for that are the three 8-byte values of the
// stack's overflow area, stored once per wave and read back only where a walk overflows the LDS levels.  Same box, same
// process order, 1M-vs-1M at the driver's arguments: k_search 0.1770 / 0.1782 ms at five waves, 0.1677 / 0.1685 at six;
// forced to seven (72 VGPRs, 40 spilled) 0.1781.  tools/r5_nobox_ab.sh)
// (the several-links launch with a link's sums inside -- graph-SLAM, 128 VGPRs -- stays at four: asked for five the compiler
// spills 23 registers inside the loops, 84 links 10.0 -> 11.6 ms; LABFLAGS=-DTDTK_MULTI_WPS=5, tools/r5_multi_wps.sh)
#ifndef TDTK_MULTI_WPS
#define TDTK_MULTI_WPS 4
#endif
constexpr int MULTI_WPS = TDTK_MULTI_WPS;
#ifndef TDTK_REFILL_WPS
#define TDTK_REFILL_WPS 6        // (LABFLAGS=-DTDTK_REFILL_WPS=1: the compiler's own choice, 94 VGPRs = five waves)
#endif
template <bool COUNT, int FUSE>
constexpr int REFILL_WPS = COUNT ? 4 : ((FUSE == 0 || FUSE == 3) ? TDTK_REFILL_WPS : 1);
static int refill_waves_per_simd()
{
  static const int w = [] {
    int blocks = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, reinterpret_cast<const void*>(&k_search_refill<128, 4, 16, REFILL_WPS<false, 3>, false, 3, false>), 128, 0) != hipSuccess) {
      (void)hipGetLastError();
      return 4;
    }
    const int v = blocks * 2 / 4;          // 128-thread workgroups = two waves; four SIMDs per CU
    return v < 1 ? 4 : (v > 8 ? 8 : v);
  }();
  return w;
}
template <bool COUNT, int FUSE>
static void launch_refill128(SearchArgs& a, hipStream_t s)
{
  int qpw;
#ifdef TDTK_LAB
  if constexpr (FUSE == 0 || FUSE == 3) {
    if (const int tb = a.bounds ? 0 : refill_top_block(a.n, a.side_by_side)) {
      const uint32_t nbt = refill_grid_b(a.n, tb, &qpw, a.side_by_side);
      a.qpw = qpw; a.phases = 1; a.pool_slab = 0; a.region = 0; a.trace = 0;
      const char* lv = lab_env("TDTK_TOP_LEVELS");
      if (tb == 1024 && lv && lv[0] == '0') hipLaunchKernelGGL((k_search_refill<1024, 4, 16, 4, COUNT, FUSE, false, 4, 0, false, 0>), dim3(nbt), dim3(1024), 0, s, a);
      else if (tb == 128) hipLaunchKernelGGL((k_search_refill<128, 4, 16, 4, COUNT, FUSE, false, 4, 0, false, 127>), dim3(nbt), dim3(128), 0, s, a);
      else
      if (tb == 1024) hipLaunchKernelGGL((k_search_refill<1024, 4, 16, 4, COUNT, FUSE, false, 4, 0, false, 1023>), dim3(nbt), dim3(1024), 0, s, a);
      else hipLaunchKernelGGL((k_search_refill<512, 4, 16, 4, COUNT, FUSE, false, 4, 0, false, 511>), dim3(nbt), dim3(512), 0, s, a);
      return;
    }
  }
#endif
#ifdef TDTK_LAB
#include "lab_part_3.inc"
#endif
#ifdef TDTK_LAB
#include "lab_part_4.inc"
#endif
  const uint32_t nb = refill_grid_b(a.n, 128, &qpw, a.side_by_side);
  a.qpw = qpw;
  // Two pieces when every wave of the launch is resident at once (the XCD's waves then move through its eighth of the
  // scan together): 1M-vs-1M ICP, L2 misses 1.98 M -> 1.15 M per launch, fabric reads 239 -> 138 MB, time unchanged
  // (0.2190 -> 0.2187 ms); with several generations of waves (4M: 0.806 -> 0.821 ms, reads -35 %) the pieces only
  // cost coherence.  Non-temporal loads / stores for the query and result streams were measured too: no change in
  // misses or time.  (gpurun_out/r2n..r2r, tools/nt_probe.sh)
  // Neither do they pay beside other passes (84 link passes on 3 streams: 13.6 -> 13.2 ms without).
  {
    static std::atomic<int> launches{0};
    static const int want = [] { const char* e = lab_env("TDTK_WAVE_TRACE"); return e ? atoi(e) : -1; }();
    a.trace = (want >= 0 && launches.fetch_add(1) == want) ? 1 : 0;
  }
  a.pool_slab = 0; a.region = 0;
  const int pool_pct = (FUSE == 3) ? 0 : refill_pool_pct(a.n, a.side_by_side);
  if (pool_pct > 0) {
    const size_t R = (((a.n + 7) / 8) + 63) & ~(size_t)63;             // queries per XCD region
    const size_t wpx = (size_t)(nb >> 3) * 2;                          // waves per XCD (128-thread workgroups)
    size_t qs = (R * (size_t)(100 - pool_pct) / 100 / wpx) & ~(size_t)31;
    if (qs < 64) qs = 64;
    a.qpw = (int)qs; a.region = R;
    const char* e = lab_env("TDTK_REFILL_POOL_SLAB");
    a.pool_slab = e ? std::max(16, atoi(e)) : 64;
  }
  int ph = ((a.n + 255) / 256 >= (size_t)num_cu() * 4 * 7 || a.side_by_side > 1 || a.pool_slab) ? 1 : 2;
  // with the slab handed out expensive queries first (a.use_cost) one piece is better: the order then spans the whole
  // slab (1M: 0.2124 -> 0.2086 ms; the twenty iterations behind the initial pose 0.2425 -> 0.2326)
  if (a.use_cost) ph = 1;
  // ... and with the sums added up by the waves themselves (FUSE 3) the pieces decide which queries share a row of partial
  // sums: one piece always, so that the sums do not depend on whether the hand-out is ordered
  if (FUSE == 3) ph = 1;
  if (const char* e = lab_env("TDTK_REFILL_PHASES")) ph = atoi(e);
  if (ph < 1) ph = 1;
  while (ph > 1 && (qpw % (ph * 16)) != 0) --ph;   // pieces stay multiples of 16 queries
  a.phases = ph;
  // diagnostics: TDTK_OCC_LDS=<bytes> of unused dynamic LDS per workgroup caps the waves resident per SIMD (how the launch
  // time depends on occupancy alone); TDTK_BUCKET_PTS=8 scans buckets eight points per round trip instead of four
  static const unsigned occ_lds = [] { const char* e = lab_env("TDTK_OCC_LDS"); return e ? (unsigned)atoi(e) : 0u; }();
  const char* pe = lab_env("TDTK_BUCKET_PTS");
  const int bpts = pe ? atoi(pe) : 4;
  // the instantiation whose warm queries defer the quick check (SearchArgs::tie): a repeated pass over a tree that has the
  // split halves and the 16-bit shadow; everything else -- every cold pass -- runs the kernel without that machinery
  constexpr int SD_ = REFILL_SD<FUSE>;
  const bool defer_ok = (FUSE == 0 || FUSE == 3) && a.warm && a.tie > 0.0 && a.T.split != nullptr && a.T.q16 != nullptr && BUCKET_Q16;
#ifdef TDTK_LAB
  if (!COUNT && FUSE == 0 && bpts == 8 && refill_thresh(a.n) == 16) {
    hipLaunchKernelGGL((k_search_refill<128, SD_, 16, 1, false, 0, false, 8>), dim3(nb), dim3(128), occ_lds, s, a);
  } else if (!COUNT && FUSE == 0 && bpts == 41 && refill_thresh(a.n) == 16) {
    hipLaunchKernelGGL((k_search_refill<128, SD_, 16, 1, false, 0, false, 4, 1>), dim3(nb), dim3(128), occ_lds, s, a);
  } else if (!COUNT && FUSE == 0 && bpts == 42 && refill_thresh(a.n) == 16) {
    hipLaunchKernelGGL((k_search_refill<128, SD_, 16, 1, false, 0, false, 4, 2>), dim3(nb), dim3(128), occ_lds, s, a);
  } else if (!COUNT && FUSE == 0 && bpts == 43 && refill_thresh(a.n) == 16) {
    hipLaunchKernelGGL((k_search_refill<128, SD_, 16, 1, false, 0, false, 4, 3>), dim3(nb), dim3(128), occ_lds, s, a);
  } else if (!COUNT && FUSE == 0 && refill_thresh(a.n) == 16 && a.T.fat != nullptr && lab_env("TDTK_FAT_NODES") && lab_env("TDTK_FAT_NODES")[0] == '1') {
    // two tree levels per round trip (KdFat): a measured negative, kept selectable -- see the comment at the walk
    hipLaunchKernelGGL((k_search_refill<128, SD_, 16, 1, false, 0, false, 4, 0, true>), dim3(nb), dim3(128), occ_lds, s, a);
  } else
#endif
  switch (refill_thresh(a.n)) {
#ifdef TDTK_LAB
    case 8: hipLaunchKernelGGL((k_search_refill<128, SD_, 8, 1, COUNT, FUSE, false>), dim3(nb), dim3(128), occ_lds, s, a); break;
#endif
    case 32:
      if (defer_ok) hipLaunchKernelGGL((k_search_refill<128, SD_, 32, REFILL_WPS<COUNT, FUSE>, COUNT, FUSE, false, 4, 0, false, 0, false, false, (FUSE == 0 || FUSE == 3)>), dim3(nb), dim3(128), occ_lds, s, a);
      else hipLaunchKernelGGL((k_search_refill<128, SD_, 32, REFILL_WPS<COUNT, FUSE>, COUNT, FUSE, false>), dim3(nb), dim3(128), occ_lds, s, a);
      break;
    default:
#ifdef TDTK_LAB
      if ((FUSE == 0 || FUSE == 3) && pipe_on() && !a.skip)
        hipLaunchKernelGGL((k_search_refill<128, SD_, 16, 4, COUNT, (FUSE == 3 ? 3 : 0), false, 4, 0, false, 0, false, true>), dim3(nb), dim3(128), occ_lds, s, a);
      else
#endif
      if (defer_ok) hipLaunchKernelGGL((k_search_refill<128, SD_, 16, REFILL_WPS<COUNT, FUSE>, COUNT, FUSE, false, 4, 0, false, 0, false, false, (FUSE == 0 || FUSE == 3)>), dim3(nb), dim3(128), occ_lds, s, a);
      else
      hipLaunchKernelGGL((k_search_refill<128, SD_, 16, REFILL_WPS<COUNT, FUSE>, COUNT, FUSE, false>), dim3(nb), dim3(128), occ_lds, s, a);
      break;
  }
  if (kLab && a.trace) {
    (void)hipStreamSynchronize(s);
    const uint32_t nw = std::min<uint32_t>(nb * 2u, WTRACE_MAX);
    std::vector<unsigned long long> h(3 * (size_t)nw);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_wtrace), h.size() * sizeof(unsigned long long)) == hipSuccess)
      for (uint32_t w = 0; w < nw; w++)
        fprintf(stderr, "WTRACE %u %u %llu %llu %llu\n", w / 2u, w % 2u, h[3 * w + 2], h[3 * w], h[3 * w + 1]);
  }
}

#ifdef TDTK_LAB
#include "lab_part_5.inc"
#endif   // TDTK_LAB

// a.fuse != 0 (only where search_can_fuse(a.n)): the base pair sums come out of the search itself, one row of
// ACC_TOTAL per workgroup in a.partials (search_fused_rows(a.n) rows) -- follow with launch_final.
hipError_t launch_search(const SearchArgs& a_in, uint32_t grid, int dirmode, bool count, hipStream_t s)
{
  if (a_in.n == 0) return hipSuccess;
  SearchArgs a = a_in;
  {
    // the lane-group kernels can walk two tree levels per trip (KdFat) -- lab, on request: measured slower here too
    const char* e = lab_env("TDTK_FAT_SMALL");
    if (!(e && e[0] == '1') && pick_variant(a.n) != 20) a.T.fat = nullptr;
  }
  if (!kLab) { a.bounds = nullptr; a.trace = 0; }
  dim3 g(grid), b(SEARCH_BLOCK);
  if (dirmode == 1) hipLaunchKernelGGL((k_search<SEARCH_BLOCK, 8, false, 1, false, 1>), g, b, 0, s, a);
  else if (dirmode == 2) hipLaunchKernelGGL((k_search<SEARCH_BLOCK, 8, false, 2, false, 1>), g, b, 0, s, a);
  else {
    int v = pick_variant(a.n);
    // (an -R pass -- SearchArgs::skip -- takes the product's three families only: the lab kernels do not read the mask)
    if (a.skip && !(v == 20 || v == 4 || v == 10)) v = (a.n >= (size_t)262144) ? 20 : ((a.n >= (size_t)98304) ? 4 : 10);
    if (a.fuse && !(v == 20 || ((v == 10 || v == 4) && !count))) return hipErrorInvalidValue;
    // the product's three families: persistent lanes (20; sums by each wave over its own slab: FUSE 3), one query per
    // lane (4), four lanes per query (10) -- and their instrumented instantiations
    if (!kLab && v == 20 && !(a.fuse == 0 || a.fuse == 3)) return hipErrorInvalidValue;
#ifdef TDTK_LAB
    if (v == 30 && (!a.q_ctr || !a.q_ctr_next)) return hipErrorInvalidValue;
#endif
    if (count) {
      // the instrumented instantiation of whatever this batch would get: same traversal, same warm radius
      if (v == 20) {
        if (a.fuse == 3) launch_refill128<true, 3>(a, s);
#ifdef TDTK_LAB
        else if (a.fuse == 2) launch_refill128<true, 2>(a, s);
        else if (a.fuse) launch_refill128<true, 1>(a, s);
#endif
        else launch_refill128<true, 0>(a, s);
      }
#ifdef TDTK_LAB
      else if (v == 30) launch_stream128<true>(a, s);
      else if (v == 40) launch_step128<true, false>(a, s);
      else if (v == 41) launch_step128<true, true>(a, s);
#endif
      else hipLaunchKernelGGL((k_search<SEARCH_BLOCK, 8, true, 0, false, 1>), g, b, 0, s, a);
      return hipGetLastError();
    }
    switch (v) {
#ifdef TDTK_LAB
#include "lab_part_6.inc"
#endif
      case 20:
        if (a.fuse == 3) launch_refill128<false, 3>(a, s);
#ifdef TDTK_LAB
        else if (a.fuse == 2) launch_refill128<false, 2>(a, s);
        else if (a.fuse) launch_refill128<false, 1>(a, s);
#endif
        else launch_refill128<false, 0>(a, s);
        break;
      case 10:
#ifdef TDTK_LAB
        if (a.loop) {
          if (!a.fuse) return hipErrorInvalidValue;
          hipLaunchKernelGGL((k_search_g8<256, 16, 4, true, true>), dim3(g8_grid4(a.n)), dim3(256), 0, s, a);
          break;
        }
#endif
        if (a.fuse) hipLaunchKernelGGL((k_search_g8<256, 16, 4, true>), dim3(g8_grid4(a.n)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_search_g8<256, 16, 4>), dim3(g8_grid4(a.n)), dim3(256), 0, s, a);
        break;
      default:
#ifdef TDTK_LAB
        if (a.loop) return hipErrorInvalidValue;      // (the host-free loop is the four-lanes-per-query family's alone)
#endif
        if (a.fuse) hipLaunchKernelGGL((k_search<SEARCH_BLOCK, 4, false, 0, true, 1, 4, true>), g, b, 0, s, a);
        else hipLaunchKernelGGL((k_search<SEARCH_BLOCK, 4, false, 0, true, 1>), g, b, 0, s, a);
        break;
    }
  }
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) k_make_hot(const KdNode* __restrict__ nodes, size_t n, KdHot* __restrict__ hot, double2* __restrict__ split)
{
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const KdNode nd = nodes[i];
  KdHot h;
  memset(&h, 0, sizeof h);
  h.cx = (float)nd.cx; h.cy = (float)nd.cy; h.cz = (float)nd.cz;
  h.hx = (float)nd.hx; h.hy = (float)nd.hy; h.hz = (float)nd.hz;
  h.splitval = nd.splitval; h.c1 = nd.c1; h.c2 = nd.c2;
  h.axis = ((nd.c1 >> 30) & 1u) | (((nd.c2 >> 30) & 1u) << 1);
  hot[i] = h;
  if (split) split[i] = make_double2(nd.splitval, __hiloint2double((int)nd.c2, (int)nd.c1));     // { splitval, c1 | c2 << 32 }
}
#ifdef TDTK_LAB   // two tree levels per record: a measured negative (see the FAT walk in search_refill_body)
#include "lab_part_7.inc"
#endif   // TDTK_LAB
hipError_t launch_make_hot(const KdNode* nodes, size_t n, KdHot* hot, hipStream_t s, double2* split)
{
  if (!n) return hipSuccess;
  hipLaunchKernelGGL(k_make_hot, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, nodes, n, hot, split);
  return hipGetLastError();
}

// ---- bucket groups: the search-side layout of the leaf points (round 3) --------------------------------------------
// Every bucket is padded to a multiple of four slots in the leaf-ordered point array (the pad slots repeat the bucket's
// last point, which can never pass the strict '<' of the leaf scan a second time), so that a bucket starts at a
// multiple of four and owns (count + 3) / 4 GROUPS.  Group g = slots 4g .. 4g+3 has a 48-byte fp32 shadow record
// { x[4], y[4], z[4] }: three 16-byte loads bring four points, and a whole default bucket (<= 20 points) arrives in one
// round trip and 60 registers.  The shadow only ever REJECTS points that provably cannot pass the reference's
// `myd2 < closest_d2` (see GroupF32); whatever survives is read from the fp64 record and tested exactly, in order.
// Child references of the node records are rewritten to the padded starts, so every other kernel keeps scanning
// (start, count) runs of the same array unchanged.
__global__ void __launch_bounds__(256) k_pad_mark(const KdNode* __restrict__ nodes, size_t n, const LeafEntry* __restrict__ leaf_tab,
                                                  uint32_t cb, uint32_t cmask, uint32_t* __restrict__ ng_at)
{
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * n) return;
  const KdNode& nd = nodes[t >> 1];
  const uint32_t ref = (t & 1) ? nd.c2 : nd.c1;
  if (!(ref & REF_LEAF)) return;
  const uint32_t v = ref & REF_VAL;
  uint32_t start, count;
  if (leaf_tab) { start = (uint32_t)leaf_tab[v].start; count = (uint32_t)leaf_tab[v].count; }
  else { start = v >> cb; count = v & cmask; }
  ng_at[start] = (count + 3u) >> 2;
}
struct Q16Grid { double lo[3]; double scale; };
__global__ void __launch_bounds__(256) k_pad_fill(KdNode* __restrict__ nodes, size_t n, LeafEntry* __restrict__ leaf_tab, uint32_t cb,
                                                  uint32_t cmask, const uint32_t* __restrict__ g_at, const KdPoint* __restrict__ pts,
                                                  KdPoint* __restrict__ ptsP, float4* __restrict__ grp, uint32_t* __restrict__ q16, const Q16Grid qg)
{
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= 2 * n) return;
  KdNode& nd = nodes[t >> 1];
  const uint32_t ref = (t & 1) ? nd.c2 : nd.c1;
  if (!(ref & REF_LEAF)) return;
  const uint32_t v = ref & REF_VAL;
  uint32_t start, count;
  if (leaf_tab) { start = (uint32_t)leaf_tab[v].start; count = (uint32_t)leaf_tab[v].count; }
  else { start = v >> cb; count = v & cmask; }
  const uint32_t g0 = g_at[start], ng = (count + 3u) >> 2;
  for (uint32_t k = 0; k < ng; k++) {
    float fx[4], fy[4], fz[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
      const uint32_t src = start + min(4u * k + j, count - 1u);
      const KdPoint P = pts[src];
      ptsP[(size_t)4 * (g0 + k) + j] = P;
      fx[j] = (float)P.x; fy[j] = (float)P.y; fz[j] = (float)P.z;   // round to nearest: what GroupF32's error bound assumes
    }
    grp[(size_t)3 * (g0 + k) + 0] = make_float4(fx[0], fx[1], fx[2], fx[3]);
    grp[(size_t)3 * (g0 + k) + 1] = make_float4(fy[0], fy[1], fy[2], fy[3]);
    grp[(size_t)3 * (g0 + k) + 2] = make_float4(fz[0], fz[1], fz[2], fz[3]);
    if (q16) {
      uint32_t ix[4], iy[4], iz[4];
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const KdPoint P = ptsP[(size_t)4 * (g0 + k) + j];
        bool out = false;     // (a point of the tree is inside its root box)
        ix[j] = (uint32_t)q16_index(P.x, qg.lo[0], qg.scale, out) & 0xFFFFu;
        iy[j] = (uint32_t)q16_index(P.y, qg.lo[1], qg.scale, out) & 0xFFFFu;
        iz[j] = (uint32_t)q16_index(P.z, qg.lo[2], qg.scale, out) & 0xFFFFu;
      }
      uint32_t* w = q16 + (size_t)6 * (g0 + k);
      w[0] = ix[0] | (iy[0] << 16); w[1] = ix[1] | (iy[1] << 16); w[2] = iz[0] | (iz[1] << 16);
      w[3] = ix[2] | (iy[2] << 16); w[4] = ix[3] | (iy[3] << 16); w[5] = iz[2] | (iz[3] << 16);
    }
  }
  if (leaf_tab) leaf_tab[v].start = (int32_t)(4u * g0);
  else {
    const uint32_t nv = (ref & ~REF_VAL) | ((4u * g0) << cb) | count;
    if (t & 1) nd.c2 = nv; else nd.c1 = nv;
  }
}
// pass 1: how many groups each bucket needs, written at the bucket's start position (ng_at[0 .. M], zeroed here)
hipError_t launch_pad_mark(const KdNode* nodes, size_t n_internal, const LeafEntry* leaf_tab, uint32_t cb, uint32_t cmask, uint32_t* ng_at,
                           size_t M, hipStream_t s)
{
  hipError_t e = hipMemsetAsync(ng_at, 0, (M + 1) * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_pad_mark, dim3((uint32_t)((2 * n_internal + 255) / 256)), dim3(256), 0, s, nodes, n_internal, leaf_tab, cb, cmask, ng_at);
  return hipGetLastError();
}
// pass 2 (after an exclusive scan of ng_at into g_at): padded points, shadow groups, rewritten references
hipError_t launch_pad_fill(KdNode* nodes, size_t n_internal, LeafEntry* leaf_tab, uint32_t cb, uint32_t cmask, const uint32_t* g_at,
                           const KdPoint* pts, KdPoint* ptsP, float4* grp, hipStream_t s, uint32_t* q16, const double* q_lo, double q_scale)
{
  Q16Grid qg{};
  if (q16 && q_lo) { qg.lo[0] = q_lo[0]; qg.lo[1] = q_lo[1]; qg.lo[2] = q_lo[2]; qg.scale = q_scale; }
  hipLaunchKernelGGL(k_pad_fill, dim3((uint32_t)((2 * n_internal + 255) / 256)), dim3(256), 0, s, nodes, n_internal, leaf_tab, cb, cmask, g_at, pts,
                     ptsP, grp, (q16 && q_lo) ? q16 : nullptr, qg);
  return hipGetLastError();
}

hipError_t launch_final(const double* partials, uint32_t rows, double* d_out, hipStream_t s, int ncols)
{
  hipLaunchKernelGGL(k_final, dim3(ACC_TOTAL), dim3(256), 0, s, partials, (int)rows, d_out, ncols);
  return hipGetLastError();
}

// ---- several batches in one launch (the link passes of a graph-SLAM round) ----
// Per batch: the grid and slab length the persistent-lane kernel would get beside `concurrent` other passes.  Only for
// batches the big-batch kernel takes (search_can_fuse(n)).
// Slab length inside such a launch: the dispatcher keeps every wave slot of the chip busy with whatever batch comes
// next, so occupancy does not depend on the slab and longer slabs (fewer drains per query) win until the last
// generation of waves becomes the tail: 84 links of 1M points in launches of 64: 224 -> 12.6 ms, 320 -> 12.0, 512 -> 11.5,
// 640 -> 11.4, 1024 -> 11.5, 1536 -> 11.7; a rank's 11 links in one launch: 320 -> 1.79 ms, 448 -> 1.77, 640 -> 1.81
// (three streams: 12.6 / 1.89 on the same box).
int search_multi_class(size_t n) { const int v = pick_variant(n); return (v == 20 || v == 4 || v == 10) ? v : 0; }
// lab (TDTK_MULTI_BLOCK=64): the several-links launch in workgroups of ONE wave -- a wave slot is free again when its wave is
// done, not when its workgroup's slower wave is (the rows of partial sums are then one per wave)
static bool multi_block64()
{
#ifdef TDTK_LAB
  const char* e = lab_env("TDTK_MULTI_BLOCK");
  return e && atoi(e) == 64;
#else
  return false;
#endif
}
uint32_t search_multi_prepare(SearchArgs& a, int links_in_launch, bool long_slabs)
{
  const int v = pick_variant(a.n);
  if (v == 4) return search_grid(a.n);
  if (v == 10) { const uint32_t g = g8_grid(a.n) / 2; return g < 8 ? 8u : (g + 7) / 8 * 8; }
  int qpw;
  uint32_t nb = refill_grid_b(a.n, 128, &qpw, 2);
  if (!lab_env("TDTK_REFILL_QPW")) {
    // (long_slabs: the waves add up their own slabs -- FUSE 5 -- so the slab length decides which queries share a row of
    // partial sums; one length for every launch then, however many links share it)
    const int want = (links_in_launch > 16 || long_slabs) ? 640 : 448;
    if (want > qpw) {
      qpw = want;
      const size_t waves = (a.n + (size_t)qpw - 1) / (size_t)qpw;
      size_t b = (waves + 1) / 2;
      b = (b + 7) & ~(size_t)7;
      if (b < 8) b = 8;
      nb = (uint32_t)b;
    }
  }
  a.qpw = qpw; a.pool_slab = 0; a.region = 0; a.trace = 0; a.side_by_side = links_in_launch;
  // A wave's slab in pieces: the waves of an XCD then sweep their eighth of the link's scan together, piece by piece, and
  // the XCD's L2 has to hold 1/pieces of what it holds otherwise.  Unlike the 1M-vs-1M pair, 64 scans do not fit the
  // Infinity Cache, so these are real HBM bytes: 84 link passes of 1M queries (round 3, gpurun_out/keep/r03_ph*):
  // one piece 0.241 GB of fabric traffic per link and 62 % L2 hits, four pieces 0.162 GB and 76 % -- but every piece
  // ends with a drain, and time goes the other way: 10.86 ms (1 piece), 10.85 (2), 11.01 (4), 11.42 (8).  Two pieces are
  // free; TDTK_LINK_PHASES overrides (pieces stay multiples of 16 queries).
  int ph = 2;
  if (const char* e = lab_env("TDTK_LINK_PHASES")) ph = atoi(e);
  if (ph < 1) ph = 1;
  while (ph > 1 && (qpw % (ph * 16)) != 0) --ph;
  a.phases = ph;
  if (long_slabs && multi_block64()) nb *= 2;
  return nb;
}
int search_multi_thresh(size_t n) { return refill_thresh(n); }
hipError_t launch_search_multi(const SearchArgs* d_args, const uint32_t* d_base, int nbatch, uint32_t total_blocks, int cls, int thresh,
                               bool count, hipStream_t s, bool ordered, bool lum_sums)
{
  if (lum_sums && !(!count && cls == 20 && (thresh == 16 || thresh == 32))) return hipErrorInvalidValue;   // (ORDER instantiation; no cost bytes: slab order)
  if (!nbatch || !total_blocks) return hipSuccess;
  if (cls == 4 || cls == 10) {   // small batches: one query per lane / four lanes per query, as launch_search would pick
    const dim3 gs(total_blocks), bs(SEARCH_BLOCK);
    if (count) hipLaunchKernelGGL((k_search_multi<SEARCH_BLOCK, 8, true, false, 1>), gs, bs, 0, s, d_args, d_base, nbatch);
    else if (cls == 4) hipLaunchKernelGGL((k_search_multi<SEARCH_BLOCK, 4, false, true, 1>), gs, bs, 0, s, d_args, d_base, nbatch);
    else hipLaunchKernelGGL((k_search_g8_multi<256, 16, 4>), gs, dim3(256), 0, s, d_args, d_base, nbatch);
    return hipGetLastError();
  }
  const dim3 g(total_blocks), b(128);
#ifdef TDTK_LAB
  if (!count && lum_sums && multi_block64()) {
    if (thresh == 32) hipLaunchKernelGGL((k_search_refill_multi<64, 4, 32, 4, false, 5, true>), g, dim3(64), 0, s, d_args, d_base, nbatch);
    else hipLaunchKernelGGL((k_search_refill_multi<64, 4, 16, 4, false, 5, true>), g, dim3(64), 0, s, d_args, d_base, nbatch);
    return hipGetLastError();
  }
#endif
  if (count) {
    switch (thresh) {
#ifdef TDTK_LAB
      case 8: hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 8, 4, true, 0>), g, b, 0, s, d_args, d_base, nbatch); break;
#endif
      case 32: hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 32, 4, true, 0>), g, b, 0, s, d_args, d_base, nbatch); break;
      default: hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 16, 4, true, 0>), g, b, 0, s, d_args, d_base, nbatch); break;
    }
  } else {
    switch (thresh) {
#ifdef TDTK_LAB
      case 8:
        if (ordered) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 8, 4, false, 0, true>), g, b, 0, s, d_args, d_base, nbatch);
        else hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 8, 4, false, 0>), g, b, 0, s, d_args, d_base, nbatch);
        break;
#endif
      case 32:
#ifdef TDTK_LAB
        if (lum_sums && pipe_on()) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 32, 4, false, 5, true, true>), g, b, 0, s, d_args, d_base, nbatch);
        else
#endif
        if (lum_sums) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 32, MULTI_WPS, false, 5, true>), g, b, 0, s, d_args, d_base, nbatch);
        else if (ordered) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 32, 4, false, 0, true>), g, b, 0, s, d_args, d_base, nbatch);
        else hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 32, 4, false, 0>), g, b, 0, s, d_args, d_base, nbatch);
        break;
      default:
#ifdef TDTK_LAB
        if (lum_sums && pipe_on()) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 16, 4, false, 5, true, true>), g, b, 0, s, d_args, d_base, nbatch);
        else
#endif
        if (lum_sums) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 16, MULTI_WPS, false, 5, true>), g, b, 0, s, d_args, d_base, nbatch);
        else if (ordered) hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 16, 4, false, 0, true>), g, b, 0, s, d_args, d_base, nbatch);
        else hipLaunchKernelGGL((k_search_refill_multi<128, MULTI_SD, 16, 4, false, 0>), g, b, 0, s, d_args, d_base, nbatch);
        break;
    }
  }
  return hipGetLastError();
}

constexpr int ACC_BLOCK = 256;
uint32_t accum_grid(size_t n)
{
  size_t want = (n + ACC_BLOCK * 4 - 1) / (ACC_BLOCK * 4);   // k_accum: 4 queries per thread and step
  size_t cap = (size_t)num_cu() * 4;
  size_t nb = want < cap ? want : cap;
  return (uint32_t)(nb ? nb : 1);
}

template <unsigned WANT>
static void launch_accum_w(const AccumArgs& a, uint32_t grid, int pmode, hipStream_t s)
{
  dim3 g(grid), b(ACC_BLOCK);
  if (pmode == 0) hipLaunchKernelGGL((k_accum<ACC_BLOCK, WANT, 0>), g, b, 0, s, a);
  else if (pmode == 1) hipLaunchKernelGGL((k_accum<ACC_BLOCK, WANT, 1>), g, b, 0, s, a);
  else hipLaunchKernelGGL((k_accum<ACC_BLOCK, WANT, 2>), g, b, 0, s, a);
}

hipError_t launch_accum(const AccumArgs& a, uint32_t grid, unsigned want, int pmode, double* d_out,
                        hipStream_t s)
{
  if (want & (TDTK_WANT_GAPX | TDTK_WANT_MOM2)) {  // both are the MM + DD columns on top of the base block
    launch_accum_w<TDTK_WANT_GAPX>(a, grid, pmode, s);
  } else if (want == (TDTK_WANT_LUM | ACC_WANT_NO_CROSS)) {
    launch_accum_w<TDTK_WANT_LUM | ACC_WANT_NO_CROSS>(a, grid, pmode, s);
  } else
  switch (want & 7u) {
    case 0: launch_accum_w<0>(a, grid, pmode, s); break;
    case 1: launch_accum_w<1>(a, grid, pmode, s); break;
    case 2: launch_accum_w<2>(a, grid, pmode, s); break;
    case 3: launch_accum_w<3>(a, grid, pmode, s); break;
    case 4: launch_accum_w<4>(a, grid, pmode, s); break;
    case 5: launch_accum_w<5>(a, grid, pmode, s); break;
    case 6: launch_accum_w<6>(a, grid, pmode, s); break;
    default: launch_accum_w<7>(a, grid, pmode, s); break;
  }
  hipLaunchKernelGGL(k_final, dim3(ACC_TOTAL), dim3(256), 0, s, a.partials, (int)grid, d_out, (int)ACC_TOTAL);
  return hipGetLastError();
}

// want: TDTK_WANT_LUM | ACC_WANT_NO_CROSS (a lum6DEuler link) or TDTK_WANT_LUM or TDTK_WANT_GAPX / MOM2 or 0 (base sums)
hipError_t launch_accum_multi(const AccumArgs* d_args, const uint32_t* d_base, int nbatch, uint32_t total_blocks, unsigned want,
                              const FinalDesc* d_final, hipStream_t s, bool rows_done)
{
  if (!nbatch || !total_blocks) return hipSuccess;
  const dim3 g(total_blocks), b(ACC_BLOCK);
  if (rows_done) {}   // (the search wrote the rows itself: FUSE 5)
  else if (want & (TDTK_WANT_GAPX | TDTK_WANT_MOM2)) hipLaunchKernelGGL((k_accum_multi<ACC_BLOCK, TDTK_WANT_GAPX, 0>), g, b, 0, s, d_args, d_base, nbatch);
  else if (want == (TDTK_WANT_LUM | ACC_WANT_NO_CROSS)) hipLaunchKernelGGL((k_accum_multi<ACC_BLOCK, TDTK_WANT_LUM | ACC_WANT_NO_CROSS, 0>), g, b, 0, s, d_args, d_base, nbatch);
  else if ((want & 7u) == TDTK_WANT_LUM) hipLaunchKernelGGL((k_accum_multi<ACC_BLOCK, TDTK_WANT_LUM, 0>), g, b, 0, s, d_args, d_base, nbatch);
  else if ((want & 7u) == 0u) hipLaunchKernelGGL((k_accum_multi<ACC_BLOCK, 0u, 0>), g, b, 0, s, d_args, d_base, nbatch);
  else return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_final_multi, dim3(ACC_TOTAL, nbatch), dim3(256), 0, s, d_final);
  return hipGetLastError();
}

hipError_t launch_transform(double* x, double* y, double* z, double* nx, double* ny, double* nz,
                            size_t n, const Mat4& A, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  size_t cap = (size_t)num_cu() * 8;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(k_transform, dim3((uint32_t)nb), dim3(256), 0, s, x, y, z, nx, ny, nz, n, A);
  return hipGetLastError();
}

hipError_t launch_transform_chain_batch(const XfChainDesc* d_desc, int count, size_t max_n, hipStream_t s)
{
  if (count <= 0 || !max_n) return hipSuccess;
  size_t nb = (max_n + 255) / 256;
  size_t cap = ((size_t)num_cu() * 8 + count - 1) / count;
  if (cap < 8) cap = 8;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(k_transform_chain_batch, dim3((uint32_t)nb, (uint32_t)count), dim3(256), 0, s, d_desc);
  return hipGetLastError();
}

hipError_t launch_bin(const BinArgs& b, hipStream_t s)
{
  if (!b.n) return hipSuccess;
  size_t nb = (b.n + 255) / 256;
  size_t cap = (size_t)num_cu() * 8;
  if (nb > cap) nb = cap;
  hipError_t e = hipMemsetAsync(b.hist, 0, 32768 * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_bin_count, dim3((uint32_t)nb), dim3(256), 0, s, b);
  hipLaunchKernelGGL(k_bin_scan, dim3(1), dim3(1024), 0, s, b.hist);
  hipLaunchKernelGGL(k_bin_scatter, dim3((uint32_t)nb), dim3(256), 0, s, b);
  return hipGetLastError();
}

hipError_t launch_split_soa(const double* q, size_t n, double* x, double* y, double* z, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  size_t cap = (size_t)num_cu() * 8;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(k_split_soa, dim3((uint32_t)nb), dim3(256), 0, s, q, n, x, y, z);
  return hipGetLastError();
}

// icp6D::Point_Point_Error (icp6D.cc:293-367): sum over the pairs of exp(|p1 - p2|^2 * scale), p1 = model point
// mapped to the world, p2 = data point.  partial[block] rows, then a fixed-order fold by one workgroup.
__global__ void __launch_bounds__(256) k_pp_error(const AccumArgs a, double scale, double* __restrict__ partial)
{
  __shared__ double red[2][4];
  double se = 0.0, cn = 0.0;
  const double4* __restrict__ pts = reinterpret_cast<const double4*>(a.T.pts);
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += stride) {
    const int k = a.kpos[i];
    if (k < 0) continue;
    const double4 c = pts[k];
    double mx, my, mz;
    dev_xf3(a.A, c.x, c.y, c.z, mx, my, mz);
    const double px = mx - a.x[i], py = my - a.y[i], pz = mz - a.z[i];
    se += exp((px * px + py * py + pz * pz) * scale);
    cn += 1.0;
  }
  se = wave_sum(se); cn = wave_sum(cn);
  const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if (lane == 0) { red[0][wv] = se; red[1][wv] = cn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    partial[2 * blockIdx.x + 1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
  }
}
__global__ void __launch_bounds__(64) k_pp_final(const double* __restrict__ partial, int rows, double* __restrict__ out)
{
  if (threadIdx.x == 0) {
    double se = 0.0, cn = 0.0;
    for (int r = 0; r < rows; r++) { se += partial[2 * r]; cn += partial[2 * r + 1]; }
    out[0] = se; out[1] = cn;
  }
}
hipError_t launch_pp_error(const AccumArgs& a, uint32_t grid, double scale, double* d_partial, double* d_out, hipStream_t s)
{
  hipLaunchKernelGGL(k_pp_error, dim3(grid), dim3(256), 0, s, a, scale, d_partial);
  hipLaunchKernelGGL(k_pp_final, dim3(1), dim3(64), 0, s, d_partial, (int)grid, d_out);
  return hipGetLastError();
}

hipError_t launch_found_flags(const int* kpos, const int32_t* order, size_t n, uint32_t* flags, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_found_flags, dim3((uint32_t)nb), dim3(256), 0, s, kpos, order, n, flags);
  return hipGetLastError();
}

hipError_t launch_pair_list(const PairListArgs& a, int pmode, hipStream_t s)
{
  if (!a.n) return hipSuccess;
  size_t nb = (a.n + 255) / 256;
  if (nb > 4096) nb = 4096;
  dim3 g((uint32_t)nb), b(256);
  if (pmode == 1) hipLaunchKernelGGL(k_pair_list<1>, g, b, 0, s, a);
  else if (pmode == 2) hipLaunchKernelGGL(k_pair_list<2>, g, b, 0, s, a);
  else hipLaunchKernelGGL(k_pair_list<0>, g, b, 0, s, a);
  return hipGetLastError();
}

hipError_t launch_scatter_idx(const int* kpos, const double* d2s, const int32_t* order,
                              const KdPoint* pts, size_t n, int32_t* idx_out, double* d2_out,
                              hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  size_t cap = (size_t)num_cu() * 8;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(k_scatter_idx, dim3((uint32_t)nb), dim3(256), 0, s, kpos, d2s, order, pts, n,
                     idx_out, d2_out);
  return hipGetLastError();
}

hipError_t launch_skip_from_mask(const unsigned char* mask_bits, const int32_t* order, size_t n, unsigned char* skip, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  const size_t cap = (size_t)num_cu() * 8;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(k_skip_from_mask, dim3((uint32_t)nb), dim3(256), 0, s, mask_bits, order, n, skip);
  return hipGetLastError();
}
hipError_t launch_idx_hash(const int* kpos, const int32_t* order, const KdPoint* pts, size_t n, unsigned long long* out, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  const size_t cap = (size_t)num_cu() * 8;
  if (nb > cap) nb = cap;
  hipLaunchKernelGGL(k_idx_hash, dim3((uint32_t)nb), dim3(256), 0, s, kpos, order, pts, n, out);
  return hipGetLastError();
}

}  // namespace tdtk
