// GPU construction of the model-scan search tree.
//
// Same tree as KDTreeImpl::create (include/slam6d/kdTreeImpl.h:82-201) and therefore the same
// flattened layout kd_build.cpp produces on the host -- node for node, bucket for bucket (the
// test compares the two bitwise) -- but built level by level on the device:
//
//   * the points travel with the permutation (SoA in run order), so every pass is a coalesced
//     stream instead of an indirect gather;
//   * k_measure: one wavefront per (node, axis).  Bounding box by wave reduction; the centroid is
//     an ORDER-DEPENDENT fp64 sum in the reference (first point, then += the rest left to right,
//     kdTreeImpl.h:94-111), so it is reproduced as exactly that chain: 64 coalesced values per
//     step, parked in LDS and folded in run order.  This chain is the critical path (1M dependent
//     adds at the root; it halves per level);
//   * the in-place Hoare partition (kdTreeImpl.h:172-182) is order-equivalent to: elements already
//     on their side stay, the k-th misplaced element from the left swaps with the k-th misplaced
//     element from the right.  That is three prefix sums (rocPRIM) + a swap kernel per level for
//     ALL nodes of the level at once.
//
// One small D2H (the number of internal nodes of the level) per level is the only host sync.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"
#include "part_scan.h"

namespace tdtk {

#define WAVE 64
#define MEAS_STAGE 8     // chunks of 64 values parked in LDS together (k_measure)

struct BSeg {
  uint32_t start, n;
  int32_t parent;   // node index, -1 for the root
  uint32_t side;    // 0 -> c1, 1 -> c2
};
struct BMeas {
  double lo[3], hi[3], mean[3];
};
// what the host used to carry from level to level now stays on the device: nodes of the level, first node index
// and first bucket index of the level (lvl[L+1] is written by k_emit of level L)
struct BLevel { uint32_t nseg, node_base, leaf_base; };
#define BUILD_MAX_LEVELS 4096

// The left-to-right adds of `cnt` doubles parked in LDS at `b` (cnt a multiple of 32, wave-uniform address: every lane
// reads the same values), sixteen requested ahead of the sixteen being added, as ONE asm statement: reads, waits and adds
// together, so that no register written by a read is visible to the compiler before the wait that makes it valid.
// (Read / wait pairs in separate asm statements -- round 1's k_measure -- are only safe while the compiler never
// copies those registers in between; in k_big_stitch, with 220 live registers, it did, before the data had arrived
// whenever LDS latency went up: one wrong addend per node, only with eight builds running at once.)  The last
// iteration requests sixteen values beyond the end -- the buffer must be padded by 128 bytes -- and never adds them.
__device__ __forceinline__ void lds_chain32(double& sum, const double* b, uint32_t cnt)
{
  // v[64:95] and v[96:127] are the two groups of sixteen values (named registers, clobbered: a 128-bit asm operand has
  // no way to name its halves, and the adds need the halves of what ds_read_b128 returns); `b` must be 16-byte aligned
  uint32_t ad = (uint32_t)(size_t)(__attribute__((address_space(3))) const double*)b;   // LDS byte offset
  uint32_t pairs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cnt / 32u));
  asm volatile(
      "ds_read_b128 v[64:67], %[ad] offset:0\n\t"
      "ds_read_b128 v[68:71], %[ad] offset:16\n\t"
      "ds_read_b128 v[72:75], %[ad] offset:32\n\t"
      "ds_read_b128 v[76:79], %[ad] offset:48\n\t"
      "ds_read_b128 v[80:83], %[ad] offset:64\n\t"
      "ds_read_b128 v[84:87], %[ad] offset:80\n\t"
      "ds_read_b128 v[88:91], %[ad] offset:96\n\t"
      "ds_read_b128 v[92:95], %[ad] offset:112\n\t"
      ".Lbigchain%=:\n\t"
      "ds_read_b128 v[96:99], %[ad] offset:128\n\t"
      "ds_read_b128 v[100:103], %[ad] offset:144\n\t"
      "ds_read_b128 v[104:107], %[ad] offset:160\n\t"
      "ds_read_b128 v[108:111], %[ad] offset:176\n\t"
      "ds_read_b128 v[112:115], %[ad] offset:192\n\t"
      "ds_read_b128 v[116:119], %[ad] offset:208\n\t"
      "ds_read_b128 v[120:123], %[ad] offset:224\n\t"
      "ds_read_b128 v[124:127], %[ad] offset:240\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_add_f64 %[s], %[s], v[64:65]\n\t"
      "v_add_f64 %[s], %[s], v[66:67]\n\t"
      "v_add_f64 %[s], %[s], v[68:69]\n\t"
      "v_add_f64 %[s], %[s], v[70:71]\n\t"
      "v_add_f64 %[s], %[s], v[72:73]\n\t"
      "v_add_f64 %[s], %[s], v[74:75]\n\t"
      "v_add_f64 %[s], %[s], v[76:77]\n\t"
      "v_add_f64 %[s], %[s], v[78:79]\n\t"
      "v_add_f64 %[s], %[s], v[80:81]\n\t"
      "v_add_f64 %[s], %[s], v[82:83]\n\t"
      "v_add_f64 %[s], %[s], v[84:85]\n\t"
      "v_add_f64 %[s], %[s], v[86:87]\n\t"
      "v_add_f64 %[s], %[s], v[88:89]\n\t"
      "v_add_f64 %[s], %[s], v[90:91]\n\t"
      "v_add_f64 %[s], %[s], v[92:93]\n\t"
      "v_add_f64 %[s], %[s], v[94:95]\n\t"
      "v_add_u32 %[ad], 0x100, %[ad]\n\t"
      "ds_read_b128 v[64:67], %[ad] offset:0\n\t"
      "ds_read_b128 v[68:71], %[ad] offset:16\n\t"
      "ds_read_b128 v[72:75], %[ad] offset:32\n\t"
      "ds_read_b128 v[76:79], %[ad] offset:48\n\t"
      "ds_read_b128 v[80:83], %[ad] offset:64\n\t"
      "ds_read_b128 v[84:87], %[ad] offset:80\n\t"
      "ds_read_b128 v[88:91], %[ad] offset:96\n\t"
      "ds_read_b128 v[92:95], %[ad] offset:112\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_add_f64 %[s], %[s], v[96:97]\n\t"
      "v_add_f64 %[s], %[s], v[98:99]\n\t"
      "v_add_f64 %[s], %[s], v[100:101]\n\t"
      "v_add_f64 %[s], %[s], v[102:103]\n\t"
      "v_add_f64 %[s], %[s], v[104:105]\n\t"
      "v_add_f64 %[s], %[s], v[106:107]\n\t"
      "v_add_f64 %[s], %[s], v[108:109]\n\t"
      "v_add_f64 %[s], %[s], v[110:111]\n\t"
      "v_add_f64 %[s], %[s], v[112:113]\n\t"
      "v_add_f64 %[s], %[s], v[114:115]\n\t"
      "v_add_f64 %[s], %[s], v[116:117]\n\t"
      "v_add_f64 %[s], %[s], v[118:119]\n\t"
      "v_add_f64 %[s], %[s], v[120:121]\n\t"
      "v_add_f64 %[s], %[s], v[122:123]\n\t"
      "v_add_f64 %[s], %[s], v[124:125]\n\t"
      "v_add_f64 %[s], %[s], v[126:127]\n\t"
      "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Lbigchain%=\n\t"
      "s_waitcnt lgkmcnt(0)"
      : [s] "+v"(sum), [ad] "+v"(ad), [n] "+s"(pairs)
      :
      : "scc", "memory", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127");
}

// The same chain for a kernel that has no sixty-four registers to give away by name (k_fin_subtrees): `cnt` doubles at `b`
// (cnt a multiple of 16, any 8-byte alignment), eight requested ahead of the eight being added, the sixteen temporaries
// chosen by the compiler -- still ONE asm statement, so no register a read writes is visible before its wait.  The last
// iteration requests eight values beyond the end and never adds them: 64 bytes behind the data must be LDS.
__device__ __forceinline__ void lds_chain16(double& sum, const double* b, uint32_t cnt)
{
  uint32_t ad = (uint32_t)(size_t)(__attribute__((address_space(3))) const double*)b;   // LDS byte offset
  uint32_t pairs = (uint32_t)__builtin_amdgcn_readfirstlane((int)(cnt / 16u));
  double a0, a1, a2, a3, a4, a5, a6, a7, c0, c1, c2, c3, c4, c5, c6, c7;
  asm volatile(
      "ds_read_b64 %[a0], %[ad] offset:0\n\t"
      "ds_read_b64 %[a1], %[ad] offset:8\n\t"
      "ds_read_b64 %[a2], %[ad] offset:16\n\t"
      "ds_read_b64 %[a3], %[ad] offset:24\n\t"
      "ds_read_b64 %[a4], %[ad] offset:32\n\t"
      "ds_read_b64 %[a5], %[ad] offset:40\n\t"
      "ds_read_b64 %[a6], %[ad] offset:48\n\t"
      "ds_read_b64 %[a7], %[ad] offset:56\n\t"
      ".Lfinchain%=:\n\t"
      "ds_read_b64 %[c0], %[ad] offset:64\n\t"
      "ds_read_b64 %[c1], %[ad] offset:72\n\t"
      "ds_read_b64 %[c2], %[ad] offset:80\n\t"
      "ds_read_b64 %[c3], %[ad] offset:88\n\t"
      "ds_read_b64 %[c4], %[ad] offset:96\n\t"
      "ds_read_b64 %[c5], %[ad] offset:104\n\t"
      "ds_read_b64 %[c6], %[ad] offset:112\n\t"
      "ds_read_b64 %[c7], %[ad] offset:120\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_add_f64 %[s], %[s], %[a0]\n\t"
      "v_add_f64 %[s], %[s], %[a1]\n\t"
      "v_add_f64 %[s], %[s], %[a2]\n\t"
      "v_add_f64 %[s], %[s], %[a3]\n\t"
      "v_add_f64 %[s], %[s], %[a4]\n\t"
      "v_add_f64 %[s], %[s], %[a5]\n\t"
      "v_add_f64 %[s], %[s], %[a6]\n\t"
      "v_add_f64 %[s], %[s], %[a7]\n\t"
      "v_add_u32 %[ad], 0x80, %[ad]\n\t"
      "ds_read_b64 %[a0], %[ad] offset:0\n\t"
      "ds_read_b64 %[a1], %[ad] offset:8\n\t"
      "ds_read_b64 %[a2], %[ad] offset:16\n\t"
      "ds_read_b64 %[a3], %[ad] offset:24\n\t"
      "ds_read_b64 %[a4], %[ad] offset:32\n\t"
      "ds_read_b64 %[a5], %[ad] offset:40\n\t"
      "ds_read_b64 %[a6], %[ad] offset:48\n\t"
      "ds_read_b64 %[a7], %[ad] offset:56\n\t"
      "s_waitcnt lgkmcnt(8)\n\t"
      "v_add_f64 %[s], %[s], %[c0]\n\t"
      "v_add_f64 %[s], %[s], %[c1]\n\t"
      "v_add_f64 %[s], %[s], %[c2]\n\t"
      "v_add_f64 %[s], %[s], %[c3]\n\t"
      "v_add_f64 %[s], %[s], %[c4]\n\t"
      "v_add_f64 %[s], %[s], %[c5]\n\t"
      "v_add_f64 %[s], %[s], %[c6]\n\t"
      "v_add_f64 %[s], %[s], %[c7]\n\t"
      "s_sub_u32 %[n], %[n], 1\n\ts_cmp_lg_u32 %[n], 0\n\ts_cbranch_scc1 .Lfinchain%=\n\t"
      "s_waitcnt lgkmcnt(0)"
      : [s] "+v"(sum), [ad] "+v"(ad), [n] "+s"(pairs), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), [a4] "=&v"(a4),
        [a5] "=&v"(a5), [a6] "=&v"(a6), [a7] "=&v"(a7), [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3), [c4] "=&v"(c4),
        [c5] "=&v"(c5), [c6] "=&v"(c6), [c7] "=&v"(c7)
      :
      : "scc", "memory");
}

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const double t = __shfl_xor(v, off, WAVE); v = (t < v) ? t : v; }
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { const double t = __shfl_xor(v, off, WAVE); v = (v < t) ? t : v; }
  return v;
}
__device__ __forceinline__ double wave_add(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}
// ---- per node: bounding box + the reference's left-to-right fp64 sum ----------------------------
// One wavefront per (node, axis).  Each step the wave loads 64 consecutive values (one coalesced
// 512-B instruction, the next chunk prefetched while the current one is folded), updates the
// bounding box per lane, and parks the chunk in LDS.  The sum -- a strictly serial chain of fp64
// adds in run order, the critical path of the whole build -- then reads the values back one by
// one at a wave-uniform address (LDS broadcast, in-order returns, so the reads pipeline ahead of
// the chain) and the vector ALU issues nothing but the dependent v_add_f64 chain.
__device__ __forceinline__ void measure_body(const uint32_t vblock, const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                             const double* __restrict__ cx, const double* __restrict__ cy,
                                             const double* __restrict__ cz, BMeas* __restrict__ out, uint32_t big_min,
                                             const uint32_t min_n = 0u, const uint32_t* __restrict__ only_axis = nullptr)
{
  __shared__ alignas(16) double stage[256 / WAVE][2][MEAS_STAGE * WAVE + 16];   // + 16: lds_chain32 requests 128 bytes past the data
  // wave-uniform by construction; readfirstlane tells the compiler so
  const uint32_t w = __builtin_amdgcn_readfirstlane((vblock * 256u + threadIdx.x) / WAVE);
  const int lane = threadIdx.x & (WAVE - 1);
  if (w >= 3u * lv->nseg) return;
  const uint32_t sgi = w / 3u, ax = w % 3u;
  const uint32_t s = __builtin_amdgcn_readfirstlane(segs[sgi].start);
  const uint32_t n = __builtin_amdgcn_readfirstlane(segs[sgi].n);
  if (n >= big_min) return;   // measured by the piecewise path below (k_big_*)
  if (n < min_n || (only_axis != nullptr && only_axis[sgi] != ax)) return;      // (k_chain_exact: the big nodes' split axes only)
  const double* __restrict__ arr = ((ax == 0) ? cx : ((ax == 1) ? cy : cz)) + s;
  double(*buf)[MEAS_STAGE * WAVE + 16] = stage[threadIdx.x / WAVE];

  const double first = arr[0];
  double lo = first, hi = first;
  double sum = first;  // the sum starts from the first point (kdTreeImpl.h:97-101) ...
  // A stage = MEAS_STAGE chunks of 64 values: loaded from memory one stage ahead, parked in LDS together, folded as
  // one run of 256 dependent adds.  The per-stage costs (LDS write, fence, the first reads' round trip) are paid once
  // per 256 adds; within the run 32 LDS values are requested ahead of the chain (left to itself the compiler keeps
  // two reads outstanding and every other add waits a full LDS round trip).
  double vq[MEAS_STAGE];
#pragma unroll
  for (int j = 0; j < MEAS_STAGE; j++) vq[j] = ((uint32_t)(j * WAVE + lane) < n) ? arr[j * WAVE + lane] : first;
  int cur = 0;
  for (uint32_t base = 0; base < n; base += WAVE * MEAS_STAGE, cur ^= 1) {
    const uint32_t cnt = (n - base < WAVE * MEAS_STAGE) ? (n - base) : WAVE * MEAS_STAGE;
#pragma unroll
    for (int j = 0; j < MEAS_STAGE; j++) {
      const double v = vq[j];
      lo = (v < lo) ? v : lo;  // lanes past the end carry `first`, harmless for min/max
      hi = (hi < v) ? v : hi;
      buf[cur][j * WAVE + lane] = v;
      const uint32_t nb = base + WAVE * MEAS_STAGE + (uint32_t)j * WAVE;
      vq[j] = (nb + lane < n) ? arr[nb + lane] : first;   // the next stage
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double* __restrict__ b = buf[cur];
    if (cnt == WAVE * MEAS_STAGE && base != 0) {
      lds_chain32(sum, b, WAVE * MEAS_STAGE);
    } else {
      // first stage of a node (the sum starts FROM the first point) and the ragged last one: 16 values per round trip
      uint32_t k = (base == 0) ? 1u : 0u;
      for (; k + 16 <= cnt; k += 16) {
        double r[16];
#pragma unroll
        for (int q = 0; q < 16; q++) r[q] = b[k + q];
#pragma unroll
        for (int q = 0; q < 16; q++) sum += r[q];
      }
      for (; k < cnt; k++) sum += b[k];  // ... and adds the rest in order
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double t;
    t = __shfl_xor(lo, off, WAVE); lo = (t < lo) ? t : lo;
    t = __shfl_xor(hi, off, WAVE); hi = (hi < t) ? t : hi;
  }
  if (lane == 0) {
    out[sgi].lo[ax] = lo;
    out[sgi].hi[ax] = hi;
    out[sgi].mean[ax] = sum / (double)n;
  }
}
__global__ void __launch_bounds__(256) k_measure(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                 const double* __restrict__ cx, const double* __restrict__ cy,
                                                 const double* __restrict__ cz, BMeas* __restrict__ out, uint32_t big_min)
{
  measure_body(blockIdx.x, segs, lv, cx, cy, cz, out, big_min);
}


// The same for every node below the piecewise path's threshold, ONE wavefront per node and all three axes in it (round 3).
// k_measure's wave per (node, axis) is sized for long chains: 33 KB of LDS per workgroup and 128 registers per lane keep
// four workgroups on a compute unit, and a level of many small nodes (65 000 of 15 points at level 16 of a 1M-point tree:
// 196 000 waves) runs as 48 generations of waves that each wait out two dependent round trips to memory for a chain of a
// few adds -- 139 us for that level, 29 us for every level from 9 to 12.  Here a wave loads 64 points of its node per
// step (three coalesced instructions, the next step's requested before the current one is folded), keeps the bounding
// boxes per lane, parks the values in LDS, and lane 0 alone folds the three sums -- three independent chains side by
// side (issue-bound at 24 cycles per point: worse than three waves at 10 for a long chain, which is why the host picks
// this kernel for the levels of small nodes only).  12 KB of LDS per workgroup, eight waves per SIMD.
#define MN_CH 64
__global__ void __launch_bounds__(256) k_measure_node(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                      const double* __restrict__ cx, const double* __restrict__ cy,
                                                      const double* __restrict__ cz, BMeas* __restrict__ out, uint32_t big_min)
{
  __shared__ alignas(16) double stage[256 / WAVE][2][3][MN_CH];
  const uint32_t sgi = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / WAVE);
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  if (sgi >= lv->nseg) return;
  const uint32_t s = __builtin_amdgcn_readfirstlane(segs[sgi].start);
  const uint32_t n = __builtin_amdgcn_readfirstlane(segs[sgi].n);
  if (n >= big_min) return;   // measured by the piecewise path (k_big_*)
  const double* __restrict__ ax = cx + s;
  const double* __restrict__ ay = cy + s;
  const double* __restrict__ az = cz + s;
  double(*buf)[3][MN_CH] = stage[threadIdx.x / WAVE];
  const double fx = ax[0], fy = ay[0], fz = az[0];
  double lox = fx, hix = fx, loy = fy, hiy = fy, loz = fz, hiz = fz;
  double sx = fx, sy = fy, sz = fz;          // the sums start from the first point (kdTreeImpl.h:97-101); lane 0's count
  double vx = (lane < n) ? ax[lane] : fx, vy = (lane < n) ? ay[lane] : fy, vz = (lane < n) ? az[lane] : fz;
  int cur = 0;
  for (uint32_t base = 0; base < n; base += MN_CH, cur ^= 1) {
    const uint32_t cnt = (n - base < MN_CH) ? (n - base) : MN_CH;
    lox = (vx < lox) ? vx : lox; hix = (hix < vx) ? vx : hix;   // lanes past the end carry the first point: harmless
    loy = (vy < loy) ? vy : loy; hiy = (hiy < vy) ? vy : hiy;
    loz = (vz < loz) ? vz : loz; hiz = (hiz < vz) ? vz : hiz;
    buf[cur][0][lane] = vx; buf[cur][1][lane] = vy; buf[cur][2][lane] = vz;
    const uint32_t nb = base + MN_CH + lane;
    vx = (nb < n) ? ax[nb] : fx; vy = (nb < n) ? ay[nb] : fy; vz = (nb < n) ? az[nb] : fz;   // the next step
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const double* __restrict__ bx = buf[cur][0];
      const double* __restrict__ by = buf[cur][1];
      const double* __restrict__ bz = buf[cur][2];
      uint32_t k = (base == 0) ? 1u : 0u;    // ... and add the rest in order
      for (; k + 4 <= cnt; k += 4) {
        double rx[4], ry[4], rz[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { rx[q] = bx[k + q]; ry[q] = by[k + q]; rz[q] = bz[k + q]; }
#pragma unroll
        for (int q = 0; q < 4; q++) { sx += rx[q]; sy += ry[q]; sz += rz[q]; }
      }
      for (; k < cnt; k++) { sx += bx[k]; sy += by[k]; sz += bz[k]; }
    }
    __builtin_amdgcn_wave_barrier();
  }
  lox = wave_min(lox); hix = wave_max(hix);
  loy = wave_min(loy); hiy = wave_max(hiy);
  loz = wave_min(loz); hiz = wave_max(hiz);
  if (lane == 0) {
    BMeas m;
    m.lo[0] = lox; m.lo[1] = loy; m.lo[2] = loz;
    m.hi[0] = hix; m.hi[1] = hiy; m.hi[2] = hiz;
    m.mean[0] = sx / (double)n; m.mean[1] = sy / (double)n; m.mean[2] = sz / (double)n;
    out[sgi] = m;
  }
}

// ---- big nodes: the same left-to-right fp64 sum, without walking it one add at a time --------------------------
// The reference's centroid is s <- fl(s + x_i) over the node's points in run order: a million roundings at the root,
// each depending on the one before.  While the running sum stays inside one binade [2^e, 2^(e+1)) every rounding is
// to a multiple of the same u = 2^(e-52), and with s = M u (M an integer, 2^52 <= M < 2^53) one add is integer
// arithmetic: x / u = k + rem / 2^d exactly (d = e - exponent(x) bits shifted out of x's mantissa), and
//     M <- M + k + [rem > half]                       rem != half
//     M <- the even one of M + k, M + k + 1           rem == half (round-half-even: it depends on the parity of M)
// (mirrored for x of the other sign).  So a PIECE of the node's run -- the part of it inside one 64-aligned block of
// positions -- acts on M as "add T", as long as M stays strictly between 2^52 and 2^53 on the way; T and the lowest
// and highest offset reached are computed for every piece of every big node at once, one lane per piece, for both
// parities of the incoming M, before that M is known: all that has to be guessed is the binade, and a plain parallel
// prefix sum is accurate enough for that.  Pieces whose offsets fit with room to spare compose -- (T, lowest,
// highest, by parity) is a monoid -- so a segmented scan folds every run of such pieces into one step; the pieces in
// which the sum changes binade or sign (3-8 % on the bench's clouds, whose Morton-ordered zero-mean coordinates keep
// the sum wandering through zero) are walked add by add.  What is left of the chain is one wave per (node, axis)
// visiting those pieces: apply the folded run before it (checked EXACTLY against the real M: exponent, sign, room),
// walk its 64 points.  A run that does not fit after all is redone piece by piece.  Bit for bit the reference's sum
// by construction; tdtk_tree_verify compares every node record with the host build (tools/bigsum_probe.py).
#define BIG_CH 64u
#define BIG_MIN 8192u          // nodes from this many points on take this path (below, the chain costs < 40 us) ...
#define BIG_MIN_SMALL 2048u    // ... in a small cloud (that chain is on its critical path: 46 + 26 us of a 15K-point build); never below
                               // BIG_PB: a block of the partial sums holds the tail of one big node and the head of one other
// which of the two a build of M points uses (every kernel of the piecewise path gets it as `big_min`)
// A level takes the piecewise path while a balanced node of it holds at least build_big_level_min(M) points.  Up to a
// million points that is half the node threshold (round 3).  Beyond, the path's passes over ALL points (k_big_stats, two
// scans, k_big_emulate: 0.6-1.3 ms per level at 10M, in order from the ninth level on) cost more than the chains they
// replace -- 4.2 ns per point of the longest node: 160 us for 39K points -- so the threshold grows with the cloud
// (M / 200: never more than the eight top levels, which are the speculated ones), and the node threshold with it.
#define BIG_LEVEL_MAX 49152u
static uint32_t build_big_level_min(size_t M)
{
  const bool grow_env = [] { const char* e = lab_env("TDTK_BUILD_BIGGROW"); return !(e && e[0] == '0'); }();
  static const bool small_env = [] { const char* e = lab_env("TDTK_BUILD_BIGMIN"); return !(e && e[0] == '0'); }();
  if (small_env && M <= (size_t)BIG_MIN_SMALL * 16u) return BIG_MIN_SMALL / 2u;
  if (!grow_env) return BIG_MIN / 2u;
  const size_t t = M / 200u;
  return (uint32_t)(t < BIG_MIN / 2u ? BIG_MIN / 2u : (t > BIG_LEVEL_MAX ? BIG_LEVEL_MAX : t));
}
static uint32_t build_big_min(size_t M)
{
  static const bool small_env = [] { const char* e = lab_env("TDTK_BUILD_BIGMIN"); return !(e && e[0] == '0'); }();
  if (small_env && M <= (size_t)BIG_MIN_SMALL * 16u) return BIG_MIN_SMALL;   // (15K points 572 -> 545 us; 40K equal; 81K 710 -> 772: the side streams' chains then outlast the levels)
  const uint32_t t = build_big_level_min(M) / 3u * 2u;
  return t > BIG_MIN ? t : BIG_MIN;
}
#define BIG_ANY 0xFFFFu        // BSum.eb of the neutral element
#define BIG_BAD 0xFFFEu        // BSum.eb of a run whose pieces disagree about the binade: never applicable
struct BPiece {
  uint32_t start, len;         // positions [start, start + len) of the run; len == 0: no piece in this slot
  uint32_t ebits, flags;       // biased exponent of the predicted running sum (k_big_stats parks the node's index here); flags: 1 = it is negative, 2 = cannot be
                               // summarised, 4 = first piece of its node, 8 = walk it (includes 2)
  double lo, hi, csum, pre;    // bounding values, plain sum of the piece, plain sum of everything before it
};
struct BSum {                  // what a piece (or a run of pieces) does to M, by parity of the incoming M
  long long T0, T1, mn0, mn1, mx0, mx1;   // (scalar fields: indexing an array by the parity sends the struct to scratch)
  uint32_t eb, reset;          // exponent bits | sign << 11 it assumes; reset: nothing before it counts
  uint32_t cnt, pad;           // walked pieces in it (a plain count: `reset` does not touch it) -- their rank in the walk list
};
__device__ __forceinline__ void bsum_neutral(BSum& r) { r.T0 = r.T1 = r.mn0 = r.mn1 = r.mx0 = r.mx1 = 0; r.eb = 0xFFFFu; r.reset = 0u; r.cnt = 0u; r.pad = 0u; }
struct BPre { double v, lo, hi; uint32_t reset, pad; };   // plain running sum and running bounds since the node's first piece
struct BPreOp {
  __device__ BPre operator()(const BPre& a, const BPre& b) const
  {
    if (b.reset) return b;
    BPre r; r.v = a.v + b.v; r.lo = (b.lo < a.lo) ? b.lo : a.lo; r.hi = (a.hi < b.hi) ? b.hi : a.hi; r.reset = a.reset; r.pad = 0u;
    return r;
  }
};
struct BSumOp {
  __device__ BSum operator()(const BSum& a, const BSum& b) const
  {
    BSum r;
    if (b.reset) { r = b; r.cnt = a.cnt + b.cnt; return r; }
    if (b.eb == BIG_ANY) { r = a; r.cnt = a.cnt + b.cnt; return r; }
    if (a.eb == BIG_ANY) { r = b; r.reset = a.reset; r.cnt = a.cnt + b.cnt; return r; }
    const bool bad = a.eb != b.eb || a.eb == BIG_BAD;
    {   // incoming M even
      const bool pa = (a.T0 & 1ll) != 0;
      const long long bt = pa ? b.T1 : b.T0, bmn = pa ? b.mn1 : b.mn0, bmx = pa ? b.mx1 : b.mx0;
      r.T0 = a.T0 + bt;
      const long long lo = a.T0 + bmn, hi = a.T0 + bmx;
      r.mn0 = (lo < a.mn0) ? lo : a.mn0;
      r.mx0 = (hi > a.mx0) ? hi : a.mx0;
    }
    {   // incoming M odd
      const bool pa = ((1ll + a.T1) & 1ll) != 0;
      const long long bt = pa ? b.T1 : b.T0, bmn = pa ? b.mn1 : b.mn0, bmx = pa ? b.mx1 : b.mx0;
      r.T1 = a.T1 + bt;
      const long long lo = a.T1 + bmn, hi = a.T1 + bmx;
      r.mn1 = (lo < a.mn1) ? lo : a.mn1;
      r.mx1 = (hi > a.mx1) ? hi : a.mx1;
    }
    r.eb = bad ? BIG_BAD : a.eb;
    r.reset = a.reset;
    r.cnt = a.cnt + b.cnt; r.pad = 0u;
    return r;
  }
};
// slot of a piece: block q of positions holds at most the tail of one big node (it starts at the block's first
// position: slot 2q) and the head of the next (it starts inside the block, behind that node's first point: slot 2q+1).
// The pieces of node (start a, n points) in run order: the first one is a head piece unless a + 1 is block-aligned.
__device__ __forceinline__ uint32_t big_piece_slot(uint32_t a, uint32_t i)
{
  const uint32_t q0 = (a + 1u) / BIG_CH;
  const bool head = ((a + 1u) % BIG_CH) != 0u;
  return (i == 0 && head) ? 2u * q0 + 1u : 2u * (q0 + i);
}
__device__ __forceinline__ uint32_t big_piece_count(uint32_t a, uint32_t n) { return (a + n - 1u) / BIG_CH - (a + 1u) / BIG_CH + 1u; }

// one wave per block of 64 positions: the pieces of that block, their bounding values and plain sums along all three
// axes (which node a piece belongs to is the same for the three: one look-up, three reductions; a wave per (block, axis)
// -- round 2 -- was 47 000 waves of four dependent round trips each at 1M points, 28 us per level, now 15 600)
__global__ void __launch_bounds__(256) k_big_stats(const BSeg* __restrict__ segs, const uint32_t* __restrict__ seg_of,
                                                   const double* __restrict__ cx, const double* __restrict__ cy,
                                                   const double* __restrict__ cz, uint32_t M, uint32_t nblocks,
                                                   BPiece* __restrict__ pieces, BPre* __restrict__ prein, const uint32_t big_min,
                                                   const uint32_t* __restrict__ only_axis = nullptr)
{
  // only_axis (a speculated level: the axis every big node is cut along is known -- k_big_approx -- before its exact sum is
  // asked for): a piece's bounds and plain sum are reduced along that axis alone.  Eighteen wave reductions per block of
  // 64 points -- three axes x (min, max, sum) x two pieces -- are what this kernel's time is made of; the exact chain
  // needs two of them.  (Start, length and node of a piece are written on every axis as before: k_big_emulate and
  // k_big_list read them from axis 0.)
  const uint32_t q = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  if (q >= nblocks) return;
  const uint32_t base = q * BIG_CH;
  const uint32_t bend = (base + BIG_CH < M) ? base + BIG_CH : M;
  const uint32_t p = base + lane;
  // this lane's values first: they do not depend on the look-ups below
  const bool inb = p < bend;
  const double vx = inb ? cx[p] : 0.0, vy = inb ? cy[p] : 0.0, vz = inb ? cz[p] : 0.0;
  // the node at the block's first position: a tail piece if it is big and started earlier
  uint32_t t_start = 0, t_end = 0, t_first = 0xFFFFFFFFu, t_sid = 0;
  {
    const uint32_t sid = seg_of[base];
    if (sid != 0xFFFFFFFFu) {
      const BSeg sg = segs[sid];
      if (sg.n >= big_min && sg.start < base) {
        t_start = base; t_end = (sg.start + sg.n < bend) ? sg.start + sg.n : bend; t_sid = sid;
        if (sg.start + 1u == base) t_first = sg.start;
      }
    }
  }
  // a big node whose first point lies in this block: a head piece behind that point
  uint32_t h_start = 0, h_end = 0, h_first = 0, h_sid = 0;
  {
    uint32_t found = 0xFFFFFFFFu, fend = 0, fsid = 0;
    if (inb) {
      const uint32_t sid = seg_of[p];
      if (sid != 0xFFFFFFFFu && (p == 0 || seg_of[p - 1] != sid)) {
        const BSeg sg = segs[sid];
        if (sg.start == p && sg.n >= big_min) { found = p; fend = sg.start + sg.n; fsid = sid; }
      }
    }
    const unsigned long long any = __ballot(found != 0xFFFFFFFFu);
    if (any) {
      const int src = __ffsll((long long)any) - 1;
      const uint32_t fp = __shfl(found, src, WAVE), e = __shfl(fend, src, WAVE);
      h_sid = __shfl(fsid, src, WAVE);
      h_first = fp; h_start = fp + 1; h_end = (e < bend) ? e : bend;
      if (h_start >= h_end) { h_start = h_end = 0; }
    }
  }
  const bool in_t = inb && p >= t_start && p < t_end, in_h = inb && p >= h_start && p < h_end;
  const bool has_t = t_end > t_start, has_h = h_end > h_start;
  const double INF = 1.0 / 0.0;
  const uint32_t t_only = (only_axis != nullptr && has_t) ? only_axis[t_sid] : 3u, h_only = (only_axis != nullptr && has_h) ? only_axis[h_sid] : 3u;
#pragma unroll
  for (uint32_t ax = 0; ax < 3u; ax++) {
    const double* __restrict__ arr = (ax == 0) ? cx : ((ax == 1) ? cy : cz);
    const double v = (ax == 0) ? vx : ((ax == 1) ? vy : vz);
    double tlo = in_t ? v : INF, thi = in_t ? v : -INF, tsum = in_t ? v : 0.0;
    double hlo = in_h ? v : INF, hhi = in_h ? v : -INF, hsum = in_h ? v : 0.0;
    if (only_axis == nullptr) {
      if (has_t) { tlo = wave_min(tlo); thi = wave_max(thi); tsum = wave_add(tsum); }
      if (has_h) { hlo = wave_min(hlo); hhi = wave_max(hhi); hsum = wave_add(hsum); }
    } else {
      // (the bounds of a speculated level's nodes are not asked for again: the plain path has them)
      tlo = hlo = INF; thi = hhi = -INF;
      if (has_t && t_only == ax) tsum = wave_add(tsum); else tsum = 0.0;
      if (has_h && h_only == ax) hsum = wave_add(hsum); else hsum = 0.0;
    }
    const size_t o = ((size_t)ax * nblocks + q) * 2;
    if (lane == 0) {
      BPiece t; t.start = t_start; t.len = t_end - t_start; t.ebits = t_sid; t.flags = 0; t.lo = tlo; t.hi = thi; t.csum = tsum; t.pre = 0.0;
      BPre tp; tp.v = t.len ? tsum : 0.0; tp.lo = tlo; tp.hi = thi; tp.reset = 0u; tp.pad = 0u;
      if (t.len && t_first != 0xFFFFFFFFu) {
        t.flags = 4u; t.pre = arr[t_first]; tp.v += t.pre; tp.reset = 1u;
        tp.lo = (t.pre < tp.lo) ? t.pre : tp.lo; tp.hi = (tp.hi < t.pre) ? t.pre : tp.hi;
      }
      pieces[o] = t; prein[o] = tp;
      BPiece h; h.start = h_start; h.len = h_end - h_start; h.ebits = h_sid; h.flags = 0; h.lo = hlo; h.hi = hhi; h.csum = hsum; h.pre = 0.0;
      BPre hp; hp.v = 0.0; hp.lo = hlo; hp.hi = hhi; hp.reset = 0u; hp.pad = 0u;
      if (h.len) {
        h.flags = 4u; h.pre = arr[h_first]; hp.v = hsum + h.pre; hp.reset = 1u;
        hp.lo = (h.pre < hp.lo) ? h.pre : hp.lo; hp.hi = (hp.hi < h.pre) ? h.pre : hp.hi;
      }
      pieces[o + 1] = h; prein[o + 1] = hp;
    }
  }
}

// one lane per piece: what the piece does to the integer mantissa of the running sum, for both parities of it -- along
// the axis its node will be split on only (k_decide: splitval = mean[axis]; the bounds of all three axes stand at the
// node's last piece once the bounds scan has run), so `own` has one entry per slot, not one per slot and axis
__device__ __forceinline__ uint32_t big_split_axis(const BPre* __restrict__ preout, size_t st, uint32_t sl_last)
{
  const BPre b0 = preout[sl_last], b1 = preout[st + sl_last], b2 = preout[2 * st + sl_last];
  const double hx = 0.5 * (b0.hi - b0.lo), hy = 0.5 * (b1.hi - b1.lo), hz = 0.5 * (b2.hi - b2.lo);
  if (hx > hy) return (hx > hz) ? 0u : 2u;
  return (hy > hz) ? 1u : 2u;
}
__global__ void __launch_bounds__(64) k_big_emulate(const BSeg* __restrict__ segs, const double* __restrict__ cx,
                                                    const double* __restrict__ cy, const double* __restrict__ cz,
                                                    uint32_t nblocks, BPiece* __restrict__ pieces,
                                                    const BPre* __restrict__ preout, BSum* __restrict__ own, int dbg,
                                                    const uint32_t* __restrict__ only_axis = nullptr)
{
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= 2u * nblocks) return;
  const size_t st = (size_t)nblocks * 2;
  BSum r;
  bsum_neutral(r);
  const BPiece g = pieces[id];                      // start, len, first-of-node and the node are the same on every axis
  if (g.len == 0) { own[id] = r; return; }
  const BSeg sg = segs[g.ebits];
  const uint32_t ax = (only_axis != nullptr) ? only_axis[g.ebits] : big_split_axis(preout, st, big_piece_slot(sg.start, big_piece_count(sg.start, sg.n) - 1u));
  const size_t o = (size_t)ax * st + id;
  BPiece pc = pieces[o];
  const double* __restrict__ arr = ((ax == 0) ? cx : ((ax == 1) ? cy : cz)) + pc.start;
  const double pre = (pc.flags & 4u) ? pc.pre : (preout[o].v - pc.csum);
  const unsigned long long pb = (unsigned long long)__double_as_longlong(pre);
  const uint32_t ebits = (uint32_t)((pb >> 52) & 0x7FFu);
  const unsigned long long flip = pb & 0x8000000000000000ull;   // work on |s|: s + x = -(|s| + (-x)), rounding is symmetric
  uint32_t flags = (pc.flags & 4u) | (flip ? 1u : 0u);
  if (ebits == 0u || ebits == 0x7FFu) flags |= 2u;
  long long S0 = 0, S1 = 0, mn0 = 0, mn1 = 0, mx0 = 0, mx1 = 0;
  // eight values requested together (indices past the end read the last point again and are never used): one wait per
  // eight points instead of one per point -- the lanes of a wave read 64 different lines
  for (uint32_t i0 = 0; i0 < pc.len && !(flags & 2u); i0 += 8u) {
   double xv[8];
#pragma unroll
   for (uint32_t u = 0; u < 8u; u++) xv[u] = arr[(i0 + u < pc.len) ? i0 + u : pc.len - 1u];
#pragma unroll
   for (uint32_t u = 0; u < 8u; u++) {
    if (i0 + u >= pc.len || (flags & 2u)) break;
    const unsigned long long yb = (unsigned long long)__double_as_longlong(xv[u]) ^ flip;
    const bool neg = (yb >> 63) != 0;
    uint32_t ey = (uint32_t)((yb >> 52) & 0x7FFu);
    unsigned long long m = yb & 0x000FFFFFFFFFFFFFull;
    if (ey) m |= 0x0010000000000000ull; else ey = 1u;          // subnormal: no hidden bit, exponent of the smallest normal
    const int d = (int)ebits - (int)ey;
    if (d <= 0) { flags |= 2u; break; }                         // |x| >= 2^e: the sum leaves the binade (or x is not finite)
    if (d > 63) continue;                                       // x < u / 2^10: nothing
    const unsigned long long k = m >> d, rem = m & ((1ull << d) - 1ull), half = 1ull << (d - 1);
    const long long ks = neg ? -(long long)k : (long long)k, one = neg ? -1ll : 1ll;
    S0 += ks; S1 += ks;
    if (rem > half) { S0 += one; S1 += one; }
    else if (rem == half) {                                     // tie: to the even mantissa
      if (S0 & 1ll) S0 += one;                                  // incoming M even
      if (!(S1 & 1ll)) S1 += one;                               // incoming M odd
    }
    mn0 = (S0 < mn0) ? S0 : mn0; mx0 = (S0 > mx0) ? S0 : mx0;
    mn1 = (S1 < mn1) ? S1 : mn1; mx1 = (S1 > mx1) ? S1 : mx1;
   }
  }
  // does it fit with room to spare for the true M (the predicted one is good to a few ulps of the largest partial sum;
  // 2^32 mantissa units are nine orders more than that)?  If not, the chain walks this piece.
  if (!(flags & 2u)) {
    const long long ONE = 0x0010000000000000ll, room = 1ll << 32;
    const long long Mi = (long long)((pb & 0x000FFFFFFFFFFFFFull) | 0x0010000000000000ull);
    const long long mn = (mn0 < mn1) ? mn0 : mn1, mx = (mx0 > mx1) ? mx0 : mx1;
    if (!(Mi + mn > ONE + room && Mi + mx < 2 * ONE - room)) flags |= 8u;
  } else flags |= 8u;
  if ((dbg & 3) == 2) flags |= 8u;
  pc.ebits = ebits; pc.flags = flags;
  pieces[o] = pc;
  if (flags & 8u) { r.reset = 1u; r.cnt = 1u; own[id] = r; return; }   // a walked piece: nothing before it composes past it
  r.T0 = S0; r.T1 = S1; r.mn0 = mn0; r.mn1 = mn1; r.mx0 = mx0; r.mx1 = mx1;
  r.eb = ebits | ((flags & 1u) << 11); r.reset = (flags & 4u) ? 1u : 0u;
  own[id] = r;
}

// apply a summary to the running sum if it provably describes what the adds would do; false: it does not
__device__ __forceinline__ bool big_apply(double& sum, long long T0, long long T1, long long mn0, long long mn1, long long mx0,
                                          long long mx1, uint32_t eb)
{
  if (eb == BIG_ANY) return true;
  const unsigned long long ONE = 0x0010000000000000ull;
  // the running sum is the same in every lane of the chain's wave: say so, and the arithmetic below is scalar
  const long long sv = __double_as_longlong(sum);
  const unsigned long long sb = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)sv >> 32)) << 32) |
                                (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)sv & 0xFFFFFFFFull));
  if ((uint32_t)(sb >> 52) != eb) return false;                // exponent and sign in one compare (eb = ebits | sign << 11)
  const long long Mi = (long long)((sb & (ONE - 1ull)) | ONE);
  const bool odd = (Mi & 1ll) != 0;
  const long long T = odd ? T1 : T0, mn = odd ? mn1 : mn0, mx = odd ? mx1 : mx0;
  if (!(Mi + mn > (long long)ONE && Mi + mx < (long long)(ONE << 1))) return false;
  sum = __longlong_as_double((long long)((sb & 0xFFF0000000000000ull) | ((unsigned long long)(Mi + T) & (ONE - 1ull))));
  return true;
}
__device__ __forceinline__ long long readlane64(long long v, int lane)
{
  const int lo = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v & 0xFFFFFFFFull), lane);
  const int hi = __builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)v >> 32), lane);
  return (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo);
}
__device__ __forceinline__ bool big_apply(double& sum, const BSum S) { return big_apply(sum, S.T0, S.T1, S.mn0, S.mn1, S.mx0, S.mx1, S.eb); }

// one wave per (node, axis): the chain.  Clusters of walked pieces are visited one by one; everything between two of
// them is one step.
#define BIG_CL 4               // walked pieces folded into one LDS stage (they are consecutive in memory)
#define BIG_NQ 3               // such stages requested ahead of the one being walked
__global__ void __launch_bounds__(256) k_big_stitch(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                    const double* __restrict__ cx, const double* __restrict__ cy,
                                                    const double* __restrict__ cz, uint32_t nblocks,
                                                    const BPiece* __restrict__ pieces, const BPre* __restrict__ preout,
                                                    const BSum* __restrict__ own, const BSum* __restrict__ comp,
                                                    const uint32_t* __restrict__ list, BMeas* __restrict__ out, int dbg, const uint32_t big_min,
                                                    const uint32_t* __restrict__ only_axis = nullptr)
{
  __shared__ alignas(16) double walk[256 / WAVE][BIG_CH * BIG_CL + 16];   // + 16: the chain's last request reads past the data
  const uint32_t w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) / WAVE);
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  if (w >= 3u * lv->nseg) return;
  const uint32_t sgi = w / 3u, ax = w % 3u;
  const uint32_t a = __builtin_amdgcn_readfirstlane(segs[sgi].start), n = __builtin_amdgcn_readfirstlane(segs[sgi].n);
  if (n < big_min) return;
  const double* __restrict__ arr = (ax == 0) ? cx : ((ax == 1) ? cy : cz);
  const size_t ao = (size_t)ax * nblocks * 2;
  const BPiece* pa = pieces + ao;
  const BSum* oa = own;                            // per slot: the summaries exist along the split axis only
  const BSum* ca = comp;
  double* wbuf = walk[threadIdx.x / WAVE];
  const uint32_t np = big_piece_count(a, n);
  {
    // Only the sum along the axis the node will be split on is ever used (k_decide: splitval = mean[axis]); the bounds
    // of all three axes are already known -- the bounds scan ends at the node's last piece -- so the waves of the two
    // other axes report their bounds and leave (bit-identical tree: k_decide picks the axis from the same bounds).
    const uint32_t sl = big_piece_slot(a, np - 1u);
    const size_t st = (size_t)nblocks * 2;
    const uint32_t split = (only_axis != nullptr) ? only_axis[sgi] : big_split_axis(preout, st, sl);
    if (split != ax) {
      if (lane == 0) {
        const BPre b = preout[(size_t)ax * st + sl];
        out[sgi].lo[ax] = b.lo; out[sgi].hi[ax] = b.hi; out[sgi].mean[ax] = 0.0;
      }
      return;
    }
  }
  const double first = arr[a];
  double sum = first, lo = first, hi = first;
  int done = -1;                                   // pieces 0 .. done are in `sum`

  // the adds of `cnt` values parked in LDS, in order
  auto chain = [&](uint32_t cnt) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if ((cnt & 31u) == 0u && cnt) {
      lds_chain32(sum, wbuf, cnt);
    } else {
      uint32_t k = 0;
      for (; k + 16 <= cnt; k += 16) {
        double r[16];
#pragma unroll
        for (int t = 0; t < 16; t++) r[t] = wbuf[k + t];
#pragma unroll
        for (int t = 0; t < 16; t++) sum += r[t];
      }
      for (; k < cnt; k++) sum += wbuf[k];
    }
    __builtin_amdgcn_wave_barrier();
  };
  // pieces lo_i .. hi_i (none of them marked for walking), folded into S; redone piece by piece if S does not fit
  // the real sum after all
  auto apply_run = [&](int lo_i, int hi_i, const BSum S) {
    if ((dbg & 3) != 1 && big_apply(sum, S)) return;
    for (int i = lo_i; i <= hi_i; i++) {
      const uint32_t slot = big_piece_slot(a, (uint32_t)i);
      const BSum O = oa[slot];
      if (big_apply(sum, O)) continue;
      const uint32_t st = pa[slot].start, ln = pa[slot].len;
      wbuf[lane] = (lane < ln) ? arr[st + lane] : 0.0;
      chain(ln);
    }
  };

  // The walked pieces of all nodes stand in one list in run order (k_big_list: their rank is the count the run scan
  // carries along), so the wave goes from one walked piece of its node to the next -- 64 list entries per trip, the
  // records of the next 64 requested a trip ahead -- instead of reading the records of every piece of the node to find
  // them: on zero-mean coordinates (the sum wanders through zero: the root and its first descendants of a centred
  // cloud) 3-8 % of the pieces are walked, and a lone wave pays ~2 us for every dependent round trip to memory.
  const uint32_t sl_first = big_piece_slot(a, 0u), sl_last = big_piece_slot(a, np - 1u);
  const uint32_t pq0 = (a + 1u) / BIG_CH;
  const BSum c_last = ca[sl_last], o_last = oa[sl_last];
  const uint32_t r0 = __builtin_amdgcn_readfirstlane(ca[sl_first].cnt);
  const uint32_t r1 = __builtin_amdgcn_readfirstlane(c_last.cnt + o_last.cnt);
  {
    const BPre bb = preout[ao + sl_last];          // bounds since the node's first point (as for the two other axes)
    lo = bb.lo; hi = bb.hi;
  }
  if ((dbg & 16) && lane == 0) printf("stitch: node at %u, %u points, %u pieces, %u walked\n", a, n, np, r1 - r0);
  auto load_slots = [&](uint32_t r) -> uint32_t { return (r + lane < r1) ? list[r + lane] : 0xFFFFFFFFu; };
  uint32_t slotN = load_slots(r0), slotNN = load_slots(r0 + WAVE);
  uint32_t stN = 0, lnN = 0; BSum RN;
  auto load_recs = [&](uint32_t slot) {
    stN = 0; lnN = 0; bsum_neutral(RN);
    if (slot != 0xFFFFFFFFu) { stN = pa[slot].start; lnN = pa[slot].len; RN = ca[slot]; }
  };
  load_recs(slotN);
  for (uint32_t rr = r0; rr < r1; rr += WAVE) {
    const uint32_t slot = slotN, p_start = stN, p_len = lnN;
    const BSum R = RN;
    slotN = slotNN; slotNN = load_slots(rr + 2u * WAVE);
    load_recs(slotN);
    const bool valid = slot != 0xFFFFFFFFu;
    const int p_idx = (slot & 1u) ? 0 : (int)(slot / 2u - pq0);          // piece index in the node's run
    unsigned long long fm = __ballot(valid);
    const unsigned long long fullm = __ballot(valid && p_len == BIG_CH);
    // entry j + 1 continues entry j in memory: the next block of the same node
    const uint32_t nslot = __shfl_down(slot, 1, WAVE);
    const unsigned long long adjm = __ballot(valid && lane < WAVE - 1 && !(slot & 1u) && nslot == slot + 2u);
    // clusters: up to BIG_CL walked pieces that follow each other in memory, all of them full (a partial piece -- the
    // first or the last of the node -- goes alone); the points of the next clusters are requested before the current
    // one is walked
    struct Cl { int j, c; uint32_t cnt; double v[BIG_CL]; };
    // always BIG_CL loads, in straight-line code (a load that is not needed reads the node's first point): the compiler
    // can then count the loads in flight and wait for the oldest cluster only -- behind a branch it waits for all of them
    auto fetch_cluster = [&]() -> Cl {
      Cl q;
      const bool any = fm != 0ull;
      q.j = any ? __ffsll((long long)fm) - 1 : 0;
      const unsigned long long fr = fullm >> q.j, ad = adjm >> q.j;
      const unsigned long long gap = ~(fr & ((ad << 1) | 1ull));       // bit k: entry j + k is full and follows j + k - 1
      int c = gap ? __ffsll((long long)gap) - 1 : 64;
      if (c > BIG_CL) c = BIG_CL;
      const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)p_start, q.j), ln = (uint32_t)__builtin_amdgcn_readlane((int)p_len, q.j);
      q.cnt = (uint32_t)c * BIG_CH;
      if (c == 0) { c = 1; q.cnt = ln; }
      if (!any) { c = 0; q.cnt = 0; }
      q.c = c;
      fm &= ~(((c >= 64) ? ~0ull : ((1ull << c) - 1ull)) << q.j);
#pragma unroll
      for (int k = 0; k < BIG_CL; k++) {
        const uint32_t off = (uint32_t)k * BIG_CH + lane;
        q.v[k] = arr[(off < q.cnt) ? st + off : a];
      }
      return q;
    };
    // BIG_NQ clusters in flight, consumed and refilled in turn (the loop is unrolled so that every one of them lives in
    // registers of its own and the compiler can count the loads in flight): a lone wave needs ~2 us for a round trip
    // to memory and 0.17 us to walk a piece, and on Morton-ordered scans the walked pieces of the top levels stand
    // alone -- with three in flight the wave spent 0.45 us per walked piece, most of it waiting
    Cl q[BIG_NQ];
#pragma unroll
    for (int t = 0; t < BIG_NQ; t++) q[t] = fetch_cluster();
    bool more = true;
    while (more) {
#pragma unroll
      for (int t = 0; t < BIG_NQ; t++) {
        if (more && q[t].c) {
          const int j = q[t].j, c = q[t].c;
          // the run in front of this walked piece: fetched by lane j
          // (j is wave-uniform: v_readlane straight into scalar registers instead of a trip through the LDS crossbar)
          BSum S;
          S.T0 = readlane64(R.T0, j); S.T1 = readlane64(R.T1, j);
          S.mn0 = readlane64(R.mn0, j); S.mn1 = readlane64(R.mn1, j);
          S.mx0 = readlane64(R.mx0, j); S.mx1 = readlane64(R.mx1, j);
          S.eb = (uint32_t)__builtin_amdgcn_readlane((int)R.eb, j); S.reset = 0u; S.cnt = 0u; S.pad = 0u;
          const int idx = __builtin_amdgcn_readlane(p_idx, j);
          if (idx - 1 > done) apply_run(done + 1, idx - 1, S);
#pragma unroll
          for (int k = 0; k < BIG_CL; k++)
            if (k < c) wbuf[k * BIG_CH + lane] = q[t].v[k];
          chain(q[t].cnt);
          done = idx + c - 1;
          q[t] = fetch_cluster();
        } else {
          more = false;
        }
      }
    }
  }
  if ((int)np - 1 > done) apply_run(done + 1, (int)np - 1, BSumOp()(c_last, o_last));
  if (lane == 0) {
    out[sgi].lo[ax] = lo;
    out[sgi].hi[ax] = hi;
    out[sgi].mean[ax] = sum / (double)n;
  }
}

// ---- speculative splits (round 3) ------------------------------------------------------------------------------
// The exact centroid sum of a big node is a serial affair (k_big_stitch: 1.2 ms for the root of a 1M-point cloud, 0.6
// and 0.4 ms on the two levels below it, one to four waves busy), and everything else waits for it -- but all the build
// needs from it right away is WHERE the node is cut, and for that the plain parallel sum of the same points (the
// bounds scan carries it along: BPre.v at the node's last piece) is almost always enough: it differs from the serial
// sum in the last few bits, and the cut only changes if a point's coordinate lies between the two values.  So the
// levels are built from the plain sums, while the chain computes the exact sums of all big nodes in the background, on
// a second stream, from a snapshot of the coordinates taken before the level's partition pass moves them.  At the end
// every big node is checked -- the number of its points below the EXACT split value must be the number the build put to
// the left -- and its record gets the exact value.  A node that fails the check (none has, outside the test that forces
// one) sends the whole build through the in-order path again.  The tree is the reference's, bit for bit, either way.
struct BSpecLevel {
  BPiece* pieces; BPre* preout; BSum* own; BSum* comp; uint32_t* wlist; double* snap;   // snap: x | y | z, n1 doubles each
  BSeg* segs; uint32_t* axis; uint32_t* node; uint32_t* nleft; uint32_t* cnt; BMeas* exact;
  BPre* prein; uint32_t* seg_of;                                                          // the level's labels, as snapped
  uint32_t level, pad;
};
// bounds and plain sums of the part of a big node inside one block of BIG_PB positions (two per block: the node the block
// starts in, and a node that starts inside it -- big nodes are longer than a block, so there is no third)
struct BPart { double lo[3], hi[3], sum[3]; };
#define BIG_PB 512u
#define BIG_SPEC_MAX 24
#define SPEC_SLOTS 14          // buffers per speculated level (spec_layout)
struct BSpecAll { BSpecLevel L[BIG_SPEC_MAX]; int n; };

// (a kernel's by-value argument block read through the kernarg segment pointer: see kernarg_block in kernels.hip)
template <class T>
__device__ __forceinline__ const T& build_kernarg_block()
{
  typedef const T __attribute__((address_space(4))) * kernarg_ptr;
  kernarg_ptr p = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const T*)p;
}

// What the cut needs, and nothing else (round 3, second step): bounds and plain sums of the big nodes in two short
// passes -- per block of 512 positions, then per node over its blocks -- instead of the piece statistics and their
// prefix scan, which only the exact chain needs and which now run with it on the second stream.
// (snap != nullptr: the block also writes the level's snapshot of its 512 positions -- what k_spec_snapshot does in a
// launch of its own)
__device__ __forceinline__ void big_partials_body(const uint32_t vblock, const BSeg* __restrict__ segs, const uint32_t* __restrict__ seg_of,
                                                  const double* __restrict__ cx, const double* __restrict__ cy,
                                                  const double* __restrict__ cz, uint32_t M, BPart* __restrict__ part,
                                                  double* __restrict__ snap, uint32_t* __restrict__ seg_snap, uint32_t n1, const uint32_t big_min)
{
  constexpr int R = BIG_PB / 256;
  __shared__ uint32_t s_startB;
  __shared__ double s_red[256 / WAVE][2][9];
  const uint32_t p0 = vblock * BIG_PB;
  const uint32_t pend = (p0 + BIG_PB < M) ? p0 + BIG_PB : M;
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  if (threadIdx.x == 0) s_startB = 0xFFFFFFFFu;
  // everything this thread will need is requested before the first dependent look-up
  uint32_t sids[R], prev[R];
  double xs[R], ys[R], zs[R];
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t p = p0 + (uint32_t)r * 256u + threadIdx.x;
    const bool in = p < pend;
    sids[r] = in ? seg_of[p] : 0xFFFFFFFFu;
    prev[r] = (in && p > p0) ? seg_of[p - 1] : 0xFFFFFFFFu;
    xs[r] = in ? cx[p] : 0.0; ys[r] = in ? cy[p] : 0.0; zs[r] = in ? cz[p] : 0.0;
  }
  const uint32_t sid0 = seg_of[p0];
  if (snap) {
#pragma unroll
    for (int r = 0; r < R; r++) {
      const uint32_t p = p0 + (uint32_t)r * 256u + threadIdx.x;
      if (p < pend) { snap[p] = xs[r]; snap[(size_t)n1 + p] = ys[r]; snap[2 * (size_t)n1 + p] = zs[r]; seg_snap[p] = sids[r]; }
    }
  }
  __syncthreads();
  // piece A: the big node the block starts in
  uint32_t a_end = p0;
  if (sid0 != 0xFFFFFFFFu) {
    const BSeg sg = segs[sid0];
    if (sg.n >= big_min) a_end = (sg.start + sg.n < pend) ? sg.start + sg.n : pend;
  }
  // piece B: the first big node that starts inside the block
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t p = p0 + (uint32_t)r * 256u + threadIdx.x;
    if (p < pend && p > p0 && sids[r] != 0xFFFFFFFFu && prev[r] != sids[r]) {
      const BSeg sg = segs[sids[r]];
      if (sg.n >= big_min) atomicMin(&s_startB, p);
    }
  }
  __syncthreads();
  const uint32_t b_start = s_startB;
  const bool has[2] = {a_end > p0, b_start != 0xFFFFFFFFu};
  const double INF = 1.0 / 0.0;
  double v[2][9];
#pragma unroll
  for (int k = 0; k < 2; k++) { for (int c = 0; c < 3; c++) { v[k][c] = INF; v[k][3 + c] = -INF; v[k][6 + c] = 0.0; } }
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t p = p0 + (uint32_t)r * 256u + threadIdx.x;
    if (p >= pend) continue;
    const double x = xs[r], y = ys[r], z = zs[r];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const bool mine = (k == 0) ? (p < a_end) : (p >= b_start);
      if (!mine) continue;
      v[k][0] = (x < v[k][0]) ? x : v[k][0]; v[k][1] = (y < v[k][1]) ? y : v[k][1]; v[k][2] = (z < v[k][2]) ? z : v[k][2];
      v[k][3] = (v[k][3] < x) ? x : v[k][3]; v[k][4] = (v[k][4] < y) ? y : v[k][4]; v[k][5] = (v[k][5] < z) ? z : v[k][5];
      v[k][6] += x; v[k][7] += y; v[k][8] += z;
    }
  }
#pragma unroll
  for (int k = 0; k < 2; k++) {
    if (has[k]) {                     // (uniform over the workgroup)
      for (int c = 0; c < 3; c++) { v[k][c] = wave_min(v[k][c]); v[k][3 + c] = wave_max(v[k][3 + c]); v[k][6 + c] = wave_add(v[k][6 + c]); }
    }
    if (lane == 0) for (int c = 0; c < 9; c++) s_red[wv][k][c] = v[k][c];
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int k = (int)threadIdx.x;
    BPart o;
    for (int c = 0; c < 3; c++) {
      double lo = s_red[0][k][c], hi = s_red[0][k][3 + c], sm = s_red[0][k][6 + c];
      for (int w = 1; w < 256 / WAVE; w++) {
        lo = (s_red[w][k][c] < lo) ? s_red[w][k][c] : lo;
        hi = (hi < s_red[w][k][3 + c]) ? s_red[w][k][3 + c] : hi;
        sm += s_red[w][k][6 + c];
      }
      o.lo[c] = lo; o.hi[c] = hi; o.sum[c] = sm;
    }
    part[(size_t)vblock * 2 + k] = o;
  }
}
__global__ void __launch_bounds__(256) k_big_partials(const BSeg* __restrict__ segs, const uint32_t* __restrict__ seg_of,
                                                      const double* __restrict__ cx, const double* __restrict__ cy,
                                                      const double* __restrict__ cz, uint32_t M, BPart* __restrict__ part, uint32_t big_min)
{
  big_partials_body(blockIdx.x, segs, seg_of, cx, cy, cz, M, part, nullptr, nullptr, 0u, big_min);
}
// The front of a speculated level in ONE launch (a small scan's build is a chain of dependent launches, ~7 us apiece): the
// first nb_part workgroups are k_big_partials' and write the snapshot on the way, the rest are k_measure's (the chains of the
// level's nodes below the piecewise path) -- independent passes over the same read-only level.
__global__ void __launch_bounds__(256) k_level_front(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                     const uint32_t* __restrict__ seg_of, const double* __restrict__ cx,
                                                     const double* __restrict__ cy, const double* __restrict__ cz, uint32_t M,
                                                     BPart* __restrict__ part, double* __restrict__ snap, uint32_t* __restrict__ seg_snap,
                                                     uint32_t n1, uint32_t nb_part, BMeas* __restrict__ meas, uint32_t big_min)
{
  if (blockIdx.x < nb_part) big_partials_body(blockIdx.x, segs, seg_of, cx, cy, cz, M, part, snap, seg_snap, n1, big_min);
  else measure_body(blockIdx.x - nb_part, segs, lv, cx, cy, cz, meas, big_min);
}
// one wave per node: a big node's bounds and plain sums from its blocks' partials; keeps what the background chain and
// the final check need of this level (its node list, which axis each node is cut along)
__device__ __forceinline__ void big_approx_wave(const uint32_t i, const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                const BPart* __restrict__ part, BMeas* __restrict__ meas, const BSpecLevel& L, int fault,
                                                const uint32_t big_min)
{
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  if (i >= lv->nseg) return;
  const BSeg sg = segs[i];
  if (lane == 0) { L.segs[i] = sg; L.node[i] = 0xFFFFFFFFu; L.axis[i] = 3u; L.nleft[i] = 0u; L.cnt[i] = 0u; }
  if (sg.n < big_min) return;
  const uint32_t b0 = sg.start / BIG_PB, b1 = (sg.start + sg.n - 1u) / BIG_PB;
  const double INF = 1.0 / 0.0;
  double lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF}, sm[3] = {0.0, 0.0, 0.0};
  for (uint32_t b = b0 + lane; b <= b1; b += WAVE) {
    const BPart q = part[(size_t)b * 2 + ((b == b0 && sg.start > b0 * BIG_PB) ? 1u : 0u)];
    for (int c = 0; c < 3; c++) { lo[c] = (q.lo[c] < lo[c]) ? q.lo[c] : lo[c]; hi[c] = (hi[c] < q.hi[c]) ? q.hi[c] : hi[c]; sm[c] += q.sum[c]; }
  }
  for (int c = 0; c < 3; c++) { lo[c] = wave_min(lo[c]); hi[c] = wave_max(hi[c]); sm[c] = wave_add(sm[c]); }
  if (lane != 0) return;
  BMeas m;
  for (int c = 0; c < 3; c++) { m.lo[c] = lo[c]; m.hi[c] = hi[c]; m.mean[c] = 0.0; }
  const double hx = 0.5 * (hi[0] - lo[0]), hy = 0.5 * (hi[1] - lo[1]), hz = 0.5 * (hi[2] - lo[2]);
  const uint32_t split = (hx > hy) ? ((hx > hz) ? 0u : 2u) : ((hy > hz) ? 1u : 2u);      // big_split_axis / decide_node
  double v = sm[split] / (double)sg.n;
  if (fault) v += 0.125 * (hi[split] - v);                 // test only: a split value that cuts elsewhere (never outside the node)
  m.mean[split] = v;
  meas[i] = m;
  L.axis[i] = split;
}
__global__ void __launch_bounds__(256) k_big_approx(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                    const BPart* __restrict__ part, BMeas* __restrict__ meas, BSpecLevel L, int fault, uint32_t big_min)
{
  big_approx_wave((blockIdx.x * blockDim.x + threadIdx.x) / WAVE, segs, lv, part, meas, L, fault, big_min);
}
// The exact sums of a speculated level's big nodes as plain chains on the snapshot (round 6), for the levels whose chains
// are short enough to end before the build does: one wave per (node, split axis), nothing but its dependent adds -- where
// the piecewise path is five passes over all points per level (k_big_stats, two scans, k_big_emulate, k_big_stitch:
// 0.45 ms of a 10M-point level's bandwidth, in the background but not for free; seven dependent launches that queue up
// on the side streams of a 1M-point build).
__global__ void __launch_bounds__(256) k_chain_exact(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv,
                                                     const double* __restrict__ sx, const double* __restrict__ sy,
                                                     const double* __restrict__ sz, BMeas* __restrict__ exact,
                                                     const uint32_t* __restrict__ axis, uint32_t big_min)
{
  measure_body(blockIdx.x, segs, lv, sx, sy, sz, exact, 0xFFFFFFFFu, big_min, axis);
}
// the level as it stands before its partition pass: coordinates and labels (the exact chain reads these, later)
__global__ void k_spec_snapshot(const uint32_t* __restrict__ seg_of, const double* __restrict__ cx, const double* __restrict__ cy,
                                const double* __restrict__ cz, uint32_t M, uint32_t n1, double* __restrict__ snap,
                                uint32_t* __restrict__ seg_snap)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  snap[p] = cx[p]; snap[(size_t)n1 + p] = cy[p]; snap[2 * (size_t)n1 + p] = cz[p];
  seg_snap[p] = seg_of[p];
}
// after the level's count: where each internal big node's record is and how many points went left
__device__ __forceinline__ void spec_keep_at(const BLevel* __restrict__ lv, const uint32_t* __restrict__ kind, const uint32_t* __restrict__ irank,
                                             const uint32_t* __restrict__ nleft, const BSpecLevel& L, const uint32_t i)
{
  if (i >= lv->nseg || L.axis[i] >= 3u || !kind[i]) return;
  L.node[i] = lv->node_base + irank[i];
  L.nleft[i] = nleft[i];
}
// the check, part 1: per speculated level (blockIdx.y), how many points of each big internal node lie below its EXACT
// split value (the points are where the finished build left them; a node's points are still the run [start, start + n)).
// A workgroup covers 1024 consecutive positions: one search for the node of its first position, then every thread steps
// forward from there (the nodes of a level ascend; a big level's nodes are thousands of positions long).
__global__ void __launch_bounds__(256) k_spec_count(BSpecAll S_by_value, const BLevel* __restrict__ lvl, const double* __restrict__ cx,
                                                    const double* __restrict__ cy, const double* __restrict__ cz, uint32_t M)
{
  // (round 6: every speculated level in one pass -- the workgroup's 1024 positions are read once, x, y and z, and looked
  //  at level by level; a pass per level read the coordinates eight times over: 413 us of a 10M-point build)
  (void)S_by_value;
  const BSpecAll& S = build_kernarg_block<BSpecAll>();
  __shared__ uint32_t s_first, s_cnt;
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t p0 = blockIdx.x * 1024u;
  if (p0 >= M) return;
  double X[4], Y[4], Z[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const uint32_t p = p0 + (uint32_t)r * 256u + threadIdx.x, g = (p < M) ? p : M - 1u;
    X[r] = cx[g]; Y[r] = cy[g]; Z[r] = cz[g];
  }
  for (int ly = 0; ly < S.n; ly++) {
    const BSpecLevel& L = S.L[ly];
    const uint32_t nseg = lvl[L.level].nseg;
    if (nseg == 0) continue;
    if (threadIdx.x == 0) {
      uint32_t lo = 0, hi = nseg;                       // last node with start <= p0
      while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (L.segs[mid].start <= p0) lo = mid; else hi = mid; }
      s_first = lo; s_cnt = 0u;
    }
    __syncthreads();
    const uint32_t first = s_first;      // its count is gathered in LDS: at the top levels every workgroup adds to the same node
    uint32_t at = first;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const uint32_t p = p0 + (uint32_t)r * 256u + threadIdx.x;
      uint32_t i = 0xFFFFFFFFu;
      bool lt = false;
      if (p < M) {
        while (at + 1u < nseg && L.segs[at + 1u].start <= p) ++at;
        const BSeg sg = L.segs[at];
        if (p >= sg.start && p - sg.start < sg.n && L.node[at] != 0xFFFFFFFFu) {
          i = at;
          const uint32_t ax = L.axis[at];
          const double v = (ax == 0) ? X[r] : ((ax == 1) ? Y[r] : Z[r]);
          lt = v < L.exact[at].mean[ax];
        }
      }
      unsigned long long todo = __ballot(i != 0xFFFFFFFFu);
      while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t cur = __shfl(i, leader, WAVE);
        const bool mine = (i == cur);
        const uint32_t add = (uint32_t)__popcll(__ballot(mine && lt));
        if (lane == (uint32_t)leader && add) { if (cur == first) atomicAdd(&s_cnt, add); else atomicAdd(&L.cnt[cur], add); }
        todo &= ~__ballot(mine);
      }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd(&L.cnt[first], s_cnt);
    __syncthreads();
  }
}
// part 2: the counts must be the build's; the records get the exact split values
__global__ void k_spec_patch(BSpecAll S, const BLevel* __restrict__ lvl, KdNode* __restrict__ nodes, uint32_t* __restrict__ err)
{
  for (int l = 0; l < S.n; l++) {
    const BSpecLevel& L = S.L[l];
    const uint32_t nseg = lvl[L.level].nseg;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nseg; i += gridDim.x * blockDim.x) {
      const uint32_t nd = L.node[i];
      if (nd == 0xFFFFFFFFu) continue;
      if (L.cnt[i] != L.nleft[i]) atomicExch(err, 1u);
      nodes[nd].splitval = L.exact[i].mean[L.axis[i]];
    }
  }
}

// the walked pieces of every big node, in run order: entry `rank` of the list is the slot of the piece (the rank is the
// number of walked pieces in the slots before it, carried along by the run scan)
__global__ void k_big_list(const BSum* __restrict__ own, const BSum* __restrict__ comp, uint32_t nsl, uint32_t* __restrict__ list)
{
  const uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= nsl) return;
  if (own[o].cnt) list[comp[o].cnt] = o;
}

__global__ void k_big_dbg_compare(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv, const BMeas* __restrict__ a,
                                  const BMeas* __restrict__ b, uint32_t level, uint32_t big_min)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= lv->nseg || segs[i].n < big_min) return;
  const double hx = 0.5 * (b[i].hi[0] - b[i].lo[0]), hy = 0.5 * (b[i].hi[1] - b[i].lo[1]), hz = 0.5 * (b[i].hi[2] - b[i].lo[2]);
  const int split = (hx > hy) ? ((hx > hz) ? 0 : 2) : ((hy > hz) ? 1 : 2);
  for (int ax = 0; ax < 3; ax++)
    if ((ax == split && a[i].mean[ax] != b[i].mean[ax]) || a[i].lo[ax] != b[i].lo[ax] || a[i].hi[ax] != b[i].hi[ax])
      printf("MISMATCH level %u seg %u (start %u n %u) ax %d: mean %.17g vs chain %.17g  lo %g/%g hi %g/%g\n", level, i, segs[i].start,
             segs[i].n, ax, a[i].mean[ax], b[i].mean[ax], a[i].lo[ax], b[i].lo[ax], a[i].hi[ax], b[i].hi[ax]);
}

// ---- per node: leaf or internal, split axis / value (kdTreeImpl.h:113-170) ----------------------
__device__ __forceinline__ void decide_node(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv, uint32_t bound,
                                            const BMeas* __restrict__ meas, uint32_t bucket, uint32_t* __restrict__ kind,
                                            uint32_t* __restrict__ axis, double* __restrict__ splitval,
                                            uint32_t* __restrict__ nleft, const uint32_t i)
{
  if (i > bound) return;
  if (i >= lv->nseg) { kind[i] = 0u; return; }   // zero padding: the rank scan runs over `bound + 1` entries
  nleft[i] = 0u;                                 // filled by k_count
  const BMeas m = meas[i];
  const double hx = 0.5 * (m.hi[0] - m.lo[0]), hy = 0.5 * (m.hi[1] - m.lo[1]), hz = 0.5 * (m.hi[2] - m.lo[2]);
  int ax;
  if (hx > hy) ax = (hx > hz) ? 0 : 2;
  else ax = (hy > hz) ? 1 : 2;
  const double mx = fmax(fmax(hx, hy), hz);
  const bool leaf = (segs[i].n <= bucket) || (fabs(mx) < 0.01);
  kind[i] = leaf ? 0u : 1u;
  axis[i] = (uint32_t)ax;
  splitval[i] = m.mean[ax];
}
__global__ void k_decide(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv, uint32_t bound,
                         const BMeas* __restrict__ meas, uint32_t bucket, uint32_t* __restrict__ kind,
                         uint32_t* __restrict__ axis, double* __restrict__ splitval, uint32_t* __restrict__ nleft)
{
  decide_node(segs, lv, bound, meas, bucket, kind, axis, splitval, nleft, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- per node: write the record / register the bucket, hook it into its parent ------------------
__device__ __forceinline__ void emit_node(const BSeg* __restrict__ segs, BLevel* __restrict__ lv, const BMeas* __restrict__ meas,
                                          const uint32_t* __restrict__ kind, const uint32_t* __restrict__ axis,
                                          const double* __restrict__ splitval, const uint32_t* __restrict__ irank,
                                          KdNode* __restrict__ nodes, double* __restrict__ node_r,
                                          LeafEntry* __restrict__ leaf_tab, uint32_t* __restrict__ root_ref,
                                          uint32_t* __restrict__ max_leaf, const uint32_t i)
{
  const uint32_t nseg = lv[0].nseg, node_base = lv[0].node_base, leaf_base = lv[0].leaf_base;
  if (i == 0) {   // the next level: two children per internal node of this one
    const uint32_t n_internal = irank[nseg];
    lv[1].nseg = 2u * n_internal; lv[1].node_base = node_base + n_internal; lv[1].leaf_base = leaf_base + (nseg - n_internal);
  }
  if (i >= nseg) return;
  const BSeg sg = segs[i];
  uint32_t ref;
  if (kind[i]) {
    const BMeas m = meas[i];
    const uint32_t me = node_base + irank[i];
    KdNode nd;
    nd.cx = 0.5 * (m.lo[0] + m.hi[0]);
    nd.cy = 0.5 * (m.lo[1] + m.hi[1]);
    nd.cz = 0.5 * (m.lo[2] + m.hi[2]);
    nd.hx = 0.5 * (m.hi[0] - m.lo[0]);
    nd.hy = 0.5 * (m.hi[1] - m.lo[1]);
    nd.hz = 0.5 * (m.hi[2] - m.lo[2]);
    nd.splitval = splitval[i];
    nd.c1 = (axis[i] & 1u) ? REF_AXIS : 0u;
    nd.c2 = (axis[i] & 2u) ? REF_AXIS : 0u;
    nodes[me] = nd;
    node_r[me] = __dsqrt_rn(nd.hx * nd.hx + nd.hy * nd.hy + nd.hz * nd.hz);
    ref = me;
  } else {
    const uint32_t id = leaf_base + (i - irank[i]);  // every node is either internal or a bucket
    leaf_tab[id].start = (int32_t)sg.start;
    leaf_tab[id].count = (int32_t)sg.n;
    atomicMax(max_leaf, sg.n);
    ref = REF_LEAF | id;
  }
  if (sg.parent < 0) *root_ref = ref;
  else {
    uint32_t* slot = sg.side ? &nodes[sg.parent].c2 : &nodes[sg.parent].c1;
    *slot = (*slot & REF_AXIS) | ref;
  }
}
__global__ void k_emit(const BSeg* __restrict__ segs, BLevel* __restrict__ lv, const BMeas* __restrict__ meas,
                       const uint32_t* __restrict__ kind, const uint32_t* __restrict__ axis,
                       const double* __restrict__ splitval, const uint32_t* __restrict__ irank,
                       KdNode* __restrict__ nodes, double* __restrict__ node_r, LeafEntry* __restrict__ leaf_tab,
                       uint32_t* __restrict__ root_ref, uint32_t* __restrict__ max_leaf)
{
  emit_node(segs, lv, meas, kind, axis, splitval, irank, nodes, node_r, leaf_tab, root_ref, max_leaf,
            blockIdx.x * blockDim.x + threadIdx.x);
}
// a level of at most 1023 nodes (the first ten of any tree): decide, rank (exclusive scan of `kind` over bound + 1
// entries) and emit in one workgroup -- three dependent launches less per level, which is what a small scan's build is
// made of (~13 launches per level at ~4.5 us each)
__device__ __forceinline__ void nodes_small_body(const BSeg* __restrict__ segs, BLevel* __restrict__ lv, uint32_t bound,
                                                 const BMeas* __restrict__ meas, uint32_t bucket,
                                                 uint32_t* __restrict__ kind, uint32_t* __restrict__ axis,
                                                 double* __restrict__ splitval, uint32_t* __restrict__ nleft,
                                                 uint32_t* __restrict__ irank, KdNode* __restrict__ nodes,
                                                 double* __restrict__ node_r, LeafEntry* __restrict__ leaf_tab,
                                                 uint32_t* __restrict__ root_ref, uint32_t* __restrict__ max_leaf)
{
  __shared__ uint32_t wtot[1024 / WAVE];
  const uint32_t i = threadIdx.x;
  decide_node(segs, lv, bound, meas, bucket, kind, axis, splitval, nleft, i);
  const uint32_t k = (i <= bound) ? kind[i] : 0u;          // this thread's own store
  const unsigned long long m = __ballot(k != 0u);
  const uint32_t lane = i & (WAVE - 1), wv = i / WAVE;
  if (lane == 0) wtot[wv] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t before = 0;
  for (uint32_t w = 0; w < wv; w++) before += wtot[w];
  const uint32_t r = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (i <= bound) irank[i] = r;
  __threadfence_block();
  __syncthreads();                                          // emit reads irank[nseg] and other nodes' nothing else
  emit_node(segs, lv, meas, kind, axis, splitval, irank, nodes, node_r, leaf_tab, root_ref, max_leaf, i);
}
__global__ void __launch_bounds__(1024) k_nodes_small(const BSeg* __restrict__ segs, BLevel* __restrict__ lv, uint32_t bound,
                                                      const BMeas* __restrict__ meas, uint32_t bucket,
                                                      uint32_t* __restrict__ kind, uint32_t* __restrict__ axis,
                                                      double* __restrict__ splitval, uint32_t* __restrict__ nleft,
                                                      uint32_t* __restrict__ irank, KdNode* __restrict__ nodes,
                                                      double* __restrict__ node_r, LeafEntry* __restrict__ leaf_tab,
                                                      uint32_t* __restrict__ root_ref, uint32_t* __restrict__ max_leaf)
{
  nodes_small_body(segs, lv, bound, meas, bucket, kind, axis, splitval, nleft, irank, nodes, node_r, leaf_tab, root_ref, max_leaf);
}
// the same behind the level's k_big_approx, for a speculated level of at most 16 nodes: wave w is node w's (one launch less)
__global__ void __launch_bounds__(1024) k_nodes_small_approx(const BSeg* __restrict__ segs, BLevel* __restrict__ lv, uint32_t bound,
                                                             BMeas* __restrict__ meas, uint32_t bucket,
                                                             uint32_t* __restrict__ kind, uint32_t* __restrict__ axis,
                                                             double* __restrict__ splitval, uint32_t* __restrict__ nleft,
                                                             uint32_t* __restrict__ irank, KdNode* __restrict__ nodes,
                                                             double* __restrict__ node_r, LeafEntry* __restrict__ leaf_tab,
                                                             uint32_t* __restrict__ root_ref, uint32_t* __restrict__ max_leaf,
                                                             const BPart* __restrict__ part, BSpecLevel L, int fault, uint32_t big_min)
{
  big_approx_wave(threadIdx.x / WAVE, segs, lv, part, meas, L, fault, big_min);
  __threadfence_block();
  __syncthreads();
  nodes_small_body(segs, lv, bound, meas, bucket, kind, axis, splitval, nleft, irank, nodes, node_r, leaf_tab, root_ref, max_leaf);
}

// ---- per node: how many of its points lie below the split value (the position of the split) ---------------
// Order-independent, so no prefix sum is needed for it: a wave covers 1024 consecutive positions, nodes are
// contiguous runs of positions, one 32-bit atomicAdd per run.
#define CNT_ITERS 8
__global__ void __launch_bounds__(256) k_count(const uint32_t* __restrict__ seg_of, const uint32_t* __restrict__ kind,
                                               const uint32_t* __restrict__ axis, const double* __restrict__ splitval,
                                               const double* __restrict__ cx, const double* __restrict__ cy,
                                               const double* __restrict__ cz, uint32_t M, uint32_t* __restrict__ nleft)
{
  const uint32_t lane = threadIdx.x & (WAVE - 1);
  const uint32_t base = (blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE) * (WAVE * CNT_ITERS);
  // the loads of all CNT_ITERS rows are issued phase by phase (label -> node kind / axis / split value -> coordinate),
  // each phase CNT_ITERS independent loads deep: with 80 waves for an 81K-point scan the kernel is pure latency
  uint32_t sgs[CNT_ITERS], axs[CNT_ITERS];
  double svs[CNT_ITERS], cs[CNT_ITERS];
#pragma unroll
  for (int it = 0; it < CNT_ITERS; it++) {
    const uint32_t p = base + (uint32_t)it * WAVE + lane;
    sgs[it] = (p < M) ? seg_of[p] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int it = 0; it < CNT_ITERS; it++)
    if (sgs[it] != 0xFFFFFFFFu && !kind[sgs[it]]) sgs[it] = 0xFFFFFFFFu;
#pragma unroll
  for (int it = 0; it < CNT_ITERS; it++) {
    axs[it] = 0; svs[it] = 0.0;
    if (sgs[it] != 0xFFFFFFFFu) { axs[it] = axis[sgs[it]]; svs[it] = splitval[sgs[it]]; }
  }
#pragma unroll
  for (int it = 0; it < CNT_ITERS; it++) {
    const uint32_t p = base + (uint32_t)it * WAVE + lane;
    cs[it] = 0.0;
    if (sgs[it] != 0xFFFFFFFFu) cs[it] = (axs[it] == 0) ? cx[p] : ((axs[it] == 1) ? cy[p] : cz[p]);
  }
  // The counts of one node are gathered per workgroup before they go to memory: on the top levels every wave of the
  // launch adds to the same one or two counters (2 000 atomics on one address: 28 us at 1M points, against 10 us on the
  // levels where the nodes are many).  The node in question is the one the workgroup's first labelled position belongs to.
  __shared__ uint32_t s_node, s_cnt;
  if (threadIdx.x == 0) { s_node = 0xFFFFFFFFu; s_cnt = 0u; }
  __syncthreads();
  {
    unsigned long long any = __ballot(sgs[0] != 0xFFFFFFFFu);
    if (threadIdx.x / WAVE == 0 && any && lane == (uint32_t)(__ffsll((long long)any) - 1)) s_node = sgs[0];
  }
  __syncthreads();
  const uint32_t shared_node = s_node;
  auto flush = [&](uint32_t node, uint32_t cnt) {
    if (lane != 0 || !cnt) return;
    if (node == shared_node) atomicAdd(&s_cnt, cnt); else atomicAdd(&nleft[node], cnt);
  };
  uint32_t pend = 0xFFFFFFFFu, pcnt = 0;
#pragma unroll
  for (int it = 0; it < CNT_ITERS; it++) {
    const uint32_t sg = sgs[it];
    const bool lt = (sg != 0xFFFFFFFFu) && (cs[it] < svs[it]);
    unsigned long long todo = __ballot(sg != 0xFFFFFFFFu);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const uint32_t cur = __shfl(sg, leader, WAVE);
      const bool mine = (sg == cur);
      const uint32_t add = (uint32_t)__popcll(__ballot(mine && lt));
      if (cur != pend) {
        if (pend != 0xFFFFFFFFu) flush(pend, pcnt);
        pend = cur; pcnt = add;
      } else pcnt += add;
      todo &= ~__ballot(mine);
    }
  }
  if (pend != 0xFFFFFFFFu) flush(pend, pcnt);
  __syncthreads();
  if (threadIdx.x == 0 && s_cnt) atomicAdd(&nleft[shared_node], s_cnt);
}

// ---- per element: misplaced on the left / on the right of its node's split position --------------
__device__ __forceinline__ void misplaced_at(const uint32_t* __restrict__ seg_of, const uint32_t* __restrict__ kind,
                            const BSeg* __restrict__ segs, const uint32_t* __restrict__ axis,
                            const double* __restrict__ splitval, const uint32_t* __restrict__ nleft,
                            const double* __restrict__ cx, const double* __restrict__ cy, const double* __restrict__ cz,
                            uint32_t M, unsigned long long* __restrict__ LR, const uint32_t p)
{
  if (p > M) return;
  uint32_t l = 0, r = 0;
  if (p < M) {
    const uint32_t sg = seg_of[p];
    if (sg != 0xFFFFFFFFu && kind[sg]) {
      const uint32_t ax = axis[sg];
      const double c = (ax == 0) ? cx[p] : ((ax == 1) ? cy[p] : cz[p]);
      const bool f = c < splitval[sg];
      const bool left_region = (p - segs[sg].start) < nleft[sg];
      l = (left_region && !f) ? 1u : 0u;
      r = (!left_region && f) ? 1u : 0u;
    }
  }
  LR[p] = (unsigned long long)l | ((unsigned long long)r << 32);   // one scan serves both counts; index M = 0 (totals)
}

// k-th misplaced from the left pairs with the k-th misplaced from the right end
__global__ void k_swaplist(const uint32_t* __restrict__ seg_of, const BSeg* __restrict__ segs,
                           const unsigned long long* __restrict__ LR, const unsigned long long* __restrict__ AB,
                           uint32_t M, uint32_t* __restrict__ posL, uint32_t* __restrict__ posR)
{
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

// launched over M/2 slots (an element is misplaced at most once per level); the count sits on the device
__device__ __forceinline__ void swap_at(const uint32_t* __restrict__ posL, const uint32_t* __restrict__ posR,
                       const unsigned long long* __restrict__ nswap_ptr, uint32_t* __restrict__ perm,
                       double* __restrict__ cx, double* __restrict__ cy, double* __restrict__ cz, const uint32_t c)
{
  if (c >= (uint32_t)*nswap_ptr) return;
  const uint32_t a = posL[c], b = posR[c];
  const uint32_t pa = perm[a], pb = perm[b];
  perm[a] = pb; perm[b] = pa;
  double t;
  t = cx[a]; cx[a] = cx[b]; cx[b] = t;
  t = cy[a]; cy[a] = cy[b]; cy[b] = t;
  t = cz[a]; cz[a] = cz[b]; cz[b] = t;
}

// ---- next level: two children per internal node; relabel the elements ---------------------------
__device__ __forceinline__ void children_of(const BSeg* __restrict__ segs, const BLevel* __restrict__ lv, const uint32_t* __restrict__ kind,
                           const uint32_t* __restrict__ irank, const uint32_t* __restrict__ nleft_,
                           BSeg* __restrict__ next, uint32_t* __restrict__ err, const uint32_t i)
{
  if (i >= lv->nseg || !kind[i]) return;
  const BSeg sg = segs[i];
  uint32_t nleft = nleft_[i];
  if (nleft == 0 || nleft >= sg.n) {   // degenerate split (non-finite input): flag it, and keep the children valid
    atomicExch(err, 1u);               // ranges -- more levels of this batch are already enqueued behind this one
    nleft = (nleft == 0) ? 1u : sg.n - 1u;
  }
  const uint32_t me = lv->node_base + irank[i];
  next[2 * irank[i]] = {sg.start, nleft, (int32_t)me, 0u};
  next[2 * irank[i] + 1] = {sg.start + nleft, sg.n - nleft, (int32_t)me, 1u};
}
__device__ __forceinline__ void relabel_at(const BSeg* __restrict__ segs, const uint32_t* __restrict__ kind,
                          const uint32_t* __restrict__ irank, const uint32_t* __restrict__ nleft, uint32_t M,
                          uint32_t* __restrict__ seg_of, const uint32_t p)
{
  if (p >= M) return;
  const uint32_t sg = seg_of[p];
  if (sg == 0xFFFFFFFFu) return;
  if (!kind[sg]) { seg_of[p] = 0xFFFFFFFFu; return; }
  seg_of[p] = 2u * irank[sg] + (((p - segs[sg].start) < nleft[sg]) ? 0u : 1u);
}
// two independent passes per launch (the first `nb_first` workgroups run the one, the rest the other): every launch of
// a level waits for the one before it, and there are levels x a dozen of them
__global__ void k_misplaced_children(const uint32_t* __restrict__ seg_of, const uint32_t* __restrict__ kind,
                                     const BSeg* __restrict__ segs, const uint32_t* __restrict__ axis,
                                     const double* __restrict__ splitval, const uint32_t* __restrict__ nleft,
                                     const double* __restrict__ cx, const double* __restrict__ cy,
                                     const double* __restrict__ cz, uint32_t M, unsigned long long* __restrict__ LR,
                                     uint32_t nb_first, const BLevel* __restrict__ lv, const uint32_t* __restrict__ irank,
                                     BSeg* __restrict__ next, uint32_t* __restrict__ err)
{
  if (blockIdx.x < nb_first) misplaced_at(seg_of, kind, segs, axis, splitval, nleft, cx, cy, cz, M, LR, blockIdx.x * blockDim.x + threadIdx.x);
  else children_of(segs, lv, kind, irank, nleft, next, err, (blockIdx.x - nb_first) * blockDim.x + threadIdx.x);
}
// ... and, on a speculated level, a third: what the final check needs of the level's big internal nodes (spec_keep_at)
__global__ void k_misplaced_children_keep(const uint32_t* __restrict__ seg_of, const uint32_t* __restrict__ kind,
                                          const BSeg* __restrict__ segs, const uint32_t* __restrict__ axis,
                                          const double* __restrict__ splitval, const uint32_t* __restrict__ nleft,
                                          const double* __restrict__ cx, const double* __restrict__ cy,
                                          const double* __restrict__ cz, uint32_t M, unsigned long long* __restrict__ LR,
                                          uint32_t nb_first, const BLevel* __restrict__ lv, const uint32_t* __restrict__ irank,
                                          BSeg* __restrict__ next, uint32_t* __restrict__ err, BSpecLevel L)
{
  if (blockIdx.x < nb_first) misplaced_at(seg_of, kind, segs, axis, splitval, nleft, cx, cy, cz, M, LR, blockIdx.x * blockDim.x + threadIdx.x);
  else {
    const uint32_t i = (blockIdx.x - nb_first) * blockDim.x + threadIdx.x;
    children_of(segs, lv, kind, irank, nleft, next, err, i);
    spec_keep_at(lv, kind, irank, nleft, L, i);
  }
}
__global__ void k_swap_relabel(const uint32_t* __restrict__ posL, const uint32_t* __restrict__ posR,
                               const unsigned long long* __restrict__ nswap_ptr, uint32_t* __restrict__ perm,
                               double* __restrict__ cx, double* __restrict__ cy, double* __restrict__ cz, uint32_t nb_first,
                               const BSeg* __restrict__ segs, const uint32_t* __restrict__ kind,
                               const uint32_t* __restrict__ irank, const uint32_t* __restrict__ nleft, uint32_t M,
                               uint32_t* __restrict__ seg_of)
{
  if (blockIdx.x < nb_first) swap_at(posL, posR, nswap_ptr, perm, cx, cy, cz, blockIdx.x * blockDim.x + threadIdx.x);
  else relabel_at(segs, kind, irank, nleft, M, seg_of, (blockIdx.x - nb_first) * blockDim.x + threadIdx.x);
}

// ---- the partition of a level in TWO passes over the points (round 6) ------------------------------------------------
// The five passes above (count, misplaced flags, scan, swap list, swap + relabel) read and write a level's points
// five times; from a million points on those passes -- not their launches -- are what a level costs.  What the Hoare
// loop of kdTreeImpl.h:172-182 leaves behind needs less.  Call an element "ge" when it is not below the split value.
// Within a node [s, s + n) with nleft elements below the split value, the k-th misplaced element from the left is the
// k-th ge element of the node (every ge element in front of it lies in the left region too), and the k-th misplaced
// element from the right end is the k-th element below the split value counted from the right end.  So ONE segmented
// scan of the ge flags (geBefore(p) = ge elements of p's node in front of p) places every element in an index list
// per node -- ge elements from the front in order, the others from the back in order:
//     list[s + geBefore(p)] = p                        p is ge
//     list[s + n - 1 - ((p - s) - geBefore(p))] = p    p is below the split value
// and the node's last element knows nleft = n - (ge elements of the node).  Pass 1 (k_part_scan) is that scan with
// the flags computed on the fly (label -> node -> coordinate), decoupled look-back over tiles of PS_TILE positions
// whose status word carries flag, epoch, "a node starts in this tile" and the count since that start (27 bits) -- a
// segmented look-back ends at the first predecessor in which a node starts.  Pass 2 (k_part_swap): slot j of node
// (s, n, nleft) with j < n - nleft holds a = list[s + j], the j-th ge element; it is misplaced iff a < s + nleft, and
// its partner is list[s + (n - nleft) + j] -- both reads coalesced; the same thread gives position s + j its label
// of the next level.  12 + 4 bytes read and 4 written per point in pass 1, 8 + 4 in pass 2 plus the swaps themselves.
size_t part_state_bytes(size_t n) { return 8 * ((n + PS_TILE - 1) / PS_TILE + 1) + 64; }
__global__ void __launch_bounds__(PS_THREADS) k_part_scan(const uint32_t* __restrict__ seg_of, const uint32_t* __restrict__ kind,
                                                   const BSeg* __restrict__ segs, const uint32_t* __restrict__ axis,
                                                   const double* __restrict__ splitval, const double* __restrict__ cx,
                                                   const double* __restrict__ cy, const double* __restrict__ cz, uint32_t M,
                                                   uint32_t* __restrict__ nleft, uint32_t* __restrict__ list,
                                                   unsigned long long* __restrict__ status, uint32_t* __restrict__ counter,
                                                   uint32_t epoch, uint32_t ntiles, uint32_t* __restrict__ err)
{
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  const uint32_t tile = ps_draw_tile(counter, ntiles);
  const uint32_t wbase = tile * PS_TILE + wv * (WAVE * PS_ROWS) + lane;   // row r of this wave: wbase + 64 r
  uint32_t sg[PS_ROWS], geb[PS_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PS_ROWS; r++) { const uint32_t p = wbase + r * WAVE; sg[r] = (p < M) ? seg_of[p] : 0xFFFFFFFFu; }
#pragma unroll
  for (uint32_t r = 0; r < PS_ROWS; r++)
    if (sg[r] != 0xFFFFFFFFu && !kind[sg[r]]) sg[r] = 0xFFFFFFFFu;
  // the flags of every row (is not below the split value; is its node's first element) as bit r of two words: the
  // loads of all rows are in flight together, nothing of them stays in registers
  uint32_t gebits = 0u, headbits = 0u;
  {
    uint32_t axs[PS_ROWS], st[PS_ROWS];
    double c[PS_ROWS], sv[PS_ROWS];
#pragma unroll
    for (uint32_t r = 0; r < PS_ROWS; r++) {
      axs[r] = 0u; sv[r] = 0.0; st[r] = 0xFFFFFFFFu;
      if (sg[r] != 0xFFFFFFFFu) { axs[r] = axis[sg[r]]; sv[r] = splitval[sg[r]]; st[r] = segs[sg[r]].start; }
    }
#pragma unroll
    for (uint32_t r = 0; r < PS_ROWS; r++) {
      const uint32_t p = wbase + r * WAVE;
      c[r] = 0.0;
      if (sg[r] != 0xFFFFFFFFu) c[r] = (axs[r] == 0u) ? cx[p] : ((axs[r] == 1u) ? cy[p] : cz[p]);
    }
#pragma unroll
    for (uint32_t r = 0; r < PS_ROWS; r++) {
      if (sg[r] != 0xFFFFFFFFu && !(c[r] < sv[r])) gebits |= 1u << r;
      if (wbase + r * WAVE == st[r]) headbits |= 1u << r;
    }
  }
  uint32_t ext = 0u, win = 0u;
  ps_scan_core(gebits, headbits, tile, status, epoch, err, geb, ext, win);
#pragma unroll
  for (uint32_t r = 0; r < PS_ROWS; r++) {
    if (sg[r] == 0xFFFFFFFFu) continue;
    const uint32_t p = wbase + r * WAVE;
    const uint32_t g = geb[r] + (((ext >> r) & 1u) ? win : 0u);
    const BSeg q = segs[sg[r]];                            // (a second look: cached, and two registers per row less)
    const uint32_t j = p - q.start;
    const bool ge = (gebits >> r) & 1u;
    const uint32_t dst = ge ? (q.start + g) : (q.start + q.n - 1u - (j - g));
    list[dst] = p;
    if (j == q.n - 1u) nleft[sg[r]] = q.n - (g + (ge ? 1u : 0u));
  }
}
// pass 2: the swaps (driven from the list's ge half) and the labels of the next level; behind them, in the same launch,
// the children of the level's internal nodes (and what the final check keeps of a speculated level)
#define PW_ROWS 4u
__device__ __forceinline__ void part_swap_at(const uint32_t* __restrict__ list, const BSeg* __restrict__ segs,
                                             const uint32_t* __restrict__ kind, const uint32_t* __restrict__ irank,
                                             const uint32_t* __restrict__ nleft, uint32_t M, uint32_t* __restrict__ seg_of,
                                             uint32_t* __restrict__ perm, double* __restrict__ cx, double* __restrict__ cy,
                                             double* __restrict__ cz, const uint32_t vblock)
{
  const uint32_t base = vblock * (256u * PW_ROWS) + threadIdx.x;
  uint32_t sg[PW_ROWS], a[PW_ROWS], b[PW_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PW_ROWS; r++) {
    const uint32_t p = base + r * 256u;
    sg[r] = (p < M) ? seg_of[p] : 0xFFFFFFFFu;
    a[r] = (p < M) ? list[p] : 0u;
  }
  bool sw[PW_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PW_ROWS; r++) {
    const uint32_t p = base + r * 256u;
    sw[r] = false; b[r] = 0u;
    if (sg[r] == 0xFFFFFFFFu) continue;
    if (!kind[sg[r]]) { seg_of[p] = 0xFFFFFFFFu; continue; }
    const BSeg q = segs[sg[r]];
    const uint32_t nl = nleft[sg[r]], j = p - q.start, nge = q.n - nl;
    seg_of[p] = 2u * irank[sg[r]] + ((j < nl) ? 0u : 1u);
    if (j < nge && a[r] < q.start + nl) { sw[r] = true; b[r] = list[p + nge]; }
  }
  uint32_t pa[PW_ROWS], pb[PW_ROWS];
  double xa[PW_ROWS], xb[PW_ROWS], ya[PW_ROWS], yb[PW_ROWS], za[PW_ROWS], zb[PW_ROWS];
#pragma unroll
  for (uint32_t r = 0; r < PW_ROWS; r++)
    if (sw[r]) {
      pa[r] = perm[a[r]]; pb[r] = perm[b[r]];
      xa[r] = cx[a[r]]; xb[r] = cx[b[r]]; ya[r] = cy[a[r]]; yb[r] = cy[b[r]]; za[r] = cz[a[r]]; zb[r] = cz[b[r]];
    }
#pragma unroll
  for (uint32_t r = 0; r < PW_ROWS; r++)
    if (sw[r]) {
      perm[a[r]] = pb[r]; perm[b[r]] = pa[r];
      cx[a[r]] = xb[r]; cx[b[r]] = xa[r]; cy[a[r]] = yb[r]; cy[b[r]] = ya[r]; cz[a[r]] = zb[r]; cz[b[r]] = za[r];
    }
}
__global__ void __launch_bounds__(256) k_part_swap(const uint32_t* __restrict__ list, const BSeg* __restrict__ segs,
                                                   const uint32_t* __restrict__ kind, const uint32_t* __restrict__ irank,
                                                   const uint32_t* __restrict__ nleft, uint32_t M, uint32_t* __restrict__ seg_of,
                                                   uint32_t* __restrict__ perm, double* __restrict__ cx, double* __restrict__ cy,
                                                   double* __restrict__ cz, uint32_t nb_first, const BLevel* __restrict__ lv,
                                                   BSeg* __restrict__ next, uint32_t* __restrict__ err)
{
  if (blockIdx.x < nb_first) part_swap_at(list, segs, kind, irank, nleft, M, seg_of, perm, cx, cy, cz, blockIdx.x);
  else children_of(segs, lv, kind, irank, nleft, next, err, (blockIdx.x - nb_first) * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_part_swap_keep(const uint32_t* __restrict__ list, const BSeg* __restrict__ segs,
                                                        const uint32_t* __restrict__ kind, const uint32_t* __restrict__ irank,
                                                        const uint32_t* __restrict__ nleft, uint32_t M, uint32_t* __restrict__ seg_of,
                                                        uint32_t* __restrict__ perm, double* __restrict__ cx, double* __restrict__ cy,
                                                        double* __restrict__ cz, uint32_t nb_first, const BLevel* __restrict__ lv,
                                                        BSeg* __restrict__ next, uint32_t* __restrict__ err, BSpecLevel L)
{
  if (blockIdx.x < nb_first) part_swap_at(list, segs, kind, irank, nleft, M, seg_of, perm, cx, cy, cz, blockIdx.x);
  else {
    const uint32_t i = (blockIdx.x - nb_first) * blockDim.x + threadIdx.x;
    children_of(segs, lv, kind, irank, nleft, next, err, i);
    spec_keep_at(lv, kind, irank, nleft, L, i);
  }
}

// (its first workgroup also sets up what the levels start from -- the build's small words, the level counters, the root
// node: four memsets / copies less in front of a build that is a chain of dependent commands)
__global__ void k_init(const double* __restrict__ xyz, uint32_t M, uint32_t* __restrict__ perm,
                       uint32_t* __restrict__ seg_of, double* __restrict__ cx, double* __restrict__ cy,
                       double* __restrict__ cz, uint32_t* __restrict__ small, BLevel* __restrict__ lvl, BSeg* __restrict__ segs)
{
  if (blockIdx.x == 0) {
    if (threadIdx.x < 64) small[threadIdx.x] = 0u;
    for (uint32_t k = threadIdx.x; k < BUILD_MAX_LEVELS + 2; k += blockDim.x) {
      BLevel z = {0u, 0u, 0u};
      if (k == 0) z.nseg = 1u;
      lvl[k] = z;
    }
    if (threadIdx.x == 0) { BSeg root = {0u, M, -1, 0u}; segs[0] = root; }
  }
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  perm[p] = p; seg_of[p] = 0;
  cx[p] = xyz[3 * (size_t)p]; cy[p] = xyz[3 * (size_t)p + 1]; cz[p] = xyz[3 * (size_t)p + 2];
}
__global__ void k_points(const uint32_t* __restrict__ perm, const double* __restrict__ cx,
                         const double* __restrict__ cy, const double* __restrict__ cz, uint32_t M,
                         KdPoint* __restrict__ pts)
{
  const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  KdPoint q;
  q.x = cx[p]; q.y = cy[p]; q.z = cz[p]; q.orig = (int32_t)perm[p]; q.pad = 0;
  pts[p] = q;
}
// packed bucket references (start << cb | count) when they fit in 30 bits (same rule as kd_build.cpp)
__global__ void k_pack_refs(KdNode* __restrict__ nodes, uint32_t nnodes, const LeafEntry* __restrict__ leaf_tab,
                            uint32_t cb, uint32_t* __restrict__ root_ref)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  auto pack = [&](uint32_t ref) -> uint32_t {
    if (!(ref & REF_LEAF)) return ref;
    const LeafEntry le = leaf_tab[ref & REF_VAL];
    return (ref & (REF_LEAF | REF_AXIS)) | ((uint32_t)le.start << cb) | (uint32_t)le.count;
  };
  if (i < nnodes) {
    nodes[i].c1 = pack(nodes[i].c1);
    nodes[i].c2 = pack(nodes[i].c2);
  }
  if (i == 0) *root_ref = pack(*root_ref);
}

#include "build_finish.inc"   // the subtree finishers (k_fin_*): 730 lines of this translation unit

static inline uint32_t cdiv(size_t a, size_t b) { return (uint32_t)((a + b - 1) / b); }
static inline int bits_for(uint64_t v)
{
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b;
}

#define BCHK(expr)                                 \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) { res.err = _e; goto fail; } \
  } while (0)

// bytes of scratch a build over M points needs (every temporary, and the node / bucket records while their
// number is still unknown)
static size_t build_layout(size_t M, size_t* offs, size_t* scan_tmp_out);
static size_t spec_layout(size_t M, size_t base, size_t* SO, int* nlev_out);
// Lab library: every region of the arena is followed by 256 guard bytes, filled with a pattern before a build and looked
// at behind it -- a region written past its end (the hand-over bug of round 4: a million root records into a table of
// 8192) fails the build it happens in, whatever it would have hit.  The product's layout has no guards.
#ifdef TDTK_LAB
constexpr size_t ARENA_GUARD = 256;
static thread_local std::vector<size_t>* g_guard_sink = nullptr;      // the layout functions push the offset of every guard
struct GuardList { uint32_t n; uint32_t pad; size_t off[250]; };
__global__ void __launch_bounds__(64) k_guard_fill(char* arena, GuardList G)
{
  if (blockIdx.x < G.n) reinterpret_cast<uint32_t*>(arena + G.off[blockIdx.x])[threadIdx.x] = 0xA5C3A5C3u ^ blockIdx.x;
}
__global__ void __launch_bounds__(64) k_guard_check(const char* arena, GuardList G, uint32_t* bad)
{
  if (blockIdx.x < G.n && reinterpret_cast<const uint32_t*>(arena + G.off[blockIdx.x])[threadIdx.x] != (0xA5C3A5C3u ^ blockIdx.x))
    atomicMax(bad, blockIdx.x + 1u);
}
#else
constexpr size_t ARENA_GUARD = 0;
#endif
static inline void arena_guard(size_t& off)
{
#ifdef TDTK_LAB
  if (g_guard_sink) g_guard_sink->push_back(off);
#endif
  off += ARENA_GUARD;
}
size_t device_build_arena_bytes(size_t M, bool with_background_chain)
{
  const size_t base = build_layout(M, nullptr, nullptr);
  return with_background_chain ? spec_layout(M, base, nullptr, nullptr) : base;
}

// Builds on `s` inside the caller's scratch `arena_` (>= device_build_arena_bytes(M), reused from build to build:
// no hipMalloc / hipFree of hundreds of MB per tree, and no device-wide sync from hipFree while another thread's
// kernels run).  On success the caller owns res.{nodes,node_r,pts,leaf_tab}, from the handle pool (pool.cpp) at their sizes.
DevBuildResult device_build_tree(const double* d_xyz, size_t M_, int bucket, void* arena_, hipStream_t s, const BuildSide* side,
                                 int no_finish)
{
  DevBuildResult res{};
  const uint32_t M = (uint32_t)M_;
  char* arena = static_cast<char*>(arena_);
  KdNode* nodes = nullptr; double* node_r = nullptr; LeafEntry* leaf_tab = nullptr; KdPoint* pts = nullptr;
  KdNode* f_nodes = nullptr; double* f_r = nullptr; LeafEntry* f_leaf = nullptr;
  uint32_t node_count = 0, leaf_count = 0, depth = 0;
  uint32_t h_spec_err = 0;
  bool tail_enqueued = false;
  bool spec_on = false;
  bool spec_suspect = false;   // the failure may be the speculation's doing (see `fail`)
  size_t scan_tmp = 0;
  size_t O[48];
  (void)build_layout(M_, O, &scan_tmp);
#ifdef TDTK_LAB
  GuardList guards{};
  {
    std::vector<size_t> g;
    g_guard_sink = &g;
    // (the speculative levels' regions only when the caller has them: without background streams the arena ends at the base
    // layout -- device_build_arena_bytes(M, false))
    const size_t base = build_layout(M_, nullptr, nullptr);
    if (side && side->s2) (void)spec_layout(M_, base, nullptr, nullptr);
    g_guard_sink = nullptr;
    guards.n = (uint32_t)std::min<size_t>(g.size(), 250);
    for (uint32_t k = 0; k < guards.n; k++) guards.off[k] = g[k];
    if (guards.n) hipLaunchKernelGGL(k_guard_fill, dim3(guards.n), dim3(64), 0, s, arena, guards);
    // (TDTK_GUARD_SELFTEST=1: one word behind the sixth region is overwritten on purpose -- the build must fail)
    if (guards.n > 5 && lab_env("TDTK_GUARD_SELFTEST")) (void)hipMemsetAsync(arena + guards.off[5] + 8, 0, 4, s);
  }
#endif
  const size_t n1 = (size_t)M + 1;
  const size_t o_perm = O[0], o_segof = O[1], o_cx = O[2], o_cy = O[3], o_cz = O[4], o_f = O[5], o_F = O[6], o_isL = O[7],
               o_A = O[8], o_posL = O[9], o_posR = O[10], o_segA = O[11], o_segB = O[12], o_meas = O[13], o_kind = O[14],
               o_axis = O[15], o_split = O[16], o_irank = O[17], o_tmp = O[18], o_small = O[19], o_nodes = O[20],
               o_r = O[21], o_leaf = O[22], o_lvl = O[23];
  (void)o_f; (void)o_F;
  nodes = (KdNode*)(arena + o_nodes); node_r = (double*)(arena + o_r); leaf_tab = (LeafEntry*)(arena + o_leaf);
  BCHK((hipError_t)pool_malloc_raw((void**)&pts, sizeof(KdPoint) * (size_t)M));
  {
    uint32_t* perm = (uint32_t*)(arena + o_perm); uint32_t* seg_of = (uint32_t*)(arena + o_segof);
    double *cx = (double*)(arena + o_cx), *cy = (double*)(arena + o_cy), *cz = (double*)(arena + o_cz);
    uint32_t *nleft = (uint32_t*)(arena + o_F), *posL = (uint32_t*)(arena + o_posL), *posR = (uint32_t*)(arena + o_posR);
    BLevel* lvl = (BLevel*)(arena + o_lvl);
    unsigned long long *LR = (unsigned long long*)(arena + o_isL), *AB = (unsigned long long*)(arena + o_A);
    BSeg* segs = (BSeg*)(arena + o_segA); BSeg* next = (BSeg*)(arena + o_segB);
    BMeas* meas = (BMeas*)(arena + o_meas);
    uint32_t *kind = (uint32_t*)(arena + o_kind), *axis = (uint32_t*)(arena + o_axis), *irank = (uint32_t*)(arena + o_irank);
    double* splitval = (double*)(arena + o_split);
    void* tmp = arena + o_tmp;
    BPiece* pieces = (BPiece*)(arena + O[24]);
    BPre *prein = (BPre*)(arena + O[25]), *preout = (BPre*)(arena + O[26]);
    BSum *own = (BSum*)(arena + O[27]), *comp = (BSum*)(arena + O[28]);
    uint32_t* wlist = (uint32_t*)(arena + O[29]);
    const uint32_t nblocks = cdiv(M, BIG_CH);
    static const bool chain_only = [] { const char* e = lab_env("TDTK_BUILD_CHAIN"); return e && e[0] == '1'; }();
    const uint32_t big_min = build_big_min(M_);
    const bool use_big = !chain_only && M >= big_min;
    const uint32_t big_level_min = build_big_level_min(M_);
    // TDTK_MEASURE=axis: round 2's wave per (node, axis) for the nodes below the piecewise path (k_measure); default: a wave
    // per node (k_measure_node).  With TDTK_BUILD_CHAIN=1 (no piecewise path: chains of any length) the long-chain kernel.
    static const bool measure_axis_env = [] { const char* e = lab_env("TDTK_MEASURE"); return e && e[0] == 'a'; }();
    const bool measure_per_axis_always = measure_axis_env || chain_only;
    const int big_dbg_all = lab_env("TDTK_BIG_DEBUG") ? atoi(lab_env("TDTK_BIG_DEBUG")) : 0;
    // speculative splits: TDTK_BUILD_SPEC=0 builds in order; TDTK_BUILD_SPEC_FAULT=1 (tests) cuts every big node elsewhere than
    // the exact sum would, so that the final check fails and the in-order path takes over
    static const bool spec_env = [] { const char* e = getenv("TDTK_BUILD_SPEC"); return !(e && e[0] == '0'); }();
    static const int spec_fault = [] { const char* e = lab_env("TDTK_BUILD_SPEC_FAULT"); return (e && e[0] == '1') ? 1 : 0; }();
    const bool spec = spec_env && side && side->s2 && use_big && !big_dbg_all;
    spec_on = spec;
    BSpecAll SP;
    SP.n = 0;
    size_t SO[BIG_SPEC_MAX * SPEC_SLOTS + 5];
    int spec_levels = 0;
    void *tmp2 = nullptr, *tmp3 = nullptr, *tmp4 = nullptr;
    if (spec) {
      (void)spec_layout(M_, build_layout(M_, nullptr, nullptr), SO, &spec_levels);
      tmp2 = arena + SO[(size_t)BIG_SPEC_MAX * SPEC_SLOTS];
      tmp3 = arena + SO[(size_t)BIG_SPEC_MAX * SPEC_SLOTS + 2];
      tmp4 = arena + SO[(size_t)BIG_SPEC_MAX * SPEC_SLOTS + 3];
    }
    const int big_dbg = big_dbg_all & (3 | 16);   // 1: never trust a folded run, 2: walk every piece, 4: garbage in the arena, 8: compare with the chain
    if (big_dbg_all & 4) BCHK(hipMemsetAsync(arena, 0xFF, build_layout(M_, nullptr, nullptr), s));
    uint32_t* small = (uint32_t*)(arena + o_small);  // [0] root_ref [1] max_leaf [2] err
    // the partition's scan in one launch while the positions fit its 27-bit counters (TDTK_OWN_SCAN=0: rocPRIM's two)
    static const bool own_scan_env = [] { const char* e = lab_env("TDTK_OWN_SCAN"); return !(e && e[0] == '0'); }();
    const bool own_scan = own_scan_env && n1 < ((size_t)1 << 27);
    const size_t o_scanstate = O[31];
    if (own_scan) BCHK(hipMemsetAsync(arena + o_scanstate, 0, std::max(scan_pair27_state_bytes(n1), part_state_bytes(n1)), s));
    hipLaunchKernelGGL(k_init, dim3(cdiv(M, 256)), dim3(256), 0, s, d_xyz, M, perm, seg_of, cx, cy, cz, small, lvl, segs);

    // Levels are enqueued in batches; how many nodes a level has, and where its node / bucket records start, is
    // known on the device only (lvl[]).  The host looks once after a first batch as deep as a balanced tree gets
    // down to bucket-sized nodes, then every two levels; per-node kernels are launched over an upper bound of the
    // level's node count (what the last look saw, doubled per level since) and return past the real count.
    // diagnostics (TDTK_BUILD_TRACE=1): an event at the start of every level on the build's stream; the elapsed times are
    // printed when the build is done -- what a level costs when no profiler is serialising the launches
    static const bool lvl_trace = [] { const char* e = lab_env("TDTK_BUILD_TRACE"); return e && (e[0] == '1' || e[0] == '2'); }();
    static const bool fin_trace = [] { const char* e = lab_env("TDTK_BUILD_TRACE"); return e && e[0] == '2'; }();
    std::vector<hipEvent_t> lvl_ev;
    uint32_t level = 0, known = 1, known_at = 0;
    uint32_t batch = 1;
    for (size_t c = (size_t)(bucket > 0 ? bucket : 1); c < M_; c <<= 1) batch++;
    batch += 2;   // mean splits do not halve: a 1M-point cloud is 18-19 levels deep, not 17, and a level past the end of the
                  // tree costs less than the host's look (it moves nothing)
    // The level from which every node is finished by one workgroup (k_fin_subtrees): the first at which a balanced node
    // holds at most FIN_HANDOFF points.  TDTK_BUILD_FINISH=0 (lab): level by level to the end.
    uint32_t fin_level = 0xFFFFFFFFu;
    const uint32_t fin_cap = fin_max_subtrees(M_);
    {
      static const bool fin_env = [] { const char* e = lab_env("TDTK_BUILD_FINISH"); return !(e && e[0] == '0'); }();
      uint32_t L = 0;
      // (round 6: a cloud of two million points and more goes on by levels until a balanced node fits a wave's finisher --
      //  two or three more levels of launches against most of the workgroup finisher's time; TDTK_BUILD_HANDOFF, lab)
      const uint32_t handoff_env = [] { const char* e = lab_env("TDTK_BUILD_HANDOFF"); return e ? (uint32_t)atoi(e) : 0u; }();
      const uint32_t handoff = handoff_env ? handoff_env : (M_ >= FIN_HANDOFF_WAVE_FROM ? FIN_HANDOFF_WAVE : FIN_HANDOFF);
      while ((M_ >> L) > handoff) L++;
      if (fin_env && no_finish == 0 && bucket >= 6 && L < 31 && (1u << L) <= fin_cap && !big_dbg_all) fin_level = L;
    }
    if (fin_level != 0xFFFFFFFFu) batch = fin_level;
    // From chain_from on the exact sums of a speculated level are plain chains in the background (k_chain_exact) -- 4.2 ns
    // per point of the level's largest node, which must end before the build does.  A balanced cloud's do; a cloud with a
    // third of its points in one tight cluster has a 1.3M-point node on levels 5 .. 10 of 4M points, and its chains (5.6 ms)
    // were what the build waited for.  So a build of two million points and more -- where a look costs a few per cent of a
    // level -- looks at the largest node before the first such level (and before every further one while it is too large):
    // beyond M / 17 points the level takes the piecewise path.
    // chain_from: the first level whose balanced node's chain (4.2 ns a point) is a small share of what a build of this
    // size takes -- 20 000 + M / 50 points, the sixth level at the latest: level 0 for a 15K-point scan, 1 at 40K, 2 at 81K,
    // 4 at 300K, 5 from 1M on (measured: 4 at 1M is 8 % slower, 3 at 300K 4 %).  On small clouds this is not about bandwidth
    // but about the HOST: a level's piecewise path is nine launches on the side streams, and enqueuing them (3-4 us
    // apiece) was what a level of a 15K .. 40K-point build took (15K: 0.50 -> 0.41 ms, lab library).
    const uint32_t chain_from = [&] {
      const char* e = lab_env("TDTK_BUILD_CHAINFROM");
      if (e) return (uint32_t)atoi(e);
      const size_t thresh = 20000u + M_ / 50u;
      uint32_t L = 0;
      while (L < 5u && (M_ >> L) > thresh) L++;
      return L;
    }();
    bool chain_ok = true;
    bool chain_look = spec && M_ >= 2000000u && fin_level != 0xFFFFFFFFu && chain_from < fin_level && chain_from < (uint32_t)spec_levels;
    if (chain_look) batch = chain_from;
    // The host's looks at the device land in pinned memory when the caller has some (a copy into pageable memory is a
    // synchronisation of its own): h_small = small[0 .. 7] (root reference, largest bucket, error word, check word, the
    // hand-over level's largest node and node count), hl = the level counters.
    std::vector<unsigned char> h_pageable;
    unsigned char* const hst = (side && side->h_pin) ? static_cast<unsigned char*>(side->h_pin) : (h_pageable.resize(65536), h_pageable.data());
    uint32_t* const h_small = reinterpret_cast<uint32_t*>(hst);
    BLevel* const hl = reinterpret_cast<BLevel*>(hst + 64);
    static_assert(64 + sizeof(BLevel) * (BUILD_MAX_LEVELS + 2) <= 65536, "the staging block holds every level's counters");
    std::memset(hst, 0, 64);
    bool have_max = false, fin_retry = false;
    uint32_t fin_nodes = 0, fin_maxn = 0;     // nodes of the level about to be handed over, the largest of them (k_fin_maxn)
    for (;;) {
      if (level == fin_level) {
        // does the level's largest node fit a workgroup's LDS?  (An unbalanced cloud -- a real scan -- takes a level or two more.)
        if (!have_max) {
          hipLaunchKernelGGL(k_fin_maxn, dim3(1), dim3(256), 0, s, segs, lvl + level, small + 4);
          BCHK(hipMemcpyAsync(h_small, small, 32, hipMemcpyDeviceToHost, s));
          BCHK(hipStreamSynchronize(s));
        }
        have_max = false;
        const uint32_t h_max[2] = {h_small[4], h_small[5]};
        fin_nodes = h_max[1]; fin_maxn = h_max[0];
        if (h_max[0] > FIN_LDS && h_max[1] * 2u <= fin_cap && fin_level + 1u < 31u) {
          fin_level++;              // one more level by its own launches (below), then look again
          known = h_max[1]; known_at = level;
          batch = 1;
        } else if (h_max[0] > FIN_LDS || h_max[1] > fin_cap) {
          // too many nodes for the tables, or a node too large with no room for a wider level: on by levels -- and a look at
          // every batch's end whether the level has thinned out (round 6: a cloud with a third of its points in one tight
          // cluster is in that state for nine levels, until the sparse rest has ended in buckets; it used to stay by levels
          // to the last bucket, eight levels more)
          fin_level = 0xFFFFFFFFu;
          fin_retry = no_finish == 0;
          known = h_max[1]; known_at = level;
          batch = 2;
        }
      }
      if (level == fin_level) {
        // hand the level over: its nodes become the roots of subtrees.  As many workgroups as the level HAS nodes (counted
        // just above; the tables hold FIN_MAX_SUBTREES): 2^level of them -- right for a balanced tree -- is a million by the
        // time a lopsided cloud (one dense cluster that stays above FIN_LDS for sixteen levels) is handed over, and the
        // copy of that many root records ran over the arena (found by tools/fuzz_parity.py --seed 4401, round 4).
        const uint32_t tmax = fin_nodes ? fin_nodes : 1u;
        BSeg* roots = (BSeg*)(arena + O[37]);
        FinTab* ftab = (FinTab*)(arena + O[38]);
        FinOff* foff = (FinOff*)(arena + O[39]);
        BCHK(hipMemcpyAsync(roots, segs, sizeof(BSeg) * tmax, hipMemcpyDeviceToDevice, s));
        if (fin_trace) { const uint32_t magic = 0x7157u; BCHK(hipMemcpyAsync(small + 15, &magic, 4, hipMemcpyHostToDevice, s)); }
        if (lvl_trace) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); lvl_ev.push_back(e); } }
        {
          FinArgs fa;
          fa.roots = roots; fa.lvH = lvl + fin_level; fa.cx = cx; fa.cy = cy; fa.cz = cz; fa.perm = perm;
          fa.segA = (BSeg*)(arena + o_segA); fa.segB = (BSeg*)(arena + o_segB); fa.meas = meas;
          fa.kind = (uint32_t*)(arena + O[35]); fa.axis = axis; fa.splitval = splitval; fa.irank = (uint32_t*)(arena + O[36]); fa.nleft = nleft;
          fa.nodes_st = (KdNode*)(arena + O[32]); fa.r_st = (double*)(arena + O[33]); fa.leaf_st = (LeafEntry*)(arena + O[34]);
          fa.tab = ftab; fa.bucket = (uint32_t)bucket; fa.small = small;
          fa.sub_nlev = (uint32_t*)(arena + O[41]); fa.sub_maxleaf = (uint32_t*)(arena + O[42]);
          const bool half_env = [] { const char* e = lab_env("TDTK_BUILD_FINHALF"); return !(e && e[0] == '0'); }();
          // subtrees of at most FIN_LDS_HALF points two workgroups to a compute unit, then (if the level has any) the larger ones
          // ... and in front of both, the subtrees of at most FW_CAP points, a wave each
          const bool wave_env = [] { const char* e = lab_env("TDTK_BUILD_FINWAVE"); return !(e && e[0] == '0'); }();
          fa.n_lo = 0u; fa.skip_big = 1u;
          fa.dbg_levels = lab_env("TDTK_FW_DEBUG") ? (uint32_t)atoi(lab_env("TDTK_FW_DEBUG")) : 0u;
          // (by size only when the level has more subtrees than the chip has room for at once: below that the launches would
          //  run one after the other where one launch runs them side by side -- the dat/ scan 0.97 -> 1.13 ms)
          const bool staged = tmax > 1024u;
          if (wave_env && staged) { hipLaunchKernelGGL(k_fin_wave, dim3(cdiv(tmax, FW_WAVES)), dim3(WAVE * FW_WAVES), 0, s, fa); fa.n_lo = FW_CAP; }
          if (half_env && staged && fin_maxn > fa.n_lo) { hipLaunchKernelGGL(k_fin_subtrees_half, dim3(tmax), dim3(FIN_T), 0, s, fa); fa.n_lo = FIN_LDS_HALF; }
          fa.skip_big = 0u;
          if (fin_maxn > fa.n_lo) hipLaunchKernelGGL(k_fin_subtrees, dim3(tmax), dim3(FIN_T), 0, s, fa);
        }
        uint32_t* ftot = (uint32_t*)(arena + O[40]);
        hipLaunchKernelGGL(k_fin_offsets, dim3(FIN_LV + 1u), dim3(FO_T), 0, s, ftab, foff, lvl + fin_level, ftot, (const uint32_t*)(arena + O[41]),
                           (const uint32_t*)(arena + O[42]), small + 1);
        hipLaunchKernelGGL(k_fin_place, dim3(tmax), dim3(256), 0, s, ftab, foff, ftot, roots, lvl + fin_level, (const KdNode*)(arena + O[32]),
                           (const double*)(arena + O[33]), (const LeafEntry*)(arena + O[34]), nodes, node_r, leaf_tab, small + 0);
        level = fin_level + FIN_LV + 2u;
        // what follows the levels does not wait for the host's look: the check of the speculated cuts and the point array
        // are enqueued now, their results come back with the level counters
        // (the point array first: it does not need the exact sums, and at a million points the build is waiting for the root's)
        hipLaunchKernelGGL(k_points, dim3(cdiv(M, 256)), dim3(256), 0, s, perm, cx, cy, cz, M, pts);
        if (spec && SP.n) {
          BCHK(hipEventRecord(side->e2, side->s2));
          BCHK(hipStreamWaitEvent(s, side->e2, 0));
          if (side->s3) { BCHK(hipEventRecord(side->e3, side->s3)); BCHK(hipStreamWaitEvent(s, side->e3, 0)); }
          if (side->s4) { BCHK(hipEventRecord(side->e4, side->s4)); BCHK(hipStreamWaitEvent(s, side->e4, 0)); }
          hipLaunchKernelGGL(k_spec_count, dim3(cdiv(M, 1024)), dim3(256), 0, s, SP, lvl, cx, cy, cz, M);
          hipLaunchKernelGGL(k_spec_patch, dim3(4), dim3(256), 0, s, SP, lvl, nodes, small + 3);
        }
        tail_enqueued = true;
        BCHK(hipMemcpyAsync(hl, lvl, sizeof(BLevel) * (level + 1), hipMemcpyDeviceToHost, s));
        BCHK(hipMemcpyAsync(h_small, small, 32, hipMemcpyDeviceToHost, s));
        BCHK(hipStreamSynchronize(s));
        if (spec && SP.n) h_spec_err = h_small[3];
        if (fin_trace) {
          uint32_t ph[10];
          if (hipMemcpy(ph, small + 16, sizeof ph, hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "FIN_TRACE subtree 0 (100 MHz ticks -> us):");
            const char* nm[10] = {"", "measure (wave: measure+emit)", "decide (wave: scan)", "rank (wave: children+swap)", "emit+count (wave: write-back)", "mark+children", "scan", "swaplist", "swap+relabel", ""};
            for (int k = 1; k < 9; k++) fprintf(stderr, " %s %.1f", nm[k], ph[k] / 100.0);
            fprintf(stderr, "\n");
          }
        }
        if (h_small[2] & 0x20000u) {       // a subtree deeper than the tables: level by level, then
          if (spec_on) { (void)hipStreamSynchronize(side->s2); if (side->s3) (void)hipStreamSynchronize(side->s3); if (side->s4) (void)hipStreamSynchronize(side->s4); }
          if (pts) pool_free(pts);
          for (hipEvent_t x : lvl_ev) (void)hipEventDestroy(x);
          return device_build_tree(d_xyz, M_, bucket, arena_, s, side, 1);
        }
        if (h_small[2]) {
          res.err = hipErrorInvalidValue; res.degenerate = true;
          spec_suspect = true;
          goto fail;
        }
        break;
      }
      for (uint32_t b = 0; b < batch && level < BUILD_MAX_LEVELS; b++, level++) {
        size_t bound = (size_t)known << ((level - known_at) < 31 ? (level - known_at) : 31);
        if (bound > M) bound = M;
        // the nodes of a level are the children of the level above's internal nodes: disjoint runs of more than
        // `bucket` points each, two children per run (at 1M points and buckets of 20 the doubling bound reaches the
        // point count at level 16, sixty times the nodes there are)
        const size_t cap = 2 * (M_ / ((size_t)(bucket > 0 ? bucket : 0) + 1));
        if (level > 0 && bound > cap) bound = cap;
        if (bound < 1) bound = 1;
        const BLevel* lv = lvl + level;
        if (lvl_trace) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); lvl_ev.push_back(e); } }
        // the piecewise path runs while a balanced node is at least half its threshold (below that level the chain in
        // k_measure takes every node, whatever its size: an empty pass of the piecewise kernels costs 75 us)
        const bool big_level = use_big && ((M_ >> level) >= big_level_min);
        // a wave per node while the nodes of a balanced tree hold at most 128 points (three chains in one wave issue 24
        // cycles per point where three waves need 10: level 7 of a 1M-point tree 253 us against 46, level 12 equal,
        // level 16 74 against 139)
        const bool measure_per_axis = measure_per_axis_always || (M_ >> level) > 128;
        bool spec_this = big_level && spec && SP.n < spec_levels && SP.n < BIG_SPEC_MAX;
        // a speculated level's chains of small nodes ride in the launch of its partial sums (k_level_front, below)
        static const bool merge_env = [] { const char* e = lab_env("TDTK_BUILD_MERGE"); return !(e && e[0] == '0'); }();
        const bool front_merged = spec_this && measure_per_axis && merge_env;
        bool nodes_done = false;
        if (front_merged) {}
        else if (measure_per_axis)
          hipLaunchKernelGGL(k_measure, dim3(cdiv(bound * 3 * WAVE, 256)), dim3(256), 0, s, segs, lv, cx, cy, cz, meas,
                             big_level ? big_min : 0xFFFFFFFFu);
        else
          hipLaunchKernelGGL(k_measure_node, dim3(cdiv(bound * WAVE, 256)), dim3(256), 0, s, segs, lv, cx, cy, cz, meas,
                             big_level ? big_min : 0xFFFFFFFFu);
        if (spec_this) {
          // this level from the plain sums; its exact sums on the second stream, from a snapshot of the coordinates
          BSpecLevel& L = SP.L[SP.n];
          const size_t* o = SO + (size_t)SP.n * SPEC_SLOTS;
          L.pieces = (BPiece*)(arena + o[0]); L.preout = (BPre*)(arena + o[1]); L.own = (BSum*)(arena + o[2]);
          L.comp = (BSum*)(arena + o[3]); L.wlist = (uint32_t*)(arena + o[4]); L.snap = (double*)(arena + o[5]);
          L.segs = (BSeg*)(arena + o[6]); L.axis = (uint32_t*)(arena + o[7]); L.node = (uint32_t*)(arena + o[8]);
          L.nleft = (uint32_t*)(arena + o[9]); L.cnt = (uint32_t*)(arena + o[10]); L.exact = (BMeas*)(arena + o[11]);
          L.level = level; L.pad = 0;
          SP.n++;
          const size_t nsl = (size_t)nblocks * 2 * 3, nsl1 = (size_t)nblocks * 2;
          L.prein = (BPre*)(arena + o[12]); L.seg_of = (uint32_t*)(arena + o[13]);
          BPart* part = (BPart*)(arena + SO[(size_t)BIG_SPEC_MAX * SPEC_SLOTS + 1]);
          double *sx = L.snap, *sy = L.snap + n1, *sz = L.snap + 2 * n1;
          if (front_merged) {
            const uint32_t nbp = cdiv(M, BIG_PB);
            hipLaunchKernelGGL(k_level_front, dim3(nbp + cdiv(bound * 3 * WAVE, 256)), dim3(256), 0, s, segs, lv, seg_of, cx, cy, cz, M,
                               part, L.snap, L.seg_of, (uint32_t)n1, nbp, meas, big_min);
          } else {
            hipLaunchKernelGGL(k_big_partials, dim3(cdiv(M, BIG_PB)), dim3(256), 0, s, segs, seg_of, cx, cy, cz, M, part, big_min);
          }
          if (merge_env && bound <= 16) {
            // at most sixteen nodes: their plain sums, the decisions and the records in one workgroup
            hipLaunchKernelGGL(k_nodes_small_approx, dim3(1), dim3(1024), 0, s, segs, lvl + level, (uint32_t)bound, meas, (uint32_t)bucket,
                               kind, axis, splitval, nleft, irank, nodes, node_r, leaf_tab, small + 0, small + 1, part, L, spec_fault, big_min);
            nodes_done = true;
          } else {
            hipLaunchKernelGGL(k_big_approx, dim3(cdiv(bound * WAVE, 256)), dim3(256), 0, s, segs, lv, part, meas, L, spec_fault, big_min);
          }
          if (!front_merged) hipLaunchKernelGGL(k_spec_snapshot, dim3(cdiv(M, 256)), dim3(256), 0, s, seg_of, cx, cy, cz, M, (uint32_t)n1, L.snap, L.seg_of);
          // the exact chain of this level, all of it, on the snapshot.  The chains of different levels do not depend on each
          // other: the root's (1.3 ms of one wave) runs on one background stream, every other level's on the second -- in a
          // single stream the levels' chains queue up behind the root's and together outlast the build
          // (round 6: ... and the second stream was then the build's critical path at 1M points -- the chains of levels 1 and 2,
          // 0.6 and 0.4 ms, and five more levels' 0.12 ms each in a row outlast levels + finisher by 0.8 ms: the levels below
          // the root alternate between two streams)
          const bool use3 = side->s3 && (SP.n - 1) != 0;
          const bool use4 = use3 && side->s4 && ((SP.n - 1) & 1) == 0;
          hipStream_t sb = use4 ? side->s4 : (use3 ? side->s3 : side->s2);
          void* tmpb = use4 ? tmp4 : (use3 ? tmp3 : tmp2);
          BCHK(hipEventRecord(side->e1, s));
          BCHK(hipStreamWaitEvent(sb, side->e1, 0));
          // from the sixth level on (a balanced node's chain: M / 32 adds of 4.2 ns, a seventh of what the build takes) the
          // exact sums are plain chains (TDTK_BUILD_CHAINFROM, lab: another level; 99: never)
          if (level >= chain_from && chain_ok) {
            hipLaunchKernelGGL(k_chain_exact, dim3(cdiv(bound * 3 * WAVE, 256)), dim3(256), 0, sb, L.segs, lv, sx, sy, sz, L.exact, L.axis, big_min);
          } else {
          hipLaunchKernelGGL(k_big_stats, dim3(cdiv(nblocks, 256 / WAVE)), dim3(256), 0, sb, L.segs, L.seg_of, sx, sy, sz, M,
                             nblocks, L.pieces, L.prein, big_min, (const uint32_t*)L.axis);
          size_t stb = scan_tmp;
          BCHK(rocprim::inclusive_scan(tmpb, stb, L.prein, L.preout, nsl, BPreOp(), sb));
          hipLaunchKernelGGL(k_big_emulate, dim3(cdiv(nsl1, 64)), dim3(64), 0, sb, L.segs, sx, sy, sz, nblocks,
                             L.pieces, L.preout, L.own, big_dbg, (const uint32_t*)L.axis);
          stb = scan_tmp;
          BSum ident;
          ident.T0 = ident.T1 = ident.mn0 = ident.mn1 = ident.mx0 = ident.mx1 = 0; ident.eb = BIG_ANY; ident.reset = 0u; ident.cnt = 0u; ident.pad = 0u;
          BCHK(rocprim::exclusive_scan(tmpb, stb, L.own, L.comp, ident, nsl1, BSumOp(), sb));
          hipLaunchKernelGGL(k_big_list, dim3(cdiv(nsl1, 256)), dim3(256), 0, sb, L.own, L.comp, (uint32_t)nsl1, L.wlist);
          hipLaunchKernelGGL(k_big_stitch, dim3(cdiv(bound * 3 * WAVE, 256)), dim3(256), 0, sb, L.segs, lv, sx, sy, sz,
                             nblocks, L.pieces, L.preout, L.own, L.comp, L.wlist, L.exact, big_dbg, big_min, (const uint32_t*)L.axis);
          }
        } else if (big_level) {
          const size_t nsl = (size_t)nblocks * 2 * 3;
          hipLaunchKernelGGL(k_big_stats, dim3(cdiv(nblocks, 256 / WAVE)), dim3(256), 0, s, segs, seg_of, cx, cy, cz, M,
                             nblocks, pieces, prein, big_min);
          size_t stb = scan_tmp;
          BCHK(rocprim::inclusive_scan(tmp, stb, prein, preout, nsl, BPreOp(), s));
          const size_t nsl1 = (size_t)nblocks * 2;
          hipLaunchKernelGGL(k_big_emulate, dim3(cdiv(nsl1, 64)), dim3(64), 0, s, segs, cx, cy, cz, nblocks, pieces, preout,
                             own, big_dbg);
          stb = scan_tmp;
          BSum ident;   // exclusive: comp[slot] = everything since the last reset BEFORE the slot = the run in front of it
          ident.T0 = ident.T1 = ident.mn0 = ident.mn1 = ident.mx0 = ident.mx1 = 0; ident.eb = BIG_ANY; ident.reset = 0u; ident.cnt = 0u; ident.pad = 0u;
          BCHK(rocprim::exclusive_scan(tmp, stb, own, comp, ident, nsl1, BSumOp(), s));
          hipLaunchKernelGGL(k_big_list, dim3(cdiv(nsl1, 256)), dim3(256), 0, s, own, comp, (uint32_t)nsl1, wlist);
          hipLaunchKernelGGL(k_big_stitch, dim3(cdiv(bound * 3 * WAVE, 256)), dim3(256), 0, s, segs, lv, cx, cy, cz, nblocks,
                             pieces, preout, own, comp, wlist, meas, big_dbg, big_min);
          if (big_dbg_all & 8) {
            BMeas* meas2 = (BMeas*)(arena + O[30]);
            hipLaunchKernelGGL(k_measure, dim3(cdiv(bound * 3 * WAVE, 256)), dim3(256), 0, s, segs, lv, cx, cy, cz, meas2, 0xFFFFFFFFu);
            hipLaunchKernelGGL(k_big_dbg_compare, dim3(cdiv(bound, 256)), dim3(256), 0, s, segs, lv, meas, meas2, level, big_min);
          }
        }
        size_t st = scan_tmp;
        if (nodes_done) {}
        else if (bound + 1 <= 1024) {
          hipLaunchKernelGGL(k_nodes_small, dim3(1), dim3(1024), 0, s, segs, lvl + level, (uint32_t)bound, meas, (uint32_t)bucket,
                             kind, axis, splitval, nleft, irank, nodes, node_r, leaf_tab, small + 0, small + 1);
        } else {
          hipLaunchKernelGGL(k_decide, dim3(cdiv(bound + 1, 256)), dim3(256), 0, s, segs, lv, (uint32_t)bound, meas,
                             (uint32_t)bucket, kind, axis, splitval, nleft);
          BCHK(rocprim::exclusive_scan(tmp, st, kind, irank, 0u, bound + 1, rocprim::plus<uint32_t>(), s));
          hipLaunchKernelGGL(k_emit, dim3(cdiv(bound, 256)), dim3(256), 0, s, segs, lvl + level, meas, kind, axis, splitval,
                             irank, nodes, node_r, leaf_tab, small + 0, small + 1);
        }
        // the partition pass (with no internal node at this level it moves nothing): two passes over the points while the
        // one-launch scan's conditions hold (TDTK_BUILD_PART=0, lab: round 3's five)
        const bool part2_env = [] { const char* e = lab_env("TDTK_BUILD_PART"); return !(e && e[0] == '0'); }();
        if (part2_env && own_scan && level < 255u) {
          const uint32_t ntiles = cdiv(M, PS_TILE), nbw = cdiv(M, 256u * PW_ROWS);
          uint32_t* counter = reinterpret_cast<uint32_t*>(arena + o_scanstate);
          unsigned long long* status = reinterpret_cast<unsigned long long*>(arena + o_scanstate + 64);
          hipLaunchKernelGGL(k_part_scan, dim3(ntiles), dim3(PS_THREADS), 0, s, seg_of, kind, segs, axis, splitval, cx, cy, cz, M, nleft, posL,
                             status, counter, level + 1u, ntiles, small + 2);
          if (spec_this)
            hipLaunchKernelGGL(k_part_swap_keep, dim3(nbw + cdiv(bound, 256)), dim3(256), 0, s, posL, segs, kind, irank, nleft, M, seg_of,
                               perm, cx, cy, cz, nbw, lv, next, small + 2, SP.L[SP.n - 1]);
          else
            hipLaunchKernelGGL(k_part_swap, dim3(nbw + cdiv(bound, 256)), dim3(256), 0, s, posL, segs, kind, irank, nleft, M, seg_of,
                               perm, cx, cy, cz, nbw, lv, next, small + 2);
          BSeg* t = segs; segs = next; next = t;
          continue;
        }
        hipLaunchKernelGGL(k_count, dim3(cdiv(M, 256 * CNT_ITERS)), dim3(256), 0, s, seg_of, kind, axis, splitval, cx, cy,
                           cz, M, nleft);
        {
          const uint32_t nbm = cdiv(n1, 256);
          if (spec_this)
            hipLaunchKernelGGL(k_misplaced_children_keep, dim3(nbm + cdiv(bound, 256)), dim3(256), 0, s, seg_of, kind, segs, axis,
                               splitval, nleft, cx, cy, cz, M, LR, nbm, lv, irank, next, small + 2, SP.L[SP.n - 1]);
          else
            hipLaunchKernelGGL(k_misplaced_children, dim3(nbm + cdiv(bound, 256)), dim3(256), 0, s, seg_of, kind, segs, axis,
                               splitval, nleft, cx, cy, cz, M, LR, nbm, lv, irank, next, small + 2);
        }
        if (own_scan && level < 255u) {
          BCHK(launch_scan_pair27(LR, AB, n1, arena + o_scanstate, level + 1u, small + 2, s));    // one launch (sort.hip)
        } else {
          st = scan_tmp;
          BCHK(rocprim::exclusive_scan(tmp, st, LR, AB, 0ull, n1, rocprim::plus<unsigned long long>(), s));
        }
        hipLaunchKernelGGL(k_swaplist, dim3(cdiv(M, 256)), dim3(256), 0, s, seg_of, segs, LR, AB, M, posL, posR);
        {
          const uint32_t nbs = cdiv((size_t)M / 2 + 1, 256);
          hipLaunchKernelGGL(k_swap_relabel, dim3(nbs + cdiv(M, 256)), dim3(256), 0, s, posL, posR, AB + M, perm, cx, cy, cz, nbs,
                             segs, kind, irank, nleft, M, seg_of);
        }
        BSeg* t = segs; segs = next; next = t;
      }
      // one look per batch: the level counters so far and the root reference / largest bucket / error word together -- and,
      // when the next level is the one handed to the finisher, its largest node
      if (level == fin_level || fin_retry) { hipLaunchKernelGGL(k_fin_maxn, dim3(1), dim3(256), 0, s, segs, lvl + level, small + 4); have_max = true; }
      else if (chain_look) hipLaunchKernelGGL(k_fin_maxn, dim3(1), dim3(256), 0, s, segs, lvl + level, small + 4);
      BCHK(hipMemcpyAsync(hl, lvl, sizeof(BLevel) * (level + 1), hipMemcpyDeviceToHost, s));
      BCHK(hipMemcpyAsync(h_small, small, 32, hipMemcpyDeviceToHost, s));
      BCHK(hipStreamSynchronize(s));
      const BLevel nx = hl[level];
      if (h_small[2] || (nx.nseg && level >= BUILD_MAX_LEVELS)) {
        // an empty child / non-finite coordinates -- or, on a speculated level, a cut so far off the exact one that a
        // child came out empty: only then is the in-order build asked what the input really is
        res.err = hipErrorInvalidValue; res.degenerate = true;
        spec_suspect = (h_small[2] & 0x10000u) == 0u;      // (bit 16: the one-launch scan gave up waiting, see below)
        if (h_small[2] & 0x10000u) { res.err = hipErrorLaunchTimeOut; res.degenerate = false; }
        goto fail;
      }
      if (nx.nseg == 0) break;
      known = nx.nseg; known_at = level;
      batch = 2;
      if (chain_look) {
        chain_ok = h_small[4] <= M_ / 17u;
        if (chain_ok || level >= (uint32_t)spec_levels || level >= fin_level) chain_look = false;
        batch = chain_look ? 1u : ((fin_level != 0xFFFFFFFFu && level < fin_level) ? fin_level - level : 2u);
      }
      if (fin_retry) {
        if (h_small[4] <= FIN_LDS && h_small[5] <= fin_cap && level + FIN_LV + 3u < BUILD_MAX_LEVELS) { fin_level = level; fin_retry = false; }   // hand over now (the block at the loop's top)
        else have_max = false;
      }
    }
    if (lvl_trace && !lvl_ev.empty()) {
      hipEvent_t e;
      if (hipEventCreate(&e) == hipSuccess) { (void)hipEventRecord(e, s); (void)hipEventSynchronize(e); lvl_ev.push_back(e); }
      fprintf(stderr, "BUILD_TRACE M=%u:", M);
      for (size_t k = 0; k + 1 < lvl_ev.size(); k++) { float ms = 0; (void)hipEventElapsedTime(&ms, lvl_ev[k], lvl_ev[k + 1]); fprintf(stderr, " L%zu %.0f", k, ms * 1e3f); }
      float tot = 0; (void)hipEventElapsedTime(&tot, lvl_ev.front(), lvl_ev.back());
      fprintf(stderr, " | levels total %.0f us\n", tot * 1e3f);
      for (hipEvent_t x : lvl_ev) (void)hipEventDestroy(x);
    }
    {
      // the first level without nodes: its counters are the totals, its index the depth of the tree
      depth = 0;
      while (depth < level && hl[depth].nseg) depth++;
      node_count = hl[depth].node_base; leaf_count = hl[depth].leaf_base;
    }
    res.max_leaf = h_small[1];
    res.cb = bits_for(res.max_leaf);
    res.table_mode = (bits_for(M) + res.cb) > 30;
    if (spec && SP.n && !tail_enqueued) {
      // the exact sums have to be there now: check every big node's cut against them, give its record the exact value
      BCHK(hipEventRecord(side->e2, side->s2));
      BCHK(hipStreamWaitEvent(s, side->e2, 0));
      if (side->s3) { BCHK(hipEventRecord(side->e3, side->s3)); BCHK(hipStreamWaitEvent(s, side->e3, 0)); }
          if (side->s4) { BCHK(hipEventRecord(side->e4, side->s4)); BCHK(hipStreamWaitEvent(s, side->e4, 0)); }
      hipLaunchKernelGGL(k_spec_count, dim3(cdiv(M, 1024)), dim3(256), 0, s, SP, lvl, cx, cy, cz, M);
      hipLaunchKernelGGL(k_spec_patch, dim3(4), dim3(256), 0, s, SP, lvl, nodes, small + 3);
      BCHK(hipMemcpyAsync(&h_spec_err, small + 3, 4, hipMemcpyDeviceToHost, s));
    }
    if (!tail_enqueued) hipLaunchKernelGGL(k_points, dim3(cdiv(M, 256)), dim3(256), 0, s, perm, cx, cy, cz, M, pts);
    if (!res.table_mode)
      hipLaunchKernelGGL(k_pack_refs, dim3(cdiv(node_count ? node_count : 1, 256)), dim3(256), 0, s, nodes, node_count,
                         leaf_tab, (uint32_t)res.cb, small + 0);
    // (the finisher's path has looked at everything already -- the check word, and a root that is a node keeps its reference
    // through k_pack_refs: what is enqueued from here on is waited for by the caller's own synchronisation, tree_finish)
    const bool no_last_look = tail_enqueued && !res.table_mode && !(h_small[0] & REF_LEAF);
    if (no_last_look) res.root_ref = h_small[0];
    else BCHK(hipMemcpyAsync(&res.root_ref, small + 0, 4, hipMemcpyDeviceToHost, s));
    // the records move out of the scratch into allocations of their exact size
    if (node_count) {
      BCHK((hipError_t)pool_malloc_raw((void**)&f_nodes, sizeof(KdNode) * (size_t)node_count));
      BCHK((hipError_t)pool_malloc_raw((void**)&f_r, sizeof(double) * (size_t)node_count));
      BCHK(hipMemcpyAsync(f_nodes, nodes, sizeof(KdNode) * (size_t)node_count, hipMemcpyDeviceToDevice, s));
      BCHK(hipMemcpyAsync(f_r, node_r, sizeof(double) * (size_t)node_count, hipMemcpyDeviceToDevice, s));
    }
    if (res.table_mode) {
      BCHK((hipError_t)pool_malloc_raw((void**)&f_leaf, sizeof(LeafEntry) * (size_t)leaf_count));
      BCHK(hipMemcpyAsync(f_leaf, leaf_tab, sizeof(LeafEntry) * (size_t)leaf_count, hipMemcpyDeviceToDevice, s));
    }
    if (!no_last_look) BCHK(hipStreamSynchronize(s));
    BCHK(hipGetLastError());
    if (kLab && lab_env("TDTK_SPEC_DUMP")) {
      (void)hipStreamSynchronize(s);
      fprintf(stderr, "SPEC_DUMP M=%u spec %d SP.n %d tail_enqueued %d fin_level %u level %u h_spec_err %u depth %u\n", M, (int)spec, (int)SP.n, (int)tail_enqueued,
              fin_level, level, h_spec_err, depth);
      for (int l = 0; l < (int)SP.n; l++) {
        const BSpecLevel& L = SP.L[l];
        uint32_t nd[4], nlf[4], cn[4], axs[4]; BMeas ex[4]; BSeg sg[4];
        (void)hipMemcpy(nd, L.node, sizeof nd, hipMemcpyDeviceToHost); (void)hipMemcpy(nlf, L.nleft, sizeof nlf, hipMemcpyDeviceToHost);
        (void)hipMemcpy(cn, L.cnt, sizeof cn, hipMemcpyDeviceToHost); (void)hipMemcpy(axs, L.axis, sizeof axs, hipMemcpyDeviceToHost);
        (void)hipMemcpy(ex, L.exact, sizeof ex, hipMemcpyDeviceToHost); (void)hipMemcpy(sg, L.segs, sizeof sg, hipMemcpyDeviceToHost);
        for (int k = 0; k < 4; k++)
          fprintf(stderr, "  spec level %u entry %d: node %08x seg(start %u n %u) axis %u nleft %u cnt %u exact mean (%.17g %.17g %.17g)\n", L.level, k, nd[k],
                  sg[k].start, sg[k].n, axs[k], nlf[k], cn[k], ex[k].mean[0], ex[k].mean[1], ex[k].mean[2]);
      }
    }
    if (h_spec_err) { res.err = hipErrorNotReady; spec_suspect = true; goto fail; }     // a cut the exact sum would have made elsewhere: in order, then
  }
#ifdef TDTK_LAB
  if (guards.n) {
    uint32_t* bad = (uint32_t*)(arena + O[19]) + 40;       // (a word of `small` nothing else uses)
    uint32_t h_bad = 0;
    if (spec_on) { (void)hipStreamSynchronize(side->s2); if (side->s3) (void)hipStreamSynchronize(side->s3); if (side->s4) (void)hipStreamSynchronize(side->s4); }
    (void)hipMemsetAsync(bad, 0, 4, s);
    hipLaunchKernelGGL(k_guard_check, dim3(guards.n), dim3(64), 0, s, arena, guards, bad);
    (void)hipMemcpyAsync(&h_bad, bad, 4, hipMemcpyDeviceToHost, s);
    (void)hipStreamSynchronize(s);
    if (h_bad) {
      fprintf(stderr, "tree build: arena guard %u of %u overwritten (M = %u): a region was written past its end\n", h_bad - 1u, guards.n, M);
      res.err = hipErrorAssert;
      goto fail;
    }
  }
#endif
  res.nodes = f_nodes; res.node_r = f_r; res.leaf_tab = f_leaf; res.pts = pts;
  res.n_internal = node_count; res.n_leaves = leaf_count; res.max_depth = depth;
  return res;
fail:
  if (spec_on) { (void)hipStreamSynchronize(side->s2); if (side->s3) (void)hipStreamSynchronize(side->s3); if (side->s4) (void)hipStreamSynchronize(side->s4); }   // nothing of this build may still be running in the arena
  if (f_nodes) pool_free(f_nodes);
  if (f_r) pool_free(f_r);
  if (f_leaf) pool_free(f_leaf);
  if (pts) pool_free(pts);
  if (spec_on && spec_suspect) {
    // a failed check, or a degenerate split on a tree cut at the plain sums: the in-order build decides what the input
    // really is.  (Anything else -- no memory, a failed launch -- is returned as it is: building twice would not help.)
    (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
    DevBuildResult again = device_build_tree(d_xyz, M_, bucket, arena_, s, nullptr, no_finish);
    again.respeculated = again.err == hipSuccess;     // tdtk_build_respeculated counts cuts that really were off, not bad inputs
    return again;
  }
  if (res.err == hipSuccess) res.err = hipErrorUnknown;
  return res;
}

// the per-level buffers of the speculative build behind the main layout: SO[SPEC_SLOTS l + k] for level l,
// SO[SPEC_SLOTS BIG_SPEC_MAX] the second stream's scan temporary, + 1 the block partials; returns the total size.  The
// chain is long on the top levels only (two-sided coordinates), and a level's snapshot is 28 bytes per point: the
// eight top levels are speculated, big levels below them (clouds of more than 2M points) go the in-order way.
#define BIG_SPEC_LEVELS 8
static size_t spec_layout(size_t M, size_t base, size_t* SO, int* nlev_out)
{
  const uint32_t BIG_MIN_M = build_big_min(M);
  size_t off = (base + 255) & ~(size_t)255;
  int nlev = 0;
  if (M >= BIG_MIN_M)
    for (uint32_t level = 0; level < BUILD_MAX_LEVELS && (M >> level) >= build_big_level_min(M) && nlev < BIG_SPEC_LEVELS; level++) nlev++;
  if (getenv("TDTK_BUILD_SPEC") && getenv("TDTK_BUILD_SPEC")[0] == '0') nlev = 0;
  const size_t n1 = M + 1, nsl = 6 * (M / BIG_CH + 2), nsl1 = nsl / 3 + 3;
  auto take = [&](size_t bytes, size_t* slot) { if (slot) *slot = off; off += (bytes + 255) & ~(size_t)255; arena_guard(off); };
  for (int l = 0; l < nlev; l++) {
    size_t maxseg = (l < 40) ? ((size_t)1 << l) : n1;
    if (maxseg > n1) maxseg = n1;
    size_t* o = SO ? SO + (size_t)l * SPEC_SLOTS : nullptr;
    take(sizeof(BPiece) * nsl, o ? o + 0 : nullptr);
    take(sizeof(BPre) * nsl, o ? o + 1 : nullptr);
    take(sizeof(BSum) * nsl1, o ? o + 2 : nullptr);
    take(sizeof(BSum) * nsl1, o ? o + 3 : nullptr);
    take(4 * (nsl1 + 1), o ? o + 4 : nullptr);
    take(8 * 3 * n1, o ? o + 5 : nullptr);
    take(sizeof(BSeg) * maxseg, o ? o + 6 : nullptr);
    take(4 * maxseg, o ? o + 7 : nullptr);
    take(4 * maxseg, o ? o + 8 : nullptr);
    take(4 * maxseg, o ? o + 9 : nullptr);
    take(4 * maxseg, o ? o + 10 : nullptr);
    take(sizeof(BMeas) * maxseg, o ? o + 11 : nullptr);
    take(sizeof(BPre) * nsl, o ? o + 12 : nullptr);
    take(4 * n1, o ? o + 13 : nullptr);
  }
  size_t scan_tmp = 0;
  if (nlev) {
    BSum* zs = nullptr;
    (void)rocprim::exclusive_scan(nullptr, scan_tmp, zs, zs, BSum(), nsl, BSumOp(), (hipStream_t)0);
    BPre* zp = nullptr; size_t tp = 0;
    (void)rocprim::inclusive_scan(nullptr, tp, zp, zp, nsl, BPreOp(), (hipStream_t)0);
    if (tp > scan_tmp) scan_tmp = tp;
  }
  take(scan_tmp + 256, SO ? SO + (size_t)BIG_SPEC_MAX * SPEC_SLOTS : nullptr);
  take(sizeof(BPart) * 2 * (M / BIG_PB + 2), SO ? SO + (size_t)BIG_SPEC_MAX * SPEC_SLOTS + 1 : nullptr);
  take(scan_tmp + 256, SO ? SO + (size_t)BIG_SPEC_MAX * SPEC_SLOTS + 2 : nullptr);      // the third stream's scan temporary
  take(scan_tmp + 256, SO ? SO + (size_t)BIG_SPEC_MAX * SPEC_SLOTS + 3 : nullptr);      // the fourth's
  if (nlev_out) *nlev_out = nlev;
  return off;
}

static size_t build_layout(size_t M, size_t* O, size_t* scan_tmp_out)
{
  size_t scan_tmp = 0;
  {
    uint32_t* z = nullptr;
    (void)rocprim::exclusive_scan(nullptr, scan_tmp, z, z, 0u, M + 1, rocprim::plus<uint32_t>(), (hipStream_t)0);
    unsigned long long* z8 = nullptr;
    size_t t8 = 0;
    (void)rocprim::exclusive_scan(nullptr, t8, z8, z8, 0ull, M + 1, rocprim::plus<unsigned long long>(), (hipStream_t)0);
    if (t8 > scan_tmp) scan_tmp = t8;
    const size_t nsl = 6 * (M / BIG_CH + 2);
    BPre* zp = nullptr; size_t tp = 0;
    (void)rocprim::inclusive_scan(nullptr, tp, zp, zp, nsl, BPreOp(), (hipStream_t)0);
    if (tp > scan_tmp) scan_tmp = tp;
    BSum* zs = nullptr; size_t ts = 0;
    (void)rocprim::exclusive_scan(nullptr, ts, zs, zs, BSum(), nsl, BSumOp(), (hipStream_t)0);
    if (ts > scan_tmp) scan_tmp = ts;
  }
  const size_t n1 = M + 1;
  size_t off = 0;
  int k = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; arena_guard(off); if (O) O[k] = o; k++; return o; };
  take(4 * n1); take(4 * n1); take(8 * n1); take(8 * n1); take(8 * n1);      // perm segof cx cy cz
  take(4 * n1); take(4 * n1); take(8 * n1); take(8 * n1);                    // f F LR AB
  take(4 * n1); take(4 * n1);                                                // posL posR
  take(sizeof(BSeg) * n1); take(sizeof(BSeg) * n1); take(sizeof(BMeas) * n1);
  take(4 * n1); take(4 * n1); take(8 * n1); take(4 * n1);                    // kind axis split irank
  take(scan_tmp + 256); take(256);                                          // tmp small
  take(sizeof(KdNode) * n1); take(sizeof(double) * n1); take(sizeof(LeafEntry) * n1);   // nodes node_r leaf_tab
  take(sizeof(BLevel) * (BUILD_MAX_LEVELS + 2));                            // 23 per-level counters
  const size_t nsl = 6 * (M / BIG_CH + 2);                                  // pieces of big nodes: 3 axes x 2 per block
  take(sizeof(BPiece) * nsl);                                               // 24
  take(sizeof(BPre) * nsl); take(sizeof(BPre) * nsl);                       // 25 26 plain prefix in / out
  take(sizeof(BSum) * nsl); take(sizeof(BSum) * nsl);                       // 27 28 piece summaries, folded runs
  take(4 * (nsl + 1));                                                      // 29 slots of the walked pieces, in run order
  take(lab_env("TDTK_BIG_DEBUG") && (atoi(lab_env("TDTK_BIG_DEBUG")) & 8) ? sizeof(BMeas) * n1 : 256);   // 30 debug: the chain's results
  take(std::max(scan_pair27_state_bytes(n1), part_state_bytes(n1)));        // 31 state of the one-launch scans (sort.hip's, k_part_scan's)
  // subtrees finished by one workgroup each (k_fin_*): staging of their records, their own kind / rank arrays, tables
  take(sizeof(KdNode) * n1); take(sizeof(double) * n1); take(sizeof(LeafEntry) * n1);   // 32 33 34
  const size_t fcap = fin_max_subtrees(M);
  take(4 * (n1 + fcap)); take(4 * (n1 + fcap));                                        // 35 36 kind, irank
  take(sizeof(BSeg) * fcap); take(sizeof(FinTab) * fcap); take(sizeof(FinOff) * fcap);   // 37 38 39
  take(4 * 2 * (FIN_LV + 2));                                                          // 40 the levels' totals
  take(4 * fcap); take(4 * fcap);                                                      // 41 42 the subtrees' depths, their largest buckets
  if (scan_tmp_out) *scan_tmp_out = scan_tmp;
  return off;
}

}  // namespace tdtk
