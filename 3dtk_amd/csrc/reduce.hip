// Octree reduction of a scan ("-r <voxelSize>", centre mode) on the device.
//
// The reference builds a pointer octree over the scan (BOctTree constructor,
// include/slam6d/Boctree.h:222-270: cubic root box = bbox centre, half size = largest half extent
// + 1.0), splits a cell into its occupied octants while its half size is > voxelSize
// (branch, :1163-1195; child index = (x > cx) | (y > cy) << 1 | (z > cz) << 2, :1353-1355; child
// centre = parent centre -/+ size/2 per axis, :612-657) and emits the centre of every leaf cell
// in depth-first child order (GetOctTreeCenter, :928-948).  Scan::calcReducedPoints
// (src/slam6d/scan.cc:577-603) stores those centres as "xyz reduced".
//
// Nothing in that result depends on the pointer tree: a point's leaf cell is the sequence of D
// child indices it takes on the way down (D = number of halvings until size <= voxelSize, the same
// for every point), the depth-first order of the leaves is the ascending order of that sequence
// read as a base-8 number, and the leaf centre is a function of the sequence alone.  So:
//   k_oct_keys     one thread per point walks the D levels (same fp64 compares and centre updates)
//                  and writes the 3D-bit key;
//   rocPRIM        radix-sorts the keys (3D bits);
//   k_oct_heads    flags the first key of every run, rocPRIM exclusive scan gives output slots;
//   k_oct_centres  one thread per run head replays the centre chain from the key.
// fp64, no FMA; size/2.0 is exact, centre -/+ size/2 is the same single rounded add the reference does.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"

namespace tdtk {

// ---- bounding box of an AoS point array (min/max are order independent, hence exact) ------
constexpr int BBOX_BLOCKS = 1024;

__device__ __forceinline__ double wave_min(double v)
{
  for (int o = 32; o; o >>= 1) { const double w = __shfl_xor(v, o); v = (w < v) ? w : v; }
  return v;
}
__device__ __forceinline__ double wave_max(double v)
{
  for (int o = 32; o; o >>= 1) { const double w = __shfl_xor(v, o); v = (w > v) ? w : v; }
  return v;
}

// partial[b][6] = lo xyz, hi xyz of the block's grid-stride slice
__global__ __launch_bounds__(256) void k_bbox_partial(const double* __restrict__ xyz, size_t n,
                                                      double* __restrict__ partial)
{
  __shared__ double sh[4][6];
  double lo[3] = {xyz[0], xyz[1], xyz[2]}, hi[3] = {xyz[0], xyz[1], xyz[2]};
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    for (int a = 0; a < 3; a++) {
      const double v = xyz[3 * i + a];
      lo[a] = (v < lo[a]) ? v : lo[a];
      hi[a] = (v > hi[a]) ? v : hi[a];
    }
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  for (int a = 0; a < 3; a++) {
    const double mn = wave_min(lo[a]), mx = wave_max(hi[a]);
    if (l == 0) { sh[w][a] = mn; sh[w][3 + a] = mx; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = sh[0][threadIdx.x];
    for (int k = 1; k < 4; k++) {
      const double u = sh[k][threadIdx.x];
      v = (threadIdx.x < 3) ? ((u < v) ? u : v) : ((u > v) ? u : v);
    }
    partial[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

__global__ __launch_bounds__(64) void k_bbox_final(const double* __restrict__ partial, int nb,
                                                   double* __restrict__ box)
{
  for (int a = 0; a < 6; a++) {
    double v = partial[a];
    for (int b = threadIdx.x; b < nb; b += 64) {
      const double u = partial[(size_t)b * 6 + a];
      v = (a < 3) ? ((u < v) ? u : v) : ((u > v) ? u : v);
    }
    v = (a < 3) ? wave_min(v) : wave_max(v);
    if (threadIdx.x == 0) box[a] = v;
  }
}

size_t bbox_temp_bytes() { return (size_t)BBOX_BLOCKS * 6 * sizeof(double); }

// box[0..2] = min, box[3..5] = max of xyz[n][3] (device memory); n >= 1
hipError_t launch_bbox(const double* d_xyz, size_t n, double* d_partial, double* d_box, hipStream_t s)
{
  size_t nb = (n + 255) / 256;
  if (nb > (size_t)BBOX_BLOCKS) nb = BBOX_BLOCKS;
  hipLaunchKernelGGL(k_bbox_partial, dim3((uint32_t)nb), dim3(256), 0, s, d_xyz, n, d_partial);
  hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(64), 0, s, d_partial, (int)nb, d_box);
  return hipGetLastError();
}

// ---- octree cells --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_oct_keys(const double* __restrict__ xyz, size_t n, OctRoot R,
                                                  uint64_t* __restrict__ keys)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    double cx = R.center[0], cy = R.center[1], cz = R.center[2], size = R.size;
    uint64_t key = 0;
    for (int d = 0; d < R.depth; d++) {
      // the array constructor Scan::calcReducedPoints uses cuts with `p < centre` | `p >= centre` (fullsort / sort,
      // Boctree.h:1737-1816), not with childIndex's strict `>`: a point exactly on a centre plane goes up
      const bool bx = !(px < cx), by = !(py < cy), bz = !(pz < cz);
      key = (key << 3) | (uint64_t)((int)bx | ((int)by << 1) | ((int)bz << 2));
      const double h = size / 2.0;                               // childcenter, Boctree.h:612-657
      cx = bx ? cx + h : cx - h;
      cy = by ? cy + h : cy - h;
      cz = bz ? cz + h : cz - h;
      size = h;
    }
    keys[i] = key;
  }
}

__global__ __launch_bounds__(256) void k_oct_heads(const uint64_t* __restrict__ keys, size_t n,
                                                   uint32_t* __restrict__ flags)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride)
    flags[i] = (i < n && (i == 0 || keys[i] != keys[i - 1])) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_oct_centres(const uint64_t* __restrict__ keys,
                                                     const uint32_t* __restrict__ flags,
                                                     const uint32_t* __restrict__ slot, size_t n, OctRoot R,
                                                     double* __restrict__ out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (!flags[i]) continue;
    const uint64_t key = keys[i];
    double cx = R.center[0], cy = R.center[1], cz = R.center[2], size = R.size;
    for (int d = R.depth - 1; d >= 0; d--) {
      const int ci = (int)((key >> (3 * d)) & 7u);
      const double h = size / 2.0;
      cx = (ci & 1) ? cx + h : cx - h;
      cy = (ci & 2) ? cy + h : cy - h;
      cz = (ci & 4) ? cz + h : cz - h;
      size = h;
    }
    const size_t o = slot[i];
    out[3 * o] = cx; out[3 * o + 1] = cy; out[3 * o + 2] = cz;
  }
}

// ---- the order the reference's in-place partitions leave the points in (random modes, `-O <nrpts>`) -----------------
// BOctTree's array constructor cuts a cell with three two-pointer partitions (z, then y inside each z half, then x inside
// each quarter; Boctree.h:1737-1816), level after level, and a leaf keeps its points in the order these leave behind --
// which is what GetOctTreeRandom indexes with rand().  A two-pointer partition is order-equivalent to "what already
// lies on its side stays, the k-th misplaced element from the left swaps with the k-th misplaced from the right END";
// where the cut falls is known in advance (the number of keys with a 0 at this bit inside the segment = a lower bound in
// the SORTED key array).  So one pass per key bit, all segments of the scan at once: flag the misplaced on either side,
// one 64-bit scan numbers both kinds, pair by number, swap.  3 x depth passes of three small kernels and a scan.
__global__ __launch_bounds__(256) void k_oct_pass_flags(const uint64_t* __restrict__ keys, const uint64_t* __restrict__ sorted,
                                                        const uint32_t* __restrict__ perm, const uint32_t* __restrict__ segS,
                                                        const uint32_t* __restrict__ segE, uint32_t* __restrict__ mid,
                                                        uint64_t* __restrict__ flags, size_t n, int b)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pos <= n; pos += stride) {
    if (pos == n) { flags[n] = 0; continue; }
    const uint64_t key = keys[perm[pos]];
    const uint64_t target = (((key >> (b + 1)) << 1) | 1ull) << b;      // first key of this segment with a 1 at bit b
    uint32_t lo = segS[pos], hi = segE[pos];
    while (lo < hi) {                                                   // lower bound inside the segment
      const uint32_t m = lo + ((hi - lo) >> 1);
      if (sorted[m] < target) lo = m + 1; else hi = m;
    }
    mid[pos] = lo;
    const bool up = (key >> b) & 1ull;
    const uint64_t ml = (pos < lo && up) ? 1ull : 0ull;                 // belongs right, lies in the left part
    const uint64_t mr = (pos >= lo && !up) ? 1ull : 0ull;               // belongs left, lies in the right part
    flags[pos] = ml | (mr << 32);
  }
}
__global__ __launch_bounds__(256) void k_oct_pass_lists(const uint64_t* __restrict__ flags, const uint64_t* __restrict__ P,
                                                        uint32_t* __restrict__ segS, uint32_t* __restrict__ segE,
                                                        const uint32_t* __restrict__ mid, uint32_t* __restrict__ posL,
                                                        uint32_t* __restrict__ posR, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t pos = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pos < n; pos += stride) {
    const uint64_t f = flags[pos];
    const uint32_t s = segS[pos], e = segE[pos], m = mid[pos];
    if (f & 1ull) posL[(uint32_t)P[pos]] = (uint32_t)pos;                           // k-th misplaced from the left, k counted globally
    if (f >> 32) {
      const uint32_t from_right = (uint32_t)(P[e] >> 32) - (uint32_t)(P[pos + 1] >> 32);   // misplaced-right elements behind this one
      posR[(uint32_t)P[s] + from_right] = (uint32_t)pos;                          // its partner: the same number among the segment's left ones
    }
    if (pos < m) segE[pos] = m; else segS[pos] = m;                               // the two halves are the next pass's segments
  }
}
__global__ __launch_bounds__(256) void k_oct_pass_swap(uint32_t* __restrict__ perm, const uint32_t* __restrict__ posL,
                                                       const uint32_t* __restrict__ posR, const uint64_t* __restrict__ P, size_t n)
{
  const uint32_t total = (uint32_t)P[n];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += stride) {
    const uint32_t a = posL[k], c = posR[k];
    const uint32_t t = perm[a]; perm[a] = perm[c]; perm[c] = t;
  }
}
__global__ __launch_bounds__(256) void k_oct_iota(uint32_t* __restrict__ perm, uint32_t* __restrict__ segS, uint32_t* __restrict__ segE, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { perm[i] = (uint32_t)i; segS[i] = 0u; segE[i] = (uint32_t)n; }
}
__global__ __launch_bounds__(256) void k_oct_leaf_starts(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ slot, size_t n,
                                                         uint32_t* __restrict__ starts)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n; i += stride) {
    if (i == n) starts[slot[n]] = (uint32_t)n;
    else if (flags[i]) starts[slot[i]] = (uint32_t)i;
  }
}
__global__ __launch_bounds__(256) void k_oct_gather(const double* __restrict__ xyz, const uint32_t* __restrict__ perm,
                                                    const uint32_t* __restrict__ sel, size_t m, double* __restrict__ out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
    const size_t src = perm[sel[j]];
    out[3 * j] = xyz[3 * src]; out[3 * j + 1] = xyz[3 * src + 1]; out[3 * j + 2] = xyz[3 * src + 2];
  }
}

size_t scan_u64_temp_bytes(size_t n)
{
  size_t tmp = 0;
  uint64_t* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, tmp, p, p, (uint64_t)0, n, rocprim::plus<uint64_t>(), (hipStream_t)0);
  return tmp;
}

size_t oct_sort_temp_bytes(size_t n)
{
  size_t tmp = 0;
  uint64_t* p = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, tmp, p, p, n, 0, 64, (hipStream_t)0);
  return tmp;
}

static inline uint32_t grid_for(size_t n)
{
  size_t nb = (n + 255) / 256;
  return (uint32_t)(nb > 4096 ? 4096 : (nb ? nb : 1));
}

hipError_t launch_oct_keys_sorted(const double* d_xyz, size_t n, const OctRoot& R, uint64_t* keys_a,
                                  uint64_t* keys_b, void* tmp, size_t tmp_bytes, hipStream_t s)
{
  hipLaunchKernelGGL(k_oct_keys, dim3(grid_for(n)), dim3(256), 0, s, d_xyz, n, R, keys_a);
  hipError_t e = rocprim::radix_sort_keys(tmp, tmp_bytes, keys_a, keys_b, n, 0, (unsigned)(3 * R.depth), s);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

hipError_t launch_oct_heads(const uint64_t* keys, size_t n, uint32_t* flags, hipStream_t s)
{
  hipLaunchKernelGGL(k_oct_heads, dim3(grid_for(n + 1)), dim3(256), 0, s, keys, n, flags);
  return hipGetLastError();
}

hipError_t launch_oct_centres(const uint64_t* keys, const uint32_t* flags, const uint32_t* slot, size_t n,
                              const OctRoot& R, double* out, hipStream_t s)
{
  hipLaunchKernelGGL(k_oct_centres, dim3(grid_for(n)), dim3(256), 0, s, keys, flags, slot, n, R, out);
  return hipGetLastError();
}

// perm[pos] = caller index of the point at position pos of the reference's leaf order.  keys: per point (caller order);
// sorted: the same keys sorted; work: 6 n x u32 (perm is separate), flags / P: (n + 1) x u64 each.
hipError_t launch_oct_leaf_order(const uint64_t* keys, const uint64_t* sorted, size_t n, int depth, uint32_t* perm, uint32_t* work,
                                 uint64_t* flags, uint64_t* P, void* tmp, size_t tmp_bytes, hipStream_t s)
{
  uint32_t *segS = work, *segE = work + n, *mid = work + 2 * n, *posL = work + 3 * n, *posR = work + 4 * n;
  hipLaunchKernelGGL(k_oct_iota, dim3(grid_for(n)), dim3(256), 0, s, perm, segS, segE, n);
  for (int b = 3 * depth - 1; b >= 0; b--) {
    hipLaunchKernelGGL(k_oct_pass_flags, dim3(grid_for(n + 1)), dim3(256), 0, s, keys, sorted, perm, segS, segE, mid, flags, n, b);
    hipError_t e = rocprim::exclusive_scan(tmp, tmp_bytes, flags, P, (uint64_t)0, n + 1, rocprim::plus<uint64_t>(), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_oct_pass_lists, dim3(grid_for(n)), dim3(256), 0, s, flags, P, segS, segE, mid, posL, posR, n);
    hipLaunchKernelGGL(k_oct_pass_swap, dim3(grid_for(n)), dim3(256), 0, s, perm, posL, posR, P, n);
  }
  return hipGetLastError();
}
hipError_t launch_oct_leaf_starts(const uint32_t* flags, const uint32_t* slot, size_t n, uint32_t* starts, hipStream_t s)
{
  hipLaunchKernelGGL(k_oct_leaf_starts, dim3(grid_for(n + 1)), dim3(256), 0, s, flags, slot, n, starts);
  return hipGetLastError();
}
hipError_t launch_oct_gather(const double* xyz, const uint32_t* perm, const uint32_t* sel, size_t m, double* out, hipStream_t s)
{
  if (!m) return hipSuccess;
  hipLaunchKernelGGL(k_oct_gather, dim3(grid_for(m)), dim3(256), 0, s, xyz, perm, sel, m, out);
  return hipGetLastError();
}

}  // namespace tdtk
