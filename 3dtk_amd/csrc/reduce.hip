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
      const bool bx = px > cx, by = py > cy, bz = pz > cz;      // Boctree.h:1353-1355
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

}  // namespace tdtk
