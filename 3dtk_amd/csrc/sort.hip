// Spatial (Morton) ordering of a data scan on the device: 30-bit keys on the scan's own bounding
// box, stable LSD radix sort of (key, caller index) pairs with rocPRIM, gather into SoA.  Stable +
// ascending input indices = ties broken by caller index, i.e. exactly the order the host sort of
// (code << 32 | index) produced in the first version -- deterministic run to run.  This is a
// once-per-scan utility, not the hot path; it replaced a 100 ms std::sort per 1M points.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "kernels.h"

namespace tdtk {

__device__ __forceinline__ uint32_t spread10(uint32_t v)
{
  v &= 0x3FFu;
  v = (v | (v << 16)) & 0x30000FFu;
  v = (v | (v << 8)) & 0x300F00Fu;
  v = (v | (v << 4)) & 0x30C30C3u;
  v = (v | (v << 2)) & 0x9249249u;
  return v;
}

// box = device-resident (min xyz, max xyz) from launch_bbox
__global__ void k_morton_keys(const double* __restrict__ xyz, size_t n, const double* __restrict__ box,
                              uint32_t* __restrict__ keys, uint32_t* __restrict__ idx)
{
  const double lx = box[0], ly = box[1], lz = box[2];
  const double ex = box[3] - lx, ey = box[4] - ly, ez = box[5] - lz;
  const double sx = (ex > 0 && isfinite(ex)) ? 1023.999 / ex : 0.0;
  const double sy = (ey > 0 && isfinite(ey)) ? 1023.999 / ey : 0.0;
  const double sz = (ez > 0 && isfinite(ez)) ? 1023.999 / ez : 0.0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double f0 = (xyz[3 * i] - lx) * sx, f1 = (xyz[3 * i + 1] - ly) * sy, f2 = (xyz[3 * i + 2] - lz) * sz;
    if (!(f0 >= 0)) f0 = 0;
    if (!(f1 >= 0)) f1 = 0;
    if (!(f2 >= 0)) f2 = 0;
    if (f0 > 1023) f0 = 1023;
    if (f1 > 1023) f1 = 1023;
    if (f2 > 1023) f2 = 1023;
    keys[i] = spread10((uint32_t)f0) | (spread10((uint32_t)f1) << 1) | (spread10((uint32_t)f2) << 2);
    idx[i] = (uint32_t)i;
  }
}

__global__ void k_gather_soa(const double* __restrict__ src, const uint32_t* __restrict__ order, size_t n,
                             double* __restrict__ x, double* __restrict__ y, double* __restrict__ z)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const size_t i = order[j];
    x[j] = src[3 * i]; y[j] = src[3 * i + 1]; z[j] = src[3 * i + 2];
  }
}

size_t scan_u32_temp_bytes(size_t n)
{
  size_t tmp = 0;
  uint32_t* p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, tmp, p, p, 0u, n, rocprim::plus<uint32_t>(), (hipStream_t)0);
  return tmp;
}
hipError_t launch_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* tmp, size_t tmp_bytes, hipStream_t s)
{
  if (!n) return hipSuccess;
  return rocprim::exclusive_scan(tmp, tmp_bytes, in, out, 0u, n, rocprim::plus<uint32_t>(), s);
}

// ---- exclusive scan of packed counter pairs, one launch (round 3) ---------------------------------------------------
// The partition passes of the two tree builders scan one 64-bit word per position -- two 32-bit counts side by side --
// once or twice per level: with rocPRIM that is two launches per scan (the look-back state's initialisation, then the
// scan), and a level is a dozen dependent launches to begin with.  This is the same decoupled look-back in one launch:
// a tile's status word carries flag, epoch and BOTH partial counts (27 bits each: up to 2^27 positions), so one 64-bit
// atomic publishes a tile and nothing needs initialising between launches -- a word from an earlier launch has another
// epoch (1..255 per memset of the state; the builders memset once per build and use a fresh epoch per scan).  Tiles take
// their numbers from a counter in the order in which their workgroups start (the one that draws the last number puts
// the counter back), so a tile only ever waits for tiles that are already running; a whole wave looks back, 64
// predecessors per round trip.
#define SP_PER 32u            // words per thread
#define SP_TILE (256u * SP_PER)
#ifndef WAVE
#define WAVE 64
#endif
__device__ __forceinline__ unsigned long long sp_pack(unsigned long long v, uint32_t flag, uint32_t epoch)
{
  return ((unsigned long long)flag << 62) | ((unsigned long long)epoch << 54) | (((v >> 32) & 0x7FFFFFFull) << 27) | (v & 0x7FFFFFFull);
}
__device__ __forceinline__ unsigned long long sp_value(unsigned long long w) { return (((w >> 27) & 0x7FFFFFFull) << 32) | (w & 0x7FFFFFFull); }
__device__ __forceinline__ unsigned long long sp_wave_incl(unsigned long long v, uint32_t lane)
{
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const unsigned long long t = (unsigned long long)__shfl_up((long long)v, off, WAVE);
    if (lane >= (uint32_t)off) v += t;
  }
  return v;
}
__global__ void __launch_bounds__(256) k_scan_pair27(const unsigned long long* __restrict__ in, unsigned long long* __restrict__ out,
                                                     uint32_t n, unsigned long long* __restrict__ status, uint32_t* __restrict__ counter,
                                                     uint32_t epoch, uint32_t ntiles, uint32_t* __restrict__ err,
                                                     const uint32_t* __restrict__ gate)
{
  // (a caller's "nothing to scan" word: every workgroup leaves before it draws a tile number, the counter and the epochs stay
  //  as they are -- a skipped epoch is simply never seen)
  if (gate != nullptr && *gate == 0u) return;
  __shared__ uint32_t s_tile;
  __shared__ unsigned long long s_wave[256 / WAVE], s_excl;
  const uint32_t tid = threadIdx.x, lane = tid & (WAVE - 1), wv = tid / WAVE;
  if (tid == 0) {
    const uint32_t t = atomicAdd(counter, 1u);
    if (t == ntiles - 1u) atomicExch(counter, 0u);     // every tile has drawn: ready for the next launch
    s_tile = t;
  }
  __syncthreads();
  const uint32_t tile = s_tile;
  // thread t takes words t, t + 256, ... of every 256-word row?  No: SP_PER CONSECUTIVE words per thread would make the
  // loads of a wave stride by 256 bytes; rows of 256 keep them coalesced, and the scan order is row-major over
  // (thread-chunk): each thread owns SP_PER consecutive words, read as SP_PER / 2 16-byte loads
  const uint32_t base = tile * SP_TILE + tid * SP_PER;
  unsigned long long v[SP_PER], tot = 0;
#pragma unroll
  for (uint32_t k = 0; k < SP_PER; k += 2) {
    if (base + k + 1u < n) { const ulonglong2 x = *reinterpret_cast<const ulonglong2*>(in + base + k); v[k] = x.x; v[k + 1] = x.y; }
    else { v[k] = (base + k < n) ? in[base + k] : 0ull; v[k + 1] = 0ull; }
  }
#pragma unroll
  for (uint32_t k = 0; k < SP_PER; k++) { const unsigned long long x = v[k]; v[k] = tot; tot += x; }      // exclusive within the thread
  const unsigned long long incl = sp_wave_incl(tot, lane);
  if (lane == WAVE - 1) s_wave[wv] = incl;
  __syncthreads();
  unsigned long long woff = 0, ttot = 0;
#pragma unroll
  for (uint32_t w = 0; w < 256 / WAVE; w++) { if (w < wv) woff += s_wave[w]; ttot += s_wave[w]; }
  if (wv == 0) {
    if (tile == 0) {
      if (lane == 0) { __hip_atomic_store(&status[0], sp_pack(ttot, 2u, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_excl = 0ull; }
    } else {
      if (lane == 0) __hip_atomic_store(&status[tile], sp_pack(ttot, 1u, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long excl = 0;
      int pos = (int)tile - 1;
      uint32_t spins = 0;
      for (;;) {
        const int idx = pos - (int)lane;
        unsigned long long w = (idx >= 0) ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                          : sp_pack(0ull, 2u, epoch);       // before the first tile: prefix 0
        const bool valid = ((uint32_t)(w >> 54) & 0xFFu) == epoch && (w >> 62) != 0ull;
        // the nearest predecessor whose PREFIX is known ends the look-back; everything nearer must at least have its
        // aggregate out
        const unsigned long long pm = __ballot(valid && (w >> 62) == 2ull);
        const unsigned long long vm = __ballot(valid);
        const int p = pm ? (__ffsll((long long)pm) - 1) : 64;
        const unsigned long long need = (p >= 64) ? ~0ull : ((2ull << p) - 1ull);
        if ((vm & need) != need) {                       // somebody in front has not published yet
          if (++spins > (1u << 22)) { if (lane == 0) atomicOr(err, 0x10000u); break; }   // its own bit: not the callers' "degenerate input" flag
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        unsigned long long part = ((int)lane <= p) ? sp_value(w) : 0ull;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += (unsigned long long)__shfl_xor((long long)part, off, WAVE);
        excl += part;
        if (p < 64) break;
        pos -= WAVE;
      }
      if (lane == 0) {
        __hip_atomic_store(&status[tile], sp_pack(excl + ttot, 2u, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_excl = excl;
      }
    }
  }
  __syncthreads();
  const unsigned long long off0 = s_excl + woff + (incl - tot);
#pragma unroll
  for (uint32_t k = 0; k < SP_PER; k += 2) {
    if (base + k + 1u < n) { ulonglong2 y; y.x = off0 + v[k]; y.y = off0 + v[k + 1]; *reinterpret_cast<ulonglong2*>(out + base + k) = y; }
    else if (base + k < n) out[base + k] = off0 + v[k];
  }
}
size_t scan_pair27_state_bytes(size_t n) { return 8 * ((n + SP_TILE - 1) / SP_TILE + 1) + 64; }
// state: scan_pair27_state_bytes(n) bytes, zeroed once (hipMemsetAsync) before the first of up to 255 scans; epoch = 1, 2, ...
// n < 2^27 and every count < 2^27 (the callers scan 0/1 flags over at most n positions); err: bit 16 is set if a tile gave up
// waiting for its predecessors (a stalled tile under a debugger / preemption; never seen) -- the offsets are wrong then
hipError_t launch_scan_pair27(const unsigned long long* in, unsigned long long* out, size_t n, void* state, uint32_t epoch,
                              uint32_t* err, hipStream_t s, const uint32_t* gate)
{
  if (!n) return hipSuccess;
  const uint32_t ntiles = (uint32_t)((n + SP_TILE - 1) / SP_TILE);
  unsigned long long* status = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(state) + 64);
  uint32_t* counter = reinterpret_cast<uint32_t*>(state);
  hipLaunchKernelGGL(k_scan_pair27, dim3(ntiles), dim3(256), 0, s, in, out, (uint32_t)n, status, counter, epoch, ntiles, err, gate);
  return hipGetLastError();
}

size_t morton_sort_temp_bytes(size_t n)
{
  size_t tmp = 0;
  uint32_t* p = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, tmp, p, p, p, p, n, 0, 30, (hipStream_t)0);
  return tmp;
}

// keys_a/idx_a are filled, sorted into keys_b/idx_b
hipError_t launch_morton_order(const double* d_xyz, size_t n, const double* d_box, uint32_t* keys_a, uint32_t* idx_a, uint32_t* keys_b, uint32_t* idx_b, void* d_tmp,
                               size_t tmp_bytes, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_morton_keys, dim3((uint32_t)nb), dim3(256), 0, s, d_xyz, n, d_box, keys_a, idx_a);
  hipError_t e = rocprim::radix_sort_pairs(d_tmp, tmp_bytes, keys_a, keys_b, idx_a, idx_b, n, 0, 30, s);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}

// the inverse of the Morton gather: sorted SoA back to the caller's order, AoS (input of the tree builder)
__global__ void k_unsort_aos(const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z,
                             const int32_t* __restrict__ order, size_t n, double* __restrict__ out)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
    const size_t i = (size_t)order[j];
    out[3 * i] = x[j]; out[3 * i + 1] = y[j]; out[3 * i + 2] = z[j];
  }
}
hipError_t launch_unsort_aos(const double* x, const double* y, const double* z, const int32_t* order, size_t n,
                             double* out, hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_unsort_aos, dim3((uint32_t)nb), dim3(256), 0, s, x, y, z, order, n, out);
  return hipGetLastError();
}

hipError_t launch_gather_soa(const double* d_src, const uint32_t* order, size_t n, double* x, double* y, double* z,
                             hipStream_t s)
{
  if (!n) return hipSuccess;
  size_t nb = (n + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(k_gather_soa, dim3((uint32_t)nb), dim3(256), 0, s, d_src, order, n, x, y, z);
  return hipGetLastError();
}

}  // namespace tdtk
