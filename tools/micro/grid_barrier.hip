// A persistent kernel of G workgroups x 256 threads that runs P phases over an array of N doubles with a grid barrier between
// phases: how long is a phase (barrier + fences + a pass over ~2 MB of data the other workgroups wrote)?  Decides whether the
// top levels of a small scan's tree build (eleven dependent launches per level today, ~7 us each) belong in one launch.
// Every phase, element i becomes f(a[j]) of an element j another workgroup wrote in the phase before: a stale read shows in
// the checksum.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct Bar { unsigned count; unsigned gen; unsigned pad[30]; };
template <int MODE>
__device__ __forceinline__ void grid_barrier(Bar* b, unsigned nwg, unsigned& my_gen)
{
  // MODE 0: fence seq_cst agent on both sides (write back + invalidate the XCD's L2 in every workgroup)
  // MODE 1: release before, acquire after (the same instructions on this target, stated separately)
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    const unsigned t = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nwg - 1u) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, my_gen + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) __builtin_amdgcn_s_sleep(1);
    }
    if (MODE == 0) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  my_gen++;
  __syncthreads();
}
template <int MODE>
__global__ void __launch_bounds__(256) k_phases(double* a, double* b, unsigned n, int phases, Bar* bar)
{
  unsigned gen = 0;
  double *src = a, *dst = b;
  for (int p = 0; p < phases; p++) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
      const unsigned j = (i + 7919u * 256u) % n;          // written by another workgroup in the phase before
      dst[i] = src[j] * 1.0000001 + 1.0;
    }
    grid_barrier<MODE>(bar, gridDim.x, gen);
    double* t = src; src = dst; dst = t;
  }
}
__global__ void k_one(const double* src, double* dst, unsigned n)
{
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
    const unsigned j = (i + 7919u * 256u) % n;
    dst[i] = src[j] * 1.0000001 + 1.0;
  }
}
// ---- round 5: the two forms the round-4 review asked for ---------------------------------------------------------------
// MODE 2: the grid confined to ONE XCD (block b of the launch runs on XCD b % 8: only every eighth block works, the others leave
//         at once) -- the workgroups share one L2, so nothing has to be written back; a workgroup waits for its own stores
//         (s_waitcnt vmcnt(0)), joins the barrier with relaxed atomics and invalidates its CU's L1 (buffer_inv sc1) behind it.
// MODE 3: one XCD, and the data that crosses the barrier is READ past the L1 (sc1 loads = relaxed agent-scope atomic loads):
//         no invalidate at all.
// MODE 4: all eight XCDs, the crossing data written with sc0 sc1 stores and read with sc0 sc1 loads (relaxed system-scope
//         atomics: nothing of it stays in an L1 or a non-coherent L2 line), no fence in the barrier.
template <int MODE>
__device__ __forceinline__ void grid_barrier_light(Bar* b, unsigned nwg, unsigned& my_gen)
{
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == nwg - 1u) {
      __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&b->gen, my_gen + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_gen) __builtin_amdgcn_s_sleep(1);
    }
  }
  my_gen++;
  __syncthreads();
  if (MODE == 2) asm volatile("buffer_inv sc1" ::: "memory");
}
template <int MODE>
__global__ void __launch_bounds__(256) k_phases_light(double* a, double* b, unsigned n, int phases, Bar* bar, unsigned stride_blocks)
{
  if (blockIdx.x % stride_blocks != 0) return;            // (MODE 2 / 3: only the blocks of XCD 0 take part)
  const unsigned wg = blockIdx.x / stride_blocks, nwg = gridDim.x / stride_blocks;
  unsigned gen = 0;
  double *src = a, *dst = b;
  for (int p = 0; p < phases; p++) {
    for (unsigned i = wg * 256u + threadIdx.x; i < n; i += nwg * 256u) {
      const unsigned j = (i + 7919u * 256u) % n;          // written by another workgroup in the phase before
      double v;
      if (MODE == 3) v = __hip_atomic_load(&src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (MODE == 4) v = __hip_atomic_load(&src[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else v = src[j];
      v = v * 1.0000001 + 1.0;
      if (MODE == 4) __hip_atomic_store(&dst[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else dst[i] = v;
    }
    grid_barrier_light<MODE>(bar, nwg, gen);
    double* t = src; src = dst; dst = t;
  }
}
int main(int argc, char** argv)
{
  const unsigned n = argc > 1 ? (unsigned)atoi(argv[1]) : 81000 * 3;     // doubles per phase (default: the coordinates of an 81K-point scan)
  double *a, *b; Bar* bar;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&bar, sizeof(Bar));
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int P = 60;
  std::vector<double> ref(n), got(n);
  // reference: P dependent launches
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(a, 0, n * 8); hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    double *src = a, *dst = b;
    for (int p = 0; p < P; p++) { k_one<<<128, 256, 0, s>>>(src, dst, n); double* t = src; src = dst; dst = t; }
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("%d dependent launches: %.2f us per phase\n", P, std::chrono::duration<double, std::micro>(t1 - t0).count() / P);
    hipMemcpy(ref.data(), (P % 2) ? b : a, n * 8, hipMemcpyDeviceToHost);
  }
  for (int mode = 0; mode < 2; mode++)
    for (unsigned G : {32u, 64u, 128u, 256u}) {
      for (int rep = 0; rep < 2; rep++) {
        hipMemset(a, 0, n * 8); hipMemset(bar, 0, sizeof(Bar));
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (mode == 0) k_phases<0><<<G, 256, 0, s>>>(a, b, n, P, bar); else k_phases<1><<<G, 256, 0, s>>>(a, b, n, P, bar);
        hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpy(got.data(), (P % 2) ? b : a, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (unsigned i = 0; i < n; i++) bad += got[i] != ref[i];
        if (rep == 1) printf("one launch, %3u workgroups, fence mode %d: %.2f us per phase (%zu stale values)\n", G, mode,
                             std::chrono::duration<double, std::micro>(t1 - t0).count() / P, bad);
      }
    }
  for (int mode = 2; mode <= 4; mode++)
    for (unsigned G : {16u, 32u, 64u, 128u}) {
      if (mode == 4 && G < 64u) continue;
      const unsigned stride = (mode == 4) ? 1u : 8u;
      for (int rep = 0; rep < 2; rep++) {
        hipMemset(a, 0, n * 8); hipMemset(bar, 0, sizeof(Bar));
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        if (mode == 2) k_phases_light<2><<<G * stride, 256, 0, s>>>(a, b, n, P, bar, stride);
        else if (mode == 3) k_phases_light<3><<<G * stride, 256, 0, s>>>(a, b, n, P, bar, stride);
        else k_phases_light<4><<<G * stride, 256, 0, s>>>(a, b, n, P, bar, stride);
        hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpy(got.data(), (P % 2) ? b : a, n * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (unsigned i = 0; i < n; i++) bad += got[i] != ref[i];
        if (rep == 1) printf("one launch, %3u workgroups%s, light barrier mode %d: %.2f us per phase (%zu stale values)\n", G,
                             mode == 4 ? " on all XCDs" : " on ONE XCD", mode, std::chrono::duration<double, std::micro>(t1 - t0).count() / P, bad);
      }
    }
  return 0;
}
