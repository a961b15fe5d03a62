// A persistent kernel of G workgroups x 256 threads that runs P phases over an array of N doubles with a grid barrier between
// phases: how long is a phase (barrier + fences + a pass over ~2 MB of data the other workgroups wrote)?  Decides whether the
// top levels of a small scan's tree build (eleven dependent launches per level today, ~7 us each) belong in one launch.
// Every phase, element i becomes f(a[j]) of an element j another workgroup wrote in the phase before: a stale read shows in
// the checksum.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
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
int main()
{
  const unsigned n = 81000 * 3;
  double *a, *b; Bar* bar;
  hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&bar, sizeof(Bar));
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int P = 60;
  std::vector<double> ref(n), got(n);
  // reference: P dependent launches
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  for (int rep = 0; rep < 3; rep++) {
    hipMemset(a, 0, n * 8);
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
        hipStreamSynchronize(0);
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
  return 0;
}
