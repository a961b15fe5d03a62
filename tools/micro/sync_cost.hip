// One ICP-like round trip: launch a kernel that leaves its result in pinned host memory, wait for it on the host, launch
// the next.  How long is a round with hipStreamSynchronize against polling the pinned words themselves?
// (work: a dependent chain of `spin` clock ticks so that the kernel lasts ~20 us like a small scan's search)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void work(volatile double* out, double v, long long ticks)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x < 8) out[threadIdx.x] = v + threadIdx.x;
}
__global__ void work2(double* tmp, double v, long long ticks)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) {}
  if (threadIdx.x < 8) tmp[threadIdx.x] = v + threadIdx.x;
}
__global__ void fin(const double* tmp, volatile double* out) { if (threadIdx.x < 8) out[threadIdx.x] = tmp[threadIdx.x]; }
int main(int argc, char** argv)
{
  const double us = argc > 1 ? atof(argv[1]) : 20.0;
  const long long ticks = (long long)(us * 100.0);   // wall_clock64: 100 MHz
  double* h; hipHostMalloc(&h, 64, hipHostMallocDefault);
  double* d; hipMalloc(&d, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 2000;
  union { uint64_t u; double d; } sent; sent.u = 0x7FF8DEADBEEF0001ull;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { work<<<1, 64, 0, s>>>(h, (double)i, ticks); hipStreamSynchronize(s); if (h[7] != i + 7.0) printf("bad\n"); }
    auto t1 = std::chrono::steady_clock::now();
    printf("one kernel, hipStreamSynchronize : %.2f us per round (kernel %.0f)\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N, us);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < 8; k++) h[k] = sent.d;
      work<<<1, 64, 0, s>>>(h, (double)i, ticks);
      volatile uint64_t* hv = (volatile uint64_t*)h;
      for (;;) { bool all = true; for (int k = 0; k < 8; k++) all &= hv[k] != sent.u; if (all) break; }
      if (h[7] != i + 7.0) printf("bad\n");
    }
    t1 = std::chrono::steady_clock::now();
    printf("one kernel, polling pinned words : %.2f us per round\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    hipStreamSynchronize(s);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) { work2<<<1, 64, 0, s>>>(d, (double)i, ticks); fin<<<1, 64, 0, s>>>(d, h); hipStreamSynchronize(s); if (h[7] != i + 7.0) printf("bad\n"); }
    t1 = std::chrono::steady_clock::now();
    printf("two kernels, hipStreamSynchronize: %.2f us per round\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      for (int k = 0; k < 8; k++) h[k] = sent.d;
      work2<<<1, 64, 0, s>>>(d, (double)i, ticks); fin<<<1, 64, 0, s>>>(d, h);
      volatile uint64_t* hv = (volatile uint64_t*)h;
      for (;;) { bool all = true; for (int k = 0; k < 8; k++) all &= hv[k] != sent.u; if (all) break; }
      if (h[7] != i + 7.0) printf("bad\n");
    }
    t1 = std::chrono::steady_clock::now();
    printf("two kernels, polling pinned words: %.2f us per round\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
    hipStreamSynchronize(s);
  }
  return 0;
}
