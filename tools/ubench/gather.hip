// Microbenchmark (scratch): cost of a wave-level 16-byte-per-lane gather from an L2-resident
// table on gfx950, by address pattern.  hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void __launch_bounds__(256) k(const double2* __restrict__ buf, unsigned mask, int iters, int mode,
                                          double* out)
{
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned lane = threadIdx.x & 63, wave = tid >> 6;
  unsigned s = tid * 2654435761u + 12345u;
  double a = 0, b = 0;
  for (int i = 0; i < iters; i++) {
    s = s * 1664525u + 1013904223u;
    unsigned w = (wave * 2246822519u + i * 3266489917u);
    unsigned idx;
    if (mode == 0) idx = s;                                   // every lane its own random 16 B
    else if (mode == 1) idx = w;                              // whole wave one address
    else if (mode == 2) idx = w + (lane >> 4) * 977u;         // 4 distinct addresses per wave
    else if (mode == 3) idx = w + lane;                       // coalesced 1 KB
    else idx = w + (lane >> 2) * 977u;                        // 16 distinct addresses per wave
    const double2 v = buf[idx & mask];
    a += v.x; b += v.y;
  }
  if (a + b == 123.456) out[tid] = a;
}
int main()
{
  const size_t n = 1u << 18;  // 256K x 16 B = 4 MB (L2 resident per XCD... 32 MB aggregate)
  double2* d; double* o;
  hipMalloc(&d, n * sizeof(double2)); hipMalloc(&o, 1 << 24);
  hipMemset(d, 0, n * sizeof(double2));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 8, iters = 2000;
  const char* names[] = {"random/lane", "wave-uniform", "4 addrs/wave", "coalesced 1KB", "16 addrs/wave"};
  for (int mode = 0; mode < 5; mode++) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, (unsigned)(n - 1), 100, mode, o);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, d, (unsigned)(n - 1), iters, mode, o);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters;             // wave-level load instructions
    double cyc_per = ms * 1e-3 * 2.4e9 * 256 / winstr;      // CU-cycles per wave-instr (at 2.4 GHz)
    printf("%-14s %8.3f ms  %7.1f G lane-loads/s  %6.2f TB/s  %6.1f CU-cycles per wave-load\n", names[mode], ms,
           winstr * 64 / ms * 1e-6, winstr * 1024 / ms * 1e-9, cyc_per);
  }
  return 0;
}
