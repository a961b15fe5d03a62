// Dependent v_add_f64 chain: cycles per add when nothing else is in the way (one wave, operands in registers).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/addchain tools/ubench/addchain.hip && /tmp/addchain
#include <cstdio>
#include <hip/hip_runtime.h>

__global__ void k_chain(const double* __restrict__ in, double* __restrict__ out, long long* __restrict__ cycles, int reps)
{
  double r[32];
#pragma unroll
  for (int j = 0; j < 32; j++) r[j] = in[j];
  double sum = in[32];
  const long long t0 = clock64();
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int j = 0; j < 32; j++) sum += r[j];
  }
  const long long t1 = clock64();
  out[threadIdx.x] = sum;
  if (threadIdx.x == 0) *cycles = t1 - t0;
}

int main()
{
  double h[33];
  for (int i = 0; i < 33; i++) h[i] = 1.0 + i * 1e-3;
  double *d_in, *d_out;
  long long* d_c;
  hipMalloc((void**)&d_in, sizeof h); hipMalloc((void**)&d_out, 64 * sizeof(double)); hipMalloc((void**)&d_c, 8);
  hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
  const int reps = 1 << 15;
  for (int rep = 0; rep < 3; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, 0, d_in, d_out, d_c, reps);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    long long c = 0;
    hipMemcpy(&c, d_c, 8, hipMemcpyDeviceToHost);
    const double adds = 32.0 * reps;
    printf("%.0f dependent v_add_f64: %.3f ms = %.2f ns per add; clock64 ticks per add %.2f\n", adds, ms, ms * 1e6 / adds, (double)c / adds);
  }
  return 0;
}
