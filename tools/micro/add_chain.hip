// How long is one link of a dependent v_add_f64 chain on a lone gfx950 wave, and does it depend on how many lanes are
// enabled?  (k_measure / k_big_stitch run such chains with wave-uniform values: every lane computes the same sum.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void chain(const double* __restrict__ x, double* out, long long* cyc, int n, unsigned long long mask_lo, int mode)
{
  const int lane = threadIdx.x & 63;
  double s = x[0];
  double v[16];
  for (int i = 0; i < 16; i++) v[i] = x[i + 1];
  const bool on = (mask_lo >> lane) & 1ull;
  long long t0 = 0, t1 = 0;
  if (on) {
    t0 = __builtin_readcyclecounter();
    if (mode == 0) {
      for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_f64 %0, %0, %1" : "+v"(s) : "v"(v[i]));
      }
    } else if (mode == 1) {
      for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_fma_f64 %0, %0, 1.0, %1" : "+v"(s) : "v"(v[i]));
      }
    } else {
      float f = (float)s;
      for (int it = 0; it < n; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f) : "v"((float)v[i]));
      }
      s = f;
    }
    t1 = __builtin_readcyclecounter();
  }
  if (lane == 0) { out[blockIdx.x] = s; cyc[blockIdx.x] = t1 - t0; }
}
int main()
{
  double h[17]; for (int i = 0; i < 17; i++) h[i] = 1.0 + i * 0.37;
  double *x, *o; long long* c;
  hipMalloc(&x, sizeof h); hipMalloc(&o, 8 * 1024); hipMalloc(&c, 8 * 1024);
  hipMemcpy(x, h, sizeof h, hipMemcpyHostToDevice);
  const int n = 4096;
  const unsigned long long masks[] = {~0ull, 0xFFFFFFFFull, 0xFFFFull, 1ull};
  const char* names[] = {"64 lanes", "32 lanes", "16 lanes", "1 lane"};
  for (int mode = 0; mode < 3; mode++)
    for (int m = 0; m < 4; m++) {
      for (int rep = 0; rep < 2; rep++) chain<<<1, 64>>>(x, o, c, n, masks[m], mode);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); chain<<<1, 64>>>(x, o, c, n, masks[m], mode); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
      printf("%s %-9s: %.2f counter ticks per add, %.2f ns per add (event time)\n", mode == 0 ? "v_add_f64" : mode == 1 ? "v_fma_f64" : "v_add_f32",
             names[m], (double)cy / (16.0 * n), ms * 1e6 / (16.0 * n));
    }
  return 0;
}
