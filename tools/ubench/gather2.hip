// Microbenchmark: what a divergent (one record per lane) fetch costs on gfx950 by record shape -- is the price per
// load instruction, per byte, or per cache line?  Table of 4 MB (L2-resident).  Loads are independent (throughput, not
// latency); 8 waves per SIMD.
// GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/gather2 tools/ubench/gather2.hip && /tmp/gather2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(const char* __restrict__ buf, unsigned nrec, int iters, float* out)
{
  unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned s = tid * 2654435761u + 12345u;
  float a = 0;
  for (int i = 0; i < iters; i++) {
    s = s * 1664525u + 1013904223u;
    const unsigned r = (unsigned)(((unsigned long long)s * nrec) >> 32);
    if (MODE == 0) { const f4 v = *(const f4*)(buf + (size_t)r * 16); a += (v.x + v.y) + (v.z + v.w); }                       // 16 B
    if (MODE == 1) { const char* p = buf + (size_t)r * 32; const f4 v = *(const f4*)p, w = *(const f4*)(p + 16); a += (v.x + v.y) + (v.z + v.w) + (w.x + w.y) + (w.z + w.w); }   // 32 B record, 2 loads
    if (MODE == 2) { const char* p = buf + (size_t)r * 48; const f4 v = *(const f4*)p; const f2 u = *(const f2*)(p + 16); const f4 w = *(const f4*)(p + 32); a += (v.x + v.y) + (v.z + v.w) + (u.x + u.y) + (w.x + w.y) + (w.z + w.w); }  // 48 B record, 16+8+16
    if (MODE == 3) { const f2 v = *(const f2*)(buf + (size_t)r * 8); a += v.x + v.y; }                        // 8 B
    if (MODE == 4) { a += *(const float*)(buf + (size_t)r * 4); }                                                // 4 B
    if (MODE == 5) { const char* p = buf + (size_t)r * 128; f4 v[8];                                            // 4 points of 32 B: 8 loads, 128 contiguous bytes
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = *(const f4*)(p + 16 * j);
#pragma unroll
      for (int j = 0; j < 8; j++) a += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
    if (MODE == 6) { const char* p = buf + (size_t)r * 96; f4 v[6];                                             // 4 points of 24 B: 6 loads, 96 contiguous bytes
#pragma unroll
      for (int j = 0; j < 6; j++) v[j] = *(const f4*)(p + 16 * j);
#pragma unroll
      for (int j = 0; j < 6; j++) a += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
    if (MODE == 7) { const char* p = buf + (size_t)r * 64; const f4 v = *(const f4*)p, w = *(const f4*)(p + 16), x = *(const f4*)(p + 32); a += (v.x + v.y) + (v.z + v.w) + (w.x + w.y) + (w.z + w.w) + (x.x + x.y) + (x.z + x.w); }  // 48 of a 64 B record
    if (MODE == 8) { const char* p = buf + (size_t)r * 32; f4 v[8];                                             // 4 points of 32 B from any 32 B-aligned start
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = *(const f4*)(p + 16 * j);
#pragma unroll
      for (int j = 0; j < 8; j++) a += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
    if (MODE == 9) { const char* p = buf + (size_t)r * 8; f4 v[6];                                              // 4 points of 24 B from any 8 B-aligned start
#pragma unroll
      for (int j = 0; j < 6; j++) v[j] = *(const f4*)(p + 16 * j);
#pragma unroll
      for (int j = 0; j < 6; j++) a += (v[j].x + v[j].y) + (v[j].z + v[j].w); }
  }
  if (a == 123.456f) out[tid] = a;
}
template <int MODE>
void run(const char* name, const char* d, size_t bytes, int recsize, int loads, float* o)
{
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int blocks = 256 * 8, iters = 1000;
  const unsigned nrec = (unsigned)(bytes / recsize);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, nrec, 50, o);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, nrec, iters, o);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double wrec = (double)blocks * 4 * iters;            // wave-level record fetches
  const double cyc = ms * 1e-3 * 2.4e9 * 256 / wrec;         // CU-cycles per wave-record (2.4 GHz, 256 CUs)
  printf("%-44s %7.3f ms  %7.1f CU-cycles per wave-record  %6.1f per load instruction  %6.2f TB/s useful\n", name, ms, cyc, cyc / loads,
         wrec * 64 * recsize / ms * 1e-9);
}
int main()
{
  const size_t bytes = 4u << 20;
  char* d; float* o;
  (void)hipMalloc(&d, bytes + 4096); (void)hipMalloc(&o, 1 << 24); (void)hipMemset(d, 0, bytes + 4096);
  run<0>("16 B record, 1 x b128", d, bytes, 16, 1, o);
  run<1>("32 B record, 2 x b128", d, bytes, 32, 2, o);
  run<2>("48 B record, b128 + b64 + b128", d, bytes, 48, 3, o);
  run<7>("48 B of a 64 B record, 3 x b128", d, bytes, 64, 3, o);
  run<3>("8 B record, 1 x b64", d, bytes, 8, 1, o);
  run<4>("4 B record, 1 x b32", d, bytes, 4, 1, o);
  run<5>("4 points of 32 B (128 B contiguous), 8 x b128", d, bytes, 128, 8, o);
  run<6>("4 points of 24 B (96 B contiguous), 6 x b128", d, bytes, 96, 6, o);
  run<8>("4 points of 32 B, any 32 B-aligned start, 8 x b128", d, bytes, 32, 8, o);
  run<9>("4 points of 24 B, any 8 B-aligned start, 6 x b128", d, bytes, 8, 6, o);
  return 0;
}
