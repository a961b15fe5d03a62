// Microbenchmark (round 5): what ONE TRIP of k_search's node walk / bucket scan costs a wave as a function of how the record
// is fetched, in the kernel's own regime -- dependent chains (the next record's index comes out of the loaded data), 4 waves
// per SIMD, a part-filled mask, a table that lives in L2 / the Infinity Cache / HBM.
//   mode 0  one 16-byte load per lane (the floor: one line touched once)
//   mode 1  three 16-byte loads per lane from its own 48-byte record (today's divergent node visit: 3 instructions, the 2nd and
//           3rd hit a line whose fill is pending)
//   mode 2  the same 48 bytes fetched ONCE per line: lane l of instruction k loads piece l % 3 of the record of lane 21 k + l / 3
//           straight into LDS (global_load_lds_dwordx4: destination = wave base + lane x 16, i.e. the records land contiguously),
//           the owner reads its record back with three ds_read_b128
//   mode 3  mode 1 with the first load waited for before the other two are issued (they are L1 hits then: two round trips)
//   mode 4  fifteen 16-byte loads per lane of 240 contiguous bytes (a bucket's five shadow groups)
//   mode 5  the same 240 bytes cooperatively: 15 lanes per bucket, 4 buckets per instruction, into LDS; owner reads 15 x b128
//   mode 6  64-byte-stride records, quad-cooperative: lanes 4j..4j+3 of instruction k load the record of lane 16 k + j into LDS
// GPU box: hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_fetch tools/micro/chain_fetch.hip && /tmp/chain_fetch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef const u4 __attribute__((address_space(1))) * gp4;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void glds16(const char* g, unsigned lds_byte)
{
  // wave-uniform LDS base in M0, per-lane global address; EXEC masks the lanes
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(lds_byte) : "memory");
}

template <int MODE>
__global__ void __launch_bounds__(128, 1) k(const char* __restrict__ buf, unsigned nrec, int trips, int nact, unsigned* out)
{
  __shared__ u4 lds[2][MODE == 5 ? 4 : 256];      // per wave: 4 KB
  __shared__ u4 lds5[2][MODE == 5 ? 512 : 4];     // mode 5: 8 KB per wave (32 buckets x 256 B at a time: 16 waves per CU must still fit 160 KB)
  const unsigned lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool act = (int)lane < nact;
  unsigned r = ((blockIdx.x * 128 + threadIdx.x) * 2654435761u) % nrec;
  unsigned acc = 0;
  const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(&lds[wv][0]));      // LDS byte address (generic -> low 32 bits are the LDS offset on gfx9)
  const unsigned lds5_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)(&lds5[wv][0]));
  for (int t = 0; t < trips; t++) {
    unsigned v = 0;
    if (MODE == 0) { if (act) { const u4 a = *(gp4)(buf + (size_t)r * 48); v = a.x ^ a.w; } }
    if (MODE == 1) { if (act) { const char* p = buf + (size_t)r * 48; const u4 a = *(gp4)p, b = *(gp4)(p + 16), c = *(gp4)(p + 32); v = a.x ^ b.y ^ c.z ^ a.w; } }
    if (MODE == 3) { if (act) { const char* p = buf + (size_t)r * 48; u4 a = *(gp4)p; asm volatile("s_waitcnt vmcnt(0)" : "+v"(a)); const u4 b = *(gp4)(p + 16), c = *(gp4)(p + 32); v = a.x ^ b.y ^ c.z ^ a.w; } }
    if (MODE == 2) {
      // every lane publishes its record index; loader lane l of instruction k serves owner 21 k + l / 3
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const unsigned owner = 21u * kk + lane / 3u, piece = lane % 3u;
        const unsigned ro = (unsigned)__shfl((int)r, (int)(owner & 63u), 64);
        const bool on = lane < 63u && (int)owner < nact && owner < 64u;
        if (on) glds16(buf + (size_t)ro * 48 + piece * 16, lds_base + kk * 1008u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (act) {
        const unsigned kk = lane / 21u, j = lane - 21u * kk;
        const u4* p = &lds[wv][0] + (kk * 1008u + j * 48u) / 16u;
        const u4 a = p[0], b = p[1], c = p[2];
        v = a.x ^ b.y ^ c.z ^ a.w;
      }
    }
    if (MODE == 6) {
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const unsigned owner = 16u * kk + (lane >> 2), piece = lane & 3u;
        const unsigned ro = (unsigned)__shfl((int)r, (int)owner, 64);
        if ((int)owner < nact && piece < 3u) glds16(buf + (size_t)ro * 64 + piece * 16, lds_base + kk * 1024u);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (act) { const u4* p = &lds[wv][0] + lane * 4u; const u4 a = p[0], b = p[1], c = p[2]; v = a.x ^ b.y ^ c.z ^ a.w; }
    }
    if (MODE == 4) { if (act) { const char* p = buf + (size_t)r * 48; u4 g[15];
#pragma unroll
        for (int j = 0; j < 15; j++) g[j] = *(gp4)(p + 16 * j);
#pragma unroll
        for (int j = 0; j < 15; j++) v ^= g[j].x + g[j].w; } }
    if (MODE == 5) {
      // 32 owners at a time (8 KB of LDS per wave): 8 instructions of 4 buckets each, then the owners read theirs back
#pragma unroll
      for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int kk = 0; kk < 8; kk++) {
          const unsigned owner = 32u * h + 4u * kk + (lane >> 4), piece = lane & 15u;
          const unsigned ro = (unsigned)__shfl((int)r, (int)owner, 64);
          // (a 16th lane of each group idles: 15 pieces; the LDS image of instruction kk is 4 x 256 B with 16 B holes)
          if ((int)owner < nact && piece < 15u) glds16(buf + (size_t)ro * 48 + piece * 16, lds5_base + kk * 1024u);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (act && (lane >> 5) == (unsigned)h) { const unsigned l5 = lane & 31u; const u4* p = &lds5[wv][0] + ((l5 >> 2) * 1024u + (l5 & 3u) * 256u) / 16u;
#pragma unroll
          for (int j = 0; j < 15; j++) { const u4 g = p[j]; v ^= g.x + g.w; } }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    acc ^= v;
    r = (v ^ (r * 1664525u + 1013904223u)) % nrec;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
double run(const char* d, size_t bytes, int nact, unsigned* o, int recsize)
{
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int blocks = 256 * 2 * 4, trips = 400;                    // 4 waves per SIMD: 16 waves per CU = 8 workgroups of 2 waves
  const unsigned nrec = (unsigned)(bytes / recsize) - 16;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, d, nrec, 20, nact, o);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(128), 0, 0, d, nrec, trips, nact, o);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e-3 * 2.4e9 / trips;          // shader cycles (at 2.4 GHz) per trip of a wave, 4 waves per SIMD resident
}

int main()
{
  const size_t sizes[] = {(size_t)3 << 20, (size_t)64 << 20, (size_t)1 << 30};
  const char* where[] = {"3 MB (L2)", "64 MB (Infinity Cache)", "1 GB (HBM)"};
  unsigned* o; CK(hipMalloc(&o, 64));
  for (int s = 0; s < 3; s++) {
    char* d; CK(hipMalloc(&d, sizes[s] + 4096));
    std::vector<unsigned> h(sizes[s] / 4);
    unsigned x = 12345u;
    for (auto& w : h) { x = x * 1664525u + 1013904223u; w = x; }
    CK(hipMemcpy(d, h.data(), sizes[s], hipMemcpyHostToDevice));
    for (int nact : {64, 32, 16}) {
      printf("table %-24s active lanes %2d | cycles per trip:  one16 %6.0f   3x16 own %6.0f   3x16 first-then-two %6.0f   coop48->LDS %6.0f   quad64->LDS %6.0f   15x16 own %6.0f   coop240->LDS %6.0f\n",
             where[s], nact, run<0>(d, sizes[s], nact, o, 48), run<1>(d, sizes[s], nact, o, 48), run<3>(d, sizes[s], nact, o, 48), run<2>(d, sizes[s], nact, o, 48),
             run<6>(d, sizes[s], nact, o, 64), run<4>(d, sizes[s], nact, o, 48), run<5>(d, sizes[s], nact, o, 48));
      fflush(stdout);
    }
    CK(hipFree(d));
  }
  return 0;
}
