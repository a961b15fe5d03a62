// What does rocprofv3's WRITE_SIZE count?  Three kernels with known stores (round 6):
//   k_copy      256 MiB read + 256 MiB written, 16-byte stores, coalesced
//   k_scatter4  16 Mi threads, each ONE 4-byte store at a permuted position of a 64 MiB array (every 64-byte line gets 16
//               stores from 16 different waves: partial-line writes)
//   k_scatter1  the same with 1-byte stores into a 16 MiB array
// run under: rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -- ./write_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__device__ __forceinline__ uint32_t perm(uint32_t i, uint32_t mask) { return (i * 2654435761u + 12345u) & mask; }   // odd multiplier: a bijection mod 2^k
__global__ void k_scatter4(uint32_t* out, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[perm(i, n - 1)] = i;
}
__global__ void k_scatter1(unsigned char* out, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[perm(i, n - 1)] = (unsigned char)i;
}
int main()
{
  const size_t bytes = 256u << 20;
  void *a, *b;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes);
  hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    k_copy<<<4096, 256>>>((const uint4*)a, (uint4*)b, bytes / 16);
    k_scatter4<<<(16u << 20) / 256, 256>>>((uint32_t*)b, 16u << 20);
    k_scatter1<<<(16u << 20) / 256, 256>>>((unsigned char*)a, 16u << 20);
  }
  hipDeviceSynchronize();
  printf("k_copy: 268435456 B written; k_scatter4: 67108864 B written in 16777216 stores; k_scatter1: 16777216 B in 16777216 stores\n");
  return 0;
}
