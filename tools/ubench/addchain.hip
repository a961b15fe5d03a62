// Dependent v_add_f64 chain, one wave: (A) operands in registers -- the floor; (B) operands streamed from LDS with
// the read pattern of k_measure (groups of 8 ds_read_b128 at a wave-uniform address, one wait per group), LDS
// filled once -- what the read pattern alone costs; (C) as B with ds_read_b64 x16 per group.
// GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/addchain tools/ubench/addchain.hip && /tmp/addchain
#include <cstdio>
#include <hip/hip_runtime.h>

typedef double d2_t __attribute__((ext_vector_type(2)));
#define LDS_READ_GROUP(R, addr)                                                                                     \
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"         \
               "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t" \
               "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112"                                     \
               : "=&v"(R[0]), "=&v"(R[1]), "=&v"(R[2]), "=&v"(R[3]), "=&v"(R[4]), "=&v"(R[5]), "=&v"(R[6]), "=&v"(R[7]) \
               : "v"(addr))
#define LDS_WAIT_GROUP(R, n)                                                                                   \
  asm volatile("s_waitcnt lgkmcnt(" #n ")"                                                                     \
               : "+v"(R[0]), "+v"(R[1]), "+v"(R[2]), "+v"(R[3]), "+v"(R[4]), "+v"(R[5]), "+v"(R[6]), "+v"(R[7]))

__global__ void k_chain_reg(const double* __restrict__ in, double* __restrict__ out, int reps)
{
  double r[32];
#pragma unroll
  for (int j = 0; j < 32; j++) r[j] = in[j];
  double sum = in[32];
  for (int i = 0; i < reps; i++) {
#pragma unroll
    for (int j = 0; j < 32; j++) sum += r[j];
  }
  out[threadIdx.x] = sum;
}

__global__ void k_chain_lds(const double* __restrict__ in, double* __restrict__ out, int reps)
{
  __shared__ double buf[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) buf[i] = in[i & 31];
  __syncthreads();
  double sum = in[32];
  const uint32_t a0 = (uint32_t)(size_t)(__attribute__((address_space(3))) const double*)buf;
  for (int i = 0; i < reps; i++) {      // 2048 values per repetition: 128 groups of 16
    uint32_t a = a0;
    d2_t ra[8], rb[8];
    LDS_READ_GROUP(ra, a);
#pragma unroll 4
    for (int g = 0; g < 128; g += 2) {
      a += 128u;
      LDS_READ_GROUP(rb, a);
      LDS_WAIT_GROUP(ra, 8);
#pragma unroll
      for (int q = 0; q < 8; q++) { sum += ra[q].x; sum += ra[q].y; }
      a += 128u;
      LDS_READ_GROUP(ra, (g + 2 < 128) ? a : a0);
      LDS_WAIT_GROUP(rb, 8);
#pragma unroll
      for (int q = 0; q < 8; q++) { sum += rb[q].x; sum += rb[q].y; }
    }
    LDS_WAIT_GROUP(ra, 0);
  }
  out[threadIdx.x] = sum;
}

template <typename F>
static void run(const char* name, F launch, double adds)
{
  for (int rep = 0; rep < 3; rep++) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep == 2) printf("%-28s %.0f dependent v_add_f64: %.3f ms = %.2f ns per add\n", name, adds, ms, ms * 1e6 / adds);
  }
}

int main()
{
  double h[33];
  for (int i = 0; i < 33; i++) h[i] = 1.0 + i * 1e-3;
  double *d_in, *d_out;
  hipMalloc((void**)&d_in, sizeof h); hipMalloc((void**)&d_out, 64 * sizeof(double));
  hipMemcpy(d_in, h, sizeof h, hipMemcpyHostToDevice);
  run("registers:", [&] { hipLaunchKernelGGL(k_chain_reg, dim3(1), dim3(64), 0, 0, d_in, d_out, 1 << 15); }, 32.0 * (1 << 15));
  run("LDS groups of 8 x b128:", [&] { hipLaunchKernelGGL(k_chain_lds, dim3(1), dim3(64), 0, 0, d_in, d_out, 512); }, 2048.0 * 512);
  return 0;
}
