// How long does a letter take?  A kernel of one wave watches a word (system-scope relaxed loads) and answers into a word of
// pinned host memory; the host writes the first word and watches the second.
//   A  the watched word in pinned HOST memory, the kernel resident over all letters
//   B  the same, but a FRESH kernel per letter (launched well before its letter is written)
//   C  the watched word in fine-grained DEVICE memory that the host stores to directly (large BAR), fresh kernel per letter
//   D  launch + answer (no letter): what a dependent launch costs end to end
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void waiter(const unsigned long long* in, unsigned long long* out, unsigned long long first, int letters)
{
  for (int k = 0; k < letters; k++) {
    while (__hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != first + k) __builtin_amdgcn_s_sleep(1);
    __hip_atomic_store(out, first + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
__global__ void answer(unsigned long long* out, unsigned long long k) { __hip_atomic_store(out, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void spin_us(double us) { const double t0 = now_us(); while (now_us() - t0 < us) {} }
int main()
{
  unsigned long long *in, *out, *din = nullptr;
  hipHostMalloc((void**)&in, 64, hipHostMallocCoherent); hipHostMalloc((void**)&out, 64, hipHostMallocCoherent);
  *in = 0; *out = 0;
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 2000;
  waiter<<<1, 64, 0, s>>>(in, out, 1, N);
  double sum = 0;
  for (int k = 1; k <= N; k++) {
    const double t0 = now_us();
    *(volatile unsigned long long*)in = (unsigned long long)k;
    while (*(volatile unsigned long long*)out != (unsigned long long)k) { __builtin_ia32_pause(); }
    if (k > 100) sum += now_us() - t0;
  }
  hipStreamSynchronize(s);
  printf("A  resident kernel, letter in pinned host memory: %.2f us per round trip\n", sum / (N - 100));
  sum = 0;
  for (int k = 1; k <= 300; k++) {
    const unsigned long long v = 1000000ull + k;
    waiter<<<1, 64, 0, s>>>(in, out, v, 1);
    spin_us(40.0);                                  // the kernel is up and watching by now
    const double t0 = now_us();
    *(volatile unsigned long long*)in = v;
    while (*(volatile unsigned long long*)out != v) { __builtin_ia32_pause(); }
    sum += now_us() - t0;
  }
  hipStreamSynchronize(s);
  printf("B  fresh kernel per letter, letter in pinned host memory: %.2f us per round trip\n", sum / 300);
  if (hipExtMallocWithFlags((void**)&din, 64, hipDeviceMallocFinegrained) == hipSuccess) {
    hipMemset(din, 0, 64); hipDeviceSynchronize();
    hipPointerAttribute_t at{};
    hipPointerGetAttributes(&at, din);
    printf("   fine-grained device memory: hostPointer %p devicePointer %p\n", at.hostPointer, at.devicePointer);
    if (at.hostPointer) {
      volatile unsigned long long* h = (volatile unsigned long long*)at.hostPointer;
      sum = 0;
      for (int k = 1; k <= 300; k++) {
        const unsigned long long v = 2000000ull + k;
        waiter<<<1, 64, 0, s>>>(din, out, v, 1);
        spin_us(40.0);
        const double t0 = now_us();
        *h = v;
        while (*(volatile unsigned long long*)out != v) { __builtin_ia32_pause(); }
        sum += now_us() - t0;
      }
      hipStreamSynchronize(s);
      printf("C  fresh kernel per letter, letter stored by the host into fine-grained DEVICE memory: %.2f us per round trip\n", sum / 300);
    }
  } else printf("C  hipExtMallocWithFlags(finegrained) failed\n");
  sum = 0;
  for (int k = 1; k <= 500; k++) {
    const double t0 = now_us();
    answer<<<1, 64, 0, s>>>(out, 3000000ull + k);
    while (*(volatile unsigned long long*)out != 3000000ull + k) { __builtin_ia32_pause(); }
    sum += now_us() - t0;
  }
  printf("D  launch + answer: %.2f us\n", sum / 500);
  return 0;
}
