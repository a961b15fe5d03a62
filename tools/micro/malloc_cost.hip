// What hipMalloc / hipFree cost on this box, by size (the tree build allocates six arrays and frees one per tree).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main()
{
  hipFree(0);
  const size_t sizes[] = {1u << 20, 5u << 20, 13u << 20, 35u << 20, 128u << 20};
  for (size_t sz : sizes) {
    double tm = 0, tf = 0;
    const int reps = 20;
    for (int r = 0; r < reps; r++) {
      void* p = nullptr;
      auto t0 = std::chrono::steady_clock::now();
      hipMalloc(&p, sz);
      auto t1 = std::chrono::steady_clock::now();
      hipMemsetAsync(p, 0, 64, 0); hipStreamSynchronize(0);
      auto t2 = std::chrono::steady_clock::now();
      hipFree(p);
      auto t3 = std::chrono::steady_clock::now();
      tm += std::chrono::duration<double, std::micro>(t1 - t0).count();
      tf += std::chrono::duration<double, std::micro>(t3 - t2).count();
    }
    printf("%4zu MB: hipMalloc %.1f us, hipFree %.1f us\n", sz >> 20, tm / reps, tf / reps);
  }
  return 0;
}
