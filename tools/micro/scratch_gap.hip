// Does a kernel that uses scratch (a private segment: the call frame of a non-inlined device function) start later behind
// its predecessor on the same stream than one that does not?  200 dependent launches each; also: big kernarg (1.3 KB).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { double v[170]; };
__device__ __noinline__ void helper(int* p, int i) { p[i & 1] += i; }
__global__ void plain(int* p) { if (threadIdx.x == 0) p[0] += 1; }
__global__ void with_scratch(int* p) { if (threadIdx.x == 0) helper(p, (int)blockIdx.x + 1); }
__global__ void big_arg(int* p, Big b) { if (threadIdx.x == 0) p[0] += (int)b.v[3]; }
__global__ void big_arg_scratch(int* p, Big b) { if (threadIdx.x == 0) helper(p, (int)b.v[3] + 1); }
template <typename F> static void run(const char* name, F f, hipStream_t s)
{
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; i++) f();
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    if (rep == 2) printf("%-28s %.2f us per launch\n", name, std::chrono::duration<double, std::micro>(t1 - t0).count() / 200);
  }
}
int main()
{
  int* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  Big b{}; b.v[3] = 1.0;
  run("plain <<<1,64>>>", [&] { plain<<<1, 64, 0, s>>>(d); }, s);
  run("scratch <<<1,64>>>", [&] { with_scratch<<<1, 64, 0, s>>>(d); }, s);
  run("plain <<<320,256>>>", [&] { plain<<<320, 256, 0, s>>>(d); }, s);
  run("scratch <<<320,256>>>", [&] { with_scratch<<<320, 256, 0, s>>>(d); }, s);
  run("big kernarg <<<320,256>>>", [&] { big_arg<<<320, 256, 0, s>>>(d, b); }, s);
  run("big kernarg + scratch", [&] { big_arg_scratch<<<320, 256, 0, s>>>(d, b); }, s);
  run("alternating plain/scratch", [&] { plain<<<320, 256, 0, s>>>(d); with_scratch<<<320, 256, 0, s>>>(d); }, s);
  return 0;
}
