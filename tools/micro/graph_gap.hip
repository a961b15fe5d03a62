// Dependent tiny kernels on one stream: launched one by one against replayed as a captured hipGraph -- how long is the
// gap between two of them in either mode?  (The tree build is ~190 such launches.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(int* p) { if (threadIdx.x == 0) p[0] += 1; }
int main()
{
  int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int N = 200;
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) tiny<<<1, 64, 0, s>>>(d);
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("stream: %.2f us per launch\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  }
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < N; i++) tiny<<<1, 64, 0, s>>>(d);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("graph : %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / N);
  }
  // ten-kernel graph replayed twenty times (a level's worth per launch)
  hipGraph_t g2; hipGraphExec_t ge2;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 10; i++) tiny<<<1, 64, 0, s>>>(d);
  hipStreamEndCapture(s, &g2);
  hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0);
  for (int rep = 0; rep < 3; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; i++) hipGraphLaunch(ge2, s);
    hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    printf("graph of 10 x 20: %.2f us per kernel\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 200);
  }
  int h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost); printf("count %d\n", h);
  return 0;
}
