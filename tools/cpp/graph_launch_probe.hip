// dev probe: host cost and end-to-end time of 7 small dependent kernels, launched one by one vs replayed as a hipGraph.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/graph_launch tools/cpp/graph_launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k_small(float *p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  float *d;
  const int n = 1 << 20;
  (void)hipMalloc(&d, n * 4);
  (void)hipMemset(d, 0, n * 4);
  hipStream_t s;
  (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  auto seven = [&] {
    for (int k = 0; k < 7; ++k) hipLaunchKernelGGL(k_small, dim3(n / 256), dim3(256), 0, s, d, n);
  };
  for (int i = 0; i < 50; ++i) seven();
  (void)hipStreamSynchronize(s);
  const int iters = 1000;
  double sub = 0, tot = 0;
  for (int i = 0; i < iters; ++i) {
    double t0 = now_us();
    seven();
    double t1 = now_us();
    (void)hipStreamSynchronize(s);
    double t2 = now_us();
    sub += t1 - t0, tot += t2 - t0;
  }
  printf("eager : submit %.1f us, complete %.1f us per 7 kernels\n", sub / iters, tot / iters);
  hipGraph_t g;
  hipGraphExec_t ge;
  (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  seven();
  (void)hipStreamEndCapture(s, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 50; ++i) (void)hipGraphLaunch(ge, s);
  (void)hipStreamSynchronize(s);
  sub = tot = 0;
  for (int i = 0; i < iters; ++i) {
    double t0 = now_us();
    (void)hipGraphLaunch(ge, s);
    double t1 = now_us();
    (void)hipStreamSynchronize(s);
    double t2 = now_us();
    sub += t1 - t0, tot += t2 - t0;
  }
  printf("graph : submit %.1f us, complete %.1f us per 7 kernels\n", sub / iters, tot / iters);
  return 0;
}
