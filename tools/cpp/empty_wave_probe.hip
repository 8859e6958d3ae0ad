// Dev tool: what do workgroups that leave after one load cost?  (k_raster: three tiles in four of a batch are empty)
// build: hipcc -O3 --offload-arch=gfx950 -o build/empty_wave_probe tools/cpp/empty_wave_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void k64(const uint2 *info, float *out) {
  __shared__ unsigned long long s[1056]; // (the LDS allocation of k_raster<1>)
  const uint2 t = info[blockIdx.x];
  if (t.x == 0u) return;
  s[threadIdx.x] = t.y;
  __builtin_amdgcn_wave_barrier();
  out[blockIdx.x * 64 + threadIdx.x] = (float)s[63 - threadIdx.x];
}
int main() {
  for (int n : {262144, 1048576}) {
    uint2 *info;
    float *out;
    (void)hipMalloc(&info, n * sizeof(uint2));
    (void)hipMalloc(&out, (size_t)n * 64 * sizeof(float));
    (void)hipMemset(info, 0, n * sizeof(uint2));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      for (int i = 0; i < 10; ++i) k64<<<n, 64>>>(info, out);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("%d workgroups of one wave, all leaving after one load: %.1f us per launch = %.2f ns per workgroup\n", n, ms * 100, ms * 1e5 / n);
    }
    (void)hipFree(info), (void)hipFree(out);
  }
  return 0;
}
