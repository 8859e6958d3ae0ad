// Dev probe: where do the 64 lanes of one global_load_lds_dwordx3 land in LDS?  (prints the LDS dword index of every source dword)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float *g, float *o) {
  __shared__ float s[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) s[i] = -1.0f;
  __syncthreads();
  __builtin_amdgcn_global_load_lds(g + threadIdx.x * 3, (__attribute__((address_space(3))) void *)s, 12, 0, 0);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) o[i] = s[i];
}
int main() {
  std::vector<float> h(256);
  for (int i = 0; i < 256; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, 1024), hipMalloc(&o, 4096);
  hipMemcpy(g, h.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, g, o);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost);
  for (int i = 0; i < 272; ++i) std::printf("%d%c", (int)r[i], (i % 16 == 15) ? '\n' : ' ');
  return 0;
}
