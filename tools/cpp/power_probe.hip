// dev experiment: does a background HBM write stream (few workgroups) slow a pure-VALU kernel on the other CUs' issue ports?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_fma(float *o, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fmaf(a0, 1.0001f, 0.5f), a1 = __builtin_fmaf(a1, 1.0001f, 0.5f), a2 = __builtin_fmaf(a2, 1.0001f, 0.5f), a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
    a4 = __builtin_fmaf(a4, 1.0001f, 0.5f), a5 = __builtin_fmaf(a5, 1.0001f, 0.5f), a6 = __builtin_fmaf(a6, 1.0001f, 0.5f), a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k_store(f4 *p, size_t n4, int reps) { // persistent streaming writer
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (int r = 0; r < reps; ++r)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(v, p + i);
}
int main() {
  float *o; hipMalloc(&o, 4096 * 256 * 4);
  f4 *buf; size_t bytes = 4ull << 30; hipMalloc(&buf, bytes);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t e0, e1, w0, w1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&w0); hipEventCreate(&w1);
  const int iters = 8192;
  for (int wgs : {0, 32, 64, 128, 256, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      if (wgs) { hipEventRecord(w0, sb); k_store<<<wgs, 256, 0, sb>>>(buf, bytes / 16, 2); hipEventRecord(w1, sb); }
      hipEventRecord(e0, sa);
      k_fma<<<4096, 256, 0, sa>>>(o, iters);
      hipEventRecord(e1, sa);
      hipDeviceSynchronize();
      float ms, wms = 0; hipEventElapsedTime(&ms, e0, e1); if (wgs) hipEventElapsedTime(&wms, w0, w1);
      if (rep) printf("writer WGs %4d: k_fma %.3f ms (%.1f Gwave-instr/s)   writer %.2f ms = %.2f TB/s\n", wgs, ms, 4096.0 * 4 * iters * 8 / ms / 1e6, wms, wms > 0 ? 2.0 * bytes / wms / 1e9 : 0.0);
    }
  }
  return 0;
}
