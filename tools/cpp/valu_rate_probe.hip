// Dev tool: VALU issue rate of one SIMD against the number of resident waves (settles the denominator of valu_frac).
// build: hipcc -O3 --offload-arch=gfx950 -o build/valu_rate_probe tools/cpp/valu_rate_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int KIND> __global__ void k(float *out, int iters) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float b = 1.0000001f, c = 1e-9f;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) { // 8 independent v_fma_f32 per iteration x 8
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a0 = __builtin_fmaf(a0, b, c), a1 = __builtin_fmaf(a1, b, c), a2 = __builtin_fmaf(a2, b, c), a3 = __builtin_fmaf(a3, b, c);
        a4 = __builtin_fmaf(a4, b, c), a5 = __builtin_fmaf(a5, b, c), a6 = __builtin_fmaf(a6, b, c), a7 = __builtin_fmaf(a7, b, c);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      }
    } else if (KIND == 1) { // one dependent chain of v_fma_f32
#pragma unroll
      for (int u = 0; u < 64; ++u) { a0 = __builtin_fmaf(a0, b, c); asm volatile("" : "+v"(a0)); }
    } else if (KIND == 2) { // f64 multiplies, 4 independent
#pragma unroll
      for (int u = 0; u < 16; ++u) { d0 *= 1.0000001, d1 *= 1.0000001, d2 *= 1.0000001, d3 *= 1.0000001; asm volatile("" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)); }
    } else if (KIND == 3) { // packed f32 fma, 4 independent pairs
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
      const f2 bb = {b, b}, cc = {c, c};
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        p0 = __builtin_elementwise_fma(p0, bb, cc), p1 = __builtin_elementwise_fma(p1, bb, cc), p2 = __builtin_elementwise_fma(p2, bb, cc), p3 = __builtin_elementwise_fma(p3, bb, cc);
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
      }
      a0 = p0.x, a1 = p0.y, a2 = p1.x, a3 = p1.y, a4 = p2.x, a5 = p2.y, a6 = p3.x, a7 = p3.y;
    } else if (KIND == 4) { // v_rcp_f32, 4 independent
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a0 = __builtin_amdgcn_rcpf(a0), a1 = __builtin_amdgcn_rcpf(a1), a2 = __builtin_amdgcn_rcpf(a2), a3 = __builtin_amdgcn_rcpf(a3);
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}
template <int KIND> void run(const char *name, float *d, int per_iter) {
  const int iters = 20000;
  for (int wps : {1, 2, 3, 4, 6, 8}) { // waves per SIMD: one workgroup of 256 * wps threads per CU (4 SIMDs)
    const int threads = 256, blocks = 256 * wps; // wps workgroups of 4 waves per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    k<KIND><<<blocks, threads>>>(d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, threads>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * per_iter * wps; // wave-instructions issued on one SIMD
    printf("%-14s waves/SIMD %d: %.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, wps, ms, ms * 1e-3 * 2.4e9 / inst_per_simd);
  }
}
int main() {
  float *d;
  hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
  run<0>("fma_f32 x8", d, 64);
  run<1>("fma_f32 chain", d, 64);
  run<2>("mul_f64 x4", d, 64);
  run<3>("pk_fma_f32 x4", d, 64);
  run<4>("rcp_f32 x4", d, 64);
  return 0;
}
