// dev experiment (not part of the library): candidate exact sequences checked on the device
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#pragma clang fp contract(off)
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ bool fast_pos(float x) { return ((f2u(x) >> 23) - 27u) <= 200u; }
__device__ __forceinline__ float rcp_core(float x) { float r = __builtin_amdgcn_rcpf(x); return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r); }
__device__ __forceinline__ float sqrtA(float x) { // rsq only
  float r = __builtin_amdgcn_rsqf(x), s = x * r, h = 0.5f * r;
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__device__ __forceinline__ float sqrtB(float x) { // rsq only, two residual steps
  float r = __builtin_amdgcn_rsqf(x), s = x * r, h = 0.5f * r;
  s = __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__global__ void k_sqrt(unsigned long long *out) {
  unsigned long long bA = 0, bB = 0, bC = 0, n = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (uint32_t)i);
    if (!fast_pos(x)) continue;
    ++n;
    const float ref = __builtin_sqrtf(x);
    bA += f2u(sqrtA(x)) != f2u(ref);
    bB += f2u(sqrtB(x)) != f2u(ref);
    bC += f2u(rcp_core(sqrtA(x))) != f2u(1.0f / ref);
  }
  atomicAdd(&out[0], n); atomicAdd(&out[1], bA); atomicAdd(&out[2], bB); atomicAdd(&out[3], bC);
}
// Markstein division with y = RN(1/b)
__device__ __forceinline__ float div_y(float a, float b, float y) {
  float q = a * y;
  q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
}
__device__ __forceinline__ float div_y1(float a, float b, float y) { // one step only
  float q = a * y;
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
}
__device__ __forceinline__ uint64_t mix(uint64_t z) { z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
__global__ void k_div(unsigned long long *out, uint64_t per_thread, int mode) {
  unsigned long long b2 = 0, b1 = 0, n = 0;
  uint64_t st = mix(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x + 12345ull * mode);
  for (uint64_t i = 0; i < per_thread; ++i) {
    st = mix(st);
    uint32_t ua = (uint32_t)st, ub = (uint32_t)(st >> 32);
    // exponents within +-60 of the bias, random sign and mantissa
    uint32_t ea = 127 - 60 + ((ua >> 23) & 0xff) % 121, eb = 127 - 60 + ((ub >> 23) & 0xff) % 121;
    if (mode == 1) { ea = 127 + (ea & 7); eb = 127 + (eb & 3); }           // near 1: many exactly representable quotients
    if (mode == 2) { ua &= 0x807fffffu & ~0x7ff000u; ub &= ~0x7fff00u; }   // few mantissa bits
    ua = (ua & 0x807fffffu) | (ea << 23), ub = (ub & 0x807fffffu) | (eb << 23);
    float a = __builtin_bit_cast(float, ua), b = __builtin_bit_cast(float, ub);
    if (mode == 3) { a = (float)(ua & 0xff); b = 255.0f; }                 // texel / 255
    const float y = rcp_core(b), ref = a / b;
    b2 += f2u(div_y(a, b, y)) != f2u(ref);
    b1 += f2u(div_y1(a, b, y)) != f2u(ref);
    ++n;
  }
  atomicAdd(&out[0], n); atomicAdd(&out[1], b2); atomicAdd(&out[2], b1);
}
// issue-rate microbenchmark: dependent-free chains of v_rcp vs v_fma vs v_sqrt vs v_mul_f64 vs v_pk_fma
template <int OP> __global__ void k_rate(float *o, int iters) {
  float a0 = threadIdx.x * 1e-3f + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
  for (int i = 0; i < iters; ++i) {
#define R8(F) a0 = F(a0), a1 = F(a1), a2 = F(a2), a3 = F(a3), a4 = F(a4), a5 = F(a5), a6 = F(a6), a7 = F(a7);
    if (OP == 0) { R8(__builtin_amdgcn_rcpf) }
    if (OP == 1) { a0 = __builtin_fmaf(a0, 1.0001f, 0.5f), a1 = __builtin_fmaf(a1, 1.0001f, 0.5f), a2 = __builtin_fmaf(a2, 1.0001f, 0.5f), a3 = __builtin_fmaf(a3, 1.0001f, 0.5f), a4 = __builtin_fmaf(a4, 1.0001f, 0.5f), a5 = __builtin_fmaf(a5, 1.0001f, 0.5f), a6 = __builtin_fmaf(a6, 1.0001f, 0.5f), a7 = __builtin_fmaf(a7, 1.0001f, 0.5f); }
    if (OP == 2) { R8(__builtin_amdgcn_sqrtf) }
    if (OP == 3) { d0 = d0 * 1.0000001, d1 = d1 * 1.0000001, d2 = d2 * 1.0000001, d3 = d3 * 1.0000001; d0 = d0 * 0.9999999, d1 = d1 * 0.9999999, d2 = d2 * 0.9999999, d3 = d3 * 0.9999999; }
    if (OP == 4) { R8(__builtin_amdgcn_rsqf) }
  }
  o[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3);
}
int main() {
  unsigned long long *d, h[4];
  hipMalloc(&d, 32);
  hipMemset(d, 0, 32);
  k_sqrt<<<8192, 256>>>(d);
  hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
  printf("sqrt: n=%llu  rsq-only mismatches=%llu  two-step=%llu  rcp(sqrtA)=%llu\n", h[0], h[1], h[2], h[3]);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(d, 0, 32);
    k_div<<<8192, 256>>>(d, 8192, mode);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("div mode %d: n=%llu  markstein(2 steps) mismatches=%llu  1 step=%llu\n", mode, h[0], h[1], h[2]);
  }
  float *o; hipMalloc(&o, 4 * 1024 * 256 * 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[5] = {"v_rcp_f32", "v_fma_f32", "v_sqrt_f32", "v_mul_f64", "v_rsq_f32"};
  for (int op = 0; op < 5; ++op) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      const int iters = 4096;
      if (op == 0) k_rate<0><<<4096, 256>>>(o, iters);
      if (op == 1) k_rate<1><<<4096, 256>>>(o, iters);
      if (op == 2) k_rate<2><<<4096, 256>>>(o, iters);
      if (op == 3) k_rate<3><<<4096, 256>>>(o, iters);
      if (op == 4) k_rate<4><<<4096, 256>>>(o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.3f ms for %d x 8 ops x 4096 WGs x 4 waves -> %.1f Gwave-instr/s\n", names[op], ms, iters, 4096.0 * 4 * iters * 8 / ms / 1e6);
    }
  }
  return 0;
}
