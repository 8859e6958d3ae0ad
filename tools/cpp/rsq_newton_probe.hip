// dev probe: can 1 / RN(sqrt(d)) (glm::inversesqrt's two roundings) be had from ONE transcendental?  The kernels use
// rcp_core(sqrt_core(d)) = v_rsq + 4 ops, then v_rcp + 2 ops.  Candidate: the Newton step of the reciprocal started from the v_rsq
// value already at hand (r ~ 1/sqrt(d) ~ 1/s) instead of v_rcp(s).  Exhaustive over every positive binary32 d in [2^-100, 2^100].
// build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o build/rsq_newton_probe tools/cpp/rsq_newton_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
__device__ __forceinline__ float rcp_core(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float sqrt_core(float x, float &r) {
  r = __builtin_amdgcn_rsqf(x);
  float s = x * r, h = 0.5f * r;
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
__global__ void k(unsigned long long *out) {
  unsigned long long n = 0, bad1 = 0, bad2 = 0, bad3 = 0, bad_ieee = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < (1ull << 31); i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t b = (uint32_t)i;
    if ((((b >> 23) & 0xffu) - 27u) > 200u) continue;
    const float d = __builtin_bit_cast(float, b);
    float r;
    const float s = sqrt_core(d, r);
    const float ref = rcp_core(s);
    const float ieee = 1.0f / __builtin_sqrtf(d);
    bad_ieee += __builtin_bit_cast(uint32_t, ref) != __builtin_bit_cast(uint32_t, ieee);
    // candidate 1: one Newton step from r
    const float c1 = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
    bad1 += __builtin_bit_cast(uint32_t, c1) != __builtin_bit_cast(uint32_t, ref);
    if (__builtin_bit_cast(uint32_t, c1) != __builtin_bit_cast(uint32_t, ref)) {
      const unsigned long long slot = atomicAdd(&out[5], 1ull);
      if (slot < 8) out[6 + slot] = ((unsigned long long)b << 32) | __builtin_bit_cast(uint32_t, c1);
    }
    // candidate 2: two Newton steps from r (no second transcendental, 4 ops)
    const float c2 = __builtin_fmaf(__builtin_fmaf(-s, c1, 1.0f), c1, c1);
    bad2 += __builtin_bit_cast(uint32_t, c2) != __builtin_bit_cast(uint32_t, ref);
    // candidate 4: SECOND-ORDER step from r: 1/s = r (1 + e + e^2 + ...), e = 1 - s r
    {
      const float e4 = __builtin_fmaf(-s, r, 1.0f), q4 = __builtin_fmaf(e4, e4, e4);
      const float c4 = __builtin_fmaf(q4, r, r);
      if (__builtin_bit_cast(uint32_t, c4) != __builtin_bit_cast(uint32_t, ref)) atomicAdd(&out[14], 1ull);
    }
    // candidates 5 / 6: one Newton step from the v_rsq value nudged one ulp up / down (off the exact tie of the all-ones roots)
    {
      const float ru = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, r) + 1u), rd = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, r) - 1u);
      const float c5 = __builtin_fmaf(__builtin_fmaf(-s, ru, 1.0f), ru, ru), c6 = __builtin_fmaf(__builtin_fmaf(-s, rd, 1.0f), rd, rd);
      if (__builtin_bit_cast(uint32_t, c5) != __builtin_bit_cast(uint32_t, ref)) atomicAdd(&out[15], 1ull);
      if (__builtin_bit_cast(uint32_t, c6) != __builtin_bit_cast(uint32_t, ref)) atomicAdd(&out[16], 1ull);
    }
    // candidate 3: Markstein-style final correction: q = c1; q = fma(fma(-s, q, 1), c1, q)  (same as c2) — and the residual form
    const float e = __builtin_fmaf(-s, c1, 1.0f);
    const float c3 = __builtin_fmaf(e, r, c1);
    bad3 += __builtin_bit_cast(uint32_t, c3) != __builtin_bit_cast(uint32_t, ref);
    ++n;
  }
  atomicAdd(&out[0], n), atomicAdd(&out[1], bad1), atomicAdd(&out[2], bad2), atomicAdd(&out[3], bad3), atomicAdd(&out[4], bad_ieee);
}
int main() {
  unsigned long long *d, h[17] = {0};
  (void)hipMalloc(&d, sizeof h);
  (void)hipMemset(d, 0, sizeof h);
  hipLaunchKernelGGL(k, dim3(8192), dim3(256), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) { uint32_t db = (uint32_t)(h[6 + i] >> 32), cb = (uint32_t)h[6 + i]; float df, cf; memcpy(&df, &db, 4); memcpy(&cf, &cb, 4); float sf = sqrtf(df); printf("  d=%08x (%a) s=%a cand=%a ieee=%a\n", db, df, sf, cf, 1.0f / sf); }
  printf("second-order step from rsq: %llu mismatches; one step from rsq + 1 ulp: %llu, from rsq - 1 ulp: %llu\n", h[14], h[15], h[16]);
  printf("operands %llu  mismatches: 1 Newton step from rsq %llu, 2 steps %llu, residual form %llu  (reference vs IEEE %llu)\n", h[0], h[1], h[2], h[3], h[4]);
  return 0;
}
