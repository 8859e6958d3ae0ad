// dev probe: the wave-level primitives k_raster / k_bin rely on (DPP scans, ds_bpermute, ds_min_u64), checked against
// host code.  Build: hipcc --offload-arch=gfx950 -O3 -o build/wave_prims tools/cpp/wave_prims_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define SRZ_DPP(v, ctrl, rmask) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, false)
__device__ uint32_t wave_scan_add(uint32_t v) {
  v += SRZ_DPP(v, 0x111, 0xf), v += SRZ_DPP(v, 0x112, 0xf), v += SRZ_DPP(v, 0x114, 0xf), v += SRZ_DPP(v, 0x118, 0xf);
  v += SRZ_DPP(v, 0x142, 0xa);
  v += SRZ_DPP(v, 0x143, 0xc);
  return v;
}
__device__ uint32_t wave_scan_max(uint32_t v) {
  v = max(v, SRZ_DPP(v, 0x111, 0xf)), v = max(v, SRZ_DPP(v, 0x112, 0xf)), v = max(v, SRZ_DPP(v, 0x114, 0xf));
  v = max(v, SRZ_DPP(v, 0x118, 0xf));
  v = max(v, SRZ_DPP(v, 0x142, 0xa));
  v = max(v, SRZ_DPP(v, 0x143, 0xc));
  return v;
}
__global__ void k(const uint32_t *in, uint32_t *out) {
  __shared__ unsigned long long s[64];
  const int lane = threadIdx.x;
  const uint32_t v = in[lane];
  out[lane] = wave_scan_add(v);
  out[64 + lane] = wave_scan_max(v);
  out[128 + lane] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((v & 63u) * 4u), (int)(v * 3u));
  s[lane] = ~0ull;
  __builtin_amdgcn_wave_barrier();
  if (v & 1u) __hip_atomic_fetch_min(&s[v & 7u], ((unsigned long long)v << 32) | (uint32_t)lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  __builtin_amdgcn_wave_barrier();
  out[192 + lane] = (uint32_t)(s[lane] >> 32);
  out[256 + lane] = (uint32_t)s[lane];
}
int main() {
  uint32_t h[64], o[320], *di, *dout;
  uint32_t st = 12345;
  for (int i = 0; i < 64; ++i) st = st * 1664525u + 1013904223u, h[i] = (st >> 8) % 1000u;
  hipMalloc(&di, sizeof h), hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  int bad = 0;
  uint32_t run = 0, mx = 0;
  unsigned long long s[64];
  for (int i = 0; i < 64; ++i) s[i] = ~0ull;
  for (int i = 0; i < 64; ++i) {
    run += h[i], mx = h[i] > mx ? h[i] : mx;
    if (o[i] != run) bad++, printf("add scan lane %d: %u != %u\n", i, o[i], run);
    if (o[64 + i] != mx) bad++, printf("max scan lane %d: %u != %u\n", i, o[64 + i], mx);
    if (o[128 + i] != h[h[i] & 63] * 3u) bad++, printf("bpermute lane %d\n", i);
    if (h[i] & 1u) {
      unsigned long long key = ((unsigned long long)h[i] << 32) | (uint32_t)i;
      if (key < s[h[i] & 7]) s[h[i] & 7] = key;
    }
  }
  for (int i = 0; i < 64; ++i)
    if (o[192 + i] != (uint32_t)(s[i] >> 32) || o[256 + i] != (uint32_t)s[i]) bad++, printf("ds_min_u64 slot %d\n", i);
  printf("wave primitives: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
  return bad != 0;
}
