// dev probe: HBM write bandwidth of the framebuffer's store patterns.  4 planes x 256 frames of 1024^2 floats (4.29 GB):
//   rows   a wave-instruction stores 1 KiB of ONE framebuffer row (k_clear's pattern), workgroup = one 32-row band
//   tiles  a wave-instruction stores 8 rows x 128 B of one 32x32 tile (k_shade / k_raster write-out), workgroup = one tile
//   tilesW the same for tiles 64 / 128 pixels wide (256 / 512-byte pieces)
// Build: hipcc --offload-arch=gfx950 -O3 -o build/store_pattern tools/cpp/store_pattern_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr int W = 1024, H = 1024, F = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st(float *p, float4 v) {
  f32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(p));
}
__global__ __launch_bounds__(256) void k_rows(float *out, int n_items) { // item = (frame, band): 32 rows x W, 4 planes
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int f = it / (H / 32), b = it % (H / 32);
    float *base = out + (size_t)f * 4 * W * H + (size_t)b * 32 * W;
    for (int ly = 0; ly < 32; ++ly)
      for (int pl = 0; pl < 4; ++pl) st(base + (size_t)pl * W * H + (size_t)ly * W + threadIdx.x * 4, v);
  }
}
__global__ __launch_bounds__(256) void k_rows_pm(float *out, int n_items) { // the same bytes, plane after plane within the band
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int f = it / (H / 32), b = it % (H / 32);
    float *base = out + (size_t)f * 4 * W * H + (size_t)b * 32 * W;
    for (int pl = 0; pl < 4; ++pl)
      for (int ly = 0; ly < 32; ++ly) st(base + (size_t)pl * W * H + (size_t)ly * W + threadIdx.x * 4, v);
  }
}
__global__ __launch_bounds__(256) void k_flat(float *out, size_t n16) { // a flat fill: consecutive 16-byte units, grid-stride
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += gridDim.x * 256ull) st(out + i * 4, v);
}
__global__ __launch_bounds__(256) void k_flat_chunk(float *out, size_t n16) { // flat fill, each workgroup a contiguous 128 KiB run at a time
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  const size_t chunks = n16 / 8192;
  for (size_t c = blockIdx.x; c < chunks; c += gridDim.x)
    for (int k = 0; k < 32; ++k) st(out + (c * 8192 + k * 256 + threadIdx.x) * 4, v);
}
template <int TW> // tile = 32 rows x TW pixels; 256 threads: each stores 16 B; a row of the tile = TW*4 bytes
__global__ __launch_bounds__(256) void k_tiles(float *out, int n_items, int planes) {
  const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
  constexpr int TPR = TW / 4, ROWS_PER_PASS = 256 / TPR, TX = W / TW;
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int f = it / (TX * (H / 32)), t = it % (TX * (H / 32)), ty = t / TX, tx = t % TX;
    float *base = out + (size_t)f * 4 * W * H + (size_t)ty * 32 * W + tx * TW;
    for (int r0 = 0; r0 < 32; r0 += ROWS_PER_PASS) {
      const int ly = r0 + threadIdx.x / TPR, lx = (threadIdx.x % TPR) * 4;
      for (int pl = 0; pl < planes; ++pl) st(base + (size_t)pl * W * H + (size_t)ly * W + lx, v);
    }
  }
}
int main() {
  float *d;
  const size_t bytes = (size_t)F * 4 * W * H * 4;
  if (hipMalloc(&d, bytes) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  auto run = [&](const char *name, auto launch, double gb) {
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  %.2f TB/s\n", name, ms / 10, gb / (ms / 10) / 1e9 * 1e3 / 1e3);
  };
  const double gb = (double)bytes;
  for (int grid : {64, 256, 2048, 8192}) {
    char nm[64];
    snprintf(nm, sizeof nm, "rows  grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_rows, dim3(grid), dim3(256), 0, 0, d, F * (H / 32)); }, gb);
  }
  for (int grid : {64, 2048}) {
    char nm[64];
    snprintf(nm, sizeof nm, "rows plane-major grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_rows_pm, dim3(grid), dim3(256), 0, 0, d, F * (H / 32)); }, gb);
    snprintf(nm, sizeof nm, "flat grid-stride grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, d, bytes / 16); }, gb);
    snprintf(nm, sizeof nm, "flat 128K runs grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_flat_chunk, dim3(grid), dim3(256), 0, 0, d, bytes / 16); }, gb);
  }
  for (int grid : {1280, 16384}) {
    char nm[64];
    snprintf(nm, sizeof nm, "tiles 32px  grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_tiles<32>, dim3(grid), dim3(256), 0, 0, d, F * 32 * 32, 4); }, gb);
    snprintf(nm, sizeof nm, "tiles 64px  grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_tiles<64>, dim3(grid), dim3(256), 0, 0, d, F * 16 * 32, 4); }, gb);
    snprintf(nm, sizeof nm, "tiles 128px grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_tiles<128>, dim3(grid), dim3(256), 0, 0, d, F * 8 * 32, 4); }, gb);
    snprintf(nm, sizeof nm, "tiles 256px grid %d", grid);
    run(nm, [&] { hipLaunchKernelGGL(k_tiles<256>, dim3(grid), dim3(256), 0, 0, d, F * 4 * 32, 4); }, gb);
  }
  (void)hipFree(d);
  return 0;
}
