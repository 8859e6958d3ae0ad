// srz_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the raster + fragment-shade stage.
//
// Replaces, on the device, the hot loops of TraditionalRasterizer::draw (src/Rasterizer.cpp:183-499 of
// Liupeter01/Software-Rasterizer) and the fragment shaders they call (src/Shader.cpp, include/shader/Shader.hpp).
// Not a translation of the AVX2 code: the reference walks triangles serially and rows in parallel; here
//
//   k_setup   one thread per triangle      bbox (Triangle::calcBoundingBox) + backface test → 8-byte BBox record
//   k_bands   one WAVE per 32-row band     in-order scan of the BBox stream, ballot-compacted → per-band triangle list
//                                          (submission order preserved by construction: no atomics, no sort)
//   k_raster  one WAVE per 32x32 tile      tile z-buffer + owner-id planes in LDS; walks its band's list in order,
//                                          per triangle the 64 lanes cover 8x8 pixel blocks of bbox∩tile, run the
//                                          coverage + z-test with the reference's per-column semantics and update LDS
//                                          (a wave's LDS ops are ordered → "last writer in submission order wins"
//                                          needs no lock).  Then VISIBILITY-FIRST SHADING: each pixel's final owner is
//                                          shaded exactly once (the reference's shaders are pure functions of
//                                          (triangle,pixel), its write is an overwrite) and the four planes leave as
//                                          16-byte-per-lane non-temporal stores.  Clear is fused (LDS init).
//
// Numerics: every float op is the oracle's op in the oracle's order (oracle/srz_oracle.c): contraction is OFF,
// fused ops are explicit fmaf(), division and sqrt are the correctly rounded ones, pow is evaluated in binary64
// and rounded once.  The z-buffer is therefore expected to be bit-identical to the oracle's.
#include "srz_device.h"

#pragma clang fp contract(off)

namespace srz {

// ---- operand-order-exact min/max (SSE / std:: semantics, see oracle) ------------------------------------------
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_clamp(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
__device__ __forceinline__ float fmsubf(float a, float b, float c) { return __builtin_fmaf(a, b, -c); }
__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  float tx = ax * bx, ty = ay * by, tz = az * bz;
  return tx + ty + tz;
}
// glm::normalize: v * (1/sqrt(dot(v,v)))
__device__ __forceinline__ void normalize3(float &x, float &y, float &z) {
  float is = 1.0f / __builtin_sqrtf(dot3(x, y, z, x, y, z));
  x = x * is, y = y * is, z = z * is;
}
// NormalSIMD::normalized (src/Tools.cpp:13-24)
__device__ __forceinline__ void v_normalized(float &x, float &y, float &z) {
  float len = __builtin_sqrtf(fmaf_(x, x, fmaf_(y, y, z * z)));
  if (len > 0.0f) {
    float inv = 1.0f / len;
    x = x * inv, y = y * inv, z = z * inv;
  } else {
    x = y = z = 0.0f;
  }
}

// pow evaluated in binary64 and rounded once to binary32 (== correctly rounded powf in all but ~1e-7 of cases).
// p is a per-frame constant, so the branch is wave-uniform.
__device__ __forceinline__ float pow_cr(float x, float p) {
  if (p == __builtin_truncf(p) && p >= 0.0f && p <= 1048576.0f) {
    unsigned n = (unsigned)p;
    double b = (double)x, r = 1.0;
    while (n) {
      if (n & 1u) r = r * b;
      b = b * b;
      n >>= 1;
    }
    return (float)r;
  }
  return (float)pow((double)x, (double)p);
}

__device__ __forceinline__ int32_t cvt_rne_i32(float f) {
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
  return (int32_t)__builtin_rintf(f);
}

// ---- per-triangle constants used by both the coverage test and the shader -----------------------------------
struct TriXY {
  float ax, ay, bx, by, cx, cy, z0, z1, z2;
  float v_inv;  // 1 / fmsub(ABx,ACy,ACx*ABy)   — "V" (8-wide) path, src/Rasterizer.cpp:111-112
  float s_area; // ABx*ACy - ABy*ACx              — "S" (scalar tail) path, src/Rasterizer.cpp:61
};
__device__ __forceinline__ void tri_consts(TriXY &t) {
  float ABx = t.bx - t.ax, ABy = t.by - t.ay, ACx = t.cx - t.ax, ACy = t.cy - t.ay;
  t.v_inv = 1.0f / fmsubf(ABx, ACy, ACx * ABy);
  t.s_area = ABx * ACy - ABy * ACx;
}

// "V" semantics: barycentric(__m256) + inside mask + z (src/Rasterizer.cpp:89-127,310-326)
__device__ __forceinline__ bool cover_v(const TriXY &t, float fx, float fy, float &alpha, float &beta, float &gamma,
                                        float &z) {
  float PBx = t.bx - fx, PBy = t.by - fy, PCx = t.cx - fx, PCy = t.cy - fy, PAx = t.ax - fx, PAy = t.ay - fy;
  float aPBC = fmsubf(PBx, PCy, PCx * PBy), aPCA = fmsubf(PCx, PAy, PAx * PCy);
  alpha = aPBC * t.v_inv, beta = aPCA * t.v_inv, gamma = 1.0f - (alpha + beta);
  z = fmaf_(alpha, t.z0, fmaf_(beta, t.z1, gamma * t.z2));
  return alpha > 0.0f && alpha < 1.0f && beta > 0.0f && beta < 1.0f && gamma > 0.0f && gamma < 1.0f;
}
// "S" semantics: insideTriangle + barycentric(scalar) + z (src/Rasterizer.cpp:11-70,473)
__device__ __forceinline__ bool cover_s(const TriXY &t, float fx, float fy, float &alpha, float &beta, float &gamma,
                                        float &z) {
  float ABx = t.bx - t.ax, ABy = t.by - t.ay, BCx = t.cx - t.bx, BCy = t.cy - t.by, CAx = t.ax - t.cx, CAy = t.ay - t.cy;
  float APx = fx - t.ax, APy = fy - t.ay, BPx = fx - t.bx, BPy = fy - t.by, CPx = fx - t.cx, CPy = fy - t.cy;
  float e0 = ABx * APy - ABy * APx, e1 = BCx * BPy - BCy * BPx, e2 = CAx * CPy - CAy * CPx;
  bool inside = (e0 > 0 && e1 > 0 && e2 > 0) || (e0 < 0 && e1 < 0 && e2 < 0);
  float PAx = t.ax - fx, PAy = t.ay - fy, PBx = t.bx - fx, PBy = t.by - fy, PCx = t.cx - fx, PCy = t.cy - fy;
  float aPBC = PBx * PCy - PBy * PCx, aPCA = PCx * PAy - PCy * PAx;
  alpha = aPBC / t.s_area, beta = aPCA / t.s_area, gamma = 1.0f - alpha - beta;
  z = alpha * t.z0 + beta * t.z1 + gamma * t.z2;
  return inside;
}

// ================================================================================================================
// k_setup — per triangle: finite check, bbox, backface test (src/Triangle.cpp:147-151,243-257; Rasterizer.cpp:203)
// ================================================================================================================
template <bool STATS>
__global__ __launch_bounds__(256) void k_setup(RenderArgs a, BBox *bbox_out) {
  const FrameDesc &fd = a.frames[blockIdx.y];
  const int W = fd.width, H = fd.height;
  const float ex = fd.eye[0], ey = fd.eye[1], ez = fd.eye[2];
  unsigned long long n_culled = 0, tests = 0;
  for (uint32_t t = blockIdx.x * 256 + threadIdx.x; t < fd.n_tris; t += gridDim.x * 256) {
    const float *p = &a.tris[fd.tri_off + t].pos[0][0];
    float A0 = p[0], A1 = p[1], A2 = p[2], B0 = p[3], B1 = p[4], B2 = p[5], C0 = p[6], C1 = p[7], C2 = p[8];
    BBox bb;
    bb.sx = 1, bb.sy = 1, bb.ex = 0, bb.ey = 0;
    bool finite = __builtin_isfinite(A0) && __builtin_isfinite(A1) && __builtin_isfinite(A2) && __builtin_isfinite(B0) &&
                  __builtin_isfinite(B1) && __builtin_isfinite(B2) && __builtin_isfinite(C0) && __builtin_isfinite(C1) &&
                  __builtin_isfinite(C2);
    bool keep = false;
    if (finite) {
      float e1x = B0 - A0, e1y = B1 - A1, e1z = B2 - A2, e2x = C0 - A0, e2y = C1 - A1, e2z = C2 - A2;
      float nx = e1y * e2z - e2y * e1z, ny = e1z * e2x - e2z * e1x, nz = e1x * e2y - e2x * e1y;
      normalize3(nx, ny, nz);
      keep = !(dot3(nx, ny, nz, ex, ey, ez) > 0.0f);
    }
    if (keep) {
      float mnx = A0, mxx = A0, mny = A1, mxy = A1;
      if (B0 < mnx) mnx = B0;
      if (C0 < mnx) mnx = C0;
      if (mxx < B0) mxx = B0;
      if (mxx < C0) mxx = C0;
      if (B1 < mny) mny = B1;
      if (C1 < mny) mny = C1;
      if (mxy < B1) mxy = B1;
      if (mxy < C1) mxy = C1;
      // clamp(trunc(v),0,W-1) == trunc(clamp(v,0,W-1)) for every finite v
      bb.sx = (int16_t)(int)std_clamp(mnx, 0.0f, (float)(W - 1));
      bb.ex = (int16_t)(int)std_clamp(mxx, 0.0f, (float)(W - 1));
      bb.sy = (int16_t)(int)std_clamp(mny, 0.0f, (float)(H - 1));
      bb.ey = (int16_t)(int)std_clamp(mxy, 0.0f, (float)(H - 1));
      if (STATS) tests += (unsigned long long)(bb.ex - bb.sx + 1) * (unsigned long long)(bb.ey - bb.sy + 1);
    } else if (STATS) {
      n_culled++;
    }
    bbox_out[fd.tri_off + t] = bb;
  }
  if (STATS) {
    if (n_culled) atomicAdd(&a.stats[ST_CULLED], n_culled);
    if (tests) atomicAdd(&a.stats[ST_PIXEL_TESTS], tests);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.stats[ST_TRIS], (unsigned long long)fd.n_tris);
  }
}

// ================================================================================================================
// k_bands — one wave per (frame, local band): ordered ballot compaction of the triangles whose bbox touches the band
// ================================================================================================================
__global__ __launch_bounds__(256) void k_bands(RenderArgs a, uint32_t *band_lists, uint32_t *band_count) {
  const FrameDesc &fd = a.frames[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const uint32_t lb = blockIdx.x * WAVES_PER_WG + (threadIdx.x >> 6);
  if (lb >= fd.n_local_bands) return;
  const int band = (int)lb * a.shard_world + a.shard_rank;
  const int y0 = band * BAND, y1 = y0 + BAND - 1;
  uint32_t *list = band_lists + fd.list_off + (uint64_t)lb * fd.n_tris;
  const BBox *bbox = a.bbox + fd.tri_off;
  uint32_t cursor = 0;
  for (uint32_t base = 0; base < fd.n_tris; base += 64) {
    uint32_t t = base + lane;
    bool hit = false;
    if (t < fd.n_tris) {
      BBox bb = bbox[t];
      hit = bb.sx <= bb.ex && bb.sy <= y1 && bb.ey >= y0;
    }
    unsigned long long m = __ballot(hit);
    if (hit) list[cursor + __popcll(m & ((1ull << lane) - 1ull))] = t;
    cursor += (uint32_t)__popcll(m);
  }
  if (lane == 0) band_count[fd.count_off + lb] = cursor;
}

// ================================================================================================================
// Fragment shaders
// ================================================================================================================
struct ShadeEnv {
  const FrameDesc *fd;
  const srz_light *lights;
  const TexDesc *tex;
};

// BlinnPhong<__m256> for one light (include/shader/Shader.hpp:104-229)
__device__ __forceinline__ void v_blinn_phong(float nx, float ny, float nz, const float *ka, float kdr, float kdg, float kdb,
                                              const float *ks, const float *cam, const srz_light &L, float px, float py,
                                              float pz, float p, float &o0, float &o1, float &o2) {
  float lx = L.pos[0] - px, ly = L.pos[1] - py, lz = L.pos[2] - pz;
  float att = 1.0f / __builtin_sqrtf(fmaf_(lx, lx, ly * ly));
  float d0 = L.intensity[0] * att, d1 = L.intensity[1] * att, d2 = L.intensity[2] * att;
  float hx = lx + (cam[0] - px), hy = ly + (cam[1] - py), hz = lz + (cam[2] - pz);
  v_normalized(hx, hy, hz);
  float nlx = lx, nly = ly, nlz = lz;
  v_normalized(nlx, nly, nlz);
  float cosA = sse_max(0.0f, fmaf_(nlx, nx, fmaf_(nly, ny, nlz * nz)));
  float cosT = pow_cr(sse_max(0.0f, fmaf_(hx, nx, fmaf_(hy, ny, hz * nz))), p);
  o0 = kdr * fmaf_(ka[0], L.intensity[0], fmaf_(d0 * kdr, cosA, (d0 * ks[0]) * cosT));
  o1 = kdg * fmaf_(ka[1], L.intensity[1], fmaf_(d1 * kdg, cosA, (d1 * ks[1]) * cosT));
  o2 = kdb * fmaf_(ka[2], L.intensity[2], fmaf_(d2 * kdb, cosA, (d2 * ks[2]) * cosT));
}

// Shader::applyFragmentShader SIMD overload + simd_*_impl (src/Shader.cpp:128-386); colour out in [0,255]
__device__ __forceinline__ void v_shade(const ShadeEnv &env, int shader, const TexDesc &tx, float px, float py, float pz,
                                        float nx, float ny, float nz, float u, float v, float &r0, float &r1, float &r2) {
  const FrameDesc &fd = *env.fd;
  float c0 = 1.0f, c1 = 1.0f, c2 = 1.0f;
  if (shader == SRZ_SHADER_NORMAL) {
    c0 = (nx + 1.0f) * 0.5f, c1 = (ny + 1.0f) * 0.5f, c2 = (nz + 1.0f) * 0.5f;
  } else if (shader == SRZ_SHADER_TEXTURE || shader == SRZ_SHADER_PHONG) {
    float kd0 = 1.0f, kd1 = 1.0f, kd2 = 1.0f;
    if (shader == SRZ_SHADER_TEXTURE) {
      float tw = (float)tx.w, th = (float)tx.h;
      u = u * tw, v = v * th;
      u = sse_max(0.0f, sse_min(u, tw - 1.0f));
      v = sse_max(0.0f, sse_min(v, th - 1.0f));
      int32_t xi = cvt_rne_i32(u), yi = cvt_rne_i32(v);
      uint32_t texel = tx.bgrx[(size_t)yi * tx.w + xi];
      const float inv255 = 1.0f / 255.0f;
      kd0 = (float)(texel & 0xffu) * inv255, kd1 = (float)((texel >> 8) & 0xffu) * inv255,
      kd2 = (float)((texel >> 16) & 0xffu) * inv255;
    }
    c0 = c1 = c2 = 0.0f;
    for (uint32_t l = 0; l < fd.n_lights; ++l) {
      float o0, o1, o2;
      v_blinn_phong(nx, ny, nz, fd.ka, kd0, kd1, kd2, fd.ks, fd.eye, env.lights[l], px, py, pz, fd.p, o0, o1, o2);
      c0 = c0 + o0, c1 = c1 + o1, c2 = c2 + o2;
    }
  }
  // DISPLACEMENT / BUMP: the reference's SIMD versions are empty stubs (src/Shader.cpp:388-444) → (1,1,1) → 255
  r0 = sse_min(sse_max(c0, 0.0f), 1.0f) * 255.0f;
  r1 = sse_min(sse_max(c1, 0.0f), 1.0f) * 255.0f;
  r2 = sse_min(sse_max(c2, 0.0f), 1.0f) * 255.0f;
}

// TextureLoader::getTextureColor(vec2) (src/TextureLoader.cpp:14-31)
__device__ __forceinline__ void s_texel(const TexDesc &tx, float u, float v, float &o0, float &o1, float &o2) {
  float cu = std_clamp(u, 0.0f, 1.0f), cv = std_clamp(v, 0.0f, 1.0f);
  float fx = cu * (float)tx.w, fy = cv * (float)tx.h;
  int x = (int)fx, y = (int)fy;
  if (x < 0 || x >= tx.w || y < 0 || y >= tx.h) {
    o0 = o1 = o2 = 0.0f;
    return;
  }
  uint32_t texel = tx.bgrx[(size_t)y * tx.w + x];
  o0 = (float)(texel & 0xffu) / 255.0f, o1 = (float)((texel >> 8) & 0xffu) / 255.0f,
  o2 = (float)((texel >> 16) & 0xffu) / 255.0f;
}

// Shader::BlinnPhong scalar (src/Shader.cpp:510-543); the two std::pow(x,2) and the sqrt are binary64 there
__device__ __forceinline__ void s_blinn_phong(const float *cam, float px, float py, float pz, float nx, float ny, float nz,
                                              float kd0, float kd1, float kd2, const srz_light &L, const float *ka,
                                              const float *ks, float p, float &o0, float &o1, float &o2) {
  normalize3(nx, ny, nz);
  float ldx = L.pos[0] - px, ldy = L.pos[1] - py, ldz = L.pos[2] - pz;
  double dx = (double)(L.pos[0] - px), dy = (double)(L.pos[1] - py);
  float dsq = (float)__builtin_sqrt(dx * dx + dy * dy);
  float d0 = L.intensity[0] / dsq, d1 = L.intensity[1] / dsq, d2 = L.intensity[2] / dsq;
  float nlx = ldx, nly = ldy, nlz = ldz;
  normalize3(nlx, nly, nlz);
  float cosTheta = std_max(0.0f, dot3(nx, ny, nz, nlx, nly, nlz));
  float vx = cam[0] - px, vy = cam[1] - py, vz = cam[2] - pz;
  float hx = ldx + vx, hy = ldy + vy, hz = ldz + vz;
  normalize3(hx, hy, hz);
  float cosAlpha = std_max(0.0f, dot3(nx, ny, nz, hx, hy, hz));
  float pw = pow_cr(cosAlpha, p);
  o0 = ((ka[0] * L.intensity[0] + (cosTheta * kd0) * d0) + (pw * ks[0]) * d0) * kd0;
  o1 = ((ka[1] * L.intensity[1] + (cosTheta * kd1) * d1) + (pw * ks[1]) * d1) * kd1;
  o2 = ((ka[2] * L.intensity[2] + (cosTheta * kd2) * d2) + (pw * ks[2]) * d2) * kd2;
}

// calcBumpMapping / calcDisplacementMapping common part (src/Shader.cpp:447-507)
__device__ __forceinline__ void s_bump_common(const TexDesc &tx, float nx, float ny, float nz, float u, float v, float kh,
                                              float kn, float &ox, float &oy, float &oz, float &origin_norm) {
  float sq = __builtin_sqrtf(nx * nx + nz * nz);
  float t0 = (nx * ny) / sq, t1 = sq, t2 = (nz * ny) / sq;
  float b0 = ny * t2 - t1 * nz, b1 = nz * t0 - t2 * nx, b2 = nx * t1 - t0 * ny;
  float a0, a1, a2, u0, u1, u2, w0, w1, w2;
  s_texel(tx, u, v, a0, a1, a2);
  float on = __builtin_sqrtf(dot3(a0, a1, a2, a0, a1, a2));
  s_texel(tx, (u + 1.0f) / (float)tx.w, v, u0, u1, u2);
  s_texel(tx, u, (v + 1.0f) / (float)tx.h, w0, w1, w2);
  float dU = kh * kn * (__builtin_sqrtf(dot3(u0, u1, u2, u0, u1, u2)) - on);
  float dV = kh * kn * (__builtin_sqrtf(dot3(w0, w1, w2, w0, w1, w2)) - on);
  float l0 = -dU, l1 = -dV, l2 = 1.0f;
  ox = t0 * l0 + t1 * l1 + t2 * l2, oy = b0 * l0 + b1 * l1 + b2 * l2, oz = nx * l0 + ny * l1 + nz * l2;
  normalize3(ox, oy, oz);
  origin_norm = on;
}

// scalar applyFragmentShader + standard_*_impl + Tools::normalizedToRGB (src/Shader.cpp:547-640, src/Tools.cpp:94-104)
__device__ __forceinline__ void s_shade(const ShadeEnv &env, int shader, const TexDesc &tx, float px, float py, float pz,
                                        float nx, float ny, float nz, float u, float v, float &r0, float &r1, float &r2) {
  const FrameDesc &fd = *env.fd;
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  if (shader == SRZ_SHADER_NORMAL) {
    normalize3(nx, ny, nz);
    c0 = (nx + 1.0f) / 2.0f, c1 = (ny + 1.0f) / 2.0f, c2 = (nz + 1.0f) / 2.0f;
  } else if (shader >= SRZ_SHADER_TEXTURE && shader <= SRZ_SHADER_BUMP) {
    float kd0 = 1.0f, kd1 = 1.0f, kd2 = 1.0f;
    float sx = px, sy = py, sz = pz, snx = nx, sny = ny, snz = nz;
    if (shader != SRZ_SHADER_PHONG) s_texel(tx, u, v, kd0, kd1, kd2);
    if (shader == SRZ_SHADER_BUMP) {
      float on;
      s_bump_common(tx, nx, ny, nz, u, v, fd.kh, fd.kn, snx, sny, snz, on);
    } else if (shader == SRZ_SHADER_DISPLACEMENT) {
      float on;
      s_bump_common(tx, nx, ny, nz, u, v, fd.kh, fd.kn, snx, sny, snz, on);
      sx = px + (fd.kn * nx) * on, sy = py + (fd.kn * ny) * on, sz = pz + (fd.kn * nz) * on;
    }
    for (uint32_t l = 0; l < fd.n_lights; ++l) {
      float o0, o1, o2;
      s_blinn_phong(fd.eye, sx, sy, sz, snx, sny, snz, kd0, kd1, kd2, env.lights[l], fd.ka, fd.ks, fd.p, o0, o1, o2);
      c0 = c0 + o0, c1 = c1 + o1, c2 = c2 + o2;
    }
  }
  float q0 = std_clamp(c0, 0.0f, 1.0f) * 255.0f, q1 = std_clamp(c1, 0.0f, 1.0f) * 255.0f,
        q2 = std_clamp(c2, 0.0f, 1.0f) * 255.0f;
  r0 = (q0 == q0) ? (float)(uint32_t)q0 : 0.0f;
  r1 = (q1 == q1) ? (float)(uint32_t)q1 : 0.0f;
  r2 = (q2 == q2) ? (float)(uint32_t)q2 : 0.0f;
}

// Shade the final owner `idx` of pixel (x,y) with depth z.
__device__ __forceinline__ void shade_pixel(const RenderArgs &a, const ShadeEnv &env, uint32_t flags, uint32_t idx, int x,
                                            int y, float z, float &r0, float &r1, float &r2, bool &textured) {
  const FrameDesc &fd = *env.fd;
  const float4 *tp = reinterpret_cast<const float4 *>(a.tris + fd.tri_off + idx);
  float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3], q4 = tp[4], q5 = tp[5];
  // pos: q0.xyz q0.w q1.xy q1.zw q2.x | nrm: q2.yzw q3.xyz q3.w q4.xy | uv: q4.zw q5.xy q5.zw
  TriXY t;
  t.ax = q0.x, t.ay = q0.y, t.z0 = q0.z, t.bx = q0.w, t.by = q1.x, t.z1 = q1.y, t.cx = q1.z, t.cy = q1.w, t.z2 = q2.x;
  tri_consts(t);
  const float n0x = q2.y, n0y = q2.z, n0z = q2.w, n1x = q3.x, n1y = q3.y, n1z = q3.z, n2x = q3.w, n2y = q4.x, n2z = q4.y;
  const float u0 = q4.z, v0 = q4.w, u1 = q5.x, v1 = q5.y, u2 = q5.z, v2 = q5.w;
  const BBox bb = a.bbox[fd.tri_off + idx];
  const int vend = (flags & SRZ_UNIFIED) ? bb.ex + 1 : bb.sx + ((bb.ex - bb.sx + 1) & ~7);
  const BatchDesc bd = a.batches[fd.batch_off + a.tri_batch[fd.tri_off + idx]];
  TexDesc tx;
  tx.bgrx = nullptr, tx.w = 1, tx.h = 1;
  textured = bd.shader == SRZ_SHADER_TEXTURE || bd.shader == SRZ_SHADER_DISPLACEMENT || bd.shader == SRZ_SHADER_BUMP;
  if (textured) tx = env.tex[bd.tex_id];
  const float fx = (float)x, fy = (float)y;
  float alpha, beta, gamma, zz;
  if (x < vend) {
    cover_v(t, fx, fy, alpha, beta, gamma, zz);
    float nx = fmaf_(alpha, n0x, fmaf_(beta, n1x, gamma * n2x));
    float ny = fmaf_(alpha, n0y, fmaf_(beta, n1y, gamma * n2y));
    float nz = fmaf_(alpha, n0z, fmaf_(beta, n1z, gamma * n2z));
    v_normalized(nx, ny, nz);
    float u = fmaf_(alpha, u0, fmaf_(beta, u1, gamma * u2));
    float v = fmaf_(alpha, v0, fmaf_(beta, v1, gamma * v2));
    v_shade(env, bd.shader, tx, fx, fy, z, nx, ny, nz, u, v, r0, r1, r2);
  } else {
    cover_s(t, fx, fy, alpha, beta, gamma, zz);
    float nx = alpha * n0x + beta * n1x + gamma * n2x;
    float ny = alpha * n0y + beta * n1y + gamma * n2y;
    float nz = alpha * n0z + beta * n1z + gamma * n2z;
    normalize3(nx, ny, nz);
    float u = alpha * u0 + beta * u1 + gamma * u2;
    float v = alpha * v0 + beta * v1 + gamma * v2;
    s_shade(env, bd.shader, tx, fx, fy, z, nx, ny, nz, u, v, r0, r1, r2);
  }
}

// ================================================================================================================
// k_raster — one wave per 32x32 tile
// ================================================================================================================
__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
typedef float f32x4 __attribute__((ext_vector_type(4)));
// streaming 16-byte store: the framebuffer is written once and never re-read by this kernel
__device__ __forceinline__ void store_nt(float *p, const float4 &v) {
  f32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(p));
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_raster(RenderArgs a) {
  __shared__ __attribute__((aligned(16))) float s_z[WAVES_PER_WG][TILE * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t s_id[WAVES_PER_WG][TILE * LDS_STRIDE];

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const FrameDesc &fd = a.frames[blockIdx.z];
  const uint32_t lb = blockIdx.y;
  if (lb >= fd.n_local_bands) return;
  const int W = fd.width, H = fd.height;
  const int tx0 = ((int)blockIdx.x * WAVES_PER_WG + wave) * TILE;
  if (tx0 >= W) return; // the whole wave leaves; no workgroup barrier is used anywhere in this kernel
  const int band = (int)lb * a.shard_world + a.shard_rank;
  const int ty0 = band * BAND;
  const int tx1 = min(tx0 + TILE, W) - 1, ty1 = min(ty0 + BAND, H) - 1;
  const uint32_t flags = fd.flags | a.flags_or;
  const bool fused = (flags & SRZ_FUSED_CLEAR) != 0;
  float *zl = s_z[wave];
  uint32_t *il = s_id[wave];

  const size_t plane = (size_t)a.local_rows * (size_t)W;
  float *out0 = a.out + (size_t)blockIdx.z * a.frame_stride + ((size_t)lb * BAND) * (size_t)W; // plane 0 (z), row ty0

  // ---- phase A: tile init (fused clear → +inf, else load the in/out z plane) ---------------------------------
  for (int i = lane; i < TILE * TILE; i += 64) {
    int ly = i >> 5, lx = i & 31;
    float z = __builtin_inff();
    if (!fused && tx0 + lx <= tx1 && ty0 + ly <= ty1) z = out0[(size_t)ly * W + tx0 + lx];
    zl[ly * LDS_STRIDE + lx] = z;
    il[ly * LDS_STRIDE + lx] = NO_TRI;
  }
  __builtin_amdgcn_wave_barrier();

  // ---- phase B: walk the band's triangle list in submission order ---------------------------------------------
  const uint32_t cnt = a.band_count[fd.count_off + lb];
  const uint32_t *list = a.band_lists + fd.list_off + (uint64_t)lb * fd.n_tris;
  const BBox *bbox = a.bbox + fd.tri_off;
  const srz_tri *tris = a.tris + fd.tri_off;
  unsigned long long n_frag = 0, n_shaded = 0;

  for (uint32_t base = 0; base < cnt; base += 64) {
    uint32_t my = NO_TRI;
    BBox bb;
    bb.sx = 1, bb.ex = 0, bb.sy = 0, bb.ey = 0;
    TriXY t;
    t.ax = t.ay = t.bx = t.by = t.cx = t.cy = t.z0 = t.z1 = t.z2 = t.v_inv = t.s_area = 0.0f;
    bool hit = false;
    if (base + lane < cnt) {
      my = list[base + lane];
      bb = bbox[my];
      hit = bb.ex >= tx0 && bb.sx <= tx1;
    }
    if (hit) {
      const float *p = &tris[my].pos[0][0];
      t.ax = p[0], t.ay = p[1], t.z0 = p[2], t.bx = p[3], t.by = p[4], t.z1 = p[5], t.cx = p[6], t.cy = p[7], t.z2 = p[8];
      tri_consts(t);
    }
    const int bbx = (int)(uint16_t)bb.sx | ((int)(uint16_t)bb.ex << 16);
    const int bby = (int)(uint16_t)bb.sy | ((int)(uint16_t)bb.ey << 16);
    unsigned long long m = __ballot(hit);
    while (m) {
      const int j = __builtin_ctzll(m);
      m &= m - 1;
      // broadcast triangle j to the wave (uniform values)
      TriXY u;
      u.ax = rl_f(t.ax, j), u.ay = rl_f(t.ay, j), u.bx = rl_f(t.bx, j), u.by = rl_f(t.by, j), u.cx = rl_f(t.cx, j);
      u.cy = rl_f(t.cy, j), u.z0 = rl_f(t.z0, j), u.z1 = rl_f(t.z1, j), u.z2 = rl_f(t.z2, j);
      u.v_inv = rl_f(t.v_inv, j), u.s_area = rl_f(t.s_area, j);
      const uint32_t idx = (uint32_t)rl_i((int)my, j);
      const int px = rl_i(bbx, j), py = rl_i(bby, j);
      const int sx = (int16_t)(px & 0xffff), ex = (int16_t)(px >> 16), sy = (int16_t)(py & 0xffff), ey = (int16_t)(py >> 16);
      const int rx0 = max(sx, tx0), rx1 = min(ex, tx1), ry0 = max(sy, ty0), ry1 = min(ey, ty1);
      const int vend = (flags & SRZ_UNIFIED) ? ex + 1 : sx + ((ex - sx + 1) & ~7);
      for (int yb = ry0; yb <= ry1; yb += 8) {
        for (int xb = rx0; xb <= rx1; xb += 8) {
          const int x = xb + (lane & 7), y = yb + (lane >> 3);
          const bool act = x <= rx1 && y <= ry1;
          const int li = (y - ty0) * LDS_STRIDE + (x - tx0);
          const float fx = (float)x, fy = (float)y;
          float al, be, ga, z = 0.0f;
          bool inside = false, pass = false;
          const bool anyV = xb < vend, anyS = min(xb + 7, rx1) >= vend; // wave-uniform
          float zold = 0.0f;
          if (act) zold = zl[li];
          if (anyV && (!anyS || x < vend)) {
            inside = cover_v(u, fx, fy, al, be, ga, z);
            pass = inside && (z < zold); // strict (src/Rasterizer.cpp:334)
          }
          if (anyS && (!anyV || x >= vend)) {
            inside = cover_s(u, fx, fy, al, be, ga, z);
            pass = inside && !(z > zold); // <= passes, NaN passes (src/Rasterizer.cpp:475)
          }
          inside = inside && act;
          pass = pass && act;
          if (pass) {
            zl[li] = z;
            il[li] = idx;
          }
          if (STATS) n_frag += inside ? 1 : 0, n_shaded += pass ? 1 : 0;
        }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();

  // ---- phase C: shade every pixel's final owner once, write z + 3 colour planes ---------------------------------
  ShadeEnv env;
  env.fd = &fd, env.lights = a.lights + fd.light_off, env.tex = a.tex;
  unsigned long long n_vis = 0, n_vis_tex = 0;
  const bool vec_ok = (W & 3) == 0;
  for (int it = 0; it < 4; ++it) {
    const int ly = it * 8 + (lane >> 3), lx4 = (lane & 7) * 4;
    const int y = ty0 + ly, x4 = tx0 + lx4;
    if (y > ty1 || x4 > tx1) continue;
    const float4 z4 = *reinterpret_cast<const float4 *>(&zl[ly * LDS_STRIDE + lx4]);
    const uint4 id4 = *reinterpret_cast<const uint4 *>(&il[ly * LDS_STRIDE + lx4]);
    float *gz = out0 + (size_t)ly * W + x4;
    float4 C0 = make_float4(0.f, 0.f, 0.f, 0.f), C1 = C0, C2 = C0;
    const bool full = vec_ok && x4 + 3 <= tx1;
    if (!fused) { // keep the colour of pixels this call does not own
      if (full) {
        C0 = *reinterpret_cast<const float4 *>(gz + plane);
        C1 = *reinterpret_cast<const float4 *>(gz + 2 * plane);
        C2 = *reinterpret_cast<const float4 *>(gz + 3 * plane);
      } else {
#define SRZ_LD(K, M)                                                                                                   \
  if (x4 + K <= tx1) C0.M = gz[plane + K], C1.M = gz[2 * plane + K], C2.M = gz[3 * plane + K];
        SRZ_LD(0, x) SRZ_LD(1, y) SRZ_LD(2, z) SRZ_LD(3, w)
#undef SRZ_LD
      }
    }
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      const uint32_t id = k == 0 ? id4.x : k == 1 ? id4.y : k == 2 ? id4.z : id4.w;
      const float z = k == 0 ? z4.x : k == 1 ? z4.y : k == 2 ? z4.z : z4.w;
      if (id != NO_TRI) {
        float r0, r1, r2;
        bool textured;
        shade_pixel(a, env, flags, id, x4 + k, y, z, r0, r1, r2, textured);
        if (k == 0) C0.x = r0, C1.x = r1, C2.x = r2;
        if (k == 1) C0.y = r0, C1.y = r1, C2.y = r2;
        if (k == 2) C0.z = r0, C1.z = r1, C2.z = r2;
        if (k == 3) C0.w = r0, C1.w = r1, C2.w = r2;
        if (STATS) n_vis++, n_vis_tex += textured ? 1 : 0;
      }
    }
    if (full) {
      store_nt(gz, z4);
      store_nt(gz + plane, C0);
      store_nt(gz + 2 * plane, C1);
      store_nt(gz + 3 * plane, C2);
    } else {
#define SRZ_ST(K, M)                                                                                                   \
  if (x4 + K <= tx1) gz[K] = z4.M, gz[plane + K] = C0.M, gz[2 * plane + K] = C1.M, gz[3 * plane + K] = C2.M;
      SRZ_ST(0, x) SRZ_ST(1, y) SRZ_ST(2, z) SRZ_ST(3, w)
#undef SRZ_ST
    }
  }
  if (STATS) {
    for (int o = 32; o > 0; o >>= 1) {
      n_frag += __shfl_down(n_frag, o);
      n_shaded += __shfl_down(n_shaded, o);
      n_vis += __shfl_down(n_vis, o);
      n_vis_tex += __shfl_down(n_vis_tex, o);
    }
    if (lane == 0) {
      if (n_frag) atomicAdd(&a.stats[ST_FRAGMENTS], n_frag);
      if (n_shaded) atomicAdd(&a.stats[ST_SHADED], n_shaded);
      if (n_vis) atomicAdd(&a.stats[ST_VISIBLE], n_vis);
      if (n_vis_tex) atomicAdd(&a.stats[ST_VISIBLE_TEX], n_vis_tex);
    }
  }
}

// BGR u8 (row_stride bytes per row) → one dword per texel
__global__ void k_tex_convert(const uint8_t *bgr, int w, int h, int row_stride, uint32_t *out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= w * h) return;
  int y = i / w, x = i - y * w;
  const uint8_t *p = bgr + (size_t)y * row_stride + (size_t)x * 3;
  out[i] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16);
}

// ---- launchers ---------------------------------------------------------------------------------------------------
void launch_setup(const RenderArgs &a, int n_frames, uint32_t max_tris, bool stats, hipStream_t s) {
  if (n_frames <= 0 || max_tris == 0) return;
  dim3 grid((max_tris + 255) / 256, n_frames);
  if (grid.x > 4096) grid.x = 4096;
  BBox *bb = const_cast<BBox *>(a.bbox);
  if (stats)
    hipLaunchKernelGGL(k_setup<true>, grid, dim3(256), 0, s, a, bb);
  else
    hipLaunchKernelGGL(k_setup<false>, grid, dim3(256), 0, s, a, bb);
}

void launch_bands(const RenderArgs &a, uint32_t *band_lists, uint32_t *band_count, int n_frames, uint32_t max_local_bands,
                  hipStream_t s) {
  if (n_frames <= 0 || max_local_bands == 0) return;
  dim3 grid((max_local_bands + WAVES_PER_WG - 1) / WAVES_PER_WG, n_frames);
  hipLaunchKernelGGL(k_bands, grid, dim3(256), 0, s, a, band_lists, band_count);
}

void launch_raster(const RenderArgs &a, int n_frames, uint32_t max_local_bands, int width, bool stats, hipStream_t s) {
  if (n_frames <= 0 || max_local_bands == 0) return;
  dim3 grid((width + TILE * WAVES_PER_WG - 1) / (TILE * WAVES_PER_WG), max_local_bands, n_frames);
  if (stats)
    hipLaunchKernelGGL(k_raster<true>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_raster<false>, grid, dim3(256), 0, s, a);
}

void launch_tex_convert(const uint8_t *d_bgr, int w, int h, int row_stride, uint32_t *d_bgrx, hipStream_t s) {
  int n = w * h;
  hipLaunchKernelGGL(k_tex_convert, dim3((n + 255) / 256), dim3(256), 0, s, d_bgr, w, h, row_stride, d_bgrx);
}

} // namespace srz
