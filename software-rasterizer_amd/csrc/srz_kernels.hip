// srz_kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the raster + fragment-shade stage.
//
// Replaces, on the device, the hot loops of TraditionalRasterizer::draw (src/Rasterizer.cpp:183-499 of
// Liupeter01/Software-Rasterizer) and the fragment shaders they call (src/Shader.cpp, include/shader/Shader.hpp).
// Not a translation of the AVX2 code: the reference walks triangles serially and rows in parallel; here
//
//   k_vertex  one thread per face          (optional) the vertex stage, Scene::loadTriangleStream: meshes + matrices → srz_tri
//                                          + the dense copy of its positions
//   k_setup   one thread per triangle      reads the dense positions (36 bytes): bbox (Triangle::calcBoundingBox) + backface
//                                          test → 8-byte BBox, and — per group of 512 triangles, in LDS — the triangles
//                                          sorted by the 32-row bands they reach: binning is O(triangles)
//   k_bin     one WORKGROUP per 32-row band count / scan / fill of the band's own entries into UNORDERED per-tile lists of 4-byte
//                                          triangle indices, LDS atomics only, assembled in LDS and stored as one coalesced run;
//                                          the lists come from a pool sized by what renders need
//   k_clear   ~64 persistent workgroups    on a second stream beside k_raster / k_shade: the fused clear of every tile no
//                                          bbox reaches (16-byte non-temporal stores of +inf / 0), throttled by its grid
//   k_raster  one WAVE per touched tile    VISIBILITY, order-independent: one 64-bit key (depth | tie-break) per pixel in
//                                          LDS, every fragment is a ds_min_u64; the tile's (triangle, pixel) candidates are
//                                          flattened into items (8 pixels of a bbox row / one scalar-tail pixel) and dealt
//                                          densely to the 64 lanes.  Touched tiles nobody owns leave as the fused clear;
//                                          owned tiles write z + owner ids — 16 bits per pixel: the owner's POSITION in the tile's
//                                          triangle list, which rides in the key's tie-break below the triangle index — and append
//                                          themselves (with their list's length and offset) to their frame's work list.
//   k_raster_slow  one wave per listed tile the reference's ORDERED algorithm for what the keys cannot express (NaN / ±0
//                                          depths), for bands that did not fit the pool, and for counting runs.
//   k_shade   one WORKGROUP per owned tile VISIBILITY-FIRST SHADING: each pixel's final owner is shaded exactly once (the
//                                          reference's shaders are pure functions of (triangle,pixel) and its write is an
//                                          overwrite), pixels compacted by semantics class into dense 64-lane chunks; the tile's
//                                          listed triangles are staged in LDS once per tile (LDS-DMA) and read from there by list
//                                          position; 16-byte stores of the 3 colour planes.
//   k_resolve8                             (optional) display(): planes → BGR8.
//
// Numerics: every float op is the oracle's op in the oracle's order (oracle/srz_oracle.c): contraction is OFF,
// fused ops are explicit fmaf(), division and sqrt are the correctly rounded ones (the compiler's IEEE expansions, or
// short sequences proven / checked on the device to return the same bits), pow is evaluated in binary64 and rounded
// once.  The framebuffer is therefore expected to be bit-identical to the oracle's.
#include "srz_device.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#pragma clang fp contract(off)

namespace srz {

// Read-only inputs are read through the CONSTANT address space: wave-uniform addresses become scalar (s_load)
// loads, per-lane ones become invariant loads that never have to be ordered against the framebuffer stores.
#define SRZ_CAS __attribute__((address_space(4)))
template <typename T> __device__ __forceinline__ const SRZ_CAS T *as_const(const T *p) { return (const SRZ_CAS T *)(p); }
// native vector types (HIP's float4/uint2 classes cannot be loaded through an address-space-qualified pointer)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- operand-order-exact min/max (SSE / std:: semantics, see oracle) ------------------------------------------
__device__ __forceinline__ float sse_max(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ float sse_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_clamp(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
__device__ __forceinline__ float fmsubf(float a, float b, float c) { return __builtin_fmaf(a, b, -c); }
__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// ---- exact reciprocal / square root without the compiler's scaling + fix-up scaffolding --------------------------
// For a normal operand with 2^-100 <= |x| <= 2^100 the short sequences below return EXACTLY the correctly rounded
// 1/x and sqrt(x) — verified on MI355X against the IEEE expansions for every one of the 3.37e9 binary32 values in that
// range (k_verify_fastmath, run by tests/test_gpu_fastmath.py).  Operands outside the range (zero, denormal, huge,
// inf, NaN) take the compiler's full IEEE path, so the functions are drop-in equal to `1.0f/x` and `sqrtf(x)`.
__device__ __forceinline__ uint32_t f2u_(float f) { return __builtin_bit_cast(uint32_t, f); }
// (by-value helpers: __builtin_bit_cast applied directly to an ext-vector element reads element 0)
__device__ __forceinline__ float u2f_(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ bool fast_range(float x) { return (((f2u_(x) >> 23) & 0xffu) - 27u) <= 200u; }
__device__ __forceinline__ float rcp_core(float x) { // v_rcp_f32 + one Newton step
  float r = __builtin_amdgcn_rcpf(x);
  return __builtin_fmaf(__builtin_fmaf(-x, r, 1.0f), r, r);
}
__device__ __forceinline__ float sqrt_core(float x) { // x * v_rsq_f32(x) corrected by the exact residual
  float r = __builtin_amdgcn_rsqf(x), s = x * r, h = 0.5f * r; // (one transcendental: they issue at ~0.57x the FMA rate)
  return __builtin_fmaf(__builtin_fmaf(-s, s, x), h, s);
}
// The slow (full IEEE) variants are out-of-line and entered only when SOME lane of the wave has an out-of-range
// operand (wave-uniform branch: no if-conversion, the long expansions are not even fetched in the common case).
__device__ __attribute__((noinline)) float rcp_ieee(float x) { return 1.0f / x; }
__device__ __attribute__((noinline)) float sqrt_ieee(float x) { return __builtin_sqrtf(x); }
__device__ __attribute__((noinline)) float rsqrt2_ieee(float d) { return 1.0f / __builtin_sqrtf(d); }
__device__ __forceinline__ float rcp_rn(float x) {
  if (__ballot(!fast_range(x)) == 0ull) return rcp_core(x);
  return rcp_ieee(x);
}
__device__ __forceinline__ float sqrt_rn(float x) {
  if (__ballot(!(fast_range(x) && x > 0.0f)) == 0ull) return sqrt_core(x);
  return sqrt_ieee(x);
}
// 1.0f / sqrtf(d) with both roundings (sqrt, then reciprocal), as glm::inversesqrt / rcp_ps(sqrt_ps()) compute it
__device__ __forceinline__ float rsqrt2_rn(float d) {
  if (__ballot(!(fast_range(d) && d > 0.0f)) == 0ull) return rcp_core(sqrt_core(d)); // sqrt(d) in [2^-50,2^50]: fast range
  return rsqrt2_ieee(d);
}

// ---- math policies of the shading code ---------------------------------------------------------------------------
// The shaders are written once against a policy M that supplies the correctly rounded 1/x, sqrt(x), 1/sqrt(x):
//   BranchMath  a wave-uniform branch per operation (k_setup / k_raster: one or two operations per thread)
//   FastMath    OPTIMISTIC: always the short exact sequence, the operand checks only accumulate into `bad`; k_shade
//               re-shades the (in practice never taken) chunk with IeeeMath when any lane saw an out-of-range operand.
//               No branches inside the shader, so the independent normalisations of one pixel interleave.
//   IeeeMath    the compiler's full IEEE expansions.
// All three return identical bits wherever FastMath's checks pass (tests/test_gpu_fastmath.py).
__device__ __forceinline__ bool fast_pos(float x) { return ((f2u_(x) >> 23) - 27u) <= 200u; } // fast_range and x > 0
// a / b from y = RN(1/b) (Markstein): q0 = a*y is refined twice with the exact residual a - b*q.  With y correctly rounded
// and q1 within one ulp, RN(q1 + r*y) is the correctly rounded quotient, i.e. exactly what the IEEE division returns,
// PROVIDED nothing under- or overflows on the way: callers guarantee 2^-40 <= |b| <= 2^40 and a == +0 or
// 2^-60 <= |a| <= 2^60 (a -0 numerator would come out as +0).  Checked on device against a / b in tests/test_gpu_fastmath.py.
__device__ __forceinline__ float div_by_rcp(float a, float b, float y) {
  float q = a * y;
  q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
  return __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
}
__device__ __forceinline__ bool div_num_ok(float a) { // (int |, &: no short-circuit branches)
  return ((int)(f2u_(a) == 0u) | (int)((((f2u_(a) >> 23) & 0xffu) - 67u) <= 120u)) != 0;
}
__device__ __forceinline__ bool div_den_ok(float b) { return ((f2u_(b) >> 23) - 87u) <= 80u; } // 2^-40 <= b <= 2^40, b > 0
// sqrt(a^2 + b^2) as the scalar Blinn-Phong computes its attenuation distance (src/Shader.cpp:516-523: std::pow(x, 2) and std::sqrt
// in binary64, rounded once)
__device__ __forceinline__ float len2d_f64(float a, float b) {
  const double dx = (double)a, dy = (double)b;
  return (float)__builtin_sqrt(dx * dx + dy * dy);
}
// The same value without the compiler's binary64 square root (v_rsq_f64 + three refinements + range scaling: ~22 binary64-rate
// instructions per light in every S pixel): two Heron steps in binary64 from the binary32 root,
//   s   = RN64(a^2 + b^2) exactly as the reference has it (both squares are exact in binary64: one fma);
//   r   = the binary32 square root of RN32(s) (sqrt_core: |r - sqrt(s)| < 2^-23.4 r), h = v_rsq_f32 / 2 (|2 h r - 1| < 2^-21.5);
//   R1  = r + (s - r^2) h    : r^2 is exact (48 bits); |R1 - sqrt(s)| < (2^-21.5 * 2^-23.4 + 2^-47.8 + 2^-52) R1 < 2^-44.7 R1
//   R2  = R1 + (s - R1^2) h  : the residual is one fma (a single rounding of the exact s - R1^2); h's error now contributes 2^-66,
//         the dropped second-order term 2^-90: R2 = sqrt(s) up to ITS OWN rounding, i.e. RN64(sqrt(s)) or a neighbour of it.
// RN32(R2) is therefore the reference's RN32(RN64(sqrt(s))) unless a binary32 rounding boundary (the 29 dropped bits =
// 0x10000000) lies within an ulp of R2: `amb` is set within 8 ulps — 3e-8 of the operands — and k_shade hands that tile to the
// generic build like any operand outside FastMath's range.  `sbits` returns the bits of RN32(s) for the range tracking: inside
// [2^-100, 2^101) everything above is normal.  Checked on the device against len2d_f64 (k_verify_fastlen, tests/test_gpu_fastmath.py).
__device__ __forceinline__ float len2d_fast(float a, float b, bool &amb, uint32_t &sbits) {
  const double dx = (double)a, dy = (double)b;
  const double s = __builtin_fma(dx, dx, dy * dy);
  const float sf = (float)s;
  sbits = f2u_(sf);
  const float rs = __builtin_amdgcn_rsqf(sf), r0 = sf * rs, h = 0.5f * rs;
  const float r = __builtin_fmaf(__builtin_fmaf(-r0, r0, sf), h, r0);
  const double R = (double)r, H = (double)h;
  const double R1 = __builtin_fma(__builtin_fma(-R, R, s), H, R);
  const double R2 = __builtin_fma(__builtin_fma(-R1, R1, s), H, R1);
  const uint32_t lo = (uint32_t)__builtin_bit_cast(uint64_t, R2) & 0x1fffffffu;
  amb |= (lo - 0x0ffffff8u) <= 16u; // within 8 binary64 ulps of the binary32 midpoint pattern
  return (float)R2;
}
struct BranchMath {
  __device__ __forceinline__ float rcp(float x) { return rcp_rn(x); }
  __device__ __forceinline__ float sqrt(float x) { return sqrt_rn(x); }
  __device__ __forceinline__ float rsqrt2(float d) { return rsqrt2_rn(d); }
};
struct IeeeMath {
  __device__ __forceinline__ float div255(float a) { return a / 255.0f; }
  __device__ __forceinline__ void div3(float a0, float a1, float a2, float b, float &q0, float &q1, float &q2) {
    q0 = a0 / b, q1 = a1 / b, q2 = a2 / b;
  }
  __device__ __forceinline__ float rcp(float x) { return 1.0f / x; }
  __device__ __forceinline__ float sqrt(float x) { return __builtin_sqrtf(x); }
  __device__ __forceinline__ float rsqrt2(float d) { return 1.0f / __builtin_sqrtf(d); }
  __device__ __forceinline__ float len2d(float a, float b) { return len2d_f64(a, b); }
};
struct FastMath {
  bool bad = false; // (the rare checks: division operands, pow_fast's rounding flag)
  // The operands of the reciprocals / square roots are tracked as the running unsigned minimum and maximum of their bit patterns
  // (two instructions per operand instead of shift + subtract + compare + or; magnitudes for the signed ones) and tested ONCE per
  // pixel: all inside [2^-100, 2^101) <=> lo >= 0x0d800000 and hi <= 0x71ffffff — the criterion of fast_range / fast_pos (a
  // negative, NaN, infinite, zero or denormal operand of a positive-only operation breaks one of the two bounds)
  uint32_t lo = 0x0d800000u, hi = 0x0d800000u;
  __device__ __forceinline__ void track(uint32_t bits) { lo = lo < bits ? lo : bits, hi = hi > bits ? hi : bits; }
  __device__ __forceinline__ bool is_bad() const { return bad | (lo < 0x0d800000u) | (hi > 0x71ffffffu); }
  // texel / 255.0f for texel = 0..255: numerator and divisor are always inside div_by_rcp's range
  __device__ __forceinline__ float div255(float a) { return div_by_rcp(a, 255.0f, __builtin_bit_cast(float, 0x3b808081u)); }
  // three numerators (wave-uniform light intensities) over one positive per-pixel divisor: one reciprocal
  __device__ __forceinline__ void div3(float a0, float a1, float a2, float b, float &q0, float &q1, float &q2) {
    bad |= ((int)div_den_ok(b) & (int)div_num_ok(a0) & (int)div_num_ok(a1) & (int)div_num_ok(a2)) == 0;
    const float y = rcp_core(b);
    q0 = div_by_rcp(a0, b, y), q1 = div_by_rcp(a1, b, y), q2 = div_by_rcp(a2, b, y);
  }
  __device__ __forceinline__ float rcp(float x) {
    track(f2u_(x) & 0x7fffffffu);
    return rcp_core(x);
  }
  __device__ __forceinline__ float sqrt(float x) {
    track(f2u_(x));
    return sqrt_core(x);
  }
  __device__ __forceinline__ float rsqrt2(float d) { // sqrt(d) lies in [2^-50, 2^50]: inside rcp_core's range
    track(f2u_(d));
    return rcp_core(sqrt_core(d));
  }
  // (the scalar path's attenuation distance: two binary64 Heron steps from the binary32 root; an ambiguous rounding → `bad`)
  __device__ __forceinline__ float len2d(float a, float b) {
    uint32_t sb;
    const float d = len2d_fast(a, b, bad, sb);
    track(sb);
    return d;
  }
};
// ApproxMath — the TOLERANCE mode (SRZ_OPT_APPROX_SHADE, opt-in; never the default): the arithmetic CLASS of the reference's own
// x86 path, which shades with approximate instructions — _mm256_rcp_ps (include/shader/Shader.hpp:131, src/Tools.cpp:19,
// include/loader/TextureLoader.hpp:99, src/Rasterizer.cpp:111: 12 bits) and SVML _mm256_pow_ps (include/shader/Shader.hpp:195) —
// on the hardware's own: bare v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp), x^p = v_exp_f32(p * v_log_f32(x)), texel * (1/255), the
// binary64 pieces of the scalar path (the attenuation distance, std::pow) in binary32.  Only k_shade's LIGHTING arithmetic takes it
// (normalisations, attenuation, the power, the texel's scaling): k_raster stays exact, so z, coverage and ownership are
// bit-identical to the oracle in this mode too, and so do the barycentrics and everything interpolated with them.  Nothing can go out of a
// "fast range" here — no operand tracking, no re-shade.  tests/test_gpu_approx.py states the tolerance.
struct ApproxMath {
  __device__ __forceinline__ float div255(float a) { return a * (1.0f / 255.0f); }
  __device__ __forceinline__ void div3(float a0, float a1, float a2, float b, float &q0, float &q1, float &q2) {
    const float y = __builtin_amdgcn_rcpf(b);
    q0 = a0 * y, q1 = a1 * y, q2 = a2 * y;
  }
  // (the one reciprocal of the coverage arithmetic, tri_consts: EXACT — the barycentrics, hence depth, normals' and texture
  // coordinates' interpolation, are the oracle's bits in this mode too: a texture coordinate one ulp off can round to the
  // neighbouring texel, which no colour tolerance covers)
  __device__ __forceinline__ float rcp(float x) { return rcp_rn(x); }
  __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  __device__ __forceinline__ float rsqrt2(float d) { return __builtin_amdgcn_rsqf(d); }
  __device__ __forceinline__ float len2d(float a, float b) { return __builtin_amdgcn_sqrtf(__builtin_fmaf(a, a, b * b)); }
};

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
  float tx = ax * bx, ty = ay * by, tz = az * bz;
  return tx + ty + tz;
}
// glm::normalize: v * (1/sqrt(dot(v,v)))
template <class M> __device__ __forceinline__ void normalize3(M &m, float &x, float &y, float &z) {
  float is = m.rsqrt2(dot3(x, y, z, x, y, z));
  x = x * is, y = y * is, z = z * is;
}
// NormalSIMD::normalized (src/Tools.cpp:13-24)
template <class M> __device__ __forceinline__ void v_normalized(M &m, float &x, float &y, float &z) {
  float len = m.sqrt(fmaf_(x, x, fmaf_(y, y, z * z)));
  if (len > 0.0f) {
    float inv = m.rcp(len);
    x = x * inv, y = y * inv, z = z * inv;
  } else {
    x = y = z = 0.0f;
  }
}

// FastMath: a squared length inside the fast range has a positive, in-range square root — one check covers both steps
__device__ __forceinline__ void v_normalized(FastMath &m, float &x, float &y, float &z) {
  const float d = fmaf_(x, x, fmaf_(y, y, z * z));
  m.track(f2u_(d));
  const float inv = rcp_core(sqrt_core(d));
  x = x * inv, y = y * inv, z = z * inv;
}

// ApproxMath: one v_rsq_f32; the zero vector stays the zero vector as in NormalSIMD::normalized
__device__ __forceinline__ void v_normalized(ApproxMath &m, float &x, float &y, float &z) {
  const float d = fmaf_(x, x, fmaf_(y, y, z * z));
  const float inv = d > 0.0f ? __builtin_amdgcn_rsqf(d) : 0.0f;
  x = x * inv, y = y * inv, z = z * inv;
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

// x^150 (the reference's Shader::p, src/Shader.cpp:10) as a fixed chain: the very multiplications pow_cr's loop performs
// for n = 150 = 0b10010110, in the same order (r = x^2 · x^4 · x^16 · x^128), without the loop
// x <= 0.5 gives x^150 <= 2^-150, which rounds to +0 in binary32 (2^-150 itself is the tie below the smallest denormal: to
// even = 0): when no lane of the wave is above 0.5 — the half-vector more than 60 degrees off the normal everywhere in the
// chunk, the common case away from a highlight — the chain is skipped.  (A NaN is not "<= 0.5": it takes the chain.)
__device__ __forceinline__ float pow150_cr(float x) {
  if (__ballot(!(x <= 0.5f)) == 0ull) return 0.0f;
  const double b1 = (double)x, b2 = b1 * b1, b4 = b2 * b2, b8 = b4 * b4, b16 = b8 * b8, b32 = b16 * b16, b64 = b32 * b32,
               b128 = b64 * b64;
  double r = b2 * b4;
  r = r * b16;
  r = r * b128;
  return (float)r;
}
// x^n for an integer exponent 0..256 that is the same for the whole wave (a per-frame constant): pow_cr's square-and-multiply
// chain as a SCALAR loop — the very same binary64 multiplications in the same order, no per-lane selects
__device__ __forceinline__ float pow_int_cr(float x, float p) {
  unsigned n = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p);
  double b = (double)x, r = 1.0;
  while (n) {
    if (n & 1u) r = r * b;
    b = b * b;
    n >>= 1;
  }
  return (float)r;
}
// Compile-time knowledge a shading variant may have (k_shade picks the variant per 64-pixel chunk, wave-uniformly):
//   SH   >= 0: every pixel of the chunk uses shader type SH; -1: per-pixel type (sd.shader)
//   NL   != 0: the frame has exactly |NL| lights (1..4) (decided on the host): the light loop is unrolled, the light constants sit
//              in SGPRs;  0: run-time count.  The sign of NL selects the exponent form (pow_frame)
// x^p for a NON-INTEGER exponent 0 < p <= 4096 (wave-uniform), OPTIMISTICALLY: exp2(p * log2(x)) in binary64 — log by the
// atanh series of (m - 1) / (m + 1) on the mantissa in (sqrt(1/2), sqrt(2)], p * e kept exact, 2^r by its Taylor polynomial —
// about 45 binary64 operations where ocml's correctly rounded pow takes ~210.  Its error is below (2.4 p + 4) ulps of the binary64
// result (log2(m): 7 roundings of 2^-53 on |t| <= 0.5, times p ln 2; the two series' tails are below 2^-55; 2^r: 4 more), so the
// binary32 rounding of the result is the rounding of the exact x^p — hence of any correctly rounded or 1-ulp binary64 pow, the
// oracle's and ocml's alike — unless the result lies within 32 + 4 p binary64 ulps of a binary32 rounding boundary (for
// p = 7.5: 2.3e-7 of the operands) or in the subnormal range: then `amb` is set, and the caller hands the pixel's tile to the
// generic build (pow_cr) exactly like an operand outside FastMath's range.  x = ±0 gives +0 (p is positive and no odd integer),
// a result below 2^-151 rounds to +0; other non-positive, non-finite or NaN operands set `amb`.
// Checked on the device against pow_cr for every binary32 x in (0, 1] at several exponents (k_verify_fastpow, tests/test_gpu_fastmath.py).
__device__ __forceinline__ float pow_fast(float x, float p, bool &amb) {
  const double xd = (double)x;
  const uint64_t xb = __builtin_bit_cast(uint64_t, xd);
  int e = (int)((xb >> 52) & 0x7ffu) - 1023;
  double m = __builtin_bit_cast(double, (xb & 0x000fffffffffffffull) | 0x3ff0000000000000ull); // [1, 2)
  const bool hi = m > 1.4142135623730951;
  m = hi ? m * 0.5 : m, e += hi ? 1 : 0;
  const double f = m - 1.0, g = m + 1.0; // (exact: m has 24 significant bits)
  // s = f / g: reciprocal by Newton, then one correction of the quotient (error < 1 ulp)
  double rg = __builtin_amdgcn_rcp(g);
  rg = __builtin_fma(__builtin_fma(-g, rg, 1.0), rg, rg);
  rg = __builtin_fma(__builtin_fma(-g, rg, 1.0), rg, rg);
  double sq = f * rg;
  sq = __builtin_fma(__builtin_fma(-g, sq, f), rg, sq);
  const double z = sq * sq; // <= 0.02944
  // log(m) = 2 s (1 + z/3 + z^2/5 + ... + z^9/19): the tail is below 2^-53
  double P = 1.0 / 19.0;
  P = __builtin_fma(P, z, 1.0 / 17.0), P = __builtin_fma(P, z, 1.0 / 15.0), P = __builtin_fma(P, z, 1.0 / 13.0);
  P = __builtin_fma(P, z, 1.0 / 11.0), P = __builtin_fma(P, z, 1.0 / 9.0), P = __builtin_fma(P, z, 1.0 / 7.0);
  P = __builtin_fma(P, z, 1.0 / 5.0), P = __builtin_fma(P, z, 1.0 / 3.0), P = __builtin_fma(P, z, 1.0);
  const double t = (2.0 * sq) * P * 1.4426950408889634; // log2(m), |t| <= 0.5
  const double pd = (double)p, yh = pd * (double)e /* exact: 24 x 11 bits */, yl = pd * t;
  const double n = __builtin_rint(yh + yl), r = (yh - n) + yl; // (yh - n is exact), |r| <= 0.5 + tiny
  const double u = r * 0.6931471805599453;
  double Q = 1.0 / 6227020800.0; // exp(u), |u| <= 0.3466: Taylor to u^13 / 13!
  Q = __builtin_fma(Q, u, 1.0 / 479001600.0), Q = __builtin_fma(Q, u, 1.0 / 39916800.0), Q = __builtin_fma(Q, u, 1.0 / 3628800.0);
  Q = __builtin_fma(Q, u, 1.0 / 362880.0), Q = __builtin_fma(Q, u, 1.0 / 40320.0), Q = __builtin_fma(Q, u, 1.0 / 5040.0);
  Q = __builtin_fma(Q, u, 1.0 / 720.0), Q = __builtin_fma(Q, u, 1.0 / 120.0), Q = __builtin_fma(Q, u, 1.0 / 24.0);
  Q = __builtin_fma(Q, u, 1.0 / 6.0), Q = __builtin_fma(Q, u, 0.5), Q = __builtin_fma(Q, u, 1.0), Q = __builtin_fma(Q, u, 1.0);
  const double nc = __builtin_fmax(__builtin_fmin(n, 1000.0), -1000.0); // (keeps ldexp's exponent an int)
  const double res = __builtin_ldexp(Q, (int)nc);
  const uint32_t lo = (uint32_t)__builtin_bit_cast(uint64_t, res) & 0x1fffffffu; // the 29 bits binary32 drops
  const uint32_t dist = lo > 0x10000000u ? lo - 0x10000000u : 0x10000000u - lo; // to the rounding boundary, in binary64 ulps
  const bool zero_in = (__builtin_bit_cast(uint32_t, x) << 1) == 0u;            // x = ±0
  const bool tiny = res < 0x1p-151, normal = res >= 0x1p-120 && res < 0x1p+120;
  const bool okx = x > 0.0f && x < __builtin_inff();
  const uint32_t safe = 32u + (uint32_t)(4.0f * p); // (scalar: p is wave-uniform)
  amb |= !(zero_in | (okx & (tiny | (normal & (dist > safe)))));
  return (zero_in | tiny) ? 0.0f : (float)res;
}

// The exponent forms of the shading builds (NL = the build's compile-time light count, see k_shade):
//   NL > 0  an integer exponent 0 <= p <= 256 (decided on the host): the fixed chain for p = 150 (the value the reference ships,
//           src/Shader.cpp:10) or the scalar square-and-multiply loop
//   NL < 0  a non-integer exponent in (0, 4096] (Shader::p is a mutable static in the reference: any value is legal): pow_fast,
//           whose ambiguous cases the FastMath pass's `bad` flag takes to the generic build
//   NL = 0  the generic build, every other exponent: pow_cr
// x^p in the tolerance mode: exp2(p * log2 x) on the transcendental unit (x >= 0 here: a clamped cosine; x = 0 → log2 = -inf →
// +0 for p > 0).  p = 0 (wave-uniform) is 1 whatever x, as pow() has it
__device__ __forceinline__ float pow_approx(float x, float p) {
  if (p == 0.0f) return 1.0f;
  return __builtin_amdgcn_exp2f(p * __builtin_amdgcn_logf(x));
}
template <int NL, class M> __device__ __forceinline__ float pow_frame(M &m, float x, float p) {
  if constexpr (std::is_same<M, ApproxMath>::value) {
    return pow_approx(x, p);
  } else if constexpr (NL > 0) {
    return p == 150.0f ? pow150_cr(x) : pow_int_cr(x, p); // (wave-uniform branch: p is a per-frame scalar)
  } else if constexpr (NL < 0 && std::is_same<M, FastMath>::value) {
    return pow_fast(x, p, m.bad); // (the host sends only frames with a non-integer exponent in (0, 4096] to these builds)
  } else {
    return pow_cr(x, p);
  }
}
template <int NL> constexpr int light_count() { return NL < 0 ? -NL : NL; } // lights known at compile time (0: run-time count)

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
template <class M> __device__ __forceinline__ void tri_consts(M &m, TriXY &t) {
  float ABx = t.bx - t.ax, ABy = t.by - t.ay, ACx = t.cx - t.ax, ACy = t.cy - t.ay;
  t.v_inv = m.rcp(fmsubf(ABx, ACy, ACx * ABy));
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
// k_vertex — the vertex stage, Scene::loadTriangleStream (src/Scene.cpp:927-958): one thread per face.
//   pos' = to_vec3(NDC_MVP * (p,1)) ; pos'.z = pos'.z*scale + offset ; nrm' = to_vec3(Normal_M * (n,1)) ; uv copied
// glm's operator*(mat4,vec4) order: (m0*v.x + m1*v.y) + (m2*v.z + m3*v.w); Tools::to_vec3 divides by w (src/Tools.cpp:74-76).
// ================================================================================================================
__device__ __forceinline__ void xform_div_w(const SRZ_CAS float *m, float x, float y, float z, float &ox, float &oy, float &oz) {
  float r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float add0 = m[0 * 4 + i] * x + m[1 * 4 + i] * y;
    const float add1 = m[2 * 4 + i] * z + m[3 * 4 + i]; // * 1.0f is exact
    r[i] = add0 + add1;
  }
  ox = r[0] / r[3], oy = r[1] / r[3], oz = r[2] / r[3];
}

// Per triangle: finite check, backface test, bounding box (src/Triangle.cpp:147-151,243-257; Rasterizer.cpp:203) — shared by
// k_setup (uploaded streams) and k_vertex (scenesets: the triangle is still in registers).  Returns "kept"; a culled or
// non-finite triangle gets the empty box.
__device__ __forceinline__ bool setup_triangle(const float (&P)[9], int W, int H, float ex, float ey, float ez, bool live, BBox &bb) {
  const float A0 = P[0], A1 = P[1], A2 = P[2], B0 = P[3], B1 = P[4], B2 = P[5], C0 = P[6], C1 = P[7], C2 = P[8];
  bb.sx = 1, bb.sy = 1, bb.ex = 0, bb.ey = 0;
  const bool finite = live && __builtin_isfinite(A0) && __builtin_isfinite(A1) && __builtin_isfinite(A2) && __builtin_isfinite(B0) &&
                      __builtin_isfinite(B1) && __builtin_isfinite(B2) && __builtin_isfinite(C0) && __builtin_isfinite(C1) &&
                      __builtin_isfinite(C2);
  bool keep = false;
  if (finite) {
    float e1x = B0 - A0, e1y = B1 - A1, e1z = B2 - A2, e2x = C0 - A0, e2y = C1 - A1, e2z = C2 - A2;
    float nx = e1y * e2z - e2y * e1z, ny = e1z * e2x - e2z * e1x, nz = e1x * e2y - e2x * e1y;
    BranchMath bm;
    normalize3(bm, nx, ny, nz);
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
  }
  return keep;
}

// A 12-byte piece of the dense position stream (9 floats per triangle, srz_device.h), moved three floats at a time
// (global_load_dwordx3 / global_store_dwordx3 or wider: dword alignment is all they need)
struct F3 {
  float x, y, z;
};
__device__ __forceinline__ F3 ld3(const SRZ_CAS float *p) { return F3{p[0], p[1], p[2]}; } // (one global_load_dwordx3)
// the 9 position floats of triangle i (ax ay z0 bx by z1 cx cy z2)
__device__ __forceinline__ void load_pos9(const SRZ_CAS float *p, float (&P)[9]) {
  const F3 v0 = ld3(p), v1 = ld3(p + 3), v2 = ld3(p + 6);
  P[0] = v0.x, P[1] = v0.y, P[2] = v0.z, P[3] = v1.x, P[4] = v1.y, P[5] = v1.z, P[6] = v2.x, P[7] = v2.y, P[8] = v2.z;
}

// (bbox_out != null: the triangle's setup — cull + bounding box — is done here too, from the registers that hold it; k_chunks
// then only reduces the boxes to the chunks' row ranges and the 96-byte triangles are not read back by a k_setup)
__global__ __launch_bounds__(256) void k_vertex(const DrawDesc *draws, srz_tri *tris, float *tri_pos, const FrameDesc *frames, BBox *bbox_out) {
  const SRZ_CAS DrawDesc *d = as_const(draws) + blockIdx.y;
  const uint32_t n_faces = d->n_faces;
  const SRZ_CAS srz_vertex *verts = as_const(d->verts);
  const SRZ_CAS uint32_t *faces = as_const(d->faces);
  const float zs = d->zscale, zo = d->zoffset;
  for (uint32_t f = blockIdx.x * 256 + threadIdx.x; f < n_faces; f += gridDim.x * 256) {
    srz_tri t;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const SRZ_CAS srz_vertex *v = verts + faces[3 * f + k];
      const float px = v->pos[0], py = v->pos[1], pz = v->pos[2], nx = v->nrm[0], ny = v->nrm[1], nz = v->nrm[2];
      float x, y, z;
      xform_div_w(d->ndc_mvp, px, py, pz, x, y, z);
      t.pos[k][0] = x, t.pos[k][1] = y, t.pos[k][2] = z * zs + zo;
      xform_div_w(d->normal_m, nx, ny, nz, x, y, z);
      t.nrm[k][0] = x, t.nrm[k][1] = y, t.nrm[k][2] = z;
      t.uv[k][0] = v->uv[0], t.uv[k][1] = v->uv[1];
    }
    const size_t ti = (size_t)d->tri_off + f;
    tris[ti] = t;
    const float P[9] = {t.pos[0][0], t.pos[0][1], t.pos[0][2], t.pos[1][0], t.pos[1][1], t.pos[1][2], t.pos[2][0], t.pos[2][1], t.pos[2][2]};
    F3 *po = reinterpret_cast<F3 *>(tri_pos + ti * TRI_POS_F); // (the dense copy of the positions: what k_setup and k_raster read)
    po[0] = F3{P[0], P[1], P[2]}, po[1] = F3{P[3], P[4], P[5]}, po[2] = F3{P[6], P[7], P[8]};
    if (bbox_out) {
      const SRZ_CAS FrameDesc *fd = as_const(frames) + d->frame;
      BBox bb;
      (void)setup_triangle(P, fd->width, fd->height, fd->eye[0], fd->eye[1], fd->eye[2], true, bb);
      bbox_out[ti] = bb;
    }
  }
}

// ================================================================================================================
// Wave-level scans on the DPP crossbar (no LDS traffic): inclusive prefix over the 64 lanes.
// row_shr:1,2,4,8 scan inside each row of 16 lanes (lanes without a source read the identity 0), row_bcast:15 / :31
// carry the row totals into the rows behind.
// ================================================================================================================
#define SRZ_DPP(v, ctrl, rmask) (uint32_t) __builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, false)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) {
  v += SRZ_DPP(v, 0x111, 0xf), v += SRZ_DPP(v, 0x112, 0xf), v += SRZ_DPP(v, 0x114, 0xf), v += SRZ_DPP(v, 0x118, 0xf);
  v += SRZ_DPP(v, 0x142, 0xa); // row_bcast:15 → rows 1 and 3
  v += SRZ_DPP(v, 0x143, 0xc); // row_bcast:31 → rows 2 and 3
  return v;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
  v = max(v, SRZ_DPP(v, 0x111, 0xf)), v = max(v, SRZ_DPP(v, 0x112, 0xf)), v = max(v, SRZ_DPP(v, 0x114, 0xf));
  v = max(v, SRZ_DPP(v, 0x118, 0xf));
  v = max(v, SRZ_DPP(v, 0x142, 0xa));
  v = max(v, SRZ_DPP(v, 0x143, 0xc));
  return v;
}

// ================================================================================================================
// Tight rectangles.  The reference tests every pixel of a triangle's bounding box; a pixel outside the triangle fails the test
// and leaves no trace, so pixels that are CERTAINLY outside need not be tested — and a tile or band the triangle certainly
// misses need not list it.  slab_extent gives the extent along u of (triangle ∩ slab lo <= w <= hi), (u, w) = (x, y) or (y, x),
// from the vertices inside the slab and the edges' crossings of its two bounds (+inf / -inf when the triangle misses it).
// CONSERVATIVE: a pixel can pass the rounded inside tests (cover_v / cover_s) only within a distance eps of the true triangle.
// With D = the vertices' extent and R = D + 1 >= every |pixel - vertex| component of a TESTED pixel (the box starts at
// trunc(min): up to one pixel before the extent, never behind it — guaranteed by the `near` test below, which sends triangles
// whose clamped box lies away from their extent, i.e. wholly off screen on an axis, to the plain box: there the operands grow
// with the distance to the screen, not with D), u = 2^-24:
//   S test (signs of AB x AP): products <= D R, |error| < 2^-22 D R; a pixel delta outside edge AB has |e| = delta |AB|, and
//       |AB| >= |area2| / (sqrt2 D) >= D / 362 under the sliver guard D^2 <= 256 |area2|, so eps < 2^-13.5 R;
//   V test (signs of alpha, beta, 1 - (alpha + beta)): aPBC = fmsub(PBx,PCy,PCx*PBy) has products <= R^2, |error| < 2^-21.7 R^2,
//       and the multiplication by v_inv keeps its sign: eps < 2^-21.7 R^2 * 362 / D = 2^-13.2 R^2 / D; for gamma the two errors
//       add and v_inv itself is off by <= 2^-13.7 relative (its operands are <= D, |area2| >= D^2 / 256):
//       eps < 2^-13.2 D + 2^-12.2 R^2 / D.
// Both stay below m = D / 512 + 1 / 64 for every D >= 2^-5 (at D = 2^-5: 2^-7.1; large D: 2^-11 D) — smaller triangles (their
// boxes are at most 2 x 2 pixels: nothing to gain), slivers, non-finite or > 2^20 coordinates keep the plain box.  m also covers
// v_rcp's error in the crossings (2^-22 of a length <= D).
// ================================================================================================================
__device__ __forceinline__ bool tight_margin(float ax, float ay, float bx, float by, float cx, float cy, float area2, int bsx, int bsy,
                                             int bex, int bey, float &m) {
  const float mnx = __builtin_fminf(__builtin_fminf(ax, bx), cx), mxx = __builtin_fmaxf(__builtin_fmaxf(ax, bx), cx);
  const float mny = __builtin_fminf(__builtin_fminf(ay, by), cy), mxy = __builtin_fmaxf(__builtin_fmaxf(ay, by), cy);
  const float D = __builtin_fmaxf(mxx - mnx, mxy - mny);
  m = __builtin_fmaf(D, 0.001953125f, 0.015625f);
  // the clamped box [bsx, bex] x [bsy, bey] lies inside [min - 1, max] of the vertices on both axes (conversions are exact)
  const bool near = ((float)bsx + 1.0f >= mnx) & ((float)bex <= mxx) & ((float)bsy + 1.0f >= mny) & ((float)bey <= mxy);
  return near && D * D <= 256.0f * __builtin_fabsf(area2) && D <= 1048576.0f && D >= 0.03125f; // (false for NaN / inf)
}
__device__ __forceinline__ void slab_extent(float au, float aw, float bu, float bw, float cu, float cw, float lo, float hi, float &mn,
                                            float &mx) {
  mn = __builtin_inff(), mx = -__builtin_inff();
  auto vertex = [&](float u, float w) {
    const bool in = (w >= lo) & (w <= hi);
    mn = in ? __builtin_fminf(mn, u) : mn, mx = in ? __builtin_fmaxf(mx, u) : mx;
  };
  auto edge = [&](float pu, float pw, float qu, float qw) {
    const float inv = __builtin_amdgcn_rcpf(pw - qw), du = qu - pu;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float bound = k ? hi : lo, dp = pw - bound, dq = qw - bound;
      const float u = __builtin_fmaf(du, dp * inv, pu); // (pw == qw: no crossing, the NaN / inf is not selected)
      const bool cross = (dp < 0.0f) != (dq < 0.0f);
      mn = cross ? __builtin_fminf(mn, u) : mn, mx = cross ? __builtin_fmaxf(mx, u) : mx;
    }
  };
  vertex(au, aw), vertex(bu, bw), vertex(cu, cw);
  edge(au, aw, bu, bw), edge(bu, bw, cu, cw), edge(cu, cw, au, aw);
}
// the integer range [i0, i1] clipped to ceil(mn - m) .. floor(mx + m) (as floats first: the conversions stay in range); an
// extent that misses the range gives i0 > i1
__device__ __forceinline__ void clip_range(int &i0, int &i1, float mn, float mx, float m) {
  const float f0 = (float)i0, f1 = (float)i1;
  i0 = (int)__builtin_fminf(__builtin_fmaxf(__builtin_ceilf(mn - m), f0), f1 + 1.0f);
  i1 = (int)__builtin_fmaxf(__builtin_fminf(__builtin_floorf(mx + m), f1), f0 - 1.0f);
}

// ================================================================================================================
// bucket_group — the O(triangles) half of the binning, shared by k_setup and k_chunks: the GROUP_TRIS triangles of one group
// (GROUP_K per thread: triangle g * GROUP_TRIS + k * 256 + tid) are sorted by the local 32-row bands their bounding boxes reach, in LDS:
//   count  one LDS atomic per (triangle, band)             scan  exclusive prefix over the bands (wave 0, DPP)
//   fill   a second LDS atomic hands out the slots; the entry {triangle, tile-x range} goes to the group's own region
// and the group's descriptor row [band] = first entry << 16 | entries.  No global atomics, no cross-workgroup state.
// A group that needs more than ENT_PER_GROUP entries (or holds a triangle more than 64 local bands tall) is marked
// DESC_RAW in every band instead: k_bin's band workgroups then walk its bounding boxes themselves.
// ================================================================================================================
__device__ __forceinline__ void bucket_group(const RenderArgs &a, const uint32_t nlb, const uint32_t group, const uint32_t t0,
                                             const bool packed, const uint32_t tri_off, const BBox (&bb)[GROUP_K],
                                             const bool (&keep)[GROUP_K], const float (*P)[9], uint32_t *s_cnt, uint32_t *s_off,
                                             uint32_t *s_fill, uint32_t *s_misc) {
  const int tid = (int)threadIdx.x, lane = tid & 63;
  const int world = a.shard_world, rank = a.shard_rank;
  for (uint32_t i = (uint32_t)tid; i < nlb; i += 256u) s_cnt[i] = 0u;
  if (tid == 0) s_misc[0] = 0u;
  __syncthreads();
  // the local bands of triangle k: lb0[k] .. lb0[k] + nb[k] - 1 (band b is local iff rank_of_band(b) == rank; local index b / world: one
  // band of every group of `world` bands, so the local indices a box reaches are consecutive)
  int lb0[GROUP_K], nb[GROUP_K];
#pragma unroll
  for (int k = 0; k < (int)GROUP_K; ++k) {
    lb0[k] = 0, nb[k] = 0;
    if (keep[k]) {
      const int b0 = (int)bb[k].sy >> 5, b1 = (int)bb[k].ey >> 5;
      int g0 = b0 / world, g1 = b1 / world;
      if (band_of(g0, rank, world) < b0) ++g0;
      if (band_of(g1, rank, world) > b1) --g1;
      if (g0 <= g1) lb0[k] = g0, nb[k] = g1 - g0 + 1;
      if (nb[k] > 64) s_misc[0] = 1u, nb[k] = 0; // (benign race: every writer stores 1)
      for (int j = 0; j < nb[k]; ++j) atomicAdd(&s_cnt[lb0[k] + j], 1u);
    }
  }
  __syncthreads();
  if (tid < 64) {
    uint32_t run = 0;
    for (uint32_t i0 = 0; i0 < nlb; i0 += 64u) {
      const uint32_t i = i0 + (uint32_t)lane, v = i < nlb ? s_cnt[i] : 0u, incl = wave_scan_add(v);
      if (i < nlb) s_off[i] = s_fill[i] = run + incl - v;
      run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0 && run > ENT_PER_GROUP) s_misc[0] = 1u;
  }
  __syncthreads();
  const bool raw = s_misc[0] != 0u;
  // what a tile list holds of a triangle: its index in the frame — and, FD_PACKED, its batch above it (coalesced 2-byte loads here,
  // instead of a gather per staged triangle in k_shade)
  uint32_t tpk[GROUP_K];
#pragma unroll
  for (int k = 0; k < (int)GROUP_K; ++k) {
    const uint32_t t = t0 + (uint32_t)k * 256u + (uint32_t)tid;
    tpk[k] = t;
    if (packed && keep[k]) tpk[k] = t | ((uint32_t)as_const(a.tri_batch)[tri_off + t] << PACK_IDX_BITS);
  }
  uint32_t *desc = a.band_desc + (size_t)group * nlb;
  for (uint32_t i = (uint32_t)tid; i < nlb; i += 256u) desc[i] = raw ? DESC_RAW : ((s_off[i] << 16) | s_cnt[i]);
  if (!raw) {
    uint2 *ent = a.band_ent + (size_t)group * ENT_PER_GROUP;
#pragma unroll
    for (int k = 0; k < (int)GROUP_K; ++k) {
      const uint32_t xr = (uint32_t)((int)bb[k].sx >> 5) | ((uint32_t)((int)bb[k].ex >> 5) << 16);
      // (P: the positions, where the caller has them in registers) the tile range of a band is tightened to the triangle's
      // x-extent inside the band's rows (tight_margin): the tiles in the corners of a large bounding box drop out of the lists
      float m = 0.0f;
      const bool tight = P != nullptr && nb[k] > 0 &&
                         tight_margin(P[k][0], P[k][1], P[k][3], P[k][4], P[k][6], P[k][7],
                                      (P[k][3] - P[k][0]) * (P[k][7] - P[k][1]) - (P[k][4] - P[k][1]) * (P[k][6] - P[k][0]), bb[k].sx,
                                      bb[k].sy, bb[k].ex, bb[k].ey, m);
      for (int j = 0; j < nb[k]; ++j) {
        uint32_t xr_j = xr;
        if (tight) {
          const int b = band_of(lb0[k] + j, rank, world);
          int X0 = bb[k].sx, X1 = bb[k].ex;
          float mn, mx;
          slab_extent(P[k][0], P[k][1], P[k][3], P[k][4], P[k][6], P[k][7], (float)max(b * BAND, (int)bb[k].sy) - m,
                      (float)min(b * BAND + BAND - 1, (int)bb[k].ey) + m, mn, mx);
          clip_range(X0, X1, mn, mx, m);
          xr_j = X0 <= X1 ? ((uint32_t)(X0 >> 5) | ((uint32_t)(X1 >> 5) << 16)) : 1u; // (1: tiles 1 .. 0 = none)
        }
        ent[atomicAdd(&s_fill[lb0[k] + j], 1u)] = make_uint2(tpk[k], xr_j);
      }
    }
  }
  __syncthreads(); // (LDS is reused by this workgroup's next group)
}

// row range of the kept triangles of a 64-triangle chunk (= one wave's 64 consecutive triangles): k_raster_slow and the
// DESC_RAW walk of k_bin skip the chunks that cannot reach their rows without reading the 64 bounding boxes
__device__ __forceinline__ void chunk_rows_store(const RenderArgs &a, const uint32_t chunk_off, const uint32_t t, const uint32_t n_tris,
                                                 const bool keep, const BBox &bb) {
  int lo = keep ? (int)bb.sy : 0x7fff, hi = keep ? (int)bb.ey : -1;
  for (int o = 32; o > 0; o >>= 1) lo = min(lo, __shfl_xor(lo, o)), hi = max(hi, __shfl_xor(hi, o));
  if ((threadIdx.x & 63) == 0 && (t & ~63u) < n_tris) a.chunk_rows[chunk_off + t / 64u] = ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16);
}

__device__ __forceinline__ void reset_render_counters(const RenderArgs &a) {
  // the work lists, k_shade's cursors into them, the record pool and the slow list start empty
  if (threadIdx.x < N_WORK_LISTS) a.work_count[threadIdx.x * CNT_STRIDE] = 0u, a.work_count[threadIdx.x * CNT_STRIDE + 1] = 0u;
  if (threadIdx.x <= a.pool_sub_mask) a.pool_heads[threadIdx.x * CNT_STRIDE] = 0u;
  if (threadIdx.x == 0) *a.slow_count = 0u, *a.redo_count = 0u;
}

// ================================================================================================================
// k_setup — per triangle: finite check, bbox, backface test (src/Triangle.cpp:147-151,243-257; Rasterizer.cpp:203);
// per group of GROUP_TRIS triangles (one workgroup pass, GROUP_K triangles per thread): the band sort above
// ================================================================================================================
template <bool STATS>
__global__ __launch_bounds__(256) void k_setup(RenderArgs a, BBox *bbox_out) {
  __shared__ uint32_t s_cnt[MAX_LOCAL_BANDS], s_off[MAX_LOCAL_BANDS], s_fill[MAX_LOCAL_BANDS], s_misc[2];
  if (blockIdx.x == 0 && blockIdx.y == 0) reset_render_counters(a);
  const SRZ_CAS FrameDesc *fd = as_const(a.frames) + blockIdx.y;
  const int W = fd->width, H = fd->height;
  const uint32_t n_tris = fd->n_tris, tri_off = fd->tri_off;
  const float ex = fd->eye[0], ey = fd->eye[1], ez = fd->eye[2];
  unsigned long long n_culled = 0, tests = 0;
  const uint32_t n_groups = (n_tris + GROUP_TRIS - 1u) / GROUP_TRIS;
  for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t t0 = g * GROUP_TRIS;
    float P[GROUP_K][9];
#pragma unroll
    for (int k = 0; k < (int)GROUP_K; ++k) { // (all the loads of the thread's triangles in flight together)
      const uint32_t t = t0 + (uint32_t)k * 256u + threadIdx.x;
      load_pos9(as_const(a.tri_pos) + (size_t)(tri_off + (t < n_tris ? t : 0u)) * a.pos_stride, P[k]);
    }
    BBox bb[GROUP_K];
    bool keep[GROUP_K];
#pragma unroll
    for (int k = 0; k < (int)GROUP_K; ++k) {
      const uint32_t t = t0 + (uint32_t)k * 256u + threadIdx.x;
      const bool live = t < n_tris;
      keep[k] = setup_triangle(P[k], W, H, ex, ey, ez, live, bb[k]);
      if (STATS) {
        if (keep[k])
          tests += (unsigned long long)(bb[k].ex - bb[k].sx + 1) * (unsigned long long)(bb[k].ey - bb[k].sy + 1);
        else if (live)
          n_culled++;
      }
      if (live) bbox_out[tri_off + t] = bb[k];
      chunk_rows_store(a, fd->chunk_off, t, n_tris, keep[k], bb[k]);
    }
    bucket_group(a, fd->n_local_bands, fd->group_off + g, t0, (fd->flags & FD_PACKED) != 0u, tri_off, bb, keep, P, s_cnt, s_off, s_fill, s_misc);
  }
  if (STATS) {
    if (n_culled) atomicAdd(&a.stats[ST_CULLED], n_culled);
    if (tests) atomicAdd(&a.stats[ST_PIXEL_TESTS], tests);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&a.stats[ST_TRIS], (unsigned long long)n_tris);
  }
}

// k_chunks — scenesets: k_vertex has written the bounding boxes; what is left of k_setup is the reset of the per-render
// counters, the row range of every 64-triangle chunk and the band sort (8 bytes read per triangle instead of 96)
__global__ __launch_bounds__(256) void k_chunks(RenderArgs a) {
  __shared__ uint32_t s_cnt[MAX_LOCAL_BANDS], s_off[MAX_LOCAL_BANDS], s_fill[MAX_LOCAL_BANDS], s_misc[2];
  if (blockIdx.x == 0 && blockIdx.y == 0) reset_render_counters(a);
  const SRZ_CAS FrameDesc *fd = as_const(a.frames) + blockIdx.y;
  const uint32_t n_tris = fd->n_tris, tri_off = fd->tri_off;
  const SRZ_CAS u32x2 *bbox = as_const(reinterpret_cast<const u32x2 *>(a.bbox + tri_off));
  const uint32_t n_groups = (n_tris + GROUP_TRIS - 1u) / GROUP_TRIS;
  for (uint32_t g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const uint32_t t0 = g * GROUP_TRIS;
    BBox bb[GROUP_K];
    bool keep[GROUP_K];
#pragma unroll
    for (int k = 0; k < (int)GROUP_K; ++k) {
      const uint32_t t = t0 + (uint32_t)k * 256u + threadIdx.x;
      bb[k].sx = 1, bb[k].sy = 1, bb[k].ex = 0, bb[k].ey = 0;
      if (t < n_tris) {
        const u32x2 r = bbox[t];
        bb[k].sx = (int16_t)(r.x & 0xffff), bb[k].sy = (int16_t)(r.x >> 16), bb[k].ex = (int16_t)(r.y & 0xffff), bb[k].ey = (int16_t)(r.y >> 16);
      }
      keep[k] = bb[k].sx <= bb[k].ex; // (the empty box of a culled triangle: sx > ex)
      chunk_rows_store(a, fd->chunk_off, t, n_tris, keep[k], bb[k]);
    }
    bucket_group(a, fd->n_local_bands, fd->group_off + g, t0, (fd->flags & FD_PACKED) != 0u, tri_off, bb, keep, nullptr, s_cnt, s_off, s_fill, s_misc);
  }
}

// ================================================================================================================
// k_bin — one WORKGROUP per (frame, local band): the triangles whose bbox touches the band are appended to the lists
// of the 32x32 tiles they overlap.  The rasteriser's result does not depend on the order of a tile's list (k_raster
// resolves depth ties with the triangle index, not with the list position), so this is a plain count / scan / fill with
// LDS atomics — no ordered compaction:
//   input   the band's entries, sorted out by k_setup / k_chunks (bucket_group): per group of GROUP_TRIS triangles one descriptor
//           word (first entry, entries) — 64 groups per wave load — whose entry runs are flattened into dense batches of
//           64 (the triangles that start inside a batch mark their position, a DPP max-scan gives every lane its group,
//           its descriptor comes over the LDS crossbar): the walk reads 8 bytes per (triangle, band) pair, nothing else.
//           Groups marked DESC_RAW (huge triangles) are walked the old way: chunk row ranges, then bounding boxes.
//   pass 1  every entry bumps the LDS counter of each tile it overlaps
//   scan    exclusive prefix over the tile counters (wave 0) = the band's layout; ONE global atomic takes the band's
//           records from the sub-pool of this workgroup (k_raster_slow serves the tiles of a band that does not fit)
//   pass 2  every (triangle, tile) pair becomes one 4-byte index at pool[band base + tile offset + slot]: assembled in LDS,
//           written as one coalesced run (k_raster gathers the triangles' 36 bytes of positions and their 8-byte boxes
//           through these indices)
// ================================================================================================================
constexpr int BIN_MAX_WAVES = 8; // launched with 2, 4 or 8 waves: the walk is latency-bound, so short streams take small
                                 // workgroups (more of them resident per CU), long ones more waves per band
constexpr uint32_t BIN_STAGE = 4096; // indices of one band staged in LDS (16 KB)
// The entries a wave met on its first walk (count) are kept in LDS for its second one (fill): BIN_KEEP batches of 64 per wave in the
// eight-wave build (16 KB: the wave slots, not the LDS, bound that build to four workgroups per CU), one batch in the smaller ones (whose
// eight workgroups per CU leave 2 KB each).  A wave that met more batches, or a DESC_RAW group, walks again as before (round 6)
__host__ __device__ constexpr uint32_t bin_keep(uint32_t waves) { return waves >= 8u ? 4u : 1u; }
// dynamic LDS = (3 * tiles_x + 64 * waves + 4 + BIN_STAGE + 128 * waves * bin_keep(waves)) dwords: [count | offset | fill cursor] per tile,
// per-wave marks, stage, kept entries
__global__ __launch_bounds__(64 * BIN_MAX_WAVES) void k_bin(RenderArgs a) {
  const int BIN_WAVES = (int)(blockDim.x >> 6);
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  const uint32_t TX = a.tiles_x;
  uint32_t *s_cnt = s_dyn, *s_off = s_dyn + TX, *s_fill = s_dyn + 2 * TX, *s_marks = s_dyn + 3 * TX;
  uint32_t *s_misc = s_marks + 64 * BIN_WAVES; // [0] first index of the band in pool[] (or UNLISTED), [1] indices of the band
  uint32_t *s_stage = s_misc + 4;

  // same XCD-aware decomposition as k_raster: workgroup i bins frame (i % 8) of its group of 8 frames, so a frame's
  // records are written through the L2 of the XCD that will rasterise it
  const uint32_t wg = blockIdx.x, xcd = wg & 7u, jj = wg >> 3;
  const uint32_t frame = (jj / a.n_local_bands) * 8u + xcd, lb = jj % a.n_local_bands;
  if (frame >= a.n_frames) return; // workgroup-uniform
  const SRZ_CAS FrameDesc *fd = as_const(a.frames) + frame;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t KEEP = bin_keep((uint32_t)BIN_WAVES);
  // (8-byte entries: behind the stage, at an even dword whatever the frame's width in tiles)
  u32x2 *s_keep = reinterpret_cast<u32x2 *>(s_dyn + ((3u * TX + 64u * (uint32_t)BIN_WAVES + 4u + BIN_STAGE + 1u) & ~1u)) + (size_t)wave * KEEP * 64u;
  uint32_t n_seen = 0;    // batches this wave has met on the current walk (wave-uniform)
  bool keep_ok = true;    // ... and none of them came from a DESC_RAW group
  for (uint32_t w = threadIdx.x; w < 3 * TX; w += blockDim.x) s_dyn[w] = 0u;
  uint32_t *s_mark = s_marks + 64 * wave;
  s_mark[lane] = 0u;
  __syncthreads();
  const uint32_t n_tris = fd->n_tris;
  const bool packed = (fd->flags & FD_PACKED) != 0u; // (tile-list entries carry the batch above the index)
  const int band = band_of((int)lb, a.shard_rank, a.shard_world);
  const int y0 = band * BAND, y1 = y0 + BAND - 1;
  const SRZ_CAS u32x2 *bbox = as_const(reinterpret_cast<const u32x2 *>(a.bbox + fd->tri_off));
  const SRZ_CAS uint32_t *chunk_rows = as_const(a.chunk_rows) + fd->chunk_off;
  const uint32_t n_chunks = (n_tris + 63) / 64, n_groups = (n_tris + GROUP_TRIS - 1u) / GROUP_TRIS;
  const uint32_t nlb = a.n_local_bands;
  const SRZ_CAS uint32_t *desc = as_const(a.band_desc) + (size_t)fd->group_off * nlb + lb;
  const SRZ_CAS u32x2 *ents = reinterpret_cast<const SRZ_CAS u32x2 *>(as_const(a.band_ent) + (size_t)fd->group_off * ENT_PER_GROUP);

  // a DESC_RAW group: its 16 chunks' row ranges, then the bounding boxes of the chunks that can reach the band, the loads of
  // four candidate chunks in flight together
  auto walk_raw = [&](const uint32_t g, auto &&fn) {
    const uint32_t c0 = g * (GROUP_TRIS / 64u);
    bool cand = false;
    const uint32_t c = c0 + (uint32_t)lane;
    if (lane < (int)(GROUP_TRIS / 64u) && c < n_chunks) {
      const uint32_t r = chunk_rows[c];
      cand = (int)(int16_t)(r & 0xffffu) <= y1 && (int)(int16_t)(r >> 16) >= y0;
    }
    unsigned long long mc = __ballot(cand);
    while (mc) {
      uint32_t cj[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        ok[u] = mc != 0ull;
        cj[u] = ok[u] ? (uint32_t)__builtin_ctzll(mc) : 0u;
        mc &= mc - 1ull; // (0 stays 0)
      }
      u32x2 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = bbox[min((c0 + cj[u]) * 64u + (uint32_t)lane, n_tris - 1u)];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (!ok[u]) break; // wave-uniform
        const uint32_t t = (c0 + cj[u]) * 64u + (uint32_t)lane;
        const int sx = (int16_t)(r[u].x & 0xffff), sy = (int16_t)(r[u].x >> 16), ex = (int16_t)(r[u].y & 0xffff), ey = (int16_t)(r[u].y >> 16);
        const bool hit = t < n_tris && sx <= ex && sy <= y1 && ey >= y0; // (bbox is clamped to the frame by k_setup)
        uint32_t tp = t;
        if (packed && hit) tp |= (uint32_t)as_const(a.tri_batch)[fd->tri_off + t] << PACK_IDX_BITS;
        fn(tp, hit, sx >> 5, ex >> 5);
      }
    }
  };
  // calls fn(triangle, hit, first tile x, last tile x) wave-wide for every triangle of the frame that reaches the band: every wave
  // reads the descriptors of all groups (64 per round) and takes every BIN_WAVES-th batch of 64 entries, continuing round-robin
  // across the rounds, so the entries of a frame with few groups are still spread over all the waves
  auto walk = [&](auto &&fn) {
    const uint32_t NW = (uint32_t)BIN_WAVES;
    uint32_t first = (uint32_t)wave; // the first batch of the next round that is this wave's
    for (uint32_t r0 = 0; r0 < n_groups; r0 += 64u) {
      const uint32_t g = r0 + (uint32_t)lane;
      const uint32_t d = g < n_groups ? desc[(size_t)g * nlb] : 0u;
      const bool raw = d == DESC_RAW;
      const uint32_t cnt = raw ? 0u : (d & 0xffffu), off = d >> 16;
      const uint32_t incl = wave_scan_add(cnt), excl = incl - cnt;
      const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63), nbat = (T + 63u) >> 6;
      uint32_t b = first;
      for (; b < nbat; b += NW) {
        const uint32_t P0 = b * 64u;
        // the group that reaches into this batch from before it: the last one that starts before P0
        const unsigned long long before = __ballot(cnt != 0u && excl < P0);
        const uint32_t carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
        const uint32_t r = excl - P0;
        if (cnt != 0u && r < 64u) s_mark[r] = (uint32_t)lane + 1u;
        __builtin_amdgcn_wave_barrier();
        uint32_t m = s_mark[lane];
        __builtin_amdgcn_wave_barrier();
        s_mark[lane] = 0u;
        m = max(wave_scan_max(m), carry);
        const uint32_t sl = (m - 1u) & 63u; // the lane that holds this entry's group
        const uint32_t e_excl = (uint32_t)__builtin_amdgcn_ds_bpermute((int)sl * 4, (int)excl);
        const uint32_t e_off = (uint32_t)__builtin_amdgcn_ds_bpermute((int)sl * 4, (int)off);
        const uint32_t e = P0 + (uint32_t)lane;
        const bool valid = e < T;
        u32x2 en = {0u, 1u}; // (tiles 1 .. 0: none)
        if (valid) en = ents[(size_t)(r0 + sl) * ENT_PER_GROUP + e_off + (e - e_excl)];
        if (n_seen < KEEP) s_keep[n_seen * 64u + (uint32_t)lane] = en;
        ++n_seen;
        fn(en.x, valid, (int)(en.y & 0xffffu), (int)(en.y >> 16));
      }
      first = b - nbat; // (b is the first batch index >= nbat that is this wave's: b - nbat < NW)
      uint32_t ri = 0;  // DESC_RAW groups of the round: dealt to the waves one by one
      for (unsigned long long mr = __ballot(raw); mr != 0ull; mr &= mr - 1ull, ++ri)
        if (ri % NW == (uint32_t)wave) keep_ok = false, walk_raw(r0 + (uint32_t)__builtin_ctzll(mr), fn);
    }
  };
  // ---- pass 1: tile counts ---------------------------------------------------------------------------------------------
  walk([&](uint32_t, bool hit, int tlo, int thi) {
    if (hit)
      for (int tx = tlo; tx <= thi; ++tx) atomicAdd(&s_cnt[tx], 1u);
  });
  __syncthreads();
  // ---- scan (wave 0) + the band's allocation -----------------------------------------------------------------------------
  if (wave == 0) {
    uint32_t run = 0;
    for (uint32_t t0 = 0; t0 < TX; t0 += 64) {
      const uint32_t tx = t0 + lane;
      const uint32_t v = tx < TX ? s_cnt[tx] : 0u;
      const uint32_t incl = wave_scan_add(v);
      if (tx < TX) s_off[tx] = run + incl - v;
      run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
    if (lane == 0) {
      uint32_t base = 0;
      if (run) {
        const uint32_t sub = wg & a.pool_sub_mask;
        const uint32_t start = atomicAdd(&a.pool_heads[sub * CNT_STRIDE], run); // (also the host's measure of what the render needed)
        base = (start <= a.pool_sub_cap && run <= a.pool_sub_cap - start) ? sub * a.pool_sub_cap + start : UNLISTED;
      }
      s_misc[0] = base, s_misc[1] = run;
    }
  }
  __syncthreads();
  const uint32_t base = s_misc[0];
  {
    uint2 *ti = a.tile_info + ((size_t)frame * a.n_local_bands + lb) * TX;
    for (uint32_t tx = threadIdx.x; tx < TX; tx += blockDim.x) ti[tx] = make_uint2(s_cnt[tx], base == UNLISTED ? UNLISTED : base + s_off[tx]);
  }
  if (base == UNLISTED || s_misc[1] == 0u) return; // workgroup-uniform
  // ---- pass 2: fill ------------------------------------------------------------------------------------------------------
  // The band's lists hold 4-byte triangle indices (the rasteriser gathers the positions of an index itself): they are
  // assembled in LDS — slots handed out by LDS atomics, tile after tile — and leave as ONE run of coalesced stores.  A band
  // with more pairs than the stage holds stores its indices straight into the pool.
  uint32_t *out = a.pool + base;
  const uint32_t run = s_misc[1];
  const bool staged = run <= BIN_STAGE; // workgroup-uniform
  auto fill = [&](uint32_t t, bool hit, int tlo, int thi) {
    if (hit)
      for (int tx = tlo; tx <= thi; ++tx) {
        const uint32_t slot = s_off[tx] + atomicAdd(&s_fill[tx], 1u);
        if (staged)
          s_stage[slot] = t;
        else
          out[slot] = t;
      }
  };
  if (keep_ok && n_seen <= KEEP) { // (wave-uniform) everything this wave met on the first walk is still in LDS: no second walk
    for (uint32_t k = 0; k < n_seen; ++k) {
      const u32x2 en = s_keep[k * 64u + (uint32_t)lane];
      fill(en.x, true, (int)(en.y & 0xffffu), (int)(en.y >> 16)); // (a lane without an entry kept tiles 1 .. 0)
    }
  } else {
    n_seen = KEEP; // (nothing more is kept)
    walk(fill);
  }
  if (staged) {
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < run; i += blockDim.x) out[i] = s_stage[i];
  }
}

// ================================================================================================================
// Fragment shaders
// ================================================================================================================
struct FrameK { // per-frame constants, wave-uniform (live in SGPRs)
  float eye[3], ka[3], ks[3], p, kh, kn;
  uint32_t n_lights;
  // FD_GREY (decided on the host): ka, ks and every light's intensity have three bit-equal channels.  The three channels of a
  // Blinn-Phong sum then differ only through the surface colour kd: with kd = (1, 1, 1) — the PHONG shader — they are the SAME
  // operations on the SAME operands, so the FAST builds compute one channel and copy it (identical bits by construction).  Only
  // BEHIND a pixel's light terms: a branch inside them (one division instead of three in the scalar path) cost the TEXTURE builds
  // 1.5 % — the terms are kept one straight line so that their independent normalisations interleave
  bool grey;
  const SRZ_CAS srz_light *lights;
};
struct ShadeDesc { // what a batch's Shader object holds: type + texture (Shader::texture, width_256/height_256)
  int shader, tw, th;
  const SRZ_CAS uint32_t *tex;
};

// BlinnPhong<__m256> for one light (include/shader/Shader.hpp:104-229)
template <class M, int NL>
__device__ __forceinline__ void v_blinn_phong(M &m, float nx, float ny, float nz, const FrameK &K, float kdr, float kdg, float kdb,
                                              const SRZ_CAS srz_light *L, float px, float py, float pz, float &o0, float &o1,
                                              float &o2) {
  const float Lx = L->pos[0], Ly = L->pos[1], Lz = L->pos[2], I0 = L->intensity[0], I1 = L->intensity[1], I2 = L->intensity[2];
  float lx = Lx - px, ly = Ly - py, lz = Lz - pz;
  float att = m.rsqrt2(fmaf_(lx, lx, ly * ly));
  float d0 = I0 * att, d1 = I1 * att, d2 = I2 * att;
  float hx = lx + (K.eye[0] - px), hy = ly + (K.eye[1] - py), hz = lz + (K.eye[2] - pz);
  v_normalized(m, hx, hy, hz);
  float nlx = lx, nly = ly, nlz = lz;
  v_normalized(m, nlx, nly, nlz);
  float cosA = sse_max(0.0f, fmaf_(nlx, nx, fmaf_(nly, ny, nlz * nz)));
  float cosT = pow_frame<NL>(m, sse_max(0.0f, fmaf_(hx, nx, fmaf_(hy, ny, hz * nz))), K.p);
  o0 = kdr * fmaf_(K.ka[0], I0, fmaf_(d0 * kdr, cosA, (d0 * K.ks[0]) * cosT));
  o1 = kdg * fmaf_(K.ka[1], I1, fmaf_(d1 * kdg, cosA, (d1 * K.ks[1]) * cosT));
  o2 = kdb * fmaf_(K.ka[2], I2, fmaf_(d2 * kdb, cosA, (d2 * K.ks[2]) * cosT));
}
// The same light in two steps: everything that does not involve the surface colour kd (attenuation, the two cosines, the
// power), then the combination.  The FAST build runs step 1 for BOTH lights between issuing the texel load and using it, so
// the load's latency passes under ~200 instructions instead of ~70 (same operations on the same operands: same bits).
struct LightTerms {
  float d0, d1, d2, cosA, cosT;
};
template <class M, int NL>
__device__ __forceinline__ void v_blinn_phong_terms(M &m, float nx, float ny, float nz, const FrameK &K, const SRZ_CAS srz_light *L,
                                                    float px, float py, float pz, LightTerms &t) {
  const float Lx = L->pos[0], Ly = L->pos[1], Lz = L->pos[2], I0 = L->intensity[0], I1 = L->intensity[1], I2 = L->intensity[2];
  float lx = Lx - px, ly = Ly - py, lz = Lz - pz;
  float att = m.rsqrt2(fmaf_(lx, lx, ly * ly));
  t.d0 = I0 * att, t.d1 = I1 * att, t.d2 = I2 * att;
  float hx = lx + (K.eye[0] - px), hy = ly + (K.eye[1] - py), hz = lz + (K.eye[2] - pz);
  v_normalized(m, hx, hy, hz);
  float nlx = lx, nly = ly, nlz = lz;
  v_normalized(m, nlx, nly, nlz);
  t.cosA = sse_max(0.0f, fmaf_(nlx, nx, fmaf_(nly, ny, nlz * nz)));
  t.cosT = pow_frame<NL>(m, sse_max(0.0f, fmaf_(hx, nx, fmaf_(hy, ny, hz * nz))), K.p);
}
__device__ __forceinline__ void v_blinn_phong_combine(const FrameK &K, const SRZ_CAS srz_light *L, const LightTerms &t, float kdr,
                                                      float kdg, float kdb, float &o0, float &o1, float &o2) {
  const float I0 = L->intensity[0], I1 = L->intensity[1], I2 = L->intensity[2];
  o0 = kdr * fmaf_(K.ka[0], I0, fmaf_(t.d0 * kdr, t.cosA, (t.d0 * K.ks[0]) * t.cosT));
  o1 = kdg * fmaf_(K.ka[1], I1, fmaf_(t.d1 * kdg, t.cosA, (t.d1 * K.ks[1]) * t.cosT));
  o2 = kdb * fmaf_(K.ka[2], I2, fmaf_(t.d2 * kdb, t.cosA, (t.d2 * K.ks[2]) * t.cosT));
}

// Shader::applyFragmentShader SIMD overload + simd_*_impl (src/Shader.cpp:128-386); colour out in [0,255]
template <class M, int SH, int NL>
__device__ __forceinline__ void v_shade(M &m, const FrameK &K, const ShadeDesc &sd, float px, float py, float pz, float nx, float ny,
                                        float nz, float u, float v, float &r0, float &r1, float &r2) {
  const int shader = SH >= 0 ? SH : sd.shader;
  constexpr int NLC = light_count<NL>();
  const uint32_t n_lights = NLC > 0 ? (uint32_t)NLC : K.n_lights;
  float c0 = 1.0f, c1 = 1.0f, c2 = 1.0f;
  if (shader == SRZ_SHADER_NORMAL) {
    c0 = (nx + 1.0f) * 0.5f, c1 = (ny + 1.0f) * 0.5f, c2 = (nz + 1.0f) * 0.5f;
  } else if (shader == SRZ_SHADER_TEXTURE || shader == SRZ_SHADER_PHONG) {
    float kd0 = 1.0f, kd1 = 1.0f, kd2 = 1.0f;
    uint32_t texel = 0u;
    if (shader == SRZ_SHADER_TEXTURE) {
      float tw = (float)sd.tw, th = (float)sd.th;
      u = u * tw, v = v * th;
      u = sse_max(0.0f, sse_min(u, tw - 1.0f));
      v = sse_max(0.0f, sse_min(v, th - 1.0f));
      int32_t xi = cvt_rne_i32(u), yi = cvt_rne_i32(v);
      texel = sd.tex[(size_t)yi * sd.tw + xi];
    }
    auto decode = [&]() {
      if (shader == SRZ_SHADER_TEXTURE) {
        const float inv255 = 1.0f / 255.0f;
        kd0 = (float)(texel & 0xffu) * inv255, kd1 = (float)((texel >> 8) & 0xffu) * inv255,
        kd2 = (float)((texel >> 16) & 0xffu) * inv255;
      }
    };
    c0 = c1 = c2 = 0.0f;
    if constexpr (NLC > 0) { // every light's colour-independent terms first, the texel only after them
      LightTerms t[NLC];
#pragma unroll
      for (int l = 0; l < NLC; ++l) v_blinn_phong_terms<M, NL>(m, nx, ny, nz, K, K.lights + l, px, py, pz, t[l]);
      asm volatile("" : "+v"(t[0].cosT), "+v"(t[NLC - 1].cosT), "+v"(texel)); // (keeps the decode below the terms)
      decode();
      if (SH == SRZ_SHADER_PHONG && K.grey) { // (wave-uniform) kd = 1 and grey constants: channel 0's operations ARE the other two's
#pragma unroll
        for (int l = 0; l < NLC; ++l) {
          const float I0 = (K.lights + l)->intensity[0];
          c0 = c0 + 1.0f * fmaf_(K.ka[0], I0, fmaf_(t[l].d0 * 1.0f, t[l].cosA, (t[l].d0 * K.ks[0]) * t[l].cosT));
        }
        r0 = r1 = r2 = sse_min(sse_max(c0, 0.0f), 1.0f) * 255.0f;
        return;
      }
#pragma unroll
      for (int l = 0; l < NLC; ++l) { // (summed in the lights' order, like the loop below)
        float o0, o1, o2;
        v_blinn_phong_combine(K, K.lights + l, t[l], kd0, kd1, kd2, o0, o1, o2);
        c0 = c0 + o0, c1 = c1 + o1, c2 = c2 + o2;
      }
    } else {
      decode();
      for (uint32_t l = 0; l < n_lights; ++l) {
        float o0, o1, o2;
        v_blinn_phong<M, NL>(m, nx, ny, nz, K, kd0, kd1, kd2, K.lights + l, px, py, pz, o0, o1, o2);
        c0 = c0 + o0, c1 = c1 + o1, c2 = c2 + o2;
      }
    }
  }
  // DISPLACEMENT / BUMP: the reference's SIMD versions are empty stubs (src/Shader.cpp:388-444) → (1,1,1) → 255
  r0 = sse_min(sse_max(c0, 0.0f), 1.0f) * 255.0f;
  r1 = sse_min(sse_max(c1, 0.0f), 1.0f) * 255.0f;
  r2 = sse_min(sse_max(c2, 0.0f), 1.0f) * 255.0f;
}

// TextureLoader::getTextureColor(vec2) (src/TextureLoader.cpp:14-31)
template <class M>
__device__ __forceinline__ void s_texel(M &m, const ShadeDesc &sd, float u, float v, float &o0, float &o1, float &o2) {
  float cu = std_clamp(u, 0.0f, 1.0f), cv = std_clamp(v, 0.0f, 1.0f);
  float fx = cu * (float)sd.tw, fy = cv * (float)sd.th;
  int x = (int)fx, y = (int)fy;
  if (x < 0 || x >= sd.tw || y < 0 || y >= sd.th) {
    o0 = o1 = o2 = 0.0f;
    return;
  }
  uint32_t texel = sd.tex[(size_t)y * sd.tw + x];
  o0 = m.div255((float)(texel & 0xffu)), o1 = m.div255((float)((texel >> 8) & 0xffu)),
  o2 = m.div255((float)((texel >> 16) & 0xffu));
}

// Shader::BlinnPhong scalar (src/Shader.cpp:510-543); the two std::pow(x,2) and the sqrt are binary64 there
template <class M, int NL>
__device__ __forceinline__ void s_blinn_phong(M &m, const FrameK &K, float px, float py, float pz, float nx, float ny, float nz,
                                              float kd0, float kd1, float kd2, const SRZ_CAS srz_light *L, float &o0, float &o1,
                                              float &o2) {
  const float Lx = L->pos[0], Ly = L->pos[1], Lz = L->pos[2], I0 = L->intensity[0], I1 = L->intensity[1], I2 = L->intensity[2];
  normalize3(m, nx, ny, nz);
  float ldx = Lx - px, ldy = Ly - py, ldz = Lz - pz;
  float dsq = m.len2d(ldx, ldy); // (binary64 in the reference: std::pow(x, 2), std::sqrt)
  float d0, d1, d2;
  m.div3(I0, I1, I2, dsq, d0, d1, d2);
  float nlx = ldx, nly = ldy, nlz = ldz;
  normalize3(m, nlx, nly, nlz);
  float cosTheta = std_max(0.0f, dot3(nx, ny, nz, nlx, nly, nlz));
  float vx = K.eye[0] - px, vy = K.eye[1] - py, vz = K.eye[2] - pz;
  float hx = ldx + vx, hy = ldy + vy, hz = ldz + vz;
  normalize3(m, hx, hy, hz);
  float cosAlpha = std_max(0.0f, dot3(nx, ny, nz, hx, hy, hz));
  float pw = pow_frame<NL>(m, cosAlpha, K.p);
  o0 = ((K.ka[0] * I0 + (cosTheta * kd0) * d0) + (pw * K.ks[0]) * d0) * kd0;
  o1 = ((K.ka[1] * I1 + (cosTheta * kd1) * d1) + (pw * K.ks[1]) * d1) * kd1;
  o2 = ((K.ka[2] * I2 + (cosTheta * kd2) * d2) + (pw * K.ks[2]) * d2) * kd2;
}
// the same light in two steps (see v_blinn_phong_terms): cosA = cosTheta, cosT = pow(cosAlpha, p)
template <class M, int NL>
__device__ __forceinline__ void s_blinn_phong_terms(M &m, const FrameK &K, float px, float py, float pz, float nx, float ny, float nz,
                                                    const SRZ_CAS srz_light *L, LightTerms &t) {
  const float Lx = L->pos[0], Ly = L->pos[1], Lz = L->pos[2], I0 = L->intensity[0], I1 = L->intensity[1], I2 = L->intensity[2];
  normalize3(m, nx, ny, nz);
  float ldx = Lx - px, ldy = Ly - py, ldz = Lz - pz;
  float dsq = m.len2d(ldx, ldy); // (binary64 in the reference: std::pow(x, 2), std::sqrt)
  m.div3(I0, I1, I2, dsq, t.d0, t.d1, t.d2); // (no branch on FrameK::grey here: the terms of a pixel stay one straight line of code)
  float nlx = ldx, nly = ldy, nlz = ldz;
  normalize3(m, nlx, nly, nlz);
  t.cosA = std_max(0.0f, dot3(nx, ny, nz, nlx, nly, nlz));
  float vx = K.eye[0] - px, vy = K.eye[1] - py, vz = K.eye[2] - pz;
  float hx = ldx + vx, hy = ldy + vy, hz = ldz + vz;
  normalize3(m, hx, hy, hz);
  float cosAlpha = std_max(0.0f, dot3(nx, ny, nz, hx, hy, hz));
  t.cosT = pow_frame<NL>(m, cosAlpha, K.p);
}
__device__ __forceinline__ void s_blinn_phong_combine(const FrameK &K, const SRZ_CAS srz_light *L, const LightTerms &t, float kd0,
                                                      float kd1, float kd2, float &o0, float &o1, float &o2) {
  const float I0 = L->intensity[0], I1 = L->intensity[1], I2 = L->intensity[2];
  o0 = ((K.ka[0] * I0 + (t.cosA * kd0) * t.d0) + (t.cosT * K.ks[0]) * t.d0) * kd0;
  o1 = ((K.ka[1] * I1 + (t.cosA * kd1) * t.d1) + (t.cosT * K.ks[1]) * t.d1) * kd1;
  o2 = ((K.ka[2] * I2 + (t.cosA * kd2) * t.d2) + (t.cosT * K.ks[2]) * t.d2) * kd2;
}
// s_texel in two steps for the FAST build: the load from an address that is always inside the image (the reference's
// out-of-range case — u or v == 1.0 → index == size → black — becomes a select afterwards), so it is not inside a branch
// and can stay in flight across the lights' terms
__device__ __forceinline__ uint32_t s_texel_issue(const ShadeDesc &sd, float u, float v, bool &inside) {
  float cu = std_clamp(u, 0.0f, 1.0f), cv = std_clamp(v, 0.0f, 1.0f);
  float fx = cu * (float)sd.tw, fy = cv * (float)sd.th;
  int x = (int)fx, y = (int)fy;
  inside = !(x < 0 || x >= sd.tw || y < 0 || y >= sd.th);
  const int xs = min(max(x, 0), sd.tw - 1), ys = min(max(y, 0), sd.th - 1);
  return sd.tex[(size_t)ys * sd.tw + xs];
}

// calcBumpMapping / calcDisplacementMapping common part (src/Shader.cpp:447-507)
template <class M>
__device__ __forceinline__ void s_bump_common(M &m, const ShadeDesc &sd, float nx, float ny, float nz, float u, float v, float kh,
                                              float kn, float &ox, float &oy, float &oz, float &origin_norm) {
  float sq = __builtin_sqrtf(nx * nx + nz * nz);
  float t0 = (nx * ny) / sq, t1 = sq, t2 = (nz * ny) / sq;
  float b0 = ny * t2 - t1 * nz, b1 = nz * t0 - t2 * nx, b2 = nx * t1 - t0 * ny;
  float a0, a1, a2, u0, u1, u2, w0, w1, w2;
  s_texel(m, sd, u, v, a0, a1, a2);
  float on = __builtin_sqrtf(dot3(a0, a1, a2, a0, a1, a2));
  s_texel(m, sd, (u + 1.0f) / (float)sd.tw, v, u0, u1, u2);
  s_texel(m, sd, u, (v + 1.0f) / (float)sd.th, w0, w1, w2);
  float dU = kh * kn * (__builtin_sqrtf(dot3(u0, u1, u2, u0, u1, u2)) - on);
  float dV = kh * kn * (__builtin_sqrtf(dot3(w0, w1, w2, w0, w1, w2)) - on);
  float l0 = -dU, l1 = -dV, l2 = 1.0f;
  ox = t0 * l0 + t1 * l1 + t2 * l2, oy = b0 * l0 + b1 * l1 + b2 * l2, oz = nx * l0 + ny * l1 + nz * l2;
  normalize3(m, ox, oy, oz);
  origin_norm = on;
}

// scalar applyFragmentShader + standard_*_impl + Tools::normalizedToRGB (src/Shader.cpp:547-640, src/Tools.cpp:94-104)
template <class M, int SH, int NL>
__device__ __forceinline__ void s_shade(M &m, const FrameK &K, const ShadeDesc &sd, float px, float py, float pz, float nx, float ny,
                                        float nz, float u, float v, float &r0, float &r1, float &r2) {
  const int shader = SH >= 0 ? SH : sd.shader;
  constexpr int NLC = light_count<NL>();
  const uint32_t n_lights = NLC > 0 ? (uint32_t)NLC : K.n_lights;
  float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f;
  if (shader == SRZ_SHADER_NORMAL) {
    normalize3(m, nx, ny, nz);
    c0 = (nx + 1.0f) / 2.0f, c1 = (ny + 1.0f) / 2.0f, c2 = (nz + 1.0f) / 2.0f;
  } else if (shader >= SRZ_SHADER_TEXTURE && shader <= SRZ_SHADER_BUMP) {
    float kd0 = 1.0f, kd1 = 1.0f, kd2 = 1.0f;
    float sx = px, sy = py, sz = pz, snx = nx, sny = ny, snz = nz;
    if constexpr (NLC > 0 && (SH == SRZ_SHADER_TEXTURE || SH == SRZ_SHADER_PHONG)) {
      // FAST build: texel load issued first, every light's colour-independent terms, then the texel and the combination
      bool inside = true;
      uint32_t texel = 0u;
      if (SH == SRZ_SHADER_TEXTURE) texel = s_texel_issue(sd, u, v, inside);
      LightTerms t[NLC];
#pragma unroll
      for (int l = 0; l < NLC; ++l) s_blinn_phong_terms<M, NL>(m, K, sx, sy, sz, snx, sny, snz, K.lights + l, t[l]);
      asm volatile("" : "+v"(t[0].cosT), "+v"(t[NLC - 1].cosT), "+v"(texel));
      if (SH == SRZ_SHADER_TEXTURE) {
        kd0 = m.div255((float)(texel & 0xffu)), kd1 = m.div255((float)((texel >> 8) & 0xffu)), kd2 = m.div255((float)((texel >> 16) & 0xffu));
        if (!inside) kd0 = kd1 = kd2 = 0.0f;
      }
      if (SH == SRZ_SHADER_PHONG && K.grey) { // (wave-uniform) kd = 1 and grey constants: one channel, copied (see FrameK::grey)
#pragma unroll
        for (int l = 0; l < NLC; ++l) {
          const float I0 = (K.lights + l)->intensity[0];
          c0 = c0 + ((K.ka[0] * I0 + (t[l].cosA * 1.0f) * t[l].d0) + (t[l].cosT * K.ks[0]) * t[l].d0) * 1.0f;
        }
        const float q = std_clamp(c0, 0.0f, 1.0f) * 255.0f;
        r0 = r1 = r2 = (q == q) ? (float)(uint32_t)q : 0.0f;
        return;
      }
#pragma unroll
      for (int l = 0; l < NLC; ++l) {
        float o0, o1, o2;
        s_blinn_phong_combine(K, K.lights + l, t[l], kd0, kd1, kd2, o0, o1, o2);
        c0 = c0 + o0, c1 = c1 + o1, c2 = c2 + o2;
      }
    } else {
    if (shader != SRZ_SHADER_PHONG) s_texel(m, sd, u, v, kd0, kd1, kd2);
    if (shader == SRZ_SHADER_BUMP) {
      float on;
      s_bump_common(m, sd, nx, ny, nz, u, v, K.kh, K.kn, snx, sny, snz, on);
    } else if (shader == SRZ_SHADER_DISPLACEMENT) {
      float on;
      s_bump_common(m, sd, nx, ny, nz, u, v, K.kh, K.kn, snx, sny, snz, on);
      sx = px + (K.kn * nx) * on, sy = py + (K.kn * ny) * on, sz = pz + (K.kn * nz) * on;
    }
    for (uint32_t l = 0; l < n_lights; ++l) {
      float o0, o1, o2;
      s_blinn_phong<M, NL>(m, K, sx, sy, sz, snx, sny, snz, kd0, kd1, kd2, K.lights + l, o0, o1, o2);
      c0 = c0 + o0, c1 = c1 + o1, c2 = c2 + o2;
    }
    }
  }
  float q0 = std_clamp(c0, 0.0f, 1.0f) * 255.0f, q1 = std_clamp(c1, 0.0f, 1.0f) * 255.0f,
        q2 = std_clamp(c2, 0.0f, 1.0f) * 255.0f;
  r0 = (q0 == q0) ? (float)(uint32_t)q0 : 0.0f;
  r1 = (q1 == q1) ? (float)(uint32_t)q1 : 0.0f;
  r2 = (q2 == q2) ? (float)(uint32_t)q2 : 0.0f;
}

// A tile that has an owner goes to one of 104 work lists for k_shade: [build that shades the frame: FAST for 1 / 2 / 3 / 4 lights,
// the same for frames with BUMP / DISPLACEMENT batches, the same for frames with a non-integer exponent, generic][frame % 8]
// (frame % 8 = the XCD that rasterised it), in arrival order — k_raster's workgroups run in frame order, so every list comes
// out (roughly) frame by frame.  An entry = {frame, local band | tile x << 10 | flags (WORK_LP), entries of the tile's triangle
// list, its first index in pool[]}: everything k_shade needs to start the tile's loads travels with it, and nothing has to be
// divided out of a flat tile number (two scalar software divisions per tile otherwise).
constexpr uint32_t WORK_LP = 1u << 20; // the tile's owner ids are POSITIONS in the tile's triangle list, 16 bits each (see LP_BITS)
constexpr uint32_t WORK_LP8 = 1u << 21; // (with WORK_LP, round 6) ... 8 bits each: the list has at most LP8_MAX entries (see id_pack8)
__device__ __forceinline__ void work_append(const RenderArgs &a, uint32_t frame_flags, uint32_t frame, uint32_t entry /* frame * tiles + tile */,
                                            uint32_t lb, uint32_t tx, uint32_t wflags, uint32_t list_cnt, uint32_t list_off) {
  // (fewer than 8 frames: the tiles are dealt over the 8 lists instead, so that every XCD has work)
  const uint32_t kind = (!a.force_generic && (frame_flags & FD_FAST_SHADE) != 0u)
                            ? ((frame_flags >> FD_NL_SHIFT) & 7u) - 1u + ((frame_flags & FD_BUMPY) ? 4u : 0u) + ((frame_flags & FD_GENPOW) ? 8u : 0u)
                            : SHADE_KIND_GENERIC;
  const uint32_t sub = (a.n_frames >= 8u ? frame : frame + entry) & 7u, L = kind * 8u + sub;
  const uint32_t Ls = ((uint32_t)(a.kind_slots >> (4u * kind)) & 15u) * 8u + sub; // (the kind's slot in the lists' storage)
  a.worklist[(size_t)Ls * a.work_cap + atomicAdd(&a.work_count[L * CNT_STRIDE], 1u)] = make_uint4(frame, lb | (tx << 10) | wflags, list_cnt, list_off);
}

// bit 31 of an owner id: the pixel lies in the scalar-tail ("S") columns of its owner's bounding box
constexpr uint32_t S_CLASS_BIT = 0x80000000u;

// What the rasterisers hand to k_shade per OWNED tile: the owner of every pixel, in the tile's slot of `vis` (PIX_SLOT dwords,
// pixel p = ly * 32 + lx).  The owner is named
//   by its POSITION in the tile's triangle list, 16 bits per pixel (position | 0x8000 for the S class, 0xffff = nobody) — k_raster,
//      tiles of at most LP_MAX list entries in frames of fewer than 2^22 - 1 triangles: the position rides in the low bits of the
//      depth key's tie-break, below the triangle index, so the pixel's final key holds it for free (work entry flag WORK_LP).
//      k_shade then stages the list's triangles in LDS once per tile and every pixel reads its owner's 96 bytes from there
//      instead of gathering them from memory; and a frame of 94 k triangles pays 2 bytes per pixel like one of 5 k;
//   or by its index in the frame, 32 bits per pixel (index | S_CLASS_BIT, NO_TRI = nobody) — the ordered rasteriser, long lists.
constexpr uint32_t PIX_SLOT = TILE * TILE, PIX_BITS = 10, PIX_MASK = PIX_SLOT - 1u;
constexpr uint32_t LP_BITS = 9, LP_MAX = 1u << LP_BITS; // (idx << 9 | position) <= 0x7ffffffe for idx < 2^22 - 1 (FD_PACKED)
static_assert(31 - LP_BITS == PACK_IDX_BITS, "the tie-break holds an index of PACK_IDX_BITS bits above the list position");
// ... or, for lists of at most LP8_MAX entries (most tiles of every BASELINE config), 8 bits per pixel (position | 0x80 for the S class,
// 0xff = nobody): the ids are written by k_raster and read back by k_shade, 1 + 1 instead of 2 + 2 bytes per pixel of every owned tile —
// during the part of a step in which the memory system is what everything waits for (NOTEBOOK r6 §2)
constexpr uint32_t LP8_MAX = 127;
__device__ __forceinline__ uint32_t id_pack8(uint32_t id) { return id == NO_TRI ? 0xffu : ((id & 127u) | ((id >> 24) & 0x80u)); }
__device__ __forceinline__ uint32_t id_pack16(uint32_t id) { return id == NO_TRI ? 0xffffu : ((id & (LP_MAX - 1u)) | ((id >> 16) & 0x8000u)); }
__device__ __forceinline__ uint32_t id_unpack16(uint32_t h) { return h == 0xffffu ? NO_TRI : ((h & (LP_MAX - 1u)) | ((h & 0x8000u) << 16)); }
// the four owners of pixels p0 .. p0 + 3 of a tile (p0 % 4 == 0) into / out of its slot
__device__ __forceinline__ void ids_store4(uint32_t *slot, uint32_t p0, const uint4 &id, bool by_lp, bool lp8 = false) {
  if (lp8) {
    *reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(slot) + p0) = id_pack8(id.x) | (id_pack8(id.y) << 8) | (id_pack8(id.z) << 16) | (id_pack8(id.w) << 24);
  } else if (by_lp) {
    const u32x2 w = {id_pack16(id.x) | (id_pack16(id.y) << 16), id_pack16(id.z) | (id_pack16(id.w) << 16)};
    *reinterpret_cast<u32x2 *>(reinterpret_cast<uint16_t *>(slot) + p0) = w;
  } else {
    *reinterpret_cast<uint4 *>(slot + p0) = id;
  }
}
__device__ __forceinline__ uint4 ids_load4(const uint32_t *slot, uint32_t p0, bool by_lp) {
  if (by_lp) {
    const u32x2 w = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const uint16_t *>(slot) + p0);
    return make_uint4(id_unpack16(w.x & 0xffffu), id_unpack16(w.x >> 16), id_unpack16(w.y & 0xffffu), id_unpack16(w.y >> 16));
  }
  return *reinterpret_cast<const uint4 *>(slot + p0);
}

// Everything the shader needs about the owner triangle of one pixel, fetched in ONE round trip (7 independent loads)
struct TriFetch {
  f32x4 q0, q1, q2, q3, q4, q5;
  uint32_t batch;
};
__device__ __forceinline__ void fetch_tri(const SRZ_CAS srz_tri *tris, const SRZ_CAS uint16_t *tri_batch, uint32_t idx,
                                          TriFetch &f) {
  const SRZ_CAS f32x4 *tp = reinterpret_cast<const SRZ_CAS f32x4 *>(tris + idx);
  f.q0 = tp[0], f.q1 = tp[1], f.q2 = tp[2], f.q3 = tp[3], f.q4 = tp[4], f.q5 = tp[5];
  f.batch = tri_batch[idx];
}

struct TriAttr {
  TriXY t;
  float n0x, n0y, n0z, n1x, n1y, n1z, n2x, n2y, n2z, u0, v0, u1, v1, u2, v2;
};
template <class M> __device__ __forceinline__ void unpack_pos(M &m, const TriFetch &f, TriAttr &a) {
  // pos: q0.xyz q0.w q1.xy q1.zw q2.x
  a.t.ax = f.q0.x, a.t.ay = f.q0.y, a.t.z0 = f.q0.z, a.t.bx = f.q0.w, a.t.by = f.q1.x, a.t.z1 = f.q1.y, a.t.cx = f.q1.z,
  a.t.cy = f.q1.w, a.t.z2 = f.q2.x;
  tri_consts(m, a.t);
}
__device__ __forceinline__ void unpack_attr(const TriFetch &f, TriAttr &a) {
  // nrm: q2.yzw q3.xyz q3.w q4.xy | uv: q4.zw q5.xy q5.zw
  a.n0x = f.q2.y, a.n0y = f.q2.z, a.n0z = f.q2.w, a.n1x = f.q3.x, a.n1y = f.q3.y, a.n1z = f.q3.z, a.n2x = f.q3.w, a.n2y = f.q4.x,
  a.n2z = f.q4.y;
  a.u0 = f.q4.z, a.v0 = f.q4.w, a.u1 = f.q5.x, a.v1 = f.q5.y, a.u2 = f.q5.z, a.v2 = f.q5.w;
}
// A triangle staged in LDS (k_shade, `late` != null) hands over its positions first and its normals / texture coordinates only
// when the barycentrics exist: LDS is close, and 24 values fetched at once are 16 registers held for nothing (the dependency
// on `dep` pins the second half of the reads behind the coverage arithmetic).  From memory (`late` == null) all 96 bytes come
// in ONE round trip, up front.
__device__ __forceinline__ void early_fetch(TriFetch &f, const f32x4 *late) {
  if (late != nullptr) f.q0 = late[0], f.q1 = late[1], f.q2 = late[2];
}
__device__ __forceinline__ void late_fetch(TriFetch &f, const f32x4 *late, float dep) {
  if (late != nullptr) {
    asm volatile("" : "+v"(late) : "v"(dep));
    f.q3 = late[3], f.q4 = late[4], f.q5 = late[5];
  }
}
// Shade pixel (x,y) of depth z, owner `f`, 8-wide ("V") semantics (src/Rasterizer.cpp:380-389)
template <class M, int SH = -1, int NL = 0>
__device__ __forceinline__ void shade_pixel_v(M &m, const FrameK &K, const ShadeDesc &sd, const TriFetch &f_in, const f32x4 *late, int x,
                                              int y, float &r0, float &r1, float &r2) {
  TriAttr a;
  TriFetch f; // (a local copy: what the staged reads below define must not outlive this call — the caller's loops would carry it)
  if constexpr (std::is_same<M, ApproxMath>::value) { // (a staged triangle: nothing of f_in but the batch id is defined — copying the
    if (late == nullptr) f = f_in;                    // struct left a dead 4-byte scratch store per chunk in the tolerance builds)
  } else {
    f = f_in;
  }
  early_fetch(f, late);
  unpack_pos(m, f, a);
  const float fx = (float)x, fy = (float)y;
  float alpha, beta, gamma, zz;
  cover_v(a.t, fx, fy, alpha, beta, gamma, zz);
  late_fetch(f, late, gamma);
  unpack_attr(f, a);
  float nx = fmaf_(alpha, a.n0x, fmaf_(beta, a.n1x, gamma * a.n2x));
  float ny = fmaf_(alpha, a.n0y, fmaf_(beta, a.n1y, gamma * a.n2y));
  float nz = fmaf_(alpha, a.n0z, fmaf_(beta, a.n1z, gamma * a.n2z));
  v_normalized(m, nx, ny, nz);
  float u = fmaf_(alpha, a.u0, fmaf_(beta, a.u1, gamma * a.u2));
  float v = fmaf_(alpha, a.v0, fmaf_(beta, a.v1, gamma * a.v2));
  v_shade<M, SH, NL>(m, K, sd, fx, fy, zz, nx, ny, nz, u, v, r0, r1, r2); // zz: the depth k_raster stored (same operations)
}
// scalar-tail ("S") semantics (src/Rasterizer.cpp:470-492)
template <class M, int SH = -1, int NL = 0>
__device__ __forceinline__ void shade_pixel_s(M &m, const FrameK &K, const ShadeDesc &sd, const TriFetch &f_in, const f32x4 *late, int x,
                                              int y, float &r0, float &r1, float &r2) {
  TriAttr a;
  TriFetch f; // (a local copy: what the staged reads below define must not outlive this call — the caller's loops would carry it)
  if constexpr (std::is_same<M, ApproxMath>::value) { // (a staged triangle: nothing of f_in but the batch id is defined — copying the
    if (late == nullptr) f = f_in;                    // struct left a dead 4-byte scratch store per chunk in the tolerance builds)
  } else {
    f = f_in;
  }
  early_fetch(f, late);
  unpack_pos(m, f, a);
  const float fx = (float)x, fy = (float)y;
  float alpha, beta, gamma, zz;
  cover_s(a.t, fx, fy, alpha, beta, gamma, zz);
  late_fetch(f, late, gamma);
  unpack_attr(f, a);
  float nx = alpha * a.n0x + beta * a.n1x + gamma * a.n2x;
  float ny = alpha * a.n0y + beta * a.n1y + gamma * a.n2y;
  float nz = alpha * a.n0z + beta * a.n1z + gamma * a.n2z;
  normalize3(m, nx, ny, nz);
  float u = alpha * a.u0 + beta * a.u1 + gamma * a.u2;
  float v = alpha * a.v0 + beta * a.v1 + gamma * a.v2;
  s_shade<M, SH, NL>(m, K, sd, fx, fy, zz, nx, ny, nz, u, v, r0, r1, r2);
}

#ifdef SRZ_ISA_PROBE
// instruction-count probes (make asm-probe): the two per-pixel shading paths in isolation
__global__ void probe_v(RenderArgs a, float *o) {
  const SRZ_CAS FrameDesc *fd = as_const(a.frames);
  FrameK K;
  K.eye[0] = fd->eye[0], K.eye[1] = fd->eye[1], K.eye[2] = fd->eye[2];
  K.ka[0] = fd->ka[0], K.ka[1] = fd->ka[1], K.ka[2] = fd->ka[2];
  K.ks[0] = fd->ks[0], K.ks[1] = fd->ks[1], K.ks[2] = fd->ks[2];
  K.p = 150.0f, K.kh = fd->kh, K.kn = fd->kn, K.n_lights = 2; // constants: the static count is the dynamic path
#ifdef SRZ_PROBE_GREY
  K.grey = true;
#else
  K.grey = false;
#endif
  K.lights = as_const(a.lights);
  TriFetch tf;
  fetch_tri(as_const(a.tris), as_const(a.tri_batch), threadIdx.x, tf);
  ShadeDesc sd;
  sd.shader = SRZ_SHADER_TEXTURE, sd.tw = 1024, sd.th = 1024, sd.tex = as_const((const uint32_t *)a.vis);
  float r0, r1, r2;
#ifndef SRZ_PROBE_SH
#define SRZ_PROBE_SH -1 /* -1: the generic build's per-pixel generality; 0 / 1 / 2: the FAST variant of that shader type */
#endif
#ifdef SRZ_PROBE_APPROX /* the tolerance mode's policy */
  ApproxMath fm;
#define SRZ_PROBE_M ApproxMath
#else
  FastMath fm;
#define SRZ_PROBE_M FastMath
#endif
#ifdef SRZ_PROBE_S
  shade_pixel_s<SRZ_PROBE_M, SRZ_PROBE_SH, (SRZ_PROBE_SH >= 0 ? 2 : 0)>(fm, K, sd, tf, nullptr, threadIdx.x, blockIdx.x, r0, r1, r2);
#else
  shade_pixel_v<SRZ_PROBE_M, SRZ_PROBE_SH, (SRZ_PROBE_SH >= 0 ? 2 : 0)>(fm, K, sd, tf, nullptr, threadIdx.x, blockIdx.x, r0, r1, r2);
#endif
#ifndef SRZ_PROBE_APPROX
  if (fm.is_bad()) r0 = -1.f;
#endif
  o[threadIdx.x] = r0 + r1 + r2;
}
#endif

// ================================================================================================================
// k_raster — VISIBILITY: one wave per 32x32 tile, ORDER-INDEPENDENT
//
// The reference walks the triangles in submission order; a fragment of the 8-wide ("V") columns replaces the pixel iff
// z < zbuf, one of the scalar-tail ("S") columns iff !(z > zbuf) (src/Rasterizer.cpp:334,475).  For ordinary depths
// (not NaN, not ±0) the pixel's final state is a function of the SET of its fragments:
//     final z = the smallest z;   final owner = the LAST S fragment at that z if there is one (every later S passes <=,
//     no later V passes <), otherwise the FIRST V fragment at that z (later ones fail <); an incoming depth equal to the
//     smallest z keeps the pixel against V fragments and loses it to S fragments.
// So every pixel holds ONE 64-bit key  (order-preserving image of z) << 32 | tie-break  and every fragment is an LDS
// ds_min_u64, in any order:   tie-break = 0x7ffffffe - idx (S)  <  0x7fffffff (incoming depth)  <  0x80000000 | idx (V),
// idx = the triangle's index in the frame = its submission order.
// What the min cannot express is detected and handed to k_raster_slow (the reference's ordered algorithm): a NaN depth
// that passes an S test (S fragments with NaN z and NaN incoming depths get the smallest key, 0) and a final depth of
// ±0 (-0 and +0 compare equal as floats but not as keys); both leave their mark in the pixel's final key.
//
// Because order is irrelevant the tile's (triangle, pixel) candidates can be spread DENSELY over the lanes instead of
// giving every triangle a wave pass of its own.  Per chunk of 64 list records (one per lane, geometry of bbox ∩ tile
// computed in parallel) the work is flattened into ITEMS — a V item is an 8-pixel piece of one bbox row, an S item is one
// pixel of the <= 7 scalar-tail columns — numbered by an exclusive scan of the per-triangle item counts.  A batch of 64
// consecutive items is then one item per lane: the triangles whose first item falls into the batch mark that position in
// LDS, an inclusive max-scan over the marks tells every lane its triangle, whose record comes over the LDS crossbar
// (ds_bpermute), and the lane tests its 8 pixels (V) or its pixel (S) with the reference's arithmetic.
// ================================================================================================================
__device__ __forceinline__ float rl_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
// streaming 16-byte store: written once, never re-read by this pipeline
// (measured on MI355X, 256 frames of 1024^2, ms per step with this store as nt / plain / sc1 / sc0 sc1 / sc1 nt in k_clear:
// 1.00 / 1.24 / 1.50 / 1.53 / 1.16; nt in k_clear and sc1 in the tile write-outs 1.15: nt everywhere)
__device__ __forceinline__ void store_nt(float *p, const float4 &v) {
  f32x4 w = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(w, reinterpret_cast<f32x4 *>(p));
}
// order-preserving image of binary32 in u32 (-0 just below +0, negative NaNs below -inf, positive NaNs above +inf)
__device__ __forceinline__ uint32_t zkey_of(float z) {
  const uint32_t b = f2u(z);
  return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float z_of_key(uint32_t k) {
  return u2f(k ^ (~(uint32_t)((int32_t)k >> 31) | 0x80000000u));
}
constexpr uint32_t TB_NONE = 0x7fffffffu; // tie-break of the incoming depth
__device__ __forceinline__ float bperm_f(int addr, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ uint32_t bperm_u(int addr, uint32_t v) { return (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v); }

// WAVES = 1: the throughput build (batches: thousands of tiles in flight, one wave each).  WAVES = 4: the latency build for
// jobs of a few frames, where a frame is as slow as its heaviest tile — the four waves of a tile share its keys (the LDS
// minimum is atomic across waves), each loads the chunk's records and takes every fourth batch of 64 items and a quarter of
// the init / decode / write-out rows.
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, WAVES == 1 ? 5 : 1) void k_raster(RenderArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned long long s_key[TILE * KEY_STRIDE];
  __shared__ uint32_t s_mark_x[WAVES > 1 ? WAVES * 64 : 1]; // WAVES > 1: the item marks of each wave (one wave: the keys' pad column)
  __shared__ uint32_t s_wg[WAVES > 1 ? 3 : 1];              // WAVES > 1: redo / owner flags of the tile, the farthest depth
  constexpr int ITS = 4 / WAVES;                            // 8-row strips of the tile per wave in the row-wise phases

  const int lane = threadIdx.x & 63;
  const int wave = WAVES > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
  auto wg_barrier = [] {
    if constexpr (WAVES > 1)
      __syncthreads();
    else
      __builtin_amdgcn_wave_barrier();
  };
  // XCD-aware tile assignment.  Workgroups are dealt round-robin to the 8 XCDs in launch order, and launch order is
  // strict: one XCD whose slots are full of long tiles stalls the dispatch of everything behind it.  So workgroup i
  // renders frame (i % 8) of its group of 8 frames: the 8 XCDs walk the SAME tile sequence in lockstep (balanced by
  // construction), and all tiles of one frame — hence its tile lists — live in ONE XCD's L2.
  const uint32_t wg = blockIdx.x;
  // what k_bin asked of the record pool (final: k_bin has finished) goes to the host through mapped pinned memory; the next
  // render of this set reads it before it launches and grows the pool if a band did not fit.  Only in the latency build
  // (batches copy it on the clear's side stream): with this store compiled into the throughput build — executed by one
  // workgroup, or by none — k_raster took 340 instead of 320 µs for 256 frames
  if constexpr (WAVES > 1)
    if (wg == 0 && wave == 0 && (uint32_t)lane <= a.pool_sub_mask) a.pool_demand[lane * CNT_STRIDE] = a.pool_heads[lane * CNT_STRIDE];
  const uint32_t xcd = wg & 7u, j = wg >> 3;
  const uint32_t tiles_per_frame = a.n_local_bands * a.tiles_x;
  const uint32_t frame = (j / tiles_per_frame) * 8u + xcd, tile = j % tiles_per_frame;
  if (frame >= a.n_frames) return;
  // nothing listed for this tile: nothing to rasterise; its clear (if any) is k_clear's job — or, in a small job that has
  // no k_clear beside it (one kernel boundary less on the latency path), this wave's
  const u32x2 tinfo = as_const(reinterpret_cast<const u32x2 *>(a.tile_info))[(size_t)frame * tiles_per_frame + tile];
  const uint32_t cnt = tinfo.x;
  if (cnt == 0u && !a.clear_in_raster) return; // (three tiles in four of a batch: out after ONE load)
  const SRZ_CAS FrameDesc *fd = as_const(a.frames) + frame;
  const uint32_t flags = fd->flags | a.flags_or;
  if (cnt == 0u) {
    if (flags & SRZ_FUSED_CLEAR) {
      const int W = fd->width, H = fd->height;
      const uint32_t lb = tile / a.tiles_x;
      const int tx0 = (int)(tile % a.tiles_x) * TILE, ty0 = band_of((int)lb, a.shard_rank, a.shard_world) * BAND;
      const int tx1 = min(tx0 + TILE, W) - 1, ty1 = min(ty0 + BAND, H) - 1;
      const size_t plane = (size_t)a.local_rows * (size_t)W;
      float *out0 = a.out + (size_t)frame * a.frame_stride + (size_t)lb * BAND * (size_t)W;
      const float inf = __builtin_inff();
      const float4 inf4 = make_float4(inf, inf, inf, inf), zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int k = 0; k < ITS; ++k) {
        const int it = wave * ITS + k;
        const int ly = it * 8 + (lane >> 3), x4 = tx0 + (lane & 7) * 4;
        if (ty0 + ly > ty1 || x4 > tx1) continue;
        float *gz = out0 + (size_t)ly * W + x4;
        if (((W & 3) == 0) && x4 + 3 <= tx1) {
          store_nt(gz, inf4), store_nt(gz + plane, zero4), store_nt(gz + 2 * plane, zero4), store_nt(gz + 3 * plane, zero4);
        } else {
          for (int k = 0; k < 4 && x4 + k <= tx1; ++k) gz[k] = inf, gz[plane + k] = 0.f, gz[2 * plane + k] = 0.f, gz[3 * plane + k] = 0.f;
        }
      }
    }
    return;
  }
  const uint32_t off = tinfo.y;
  // the tie-break's payload: the triangle's index, or index << LP_BITS | its position in this tile's list (same order)
  const bool lp_mode = cnt <= LP_MAX && (fd->flags & FD_PACKED) != 0u; // wave-uniform
  const bool lp8 = lp_mode && cnt <= LP8_MAX;                          // (the ids then take 8 bits per pixel)
  const uint32_t idx_mask = (fd->flags & FD_PACKED) ? PACK_IDX_MASK : 0xffffffffu; // (a packed list entry = index | batch << 22)
  if (off == UNLISTED || a.force_ordered || (flags & SRZ_ORDERED_RASTER)) { // the reference's ordered algorithm, from the stream
    if (lane == 0 && wave == 0) a.slow_list[atomicAdd(a.slow_count, 1u)] = frame * tiles_per_frame + tile;
    return;
  }
  const uint32_t lb = tile / a.tiles_x;
  const int W = fd->width, H = fd->height;
  const int tx0 = (int)(tile % a.tiles_x) * TILE;
  const int band = band_of((int)lb, a.shard_rank, a.shard_world);
  const int ty0 = band * BAND;
  const int tx1 = min(tx0 + TILE, W) - 1, ty1 = min(ty0 + BAND, H) - 1;
  const bool fused = (flags & SRZ_FUSED_CLEAR) != 0;
  const size_t row0 = (size_t)lb * BAND;
  float *out0 = a.out + (size_t)frame * a.frame_stride + row0 * (size_t)W; // plane 0 (z), row ty0

  // (the first 64 indices of the tile's list are loaded under the tile init)
  const uint32_t i_first = as_const(a.pool)[off + min((uint32_t)lane, cnt - 1u)] & idx_mask;
  // ---- phase A: tile init (fused clear → +inf, else the in/out z plane) with the incoming-depth tie-break ----------
  if (fused) { // (wave-uniform) straight stores of one constant: nothing here waits for the index load above
    const unsigned long long k_inf = ((unsigned long long)zkey_of(__builtin_inff()) << 32) | TB_NONE;
    unsigned long long *kp = &s_key[((int)threadIdx.x >> 5) * KEY_STRIDE + ((int)threadIdx.x & 31)];
#pragma unroll
    for (int it = 0; it < TILE * TILE / (64 * WAVES); ++it) kp[it * (2 * WAVES * KEY_STRIDE)] = k_inf; // (one address, 16 offsets)
  } else {
    for (int i = (int)threadIdx.x; i < TILE * TILE; i += 64 * WAVES) {
      const int ly = i >> 5, lx = i & 31;
      float z = __builtin_inff();
      if (tx0 + lx <= tx1 && ty0 + ly <= ty1) z = out0[(size_t)ly * W + tx0 + lx];
      const uint32_t zk = (z == z) ? zkey_of(z) : 0u; // a NaN already in the buffer: only the ordered algorithm knows
      s_key[ly * KEY_STRIDE + lx] = ((unsigned long long)zk << 32) | TB_NONE;
    }
  }
  // the pad key of every row is scratch: 64 dword marks, mark r in row r / 2
  // (plain LDS accesses: the mark a triangle writes may be the one this lane reads, so the compiler keeps their order;
  // `volatile` would turn them into flat, system-scope accesses)
  uint32_t *const s_mark = WAVES > 1 ? s_mark_x + wave * 64 : reinterpret_cast<uint32_t *>(s_key);
  auto mark_at = [](uint32_t r) { return WAVES > 1 ? r : ((r >> 1) * (uint32_t)KEY_STRIDE + TILE) * 2u + (r & 1u); };
  const uint32_t my_mark = mark_at((uint32_t)lane);
  s_mark[my_mark] = 0u;
  if (WAVES > 1 && threadIdx.x < 3) s_wg[threadIdx.x] = 0u;
  wg_barrier();

  // ---- phase B: the tile's list, 64 records at a time --------------------------------------------------------------
  // geometry of one record's bbox ∩ tile in TILE-LOCAL coordinates: V columns [x0, v-1] in nseg pieces of 8, S columns [v, x1]
  // An S item is a COLUMN piece: S_ROWS vertically adjacent pixels of one scalar-tail column.  (Rounds 2-5: one pixel per item —
  // 45 instructions of item expansion per pixel test; a triangle's S columns are at most 7 wide but as tall as its box, so pieces
  // run down the column: the x-dependent halves of the three edge functions are shared by the piece's pixels.)
  constexpr uint32_t S_ROWS = 4;
  struct Geo {
    bool ok;
    uint32_t nV, nS, word; // items of the two kinds; packed x0 | y0 << 5 | v << 10 | nseg << 16 | ws << 19 | (rows - 1) << 22
    float zmin;            // nearest vertex depth
  };
  auto geometry = [&](const f32x4 &r0, const f32x4 &r1, const f32x4 &r2, float s_area, bool valid) {
    // r0 = ax ay z0 bx | r1 = by z1 cx cy | r2 = z2 bbx bby -
    const uint32_t bbx = f2u(r2.y), bby = f2u(r2.z);
    const int bsx = (int16_t)(bbx & 0xffff), bsy = (int16_t)(bbx >> 16), bex = (int16_t)(bby & 0xffff), bey = (int16_t)(bby >> 16);
    int x0 = max(bsx, tx0) - tx0, x1 = min(bex, tx1) - tx0, y0 = max(bsy, ty0) - ty0, y1 = min(bey, ty1) - ty0;
    // the rectangle that is walked: bounding box ∩ tile, tightened to the triangle (see tight_margin): the triangle's x-extent
    // inside the rectangle's rows, then its y-extent inside the remaining columns.  About half of the pixel tests of a large
    // triangle and 30 % of a small one's go away (the integer box starts at trunc(min), a column / row before the first pixel
    // centre that can be inside; a large triangle's rectangle in a tile is mostly outside it).
    {
      const float ax = r0.x, ay = r0.y, bx = r0.w, by = r1.x, cx = r1.z, cy = r1.w;
      float m, mn, mx;
      if (tight_margin(ax, ay, bx, by, cx, cy, s_area, bsx, bsy, bex, bey, m)) {
        int X0 = tx0 + x0, X1 = tx0 + x1, Y0 = ty0 + y0, Y1 = ty0 + y1;
        slab_extent(ax, ay, bx, by, cx, cy, (float)Y0 - m, (float)Y1 + m, mn, mx);
        clip_range(X0, X1, mn, mx, m);
        slab_extent(ay, ax, by, bx, cy, cx, (float)X0 - m, (float)X1 + m, mn, mx);
        clip_range(Y0, Y1, mn, mx, m);
        x0 = X0 - tx0, x1 = X1 - tx0, y0 = Y0 - ty0, y1 = Y1 - ty0;
      }
    }
    const int vend = (flags & SRZ_UNIFIED) ? bex + 1 : bsx + ((bex - bsx + 1) & ~7);
    const int v = min(max(vend - tx0, x0), x1 + 1);
    Geo g;
    g.ok = valid && x0 <= x1 && y0 <= y1;
    const uint32_t h = g.ok ? (uint32_t)(y1 - y0 + 1) : 0u, nseg = (uint32_t)(v - x0 + 7) >> 3, ws = (uint32_t)(x1 + 1 - v);
    g.nV = __umul24(h, nseg), g.nS = __umul24((h + S_ROWS - 1u) / S_ROWS, ws); // <= 128 / <= 56 per triangle
    g.word = (uint32_t)x0 | ((uint32_t)y0 << 5) | ((uint32_t)v << 10) | (nseg << 16) | (ws << 19) | (((h - 1u) & 31u) << 22);
    g.zmin = __builtin_fminf(__builtin_fminf(r0.z, r1.y), r2.x);
    return g;
  };
  const SRZ_CAS uint32_t *list = as_const(a.pool) + off; // the tile's triangle indices
  // a list entry's record: the triangle's 9 position floats from the dense stream + its 8-byte bounding box (four independent loads)
  const SRZ_CAS float *tpos = as_const(a.tri_pos) + (size_t)fd->tri_off * a.pos_stride;
  const SRZ_CAS u32x2 *tbox = as_const(reinterpret_cast<const u32x2 *>(a.bbox + fd->tri_off));
  const uint32_t pos_stride = a.pos_stride;
  auto rec_load = [&](uint32_t i, f32x4 &q0, f32x4 &q1, f32x4 &q2) { // q0 = ax ay z0 bx | q1 = by z1 cx cy | q2 = z2 bbx bby -
    const SRZ_CAS float *q = tpos + (size_t)i * pos_stride;
    const F3 v0 = ld3(q), v1 = ld3(q + 3), v2 = ld3(q + 6);
    const u32x2 bb = tbox[i];
    q0 = f32x4{v0.x, v0.y, v0.z, v1.x}, q1 = f32x4{v1.y, v1.z, v2.x, v2.y}, q2 = f32x4{v2.z, u2f_(bb.x), u2f_(bb.y), 0.0f};
  };
  // ---- depth-ordered groups for chunks with heavy overdraw -----------------------------------------------------------
  // The keys make the order of the triangles irrelevant, so a chunk of records that asks for several times the tile's area in
  // pixel tests is rasterised NEAREST FIRST, in 4 groups of equal depth range of the triangles' nearest vertex (the records
  // stay in registers); after each group the tile's farthest depth is read back, and the items of a triangle whose nearest
  // possible depth lies behind it are dropped: they could not win a pixel.  Conservative bounds (positive depths only):
  //   V items: 0 < alpha, beta, gamma < 1 holds for every fragment, so its depth is >= zmin = min(z0,z1,z2) less a few ulps:
  //            bound zmin * (1 - 2^-20)
  //   S items: the inside test is on the edge functions, not on the barycentrics, which may leave [0,1] by their rounding
  //            error <= ~6 * 2^-24 * d^2 / |area2| (d = extent of the vertices, area2 = twice the area).  Only for triangles
  //            with d^2 <= 2^10 |area2| (error < 4e-4): bound zmin - (zmax - zmin) / 256, then * (1 - 2^-20); slivers are
  //            never dropped
  uint32_t zfar = 0xffffffffu; // image of the tile's farthest depth when last read back (all ones: nothing can be dropped yet;
                               // depths only come down, so an old value stays a valid bound)
  // two-deep pipeline: the indices of chunk c + 2 and the records of chunk c + 1 are in flight while chunk c is rasterised
  f32x4 n0, n1, n2;
  bool nv = (uint32_t)lane < cnt;
  uint32_t i_cur = i_first, i_nxt = 64u < cnt ? (list[min(64u + (uint32_t)lane, cnt - 1u)] & idx_mask) : 0u;
  rec_load(i_cur, n0, n1, n2);
  for (uint32_t base = 0; base < cnt; base += 64) {
    const f32x4 r0 = n0, r1 = n1, r2 = n2;
    const uint32_t my_idx = lp_mode ? ((i_cur << LP_BITS) | (base + (uint32_t)lane)) : i_cur; // (the tie-break's payload)
    // the per-triangle constants of the two coverage tests, once per record (not stored anywhere: 8 bytes more per triangle to write and to gather)
    float rec_v_inv, rec_s_area;
    {
      TriXY k;
      k.ax = r0.x, k.ay = r0.y, k.bx = r0.w, k.by = r1.x, k.cx = r1.z, k.cy = r1.w;
      BranchMath bm;
      tri_consts(bm, k);
      rec_v_inv = k.v_inv, rec_s_area = k.s_area;
    }
    const bool valid = nv;
    nv = base + 64 + lane < cnt;
    if (base + 64 < cnt) {
      i_cur = i_nxt;
      rec_load(i_cur, n0, n1, n2);
      if (base + 128 < cnt) i_nxt = list[min(base + 128u + (uint32_t)lane, cnt - 1u)] & idx_mask;
    }
    const Geo G = geometry(r0, r1, r2, rec_s_area, valid);
    const uint32_t geom = G.word;
    uint32_t group = 0, n_groups = 1, znear = 0, znear_s = 0; // nearest possible depth of a V / S fragment, as keys (0: unknown → kept)
    if ((uint32_t)rl_i((int)wave_scan_add(8u * G.nV + S_ROWS * G.nS), 63) > 4u * TILE * TILE) { // > 4 pixel tests per pixel of the tile
      float lo = G.ok ? G.zmin : __builtin_inff(), hi = G.ok ? G.zmin : -__builtin_inff();
      for (int o = 32; o > 0; o >>= 1) lo = __builtin_fminf(lo, __shfl_xor(lo, o)), hi = __builtin_fmaxf(hi, __shfl_xor(hi, o));
      if (hi > lo && hi < __builtin_inff() && lo > -__builtin_inff()) {
        n_groups = 4;
        const float g = (G.zmin - lo) * (4.0f / (hi - lo)); // (a NaN depth → group 0)
        group = (g > 0.0f && g < 4.0f) ? (uint32_t)g : (g >= 4.0f ? 3u : 0u);
      }
    }
    if (n_groups > 1 || zfar != 0xffffffffu) {
      if (G.zmin > 0.0f) znear = zkey_of(G.zmin * 0.99999905f);
      const float zmax = __builtin_fmaxf(__builtin_fmaxf(r0.z, r1.y), r2.x);
      const float dx = __builtin_fmaxf(__builtin_fmaxf(r0.x, r0.w), r1.z) - __builtin_fminf(__builtin_fminf(r0.x, r0.w), r1.z);
      const float dy = __builtin_fmaxf(__builtin_fmaxf(r0.y, r1.x), r1.w) - __builtin_fminf(__builtin_fminf(r0.y, r1.x), r1.w);
      const float zs = (G.zmin - (zmax - G.zmin) * 0.00390625f) * 0.99999905f;
      if (dx * dx + dy * dy <= 1024.0f * __builtin_fabsf(rec_s_area) && zmax < 1e30f && zs > 0.0f) znear_s = zkey_of(zs);
    }
    for (uint32_t gi = 0; gi < n_groups; ++gi) {
    const bool mine = G.ok && group == gi;
    const uint32_t nV = (mine && znear <= zfar) ? G.nV : 0u, nS = (mine && znear_s <= zfar) ? G.nS : 0u;
    const uint32_t packed = nV | (nS << 16), incl = wave_scan_add(packed);
    const uint32_t offs = incl - packed; // exclusive: first V item | first S item << 16
    const uint32_t tot = (uint32_t)rl_i((int)incl, 63), TV = tot & 0xffffu, TS = tot >> 16;

    // ---- V items: 8 pixels of one row each (barycentric(__m256) + inside mask + z, src/Rasterizer.cpp:89-127,310-334)
    uint32_t carry = 0;
    for (uint32_t P0 = 64u * (uint32_t)wave; P0 < TV; P0 += 64u * WAVES) {
      if constexpr (WAVES > 1) { // (this wave did not run the batch before: the triangle that reaches into this one is the last to start before it)
        const unsigned long long before = __ballot(nV != 0u && (offs & 0xffffu) < P0);
        carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
      }
      const uint32_t r = (offs & 0xffffu) - P0;
      if (nV != 0u && r < 64u) s_mark[mark_at(r)] = (uint32_t)lane + 1u;
      __builtin_amdgcn_wave_barrier();
      uint32_t m = s_mark[my_mark];
      __builtin_amdgcn_wave_barrier();
      s_mark[my_mark] = 0u;
      m = max(wave_scan_max(m), carry);
      carry = (uint32_t)rl_i((int)m, 63);
      const int src = (int)((m - 1u) & 63u) * 4;
      const float ax = bperm_f(src, r0.x), ay = bperm_f(src, r0.y), z0 = bperm_f(src, r0.z), bx = bperm_f(src, r0.w);
      const float by = bperm_f(src, r1.x), z1 = bperm_f(src, r1.y), cx = bperm_f(src, r1.z), cy = bperm_f(src, r1.w);
      const float z2 = bperm_f(src, r2.x), v_inv = bperm_f(src, rec_v_inv);
      const uint32_t g = bperm_u(src, geom), idx = bperm_u(src, my_idx), o = bperm_u(src, offs);
      const uint32_t item = P0 + (uint32_t)lane, i = item - (o & 0xffffu), ns = (g >> 16) & 7u;
      // row = i / ns: (i + 0.5) / ns is at least 1 / 8 away from every integer, far more than the error of v_rcp
      const uint32_t row = (uint32_t)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)ns)), seg = i - __umul24(row, ns);
      const uint32_t yl = ((g >> 5) & 31u) + row, xs = (g & 31u) + 8u * seg;
      const int lim = item < TV ? (int)((g >> 10) & 63u) - (int)xs : 0; // pixels of this piece inside the V columns
      const float fy = (float)(ty0 + (int)yl), fx0 = (float)(tx0 + (int)xs);
      const float PBy = by - fy, PCy = cy - fy, PAy = ay - fy;
      unsigned long long *kp = s_key + ((yl << 5) + yl) + xs; // yl * KEY_STRIDE
      const uint32_t tb = 0x80000000u | idx;
      // all the arithmetic of the 8 independent pixels first, then the predicated LDS atomics
      uint32_t zk[8];
      bool inside[8];
#pragma unroll
      for (int px = 0; px < 8; ++px) {
        const float fx = fx0 + (float)px; // (exact)
        const float PBx = bx - fx, PCx = cx - fx, PAx = ax - fx;
        const float aPBC = fmsubf(PBx, PCy, PCx * PBy), aPCA = fmsubf(PCx, PAy, PAx * PCy);
        const float al = aPBC * v_inv, be = aPCA * v_inv, ga = 1.0f - (al + be);
        const float z = fmaf_(al, z0, fmaf_(be, z1, ga * z2));
        // 0<al<1 & 0<be<1 & 0<ga<1  <=>  al>0 & be>0 & ga>0 & ga<1 (al+be is rounded monotonically, so ga>0 forces
        // al,be <= al+be < 1); every compare is ordered, so NaNs reject exactly like _CMP_*_OQ
        // (a NaN alpha or beta makes gamma NaN, which fails gamma < 1: the three-way minimum may drop NaNs)
        inside[px] = (px < lim) & (__builtin_fminf(__builtin_fminf(al, be), ga) > 0.0f) & (ga < 1.0f);
        // (an inside fragment's depth is a sum of finite terms of bounded weights: ±inf on overflow at worst, never NaN)
        zk[px] = zkey_of(z);
        asm volatile("" : "+v"(zk[px])); // (keeps the depth arithmetic out of the predicated blocks: 8 pixels of straight-line code)
      }
#pragma unroll
      for (int px = 0; px < 8; ++px)
        if (inside[px])
          __hip_atomic_fetch_min(kp + px, ((unsigned long long)zk[px] << 32) | tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // ---- S items: S_ROWS pixels of one column each (insideTriangle + barycentric(scalar) + z, src/Rasterizer.cpp:11-70,465-477)
    carry = 0;
    for (uint32_t P0 = 64u * (uint32_t)wave; P0 < TS; P0 += 64u * WAVES) {
      if constexpr (WAVES > 1) {
        const unsigned long long before = __ballot(nS != 0u && (offs >> 16) < P0);
        carry = before ? 64u - (uint32_t)__builtin_clzll(before) : 0u;
      }
      const uint32_t r = (offs >> 16) - P0;
      if (nS != 0u && r < 64u) s_mark[mark_at(r)] = (uint32_t)lane + 1u;
      __builtin_amdgcn_wave_barrier();
      uint32_t m = s_mark[my_mark];
      __builtin_amdgcn_wave_barrier();
      s_mark[my_mark] = 0u;
      m = max(wave_scan_max(m), carry);
      carry = (uint32_t)rl_i((int)m, 63);
      const int src = (int)((m - 1u) & 63u) * 4;
      const float ax = bperm_f(src, r0.x), ay = bperm_f(src, r0.y), z0 = bperm_f(src, r0.z), bx = bperm_f(src, r0.w);
      const float by = bperm_f(src, r1.x), z1 = bperm_f(src, r1.y), cx = bperm_f(src, r1.z), cy = bperm_f(src, r1.w);
      const float z2 = bperm_f(src, r2.x), s_area = bperm_f(src, rec_s_area);
      const uint32_t g = bperm_u(src, geom), idx = bperm_u(src, my_idx), o = bperm_u(src, offs);
      const uint32_t item = P0 + (uint32_t)lane, i = item - (o >> 16), w = (g >> 19) & 7u;
      const uint32_t piece = (uint32_t)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)w)), col = i - __umul24(piece, w);
      const uint32_t yl0 = ((g >> 5) & 31u) + S_ROWS * piece, xl = ((g >> 10) & 63u) + col;
      const int lim = item < TS ? (int)((g >> 22) & 31u) + 1 - (int)(S_ROWS * piece) : 0; // rows of this piece inside the rectangle
      const float fy0 = (float)(ty0 + (int)yl0), fx = (float)(tx0 + (int)xl);
      // the reference's expressions per pixel; what depends on x alone is the same for the piece's pixels.  P?x = ?x - fx is the exact
      // negative of ?Px = fx - ?x, and a product of two negated factors is the product: aPBC = PBx * PCy - PBy * PCx = BPx * CPy - BPy * CPx
      const float ABx = bx - ax, ABy = by - ay, BCx = cx - bx, BCy = cy - by, CAx = ax - cx, CAy = ay - cy;
      const float APx = fx - ax, BPx = fx - bx, CPx = fx - cx;
      const float t0 = ABy * APx, t1 = BCy * BPx, t2 = CAy * CPx;
      unsigned long long *kp = s_key + ((yl0 << 5) + yl0) + xl; // yl0 * KEY_STRIDE; the next row is KEY_STRIDE keys further
      const uint32_t tb = 0x7ffffffeu - idx;
      uint32_t zk[S_ROWS];
      bool hit[S_ROWS];
#pragma unroll
      for (int k = 0; k < (int)S_ROWS; ++k) {
        const float fy = fy0 + (float)k; // (exact)
        const float APy = fy - ay, BPy = fy - by, CPy = fy - cy;
        const float e0 = ABx * APy - t0, e1 = BCx * BPy - t1, e2 = CAx * CPy - t2;
        const bool in_tri = ((e0 > 0) & (e1 > 0) & (e2 > 0)) | ((e0 < 0) & (e1 < 0) & (e2 < 0));
        const float aPBC = BPx * CPy - BPy * CPx, aPCA = CPx * APy - CPy * APx;
        const float al = aPBC / s_area, be = aPCA / s_area, ga = 1.0f - al - be;
        const float z = al * z0 + be * z1 + ga * z2;
        hit[k] = (k < lim) & in_tri; // '<=' passes and so does a NaN depth (src/Rasterizer.cpp:475): key 0 = "ask the ordered rasteriser"
        zk[k] = (z == z) ? zkey_of(z) : 0u;
        asm volatile("" : "+v"(zk[k])); // (as in the V loop: the arithmetic of the piece's pixels is one straight line)
      }
#pragma unroll
      for (int k = 0; k < (int)S_ROWS; ++k)
        if (hit[k])
          __hip_atomic_fetch_min(kp + k * KEY_STRIDE, ((unsigned long long)zk[k] << 32) | tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // the tile's farthest depth after this group (pixels outside the frame keep +inf: no dropping there); also between the
    // chunks of a long list once dropping has started paying
    if (gi + 1 < n_groups || (n_groups > 1 && base + 64 < cnt)) {
      wg_barrier(); // (every wave's minima of this group are in the keys)
      uint32_t zm = 0;
#pragma unroll
      for (int it = 0; it < 16; ++it) zm = max(zm, (uint32_t)(s_key[(it * 2 + (lane >> 5)) * KEY_STRIDE + (lane & 31)] >> 32));
      for (int o = 32; o > 0; o >>= 1) zm = max(zm, (uint32_t)__shfl_xor((int)zm, o));
      zfar = zm;
      // (the waves must agree on the bound: it decides the item counts, hence which items a batch holds — no wave may lower a
      // key before all have read them)
      wg_barrier();
    }
    } // groups
  } // chunks
  wg_barrier();

  // ---- phase C: decode the keys, write out ------------------------------------------------------------------------
  //  some final key needs the ordered algorithm  : the tile goes to k_raster_slow, nothing is written here
  //  nobody owns the tile: fused → the clear itself (z=+inf, colour 0), else the framebuffer is left untouched
  //  owned tile          : z plane + owner ids, and the tile is queued for k_shade (which writes the 3 colour planes)
  float4 z4[ITS];
  uint4 id4[ITS];
  bool any_owner = false, redo = false;
#pragma unroll
  for (int it = 0; it < ITS; ++it) {
    const int ly = (wave * ITS + it) * 8 + (lane >> 3), lx4 = (lane & 7) * 4;
    const unsigned long long *kr = &s_key[ly * KEY_STRIDE + lx4]; // (rows are 264 bytes apart: 8-byte aligned reads)
    const unsigned long long k0 = kr[0], k1 = kr[1], k2 = kr[2], k3 = kr[3];
    const uint32_t tb[4] = {(uint32_t)k0, (uint32_t)k1, (uint32_t)k2, (uint32_t)k3};
    const uint32_t zk[4] = {(uint32_t)(k0 >> 32), (uint32_t)(k1 >> 32), (uint32_t)(k2 >> 32), (uint32_t)(k3 >> 32)};
    float zz[4];
    uint32_t id[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      redo |= (zk[k] == 0u) | ((zk[k] + 0x80000001u) <= 1u); // NaN that passed / final depth ±0
      zz[k] = z_of_key(zk[k]);
      // (the payload of the tie-break: the triangle's index, or — lp_mode — index << LP_BITS | list position: the ids keep
      // either all of it or its position bits, id_pack16)
      id[k] = (tb[k] & 0x80000000u) ? (tb[k] & 0x7fffffffu) : (tb[k] == TB_NONE ? NO_TRI : ((0x7ffffffeu - tb[k]) | S_CLASS_BIT));
      any_owner |= tb[k] != TB_NONE;
    }
    z4[it] = make_float4(zz[0], zz[1], zz[2], zz[3]);
    id4[it] = make_uint4(id[0], id[1], id[2], id[3]);
  }
  bool tile_redo = __ballot(redo) != 0ull, tile_has_owner = __ballot(any_owner) != 0ull; // wave-uniform
  if constexpr (WAVES > 1) { // tile-wide: through LDS
    if (lane == 0 && tile_redo) s_wg[0] = 1u;
    if (lane == 0 && tile_has_owner) s_wg[1] = 1u;
    __syncthreads();
    tile_redo = s_wg[0] != 0u, tile_has_owner = s_wg[1] != 0u;
  }
  if (tile_redo) {
    if (lane == 0 && wave == 0) a.slow_list[atomicAdd(a.slow_count, 1u)] = frame * tiles_per_frame + tile;
    return;
  }
  const bool vec_ok = (W & 3) == 0;
  if (tile_has_owner || fused) {
    const size_t plane = (size_t)a.local_rows * (size_t)W;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t *slot = a.vis + ((size_t)frame * tiles_per_frame + tile) * PIX_SLOT;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int ly = (wave * ITS + it) * 8 + (lane >> 3), lx4 = (lane & 7) * 4;
      const int y = ty0 + ly, x4 = tx0 + lx4;
      // the owners of ALL of the tile's pixels (those beyond the frame's edge have none: their keys never got a fragment);
      // re-read by k_shade: cacheable
      if (tile_has_owner) ids_store4(slot, (uint32_t)(ly * TILE + lx4), id4[it], lp_mode, lp8);
      if (y > ty1 || x4 > tx1) continue;
      float *gz = out0 + (size_t)ly * W + x4;
      const bool full = vec_ok && x4 + 3 <= tx1;
      if (tile_has_owner) {
        if (full) {
          store_nt(gz, z4[it]); // final: k_shade recomputes the depth it needs from the owner triangle
        } else {
#define SRZ_ST(K_, M)                                                                                                  \
  if (x4 + K_ <= tx1) gz[K_] = z4[it].M;
          SRZ_ST(0, x) SRZ_ST(1, y) SRZ_ST(2, z) SRZ_ST(3, w)
#undef SRZ_ST
        }
      } else { // touched by a bbox but owned by nobody: the clear itself
        if (full) {
          store_nt(gz, z4[it]);
          store_nt(gz + plane, zero4);
          store_nt(gz + 2 * plane, zero4);
          store_nt(gz + 3 * plane, zero4);
        } else {
#define SRZ_ST(K_, M)                                                                                                  \
  if (x4 + K_ <= tx1) gz[K_] = z4[it].M, gz[plane + K_] = 0.f, gz[2 * plane + K_] = 0.f, gz[3 * plane + K_] = 0.f;
          SRZ_ST(0, x) SRZ_ST(1, y) SRZ_ST(2, z) SRZ_ST(3, w)
#undef SRZ_ST
        }
      }
    }
  }
  // owned tile → the frame's own work list (a counter per frame: one shared counter serialises ~10 ns per tile)
  if (tile_has_owner && lane == 0 && wave == 0)
    work_append(a, fd->flags, frame, frame * tiles_per_frame + tile, lb, (uint32_t)tx0 / TILE, lp_mode ? (lp8 ? WORK_LP | WORK_LP8 : WORK_LP) : 0u, cnt, off);
}

// ================================================================================================================
// k_raster_slow — the reference's ORDERED algorithm for the tiles k_raster could not finish: tiles whose band did not
// fit the record pool, tiles where a NaN or ±0 depth decides a pixel, and every touched tile of a counting run or under
// SRZ_ORDERED_RASTER.  One wave per tile (persistent grid over the list); the wave walks the FRAME's triangles in
// submission order (64 bboxes at a time, chunks whose rows miss the tile are skipped), broadcasts one hit triangle at a
// time into SGPRs and sweeps bbox ∩ tile in 64-pixel blocks against z + owner planes in LDS: a wave's LDS operations are
// ordered, so "last writer in submission order wins" needs no lock.
// ================================================================================================================
template <bool STATS>
__global__ __launch_bounds__(64) void k_raster_slow(RenderArgs a) {
  __shared__ __attribute__((aligned(16))) float zl[TILE * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) uint32_t il[TILE * LDS_STRIDE];
  const int lane = threadIdx.x & 63;
  const uint32_t tiles_per_frame = a.n_local_bands * a.tiles_x;
  const uint32_t n_slow = *as_const(a.slow_count);
  unsigned long long n_frag = 0, n_shaded = 0;
  for (uint32_t e = blockIdx.x; e < n_slow; e += gridDim.x) {
    const uint32_t entry = as_const(a.slow_list)[e];
    const uint32_t frame = entry / tiles_per_frame, tile = entry % tiles_per_frame;
    const SRZ_CAS FrameDesc *fd = as_const(a.frames) + frame;
    const uint32_t lb = tile / a.tiles_x;
    const int W = fd->width, H = fd->height;
    const uint32_t n_tris = fd->n_tris;
    const uint32_t flags = fd->flags | a.flags_or;
    const int tx0 = (int)(tile % a.tiles_x) * TILE;
    const int band = band_of((int)lb, a.shard_rank, a.shard_world);
    const int ty0 = band * BAND;
    const int tx1 = min(tx0 + TILE, W) - 1, ty1 = min(ty0 + BAND, H) - 1;
    const bool fused = (flags & SRZ_FUSED_CLEAR) != 0;
    const size_t row0 = (size_t)lb * BAND;
    float *out0 = a.out + (size_t)frame * a.frame_stride + row0 * (size_t)W;
    // ---- phase A: tile init (fused clear → +inf, else load the in/out z plane) -------------------------------
    for (int i = lane; i < TILE * TILE; i += 64) {
      const int ly = i >> 5, lx = i & 31;
      float z = __builtin_inff();
      if (!fused && tx0 + lx <= tx1 && ty0 + ly <= ty1) z = out0[(size_t)ly * W + tx0 + lx];
      zl[ly * LDS_STRIDE + lx] = z;
      il[ly * LDS_STRIDE + lx] = NO_TRI;
    }
    __builtin_amdgcn_wave_barrier();
    // ---- phase B: the frame's triangles in submission order ------------------------------------------------------
    const SRZ_CAS u32x2 *bbox = as_const(reinterpret_cast<const u32x2 *>(a.bbox + fd->tri_off));
    const SRZ_CAS float *tpos = as_const(a.tri_pos) + (size_t)fd->tri_off * a.pos_stride;
    const SRZ_CAS uint32_t *chunk_rows = as_const(a.chunk_rows) + fd->chunk_off;
    const uint32_t n_chunks = (n_tris + 63) / 64;
    for (uint32_t c = 0; c < n_chunks; ++c) {
      const uint32_t cr = chunk_rows[c]; // wave-uniform
      if ((int)(int16_t)(cr & 0xffffu) > ty1 || (int)(int16_t)(cr >> 16) < ty0) continue;
      const uint32_t my = c * 64u + (uint32_t)lane;
      const u32x2 bb = bbox[min(my, n_tris - 1u)];
      const int bsx = (int16_t)(bb.x & 0xffff), bsy = (int16_t)(bb.x >> 16), bex = (int16_t)(bb.y & 0xffff), bey = (int16_t)(bb.y >> 16);
      const bool hit = my < n_tris && bsx <= bex && bex >= tx0 && bsx <= tx1 && bey >= ty0 && bsy <= ty1;
      unsigned long long m = __ballot(hit);
      if (m == 0ull) continue;
      TriXY t;
      t.ax = t.ay = t.bx = t.by = t.cx = t.cy = t.z0 = t.z1 = t.z2 = t.v_inv = t.s_area = 0.0f;
      // Per-lane (= per-triangle) geometry of bbox ∩ tile in TILE-LOCAL coordinates, packed into one word:
      //   [4:0] x0  [9:5] x1  [14:10] y0  [19:15] y1  [25:20] v = first scalar-tail column (V part = [x0,v-1], S = [v,x1])
      //   [27:26] log2(block width) of the V sweep  [29:28] of the S sweep   (block = BW x 64/BW pixels: 8x8, 16x4, 4x16,
      //   whichever needs the fewest blocks)
      uint32_t geom = 0;
      if (hit) {
        float p[9];
        load_pos9(tpos + (size_t)my * a.pos_stride, p);
        t.ax = p[0], t.ay = p[1], t.z0 = p[2], t.bx = p[3], t.by = p[4], t.z1 = p[5], t.cx = p[6], t.cy = p[7], t.z2 = p[8];
        BranchMath bm;
        tri_consts(bm, t);
        const int x0 = max(bsx, tx0) - tx0, x1 = min(bex, tx1) - tx0, y0 = max(bsy, ty0) - ty0, y1 = min(bey, ty1) - ty0;
        const int vend = (flags & SRZ_UNIFIED) ? bex + 1 : bsx + ((bex - bsx + 1) & ~7);
        const int v = min(max(vend - tx0, x0), x1 + 1);
        const int h = y1 - y0 + 1, wv = v - x0, ws = x1 + 1 - v;
        auto pick = [h](int w) {
          const int n88 = ((w + 7) >> 3) * ((h + 7) >> 3), n164 = ((w + 15) >> 4) * ((h + 3) >> 2), n416 = ((w + 3) >> 2) * ((h + 15) >> 4);
          int l = 3;
          if (n164 < n88 && n164 <= n416) l = 4;
          if (n416 < n88 && n416 < n164) l = 2;
          return l;
        };
        geom = (uint32_t)x0 | ((uint32_t)x1 << 5) | ((uint32_t)y0 << 10) | ((uint32_t)y1 << 15) | ((uint32_t)v << 20) |
               ((uint32_t)(pick(wv) - 2) << 26) | ((uint32_t)(pick(ws) - 2) << 28);
      }
      while (m) {
        const int j = __builtin_ctzll(m);
        m &= m - 1;
        // broadcast triangle j to the wave (uniform values → SGPRs)
        TriXY u;
        u.ax = rl_f(t.ax, j), u.ay = rl_f(t.ay, j), u.bx = rl_f(t.bx, j), u.by = rl_f(t.by, j), u.cx = rl_f(t.cx, j);
        u.cy = rl_f(t.cy, j), u.z0 = rl_f(t.z0, j), u.z1 = rl_f(t.z1, j), u.z2 = rl_f(t.z2, j);
        u.v_inv = rl_f(t.v_inv, j), u.s_area = rl_f(t.s_area, j);
        const uint32_t idx = c * 64u + (uint32_t)j;
        const uint32_t g = (uint32_t)rl_i((int)geom, j);
        const int x0 = g & 31, x1 = (g >> 5) & 31, y0 = (g >> 10) & 31, y1 = (g >> 15) & 31, vx = (g >> 20) & 63;
        if (y1 < y0) continue; // (cannot happen for a hit; keeps the loops well-formed)
        if (vx > x0) { // ---- "V" sweep over columns [x0, vx-1] -------------------------------------------------
          const int lbw = (int)((g >> 26) & 3) + 2, bw = 1 << lbw, bh = 64 >> lbw;
          const int lx = lane & (bw - 1), ly = lane >> lbw;
          for (int yb = y0; yb <= y1; yb += bh) {
            const int yl = yb + ly;
            const float fy = (float)(ty0 + yl);
            const float PBy = u.by - fy, PCy = u.cy - fy, PAy = u.ay - fy;
            const bool rowok = yl <= y1;
            for (int xb = x0; xb < vx; xb += bw) {
              const int xl = xb + lx;
              const float fx = (float)(tx0 + xl);
              const float PBx = u.bx - fx, PCx = u.cx - fx, PAx = u.ax - fx;
              const float aPBC = fmsubf(PBx, PCy, PCx * PBy), aPCA = fmsubf(PCx, PAy, PAx * PCy);
              const float al = aPBC * u.v_inv, be = aPCA * u.v_inv, ga = 1.0f - (al + be);
              const float z = fmaf_(al, u.z0, fmaf_(be, u.z1, ga * u.z2));
              const int li = min(yl * LDS_STRIDE + xl, TILE * LDS_STRIDE - 1); // clamped: idle lanes read a valid word
              const float zold = zl[li];
              const bool inside = rowok & (xl < vx) & (al > 0.0f) & (be > 0.0f) & (ga > 0.0f) & (ga < 1.0f);
              const bool pass = inside & (z < zold); // strict (src/Rasterizer.cpp:334)
              if (pass) {
                zl[li] = z;
                il[li] = idx;
              }
              if (STATS) n_frag += inside ? 1 : 0, n_shaded += pass ? 1 : 0;
            }
          }
        }
        if (vx <= x1) { // ---- scalar-tail "S" sweep over columns [vx, x1] (<= 7 wide) -----------------------------
          const int lbw = (int)((g >> 28) & 3) + 2, bw = 1 << lbw, bh = 64 >> lbw;
          const int lx = lane & (bw - 1), ly = lane >> lbw;
          const float ABx = u.bx - u.ax, ABy = u.by - u.ay, BCx = u.cx - u.bx, BCy = u.cy - u.by, CAx = u.ax - u.cx,
                      CAy = u.ay - u.cy;
          for (int yb = y0; yb <= y1; yb += bh) {
            const int yl = yb + ly;
            const float fy = (float)(ty0 + yl);
            const bool rowok = yl <= y1;
            for (int xb = vx; xb <= x1; xb += bw) {
              const int xl = xb + lx;
              const float fx = (float)(tx0 + xl);
              // insideTriangle (src/Rasterizer.cpp:11-41)
              const float APx = fx - u.ax, APy = fy - u.ay, BPx = fx - u.bx, BPy = fy - u.by, CPx = fx - u.cx, CPy = fy - u.cy;
              const float e0 = ABx * APy - ABy * APx, e1 = BCx * BPy - BCy * BPx, e2 = CAx * CPy - CAy * CPx;
              const bool in_tri = ((e0 > 0) & (e1 > 0) & (e2 > 0)) | ((e0 < 0) & (e1 < 0) & (e2 < 0));
              // barycentric (scalar) + z (src/Rasterizer.cpp:43-70,473)
              const float PAx = u.ax - fx, PAy = u.ay - fy, PBx = u.bx - fx, PBy = u.by - fy, PCx = u.cx - fx, PCy = u.cy - fy;
              const float aPBC = PBx * PCy - PBy * PCx, aPCA = PCx * PAy - PCy * PAx;
              const float al = aPBC / u.s_area, be = aPCA / u.s_area, ga = 1.0f - al - be;
              const float z = al * u.z0 + be * u.z1 + ga * u.z2;
              const int li = min(yl * LDS_STRIDE + xl, TILE * LDS_STRIDE - 1);
              const float zold = zl[li];
              const bool inside = rowok & (xl <= x1) & in_tri;
              const bool pass = inside & !(z > zold); // <= passes, NaN passes (src/Rasterizer.cpp:475)
              if (pass) {
                zl[li] = z;
                il[li] = idx | S_CLASS_BIT;
              }
              if (STATS) n_frag += inside ? 1 : 0, n_shaded += pass ? 1 : 0;
            }
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    bool any_owner = false;
    for (int it = 0; it < 4; ++it) {
      const uint4 q = *reinterpret_cast<const uint4 *>(&il[(it * 8 + (lane >> 3)) * LDS_STRIDE + (lane & 7) * 4]);
      any_owner |= (q.x & q.y & q.z & q.w) != NO_TRI;
    }
    const bool tile_has_owner = __ballot(any_owner) != 0ull; // wave-uniform
    // ---- phase C: write-out (as k_raster) ---------------------------------------------------------------------------
    const bool vec_ok = (W & 3) == 0;
    if (tile_has_owner || fused) {
      const size_t plane = (size_t)a.local_rows * (size_t)W;
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int it = 0; it < 4; ++it) {
        const int ly = it * 8 + (lane >> 3), lx4 = (lane & 7) * 4;
        const int y = ty0 + ly, x4 = tx0 + lx4;
        if (y > ty1 || x4 > tx1) continue;
        const float4 z4 = *reinterpret_cast<const float4 *>(&zl[ly * LDS_STRIDE + lx4]);
        float *gz = out0 + (size_t)ly * W + x4;
        const bool full = vec_ok && x4 + 3 <= tx1;
        if (tile_has_owner) {
          if (full) {
            store_nt(gz, z4);
          } else {
#define SRZ_ST(K_, M)                                                                                                  \
  if (x4 + K_ <= tx1) gz[K_] = z4.M;
            SRZ_ST(0, x) SRZ_ST(1, y) SRZ_ST(2, z) SRZ_ST(3, w)
#undef SRZ_ST
          }
        } else {
          if (full) {
            store_nt(gz, z4);
            store_nt(gz + plane, zero4);
            store_nt(gz + 2 * plane, zero4);
            store_nt(gz + 3 * plane, zero4);
          } else {
#define SRZ_ST(K_, M)                                                                                                  \
  if (x4 + K_ <= tx1) gz[K_] = z4.M, gz[plane + K_] = 0.f, gz[2 * plane + K_] = 0.f, gz[3 * plane + K_] = 0.f;
            SRZ_ST(0, x) SRZ_ST(1, y) SRZ_ST(2, z) SRZ_ST(3, w)
#undef SRZ_ST
          }
        }
      }
    }
    if (tile_has_owner) { // the owners of all of the tile's pixels, by index (32 bits each), and the tile's work entry
      uint32_t *slot = a.vis + (size_t)entry * PIX_SLOT;
      for (int it = 0; it < 4; ++it) {
        const int ly = it * 8 + (lane >> 3), lx4 = (lane & 7) * 4;
        ids_store4(slot, (uint32_t)(ly * TILE + lx4), *reinterpret_cast<const uint4 *>(&il[ly * LDS_STRIDE + lx4]), false);
      }
      if (lane == 0) work_append(a, fd->flags, frame, entry, lb, (uint32_t)tx0 / TILE, 0u, 0u, 0u);
    }
    __builtin_amdgcn_wave_barrier(); // the planes are reused by this wave's next tile
  }
  if (STATS) {
    for (int o = 32; o > 0; o >>= 1) n_frag += __shfl_down(n_frag, o), n_shaded += __shfl_down(n_shaded, o);
    if (lane == 0) {
      if (n_frag) atomicAdd(&a.stats[ST_FRAGMENTS], n_frag);
      if (n_shaded) atomicAdd(&a.stats[ST_SHADED], n_shaded);
    }
  }
}

// ================================================================================================================
// k_shade — VISIBILITY-FIRST SHADING: one workgroup per owned tile (persistent grid over the worklist).
// The tile's owned pixels arrive COMPACTED by semantics class (the rasteriser's write-out made the two lists: V pixels, then S
// pixels), so every wave runs one of the two shader variants with full lanes instead of both under divergence;
// colours go to LDS planes and leave as coalesced 16-byte stores.
// ================================================================================================================
// k_clear — the fused clear of every tile NO bbox reaches (k_bin's tile counts), i.e. most of the framebuffer.  It runs
// on a second stream NEXT TO k_raster, which skips those tiles: no LDS and < 32 VGPRs, so its waves fit beside the
// rasteriser's on every CU and the bulk of the frame's HBM writes drains under the visibility arithmetic.
// One work item = one band of one frame, written ROW-MAJOR: a wave-instruction covers 1 KiB of one framebuffer row
// (256 consecutive pixels), so runs of untouched tiles become long contiguous DRAM bursts instead of 128-byte tile rows
// 4 KiB apart — the same bytes occupy the memory system for less time, which is what the kernels beside it pay for.
// One work item = ONE PLANE of one band of one frame (round 6; rounds 2-5: one band, its four planes interleaved row by row).  A thread writes
// its 16 bytes of 32 consecutive rows of the plane: a workgroup streams one plane's rows instead of keeping four streams 4 MB apart going, and
// the memory system — which is what the compute kernels beside the clear wait for — takes the same bytes in less time: measured beside
// k_raster / k_shade, two lanes, at each form's best grid: config 2 0.645 -> 0.738 of the roofline, config 3 0.600 -> 0.675, config 4
// 0.411 -> 0.438, config 5 0.43 -> 0.444 (NOTEBOOK r6 §6; the rows of a plane written one after the other by the whole workgroup — fewer
// stores in flight per thread — is slower again: 0.67 on config 2).
__global__ __launch_bounds__(256) void k_clear(RenderArgs a) {
  const uint32_t n_items = a.n_frames * a.n_local_bands * 4u;
  const float inf = __builtin_inff();
  // the workgroups that take part: the whole grid, or — while the set measures its grid (srz_device.h, ClearCtl) — the first `wgs` of it
  uint32_t n_wgs = gridDim.x;
  if (a.clear_wgs_dev) {
    const uint32_t w = *as_const(a.clear_wgs_dev);
    n_wgs = min(w ? w : CLEAR_CAND[0], gridDim.x);
    if (blockIdx.x >= n_wgs) return;
  }
  for (uint32_t it = blockIdx.x; it < n_items; it += n_wgs) { // it = (frame * n_local_bands + lb) * 4 + plane
    const uint32_t br = it >> 2, pl = it & 3u;
    const uint32_t f = br / a.n_local_bands, lb = br % a.n_local_bands;
    const SRZ_CAS FrameDesc *fd = as_const(a.frames) + f;
    if (!((fd->flags | a.flags_or) & SRZ_FUSED_CLEAR)) continue;
    const int W = fd->width, H = fd->height;
    const int band = band_of((int)lb, a.shard_rank, a.shard_world);
    const int rows = min(BAND, H - band * BAND);
    const size_t plane = (size_t)a.local_rows * (size_t)W;
    float *base = a.out + (size_t)f * a.frame_stride + (size_t)lb * BAND * (size_t)W + pl * plane;
    const float v = pl == 0u ? inf : 0.f; // (plane 0 = z)
    const float4 v4 = make_float4(v, v, v, v);
    const SRZ_CAS u32x2 *cnt = as_const(reinterpret_cast<const u32x2 *>(a.tile_info)) + (size_t)br * a.tiles_x;
    for (int x4 = (int)threadIdx.x * 4; x4 < W; x4 += (int)blockDim.x * 4) {
      if (cnt[(uint32_t)x4 / TILE].x != 0u) continue; // some bbox reaches this tile: the rasteriser's
      if (((W & 3) == 0)) {
        for (int ly = 0; ly < rows; ++ly) store_nt(base + (size_t)ly * W + x4, v4);
      } else { // odd widths: scalar stores, the quad may end at the frame's edge or straddle nothing else (TILE % 4 == 0)
        for (int ly = 0; ly < rows; ++ly)
          for (int k = 0; k < 4 && x4 + k < W; ++k) base[(size_t)ly * W + x4 + k] = v;
      }
    }
  }
}


// k_shade is latency-sensitive: measurably slower at 3 waves per SIMD than at 4 — hold the allocator to 128 VGPRs
#ifndef SRZ_FAST_MINW
#define SRZ_FAST_MINW 6 // (80 VGPRs: the FAST builds for 1 / 2 lights fit with a few prologue spills; 3 lights keep 5 — see k_shade)
#endif
#ifndef SRZ_STAGE_TRIS
#define SRZ_STAGE_TRIS 96 // triangles of a tile's list k_shade stages in LDS (12 KB colours + 4 KB lists + 9 KB triangles: 6 workgroups per CU)
#endif
constexpr uint32_t STAGE_TRIS = SRZ_STAGE_TRIS, STAGE_SD = 16;
// (s_tri doubles as the {pixel, index} pair list of tiles whose ids are triangle indices: 2 dwords per pixel of the tile)
static_assert(SRZ_STAGE_TRIS * 6 * 16 >= 2 * 32 * 32 * 4, "SRZ_STAGE_TRIS: s_tri must hold a tile's 1024 {pixel, index} pairs (>= 86 triangles)");
#ifndef SRZ_GENPOW_MINW
#define SRZ_GENPOW_MINW 5 // (the builds with pow_fast: 81 VGPRs for two lights; held to 80 they spill 20 bytes into the chunk loops: -3 %)
#endif
#ifndef SRZ_SHADE_MINW
#define SRZ_SHADE_MINW 4
#endif
// Two builds of the same kernel share the persistent walk over the frames' work lists:
//   FAST     (one build per light count 1..4) frames whose shading is fully described at compile time up to the shader type and
//            the exponent: FASTNL lights, an integer exponent 0..256 (FD_FAST_SHADE + the count, decided on the host).  Per 64-pixel chunk ONE wave-uniform
//            switch picks the variant compiled for the chunk's shader type (mixed chunks take one pass per type);
//            optimistic FastMath only — a tile where an operand left the fast range is handed back through redo_list.
//   generic  every other frame with per-pixel generality (any shader, any light count, any exponent), FastMath first and
//            the IEEE expansions for a tile that needs them; then the tiles the FAST build handed back, IEEE at once.
//   FASTNL < 0: the FAST build for -FASTNL lights and a non-integer exponent in (0, 4096] (pow_fast; its ambiguous roundings go to the generic build) — frames that
//            differ from the common case only in Shader::p keep the per-chunk variants, the unrolled lights and the hoisted texel
//   APPROX   (SRZ_OPT_APPROX_SHADE, one build per light count 1..4): the FAST walk with the ApproxMath policy — the tolerance mode.
//            Any exponent (x^p is exp2(p log2 x) there), never a re-shade; frames it does not cover (0 or > 4 lights, BUMP /
//            DISPLACEMENT batches) keep the exact builds.
template <bool STATS, int FASTNL, bool BUMPY = false, bool APPROX = false>
__global__ __launch_bounds__(256, (FASTNL > 0 && FASTNL <= 2 && !BUMPY) ? SRZ_FAST_MINW : (FASTNL < 0 && FASTNL >= -2) ? SRZ_GENPOW_MINW
                                  : ((FASTNL == 3 || FASTNL == -3) && !BUMPY) ? 5 : SRZ_SHADE_MINW)
void k_shade(RenderArgs a) {
  constexpr bool FAST = FASTNL != 0, GENPOW = FASTNL < 0;
  static_assert(FAST || !BUMPY, "BUMPY is a property of the FAST builds");
  static_assert(!APPROX || (FASTNL > 0 && !BUMPY && !STATS), "the tolerance mode has FAST builds for 1..4 lights only");
  static_assert(!(GENPOW && BUMPY), "frames with BUMP / DISPLACEMENT batches and a non-integer exponent take the generic build");
  __shared__ __attribute__((aligned(16))) float s_c[3][TILE * TILE];
  // the tile's owned pixels compacted by class: V entries [0, nV), S entries [nV, nV + nS), row-major each; an entry = pixel |
  // owner's list position << 10 — or, for tiles whose ids are triangle indices, {pixel, index} pairs in s_tri's memory (no
  // triangles are staged for those)
  __shared__ __attribute__((aligned(16))) uint32_t s_ent[PIX_SLOT];
  __shared__ uint32_t s_wcnt[4];                                        // per wave: V-class | S-class << 16 pixels among its 256
  // the triangles of the tile's list, staged once per tile (96 bytes each, by list position) + their batch ids, and the frame's
  // shader descriptors: a pixel takes its owner's data from here instead of gathering it from memory — 26 KB of LDS in all,
  // six workgroups per CU as before
  __shared__ __attribute__((aligned(16))) f32x4 s_tri[STAGE_TRIS * 6];
  __shared__ uint16_t s_bat[STAGE_TRIS];
  __shared__ __attribute__((aligned(16))) ShadeDescG s_sd[STAGE_SD];
  __shared__ uint32_t s_flag;                                           // "some operand left FastMath's range"

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long n_vis = 0, n_vis_tex = 0;
#ifdef SRZ_PHASE_PROBE /* dev build (tools/mkvariant.sh probe -DSRZ_PHASE_PROBE, tools/phase_probe.py): where a wave's time per tile goes */
  unsigned long long ph[6] = {0, 0, 0, 0, 0, 0}, ph_t = 0;
  // (and the clock the wave ran at: shader clocks against the constant 100 MHz counter over the wave's whole life)
  const unsigned long long ph_c0 = __builtin_amdgcn_s_memtime(), ph_r0 = __builtin_amdgcn_s_memrealtime();
#define SRZ_STAMP(K) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph[K] += t_ - ph_t, ph_t = t_; }
#else
#define SRZ_STAMP(K)
#endif

  // ---- one owned tile: x = its work-list entry (work_append).  mode: 0 = FAST variants, 1 = generic (FastMath, then IEEE if
  //      needed), 2 = IEEE at once -------------------------------------------------------------------------------------------------
  auto shade_tile = [&](const u32x4 x, auto mode_c) {
    constexpr int MODE = decltype(mode_c)::value;
    // (everything a thread derives from its index is derived HERE, per tile, from an opaque copy: hoisted out of the tile loop
    // those values live across the chunk loops, get spilled, and every tile starts with scratch round trips in front of its loads)
    int tid = (int)threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
#ifdef SRZ_PHASE_PROBE
    ph_t = __builtin_amdgcn_s_memtime(), ph[5] += 1;
#endif
    // (likewise the kernel's arguments: read from the kernarg segment per tile — scalar loads that hit the constant cache —
    // through an opaque pointer.  Loaded once at kernel entry they are ~50 scalar registers alive across everything: the compiler
    // parks them in VGPR lanes (v_writelane) and fetches them back with a VALU instruction each (v_readlane), 8 % of the
    // kernel's vector instructions)
    const SRZ_CAS RenderArgs *ap = (const SRZ_CAS RenderArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ap));
    const uint32_t tpf = ap->n_local_bands * ap->tiles_x;
    // how the tile's ids name a pixel's owner: by position in the tile's triangle list (16-bit ids) — the list's triangles are
    // staged in LDS if they fit (by_lp && staged), else looked up per pixel (by_lp) — or by index in the frame (32-bit ids)
    const bool by_lp = (x.y & WORK_LP) != 0u, staged = by_lp && x.z <= STAGE_TRIS; // workgroup-uniform
    const bool lp8 = (x.y & WORK_LP8) != 0u;                                        // (8-bit ids: id_pack8)
    const SRZ_CAS uint32_t *tlist = as_const(ap->pool) + x.w;
    // ---- 1. this thread's 4 pixels: their owner ids (nothing but the entry is needed for the address: the load is in flight
    //         under the frame descriptor's scalar loads), and the indices of the list entries whose pieces it will stage
    const int ly = wave * 8 + (lane >> 3), lx4 = (lane & 7) * 4;
    const int p0 = ly * TILE + lx4;
    // (the 16-bit codes are classified as they are — position | 0x8000 for the S class, 0xffff = nobody —, never widened; and they
    // are taken apart only at the classification below: unpacked here, the wait for this load would sit in front of the index
    // loads that follow, one more round trip in a row per tile)
    const uint32_t f = x.x, lb = x.y & 1023u, tx = (x.y >> 10) & 1023u;
    const uint32_t tile_slot = f * tpf + lb * ap->tiles_x + tx;
    uint4 id_raw = make_uint4(0u, 0u, 0u, 0u);
    {
      const uint32_t *slot = ap->vis + (size_t)tile_slot * PIX_SLOT;
      if (lp8) {
        id_raw.x = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(slot) + p0);
      } else if (by_lp) {
        const u32x2 w = *reinterpret_cast<const u32x2 *>(reinterpret_cast<const uint16_t *>(slot) + p0);
        id_raw.x = w.x, id_raw.y = w.y;
      } else {
        id_raw = *reinterpret_cast<const uint4 *>(slot + p0);
      }
    }
    const uint32_t none = lp8 ? 0xffu : by_lp ? 0xffffu : NO_TRI, sbit = lp8 ? 0x80u : by_lp ? 0x8000u : S_CLASS_BIT; // (scalar)
    constexpr int PASSES = (STAGE_TRIS * 6 + 255) / 256;
    const uint32_t n_pc = staged ? x.z * 6u : 0u;
    uint32_t ti[PASSES];
#pragma unroll
    for (int k = 0; k < PASSES; ++k) { // (passes the list does not reach are skipped by a SCALAR branch: most lists fit the first)
      ti[k] = 0u;
      if (n_pc > 256u * k) {
        const uint32_t pc = (uint32_t)tid + 256u * k;
        if (pc < n_pc) ti[k] = tlist[pc / 6u];
      }
    }
    const SRZ_CAS FrameDesc *fd = as_const(ap->frames) + f;
    const int W = fd->width, H = fd->height;
    const uint32_t tri_off = fd->tri_off, batch_off = fd->batch_off;
    const uint32_t flags = fd->flags | ap->flags_or;
    const bool fused = (flags & SRZ_FUSED_CLEAR) != 0;
    const SRZ_CAS srz_tri *tris = as_const(ap->tris) + tri_off;
    const SRZ_CAS uint16_t *tri_batch = as_const(ap->tri_batch) + tri_off;
    const SRZ_CAS ShadeDescG *sdesc = as_const(ap->sdesc) + batch_off;
    const bool sd_staged = fd->n_batches <= STAGE_SD; // workgroup-uniform

    const int band = band_of((int)lb, ap->shard_rank, ap->shard_world);
    const int tx0 = (int)tx * TILE, ty0 = band * BAND;
    const int tx1 = min(tx0 + TILE, W) - 1, ty1 = min(ty0 + BAND, H) - 1;
    const size_t plane = (size_t)ap->local_rows * (size_t)W;
    const size_t row0 = (size_t)lb * BAND;
    float *out0 = ap->out + (size_t)f * ap->frame_stride + row0 * (size_t)W;

    // (the frame's shader descriptors: loaded here, parked in LDS behind the first barrier — the store's wait for the load would
    // otherwise hold wave 0, and with it the barrier, until this load, the last one issued, is back)
    const bool sd_mine = sd_staged && (uint32_t)tid < fd->n_batches;
    ShadeDescG sd_reg;
    sd_reg.shader = 0, sd_reg.tw = sd_reg.th = 1, sd_reg._pad = 0, sd_reg.tex = nullptr;
    if (sd_mine) {
      const SRZ_CAS ShadeDescG *g = sdesc + tid;
      sd_reg.shader = g->shader, sd_reg.tw = g->tw, sd_reg.th = g->th, sd_reg.tex = g->tex;
    }

    // ---- 2. the colour staging planes start as what the pixels this call does not own must hold: 0 after the fused clear, else
    //         the colour already in the framebuffer (the z plane is not read: the shader's depth is recomputed from the owner
    //         triangle with k_raster's own operations, which is cheaper than 4 bytes per pixel of HBM read)
    const int y = ty0 + ly, x4 = tx0 + lx4;
    const bool in_tile = y <= ty1 && x4 <= tx1;
    const bool full = in_tile && ((W & 3) == 0) && x4 + 3 <= tx1;
    float *gz = out0 + (size_t)ly * W + x4;
    float4 C0 = make_float4(0.f, 0.f, 0.f, 0.f), C1 = C0, C2 = C0;
    if (!fused) { // keep the colour of pixels this call does not own
      if (full) {
        C0 = *reinterpret_cast<const float4 *>(gz + plane);
        C1 = *reinterpret_cast<const float4 *>(gz + 2 * plane);
        C2 = *reinterpret_cast<const float4 *>(gz + 3 * plane);
      } else if (in_tile) {
#define SRZ_LD(K_, M)                                                                                                  \
  if (x4 + K_ <= tx1) C0.M = gz[plane + K_], C1.M = gz[2 * plane + K_], C2.M = gz[3 * plane + K_];
        SRZ_LD(0, x) SRZ_LD(1, y) SRZ_LD(2, z) SRZ_LD(3, w)
#undef SRZ_LD
      }
    }
    *reinterpret_cast<float4 *>(&s_c[0][p0]) = C0;
    *reinterpret_cast<float4 *>(&s_c[1][p0]) = C1;
    *reinterpret_cast<float4 *>(&s_c[2][p0]) = C2;
    // classify this thread's 4 pixels: per-thread counts, one packed wave scan (DPP), the waves' totals through LDS
#ifdef SRZ_PHASE_PROBE
    asm volatile("" : "+v"(id_raw.x), "+v"(id_raw.y), "+v"(ti[0])); // (the wait for the owner ids and the first indices lands here)
    SRZ_STAMP(0) // tile start → owner ids + list indices here
#endif
    uint32_t idk[4] = {id_raw.x, id_raw.y, id_raw.z, id_raw.w};
    if (lp8)
      idk[0] = id_raw.x & 0xffu, idk[1] = (id_raw.x >> 8) & 0xffu, idk[2] = (id_raw.x >> 16) & 0xffu, idk[3] = id_raw.x >> 24;
    else if (by_lp)
      idk[0] = id_raw.x & 0xffffu, idk[1] = id_raw.x >> 16, idk[2] = id_raw.y & 0xffffu, idk[3] = id_raw.y >> 16;
    uint32_t cnt2 = 0; // V count | S count << 16 of this thread
#pragma unroll
    for (int k = 0; k < 4; ++k) cnt2 += idk[k] == none ? 0u : ((idk[k] & sbit) ? 0x10000u : 1u);
    const uint32_t incl2 = wave_scan_add(cnt2);
    if (lane == 63) s_wcnt[wave] = incl2;
    if (tid == 0) s_flag = 0;
    // ---- 1b. the tile's triangles → LDS by DMA: 16-byte piece q of list entry l per thread (six threads per triangle: coalesced
    //          96 bytes).  The pieces go from memory straight to LDS — the wave's 64 land one after the other at a wave-uniform
    //          base, which is s_tri's own order — and hold no registers while in flight, under the compaction below
    if (staged) {
#pragma unroll
      for (int k = 0; k < PASSES; ++k) {
        if (n_pc <= 256u * k) break; // scalar
        const uint32_t pc = (uint32_t)tid + 256u * k;
        if (pc < n_pc) {
          const uint32_t q = pc % 6u;
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const f32x4 *>(ap->tris + tri_off + (ti[k] & PACK_IDX_MASK)) + q,
                                           (__attribute__((address_space(3))) void *)(s_tri + (wave * 64 + 256 * k)), 16, 0, 0);
          if (q == 0u) s_bat[pc / 6u] = (uint16_t)(ti[k] >> PACK_IDX_BITS); // (by_lp = FD_PACKED: the list entry carries the batch)
        }
      }
    }
    // raw barrier: only the LDS stores above must be visible (lgkmcnt); the DMA pieces stay in flight across it.  s_barrier is
    // IntrNoMem for the compiler: the two empty asm statements are compiler-level fences that keep the LDS stores above in front
    // of it and the reads of the other waves' s_wcnt[] behind it (no instruction, vmcnt untouched)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    SRZ_STAMP(1) // → classification done, first barrier passed
    uint32_t bV = 0, bS = 0, nV = 0, nS = 0; // this wave's first slot in each list, list lengths (all wave-uniform)
#pragma unroll
    for (int w2 = 0; w2 < 4; ++w2) {
      const uint32_t t2 = s_wcnt[w2];
      bV += w2 < wave ? (t2 & 0xffffu) : 0u, bS += w2 < wave ? (t2 >> 16) : 0u, nV += t2 & 0xffffu, nS += t2 >> 16;
    }
    nV = (uint32_t)__builtin_amdgcn_readfirstlane((int)nV), nS = (uint32_t)__builtin_amdgcn_readfirstlane((int)nS);
    if (sd_mine) s_sd[tid] = sd_reg;
    uint32_t *const s_ent2 = reinterpret_cast<uint32_t *>(s_tri); // ({pixel, index} pairs: tiles without staged triangles)
    {
      uint32_t oV = bV + ((incl2 - cnt2) & 0xffffu), oS = nV + bS + ((incl2 - cnt2) >> 16);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const bool own = idk[k] != none, isS = own && (idk[k] & sbit) != 0;
        const uint32_t o = isS ? oS : oV, owner = idk[k] & ~sbit;
        if (own) {
          if (by_lp)
            s_ent[o] = (uint32_t)(p0 + k) | (owner << PIX_BITS);
          else
            s_ent2[2 * o] = (uint32_t)(p0 + k), s_ent2[2 * o + 1] = owner;
        }
        oV += (own && !isS) ? 1u : 0u, oS += isS ? 1u : 0u;
      }
    }
    // (the DMA pieces must have landed: an LDS-DMA is a pending LDS write on the vector-memory counter — the barrier's fence waits
    // for it by itself with this compiler; the explicit wait states the requirement)
    __builtin_amdgcn_s_waitcnt(0x0f70); // vmcnt(0), expcnt / lgkmcnt untouched
    __syncthreads();
    SRZ_STAMP(2) // → compaction done, staged triangles landed, second barrier passed

    // ---- 3. dense passes: the two lists are cut into 64-entry chunks dealt round-robin to the 4 waves, so a wave runs
    //         ONE shader variant per chunk with (nearly) all lanes busy; only the last chunk of each list is partial.
    const uint32_t cV = (nV + 63) >> 6, cS = (nS + 63) >> 6;
    // (one loop per class: a single loop over both keeps the registers of the V and of the S shader alive together)
    auto class_pass = [&](auto policy, auto is_v, auto staged_c, bool count) -> bool {
      using M = decltype(policy);
      constexpr bool isV = decltype(is_v)::value, STAGED = decltype(staged_c)::value;
      bool bad = false;
      // the frame's shading constants are (re)read here, after the IO phase: scalar loads from a hot line, and ~20 SGPRs
      // fewer alive across the phase that has none to spare
      FrameK K;
      K.eye[0] = fd->eye[0], K.eye[1] = fd->eye[1], K.eye[2] = fd->eye[2];
      K.ka[0] = fd->ka[0], K.ka[1] = fd->ka[1], K.ka[2] = fd->ka[2];
      K.ks[0] = fd->ks[0], K.ks[1] = fd->ks[1], K.ks[2] = fd->ks[2];
      K.p = fd->p, K.kh = fd->kh, K.kn = fd->kn, K.n_lights = fd->n_lights;
      K.grey = (fd->flags & FD_GREY) != 0u;
      K.lights = as_const(ap->lights) + fd->light_off;
      // chunk c of the tile, dealt round-robin to the waves: the S chunks first (the dearer ones: ~450 instructions against ~310), so that
      // the waves' loads differ by at most one V chunk at the barrier behind the passes
      for (uint32_t c = (uint32_t)wave; c < cV + cS; c += 4) {
        if ((c >= cS) != isV) continue;
        const uint32_t i = (isV ? c - cS : c) * 64 + lane;
        if (i >= (isV ? nV : nS)) continue;
        const uint32_t slot = isV ? i : nV + i;
        uint32_t p, id;
        if (by_lp) {
          const uint32_t en = s_ent[slot];
          p = en & PIX_MASK, id = en >> PIX_BITS;
        } else {
          p = s_ent2[2 * slot], id = s_ent2[2 * slot + 1];
        }
        float r0 = 1.f, r1 = 2.f, r2 = 3.f;
        ShadeDesc sd;
        TriFetch tf;
        const f32x4 *late = nullptr;
        if constexpr (STAGED) { // the owner's positions and its batch from the tile's staged list now, its attributes later (late_fetch)
          late = s_tri + id * 6u; // (read inside the shading variant: early_fetch / late_fetch)
          tf.batch = s_bat[id];
        } else {
          if (by_lp) id = tlist[id] & PACK_IDX_MASK; // (a list too long for the stage: position → index, then the gather)
          fetch_tri(tris, tri_batch, id, tf);
        }
        const int px = tx0 + (int)(p & 31), py = ty0 + (int)(p >> 5);
        M m;
        if constexpr (MODE == 0) {
          // The pixels of a chunk are shaded batch by batch (nearly always there is one): the batch's shader descriptor — type,
          // texture size and address — is WAVE-UNIFORM (scalar registers: LDS read + readfirstlane, or scalar loads), one
          // scalar switch picks the variant compiled for its shader type, and the lanes of other batches wait for their pass
          bool todo = true;
          for (unsigned long long tm = __ballot(true); tm != 0ull; tm = __ballot(todo)) {
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)tf.batch, __builtin_ctzll(tm)); // batch of the first pixel still to do
            if (sd_staged) {
              const ShadeDescG &g = s_sd[b0];
              const unsigned long long tp = reinterpret_cast<unsigned long long>(g.tex);
              sd.shader = __builtin_amdgcn_readfirstlane(g.shader), sd.tw = __builtin_amdgcn_readfirstlane(g.tw);
              sd.th = __builtin_amdgcn_readfirstlane(g.th);
              sd.tex = reinterpret_cast<const SRZ_CAS uint32_t *>((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)tp) |
                                                                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(tp >> 32)) << 32));
            } else {
              const SRZ_CAS ShadeDescG *g = sdesc + b0; // (uniform address: scalar loads)
              sd.shader = g->shader, sd.tw = g->tw, sd.th = g->th, sd.tex = as_const(g->tex);
            }
            if (todo && tf.batch == b0) {
              auto run = [&](auto sh) {
                if constexpr (isV)
                  shade_pixel_v<M, decltype(sh)::value, FASTNL>(m, K, sd, tf, late, px, py, r0, r1, r2);
                else
                  shade_pixel_s<M, decltype(sh)::value, FASTNL>(m, K, sd, tf, late, px, py, r0, r1, r2);
              };
              const int sh0 = sd.shader; // wave-uniform
              if (sh0 == SRZ_SHADER_TEXTURE)
                run(std::integral_constant<int, SRZ_SHADER_TEXTURE>{});
              else if (sh0 == SRZ_SHADER_PHONG)
                run(std::integral_constant<int, SRZ_SHADER_PHONG>{});
              else if (!BUMPY || sh0 == SRZ_SHADER_NORMAL)
                run(std::integral_constant<int, SRZ_SHADER_NORMAL>{});
              else if constexpr (BUMPY) { // (the builds for frames with BUMP / DISPLACEMENT batches: their scalar-tail variants
                                          // cost ~20 VGPRs more than the other three types need)
                if (sh0 == SRZ_SHADER_BUMP)
                  run(std::integral_constant<int, SRZ_SHADER_BUMP>{});
                else
                  run(std::integral_constant<int, SRZ_SHADER_DISPLACEMENT>{});
              }
              todo = false;
            }
          }
        } else {
          if (sd_staged) {
            const ShadeDescG &g = s_sd[tf.batch];
            sd.shader = g.shader, sd.tw = g.tw, sd.th = g.th, sd.tex = as_const(g.tex);
          } else {
            const SRZ_CAS ShadeDescG *g = sdesc + tf.batch;
            sd.shader = g->shader, sd.tw = g->tw, sd.th = g->th, sd.tex = as_const(g->tex);
          }
          if constexpr (isV)
            shade_pixel_v<M>(m, K, sd, tf, late, px, py, r0, r1, r2);
          else
            shade_pixel_s<M>(m, K, sd, tf, late, px, py, r0, r1, r2);
        }
        if constexpr (std::is_same<M, FastMath>::value) bad |= m.is_bad();
        s_c[0][p] = r0, s_c[1][p] = r1, s_c[2][p] = r2;
        if (STATS && count)
          n_vis++, n_vis_tex += (sd.shader == SRZ_SHADER_TEXTURE || sd.shader == SRZ_SHADER_DISPLACEMENT || sd.shader == SRZ_SHADER_BUMP);
      }
      return bad;
    };
    auto dense_passes = [&](auto policy, bool count) -> bool {
      bool bv, bs;
      if (staged) { // workgroup-uniform
        bv = class_pass(policy, std::true_type{}, std::true_type{}, count);
        bs = class_pass(policy, std::false_type{}, std::true_type{}, count);
      } else {
        bv = class_pass(policy, std::true_type{}, std::false_type{}, count);
        bs = class_pass(policy, std::false_type{}, std::false_type{}, count);
      }
      return bv | bs;
    };
    bool skip_write = false;
    if constexpr (APPROX) {
      dense_passes(ApproxMath{}, true);
      __syncthreads();
    } else if constexpr (MODE == 2) {
      if (STATS && tid == 0) atomicAdd(&ap->stats[ST_DBG_IEEE_TILES], 1ull);
      dense_passes(IeeeMath{}, true);
      __syncthreads();
    } else {
      // First with the optimistic FastMath; a tile where any operand left the fast range (degenerate normals, a light
      // straight above a pixel, ...) is shaded again with the IEEE expansions.
      const bool bad = dense_passes(FastMath{}, true);
      if (__ballot(bad) != 0ull && lane == 0) s_flag = 1u;
      __syncthreads();
      if (s_flag) { // workgroup-uniform
        if constexpr (MODE == 0) { // hand the tile to the generic build
          if (tid == 0) ap->redo_list[atomicAdd(ap->redo_count, 1u)] = make_uint4(x.x, x.y, x.z, x.w);
          skip_write = true;
        } else {
          if (STATS && tid == 0) atomicAdd(&ap->stats[ST_DBG_IEEE_TILES], 1ull);
          dense_passes(IeeeMath{}, false);
          __syncthreads();
        }
      }
    }

    SRZ_STAMP(3) // → dense passes done (and the barrier behind them)
    // ---- 4. coalesced write-out of the three colour planes ----------------------------------------------------------
    if (!skip_write) {
      C0 = *reinterpret_cast<const float4 *>(&s_c[0][p0]);
      C1 = *reinterpret_cast<const float4 *>(&s_c[1][p0]);
      C2 = *reinterpret_cast<const float4 *>(&s_c[2][p0]);
      if (full) {
        store_nt(gz + plane, C0);
        store_nt(gz + 2 * plane, C1);
        store_nt(gz + 3 * plane, C2);
      } else if (in_tile) {
#define SRZ_ST(K_, M)                                                                                                  \
  if (x4 + K_ <= tx1) gz[plane + K_] = C0.M, gz[2 * plane + K_] = C1.M, gz[3 * plane + K_] = C2.M;
        SRZ_ST(0, x) SRZ_ST(1, y) SRZ_ST(2, z) SRZ_ST(3, w)
#undef SRZ_ST
      }
    }
    SRZ_STAMP(4) // → write-out issued
  };

  // Workgroup b (on XCD b % 8) shades entries b/8, b/8 + G/8, ... of work list [this build][b % 8]: the tiles of the frames
  // that XCD rasterised (its L2 holds their pixel lists), in frame order.  The grid is LARGE (a tile or two per workgroup),
  // so the hardware dispatcher hands the work out in order and at tile granularity: all resident workgroups are on the same
  // few frames at any time (triangles + lists stay cache-resident), a CU slowed by the clear's waves simply receives fewer
  // workgroups, the kernel's tail is a tile long — and, unlike a persistent grid, CU slots keep turning over, which lets the
  // kernels of another stream (LaneRenderer: the next batch's k_bin / k_raster) in beside this one.  Measured on MI355X, 256
  // frames of 1024^2: a persistent grid of 4096 workgroups dealing every 128th tile of a frame 0.78 ms (one stream), a
  // persistent grid drawing tiles from atomic cursors the same on one stream but 7 % slower on two (it holds every CU slot to
  // its end), this walk 0.73 ms.
  constexpr uint32_t KIND = FAST ? (uint32_t)(light_count<FASTNL>() - 1) + (BUMPY ? 4u : 0u) + (GENPOW ? 8u : 0u) : SHADE_KIND_GENERIC;
  const uint32_t L = KIND * 8u + (blockIdx.x & 7u), Ls = ((uint32_t)(a.kind_slots >> (4u * KIND)) & 15u) * 8u + (blockIdx.x & 7u);
  const SRZ_CAS u32x4 *list = reinterpret_cast<const SRZ_CAS u32x4 *>(as_const(a.worklist)) + (size_t)Ls * a.work_cap;
  // the list's length and this workgroup's first entry are loaded TOGETHER (the entry's index is clamped into the list's
  // storage; it is used only if it lies below the length): one round trip instead of two before the tile's own loads start
  uint32_t w = blockIdx.x >> 3;
  u32x4 x = list[min(w, a.work_cap - 1u)];
  const uint32_t n_work = (FAST || a.force_generic || a.any_generic) ? as_const(a.work_count)[L * CNT_STRIDE] : 0u;
  while (w < n_work) {
    const u32x4 xc = x;
    w += gridDim.x >> 3;
    if (w < n_work) x = list[w];
    if constexpr (FAST)
      shade_tile(xc, std::integral_constant<int, 0>{});
    else
      shade_tile(xc, std::integral_constant<int, 1>{});
    __syncthreads(); // LDS is reused by the next tile
  }

  if constexpr (!FAST) { // the tiles the FAST build handed back (it ran before this kernel on the same stream)
    const uint32_t n_redo = *as_const(a.redo_count);
    for (uint32_t i = blockIdx.x; i < n_redo; i += gridDim.x) {
      const u32x4 xr = reinterpret_cast<const SRZ_CAS u32x4 *>(as_const(a.redo_list))[i];
      shade_tile(xr, std::integral_constant<int, 2>{});
      __syncthreads(); // LDS is reused by the next tile
    }

  }
#ifdef SRZ_PHASE_PROBE
  if (lane == 0 && FAST) {
    for (int k = 0; k < 6; ++k)
      if (ph[k]) atomicAdd(&a.stats[ST_DBG_CYC_A + k], ph[k]);
    atomicAdd(&a.stats[0], __builtin_amdgcn_s_memtime() - ph_c0), atomicAdd(&a.stats[1], __builtin_amdgcn_s_memrealtime() - ph_r0);
  }
#endif
  if (STATS) {
    for (int o = 32; o > 0; o >>= 1) {
      n_vis += __shfl_down(n_vis, o);
      n_vis_tex += __shfl_down(n_vis_tex, o);
    }
    if (lane == 0) {
      if (n_vis) atomicAdd(&a.stats[ST_VISIBLE], n_vis);
      if (n_vis_tex) atomicAdd(&a.stats[ST_VISIBLE_TEX], n_vis_tex);
    }
  }
}

// ================================================================================================================
// k_resolve8 — display()'s resolve (src/Render.cpp:61-62): cv::merge(planes 0,1,2) + convertTo(CV_8UC3) =
// saturate_cast<uchar>(cvRound(v)): round half to even, clamp to [0,255]; NaN → 0.  4 pixels per thread: three 16-byte
// plane reads → 12 output bytes (three dword stores).
// ================================================================================================================
__device__ __forceinline__ uint32_t to_u8(float v) {
  if (!(v == v)) return 0u;
  const float r = __builtin_rintf(v); // round half to even (default rounding mode)
  return r <= 0.0f ? 0u : (r >= 255.0f ? 255u : (uint32_t)r);
}
__global__ __launch_bounds__(256) void k_resolve8(const float *planes, uint8_t *out, uint32_t n_frames, uint32_t rows, uint32_t W,
                                                  uint64_t frame_stride) {
  const uint64_t quads_per_frame = (uint64_t)rows * W / 4; // W % 4 == 0 is required by the launcher
  const uint64_t total = quads_per_frame * n_frames;
  for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t f = q / quads_per_frame, i = (q - f * quads_per_frame) * 4;
    const float *p = planes + f * frame_stride + (uint64_t)rows * W; // plane 1 = c0 (plane 0 is z)
    const SRZ_CAS f32x4 *c0 = reinterpret_cast<const SRZ_CAS f32x4 *>(as_const(p + i));
    const SRZ_CAS f32x4 *c1 = reinterpret_cast<const SRZ_CAS f32x4 *>(as_const(p + (uint64_t)rows * W + i));
    const SRZ_CAS f32x4 *c2 = reinterpret_cast<const SRZ_CAS f32x4 *>(as_const(p + 2ull * rows * W + i));
    const f32x4 a = *c0, b = *c1, c = *c2;
    const uint32_t b0 = to_u8(a.x), g0 = to_u8(b.x), r0 = to_u8(c.x), b1 = to_u8(a.y), g1 = to_u8(b.y), r1 = to_u8(c.y);
    const uint32_t b2 = to_u8(a.z), g2 = to_u8(b.z), r2 = to_u8(c.z), b3 = to_u8(a.w), g3 = to_u8(b.w), r3 = to_u8(c.w);
    uint32_t *o = reinterpret_cast<uint32_t *>(out + (f * (uint64_t)rows * W + i) * 3);
    o[0] = b0 | (g0 << 8) | (r0 << 16) | (b1 << 24);
    o[1] = g1 | (r1 << 8) | (b2 << 16) | (g2 << 24);
    o[2] = r2 | (b3 << 8) | (g3 << 16) | (r3 << 24);
  }
}
// Frames whose plane size rows * W is not a multiple of 4 pixels (odd sizes: the planes behind the first are not 16-byte
// aligned and the 12-byte output groups would straddle frames): one pixel per thread, dword loads, byte stores — rare sizes,
// same result
__global__ __launch_bounds__(256) void k_resolve8_px(const float *planes, uint8_t *out, uint32_t n_frames, uint32_t rows, uint32_t W,
                                                     uint64_t frame_stride) {
  const uint64_t px_per_frame = (uint64_t)rows * W, total = px_per_frame * n_frames;
  for (uint64_t q = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; q < total; q += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t f = q / px_per_frame, i = q - f * px_per_frame;
    const SRZ_CAS float *p = as_const(planes + f * frame_stride + px_per_frame + i);
    uint8_t *o = out + q * 3;
    o[0] = (uint8_t)to_u8(p[0]), o[1] = (uint8_t)to_u8(p[px_per_frame]), o[2] = (uint8_t)to_u8(p[2 * px_per_frame]);
  }
}
void launch_resolve8(const float *planes, uint8_t *out, uint32_t n_frames, uint32_t rows, uint32_t W, uint64_t frame_stride,
                     hipStream_t s) {
  const uint64_t px = (uint64_t)n_frames * rows * W;
  if (!px) return;
  // the 4-pixel kernel walks a frame's plane as one flat run: it needs rows * W % 4 == 0 (then every plane and every frame starts
  // on a 16-byte boundary of a 16-byte aligned buffer, and every 12-byte output group on a 4-byte one) — not W % 4 == 0
  if ((((uint64_t)rows * W) & 3u) == 0 && ((uintptr_t)planes & 15u) == 0 && ((uintptr_t)out & 3u) == 0) {
    const uint32_t grid = (uint32_t)std::min<uint64_t>((px / 4 + 255) / 256, 8192);
    hipLaunchKernelGGL(k_resolve8, dim3(grid), dim3(256), 0, s, planes, out, n_frames, rows, W, frame_stride);
  } else {
    const uint32_t grid = (uint32_t)std::min<uint64_t>((px + 255) / 256, 8192);
    hipLaunchKernelGGL(k_resolve8_px, dim3(grid), dim3(256), 0, s, planes, out, n_frames, rows, W, frame_stride);
  }
}

// ================================================================================================================
// k_deinterleave — the multi-GPU exchange's second half.  The all-gather leaves every rank's shard one after the other,
//   gathered[rank][frame][plane][local band][32 rows][row bytes]        (band b = band_of(local band, rank): srz_device.h),
// and this kernel restores the reference's row-major planes,
//   full[frame][plane][band][32 rows][row bytes],
// in one pass of 16-byte (or 4-byte) units: consecutive threads copy consecutive units of one destination row.
// ================================================================================================================
template <typename U>
__global__ __launch_bounds__(256) void k_deinterleave(const U *gathered, U *full, uint32_t world, uint32_t n_fp /* frames x planes */,
                                                      uint32_t bands_per_rank, uint32_t band_units_ /* units per 32-row band */) {
  const uint64_t band_units = band_units_, total = (uint64_t)n_fp * bands_per_rank * world * band_units;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t in_band = i % band_units, b = i / band_units; // b = (frame-plane, local band, rank) of the destination
    const uint64_t fp = b / ((uint64_t)world * bands_per_rank);
    const uint32_t bb = (uint32_t)(b % ((uint64_t)world * bands_per_rank)); // the band inside its frame-plane
    const uint32_t rank = (uint32_t)rank_of_band((int)bb, (int)world), lb = bb / world;
    full[i] = __builtin_nontemporal_load(gathered + (((uint64_t)rank * n_fp + fp) * bands_per_rank + lb) * band_units + in_band);
  }
}
void launch_deinterleave(const void *gathered, void *full, uint32_t world, uint32_t n_fp, uint32_t bands_per_rank, uint32_t row_bytes,
                         hipStream_t s) {
  // unit of the copy: what a BAND (32 rows) of row_bytes and both buffers are aligned to — 16, 4 or (8-bit images of a width that is
  // not a multiple of 4) 1 byte
  const uint32_t band_bytes = (uint32_t)BAND * row_bytes;
  const uintptr_t al = (uintptr_t)gathered | (uintptr_t)full;
  const uint32_t unit = ((band_bytes & 15u) == 0 && (al & 15u) == 0) ? 16u : ((band_bytes & 3u) == 0 && (al & 3u) == 0) ? 4u : 1u;
  const uint32_t band_units = band_bytes / unit; // (k_deinterleave's row_units x BAND: it only ever uses the product)
  const uint64_t total = (uint64_t)n_fp * bands_per_rank * world * band_units;
  if (!total) return;
  const uint32_t grid = (uint32_t)std::min<uint64_t>((total + 255) / 256, 16384);
  if (unit == 16u)
    hipLaunchKernelGGL(k_deinterleave<u32x4>, dim3(grid), dim3(256), 0, s, (const u32x4 *)gathered, (u32x4 *)full, world, n_fp, bands_per_rank, band_units);
  else if (unit == 4u)
    hipLaunchKernelGGL(k_deinterleave<uint32_t>, dim3(grid), dim3(256), 0, s, (const uint32_t *)gathered, (uint32_t *)full, world, n_fp, bands_per_rank,
                       band_units);
  else
    hipLaunchKernelGGL(k_deinterleave<uint8_t>, dim3(grid), dim3(256), 0, s, (const uint8_t *)gathered, (uint8_t *)full, world, n_fp, bands_per_rank,
                       band_units);
}

// Exhaustive check of the short exact sequences above against the IEEE expansions: all 2^32 bit patterns.
// out[0] = operands tested on the fast path, out[1..3] = mismatches of rcp_rn / sqrt_rn / rsqrt2_rn (must be 0).
__global__ void k_verify_fastmath(unsigned long long *out) {
  unsigned long long bad0 = 0, bad1 = 0, bad2 = 0, n = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (uint32_t)i);
    const float a = rcp_rn(x), ra = 1.0f / x;
    bad0 += f2u_(a) != f2u_(ra);
    const float b = sqrt_rn(x), rb = __builtin_sqrtf(x);
    bad1 += f2u_(b) != f2u_(rb);
    const float c = rsqrt2_rn(x), rc = 1.0f / rb;
    bad2 += f2u_(c) != f2u_(rc);
    n += fast_range(x);
  }
  if (n) atomicAdd(&out[0], n);
  if (bad0) atomicAdd(&out[1], bad0);
  if (bad1) atomicAdd(&out[2], bad1);
  if (bad2) atomicAdd(&out[3], bad2);
}
void launch_verify_fastmath(unsigned long long *d_out, hipStream_t s) {
  hipLaunchKernelGGL(k_verify_fastmath, dim3(8192), dim3(256), 0, s, d_out);
}

// Companion check of div_by_rcp against the IEEE division: out[0] = pairs tested, out[1] = mismatches on pseudo-random
// (a, b) inside the documented ranges (uniform exponents, random mantissas, plus few-significant-bit operands whose
// quotients are exact or near-exact), out[2] = mismatches of texel / 255 over all 256 texels (must both be 0).
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull, z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull, z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ void k_verify_fastdiv(unsigned long long *out, uint32_t per_thread) {
  unsigned long long bad = 0, bad255 = 0, n = 0;
  uint64_t st = mix64(blockIdx.x * (uint64_t)blockDim.x + threadIdx.x);
  for (uint32_t i = 0; i < per_thread; ++i) {
    st = mix64(st);
    uint32_t ua = (uint32_t)st, ub = (uint32_t)(st >> 32);
    const uint32_t ea = 67u + ((ua >> 23) & 0xffu) % 121u, eb = 87u + ((ub >> 23) & 0xffu) % 81u;
    if (i & 1u) ua &= ~0x7ff000u, ub &= ~0x7fff00u; // few significant bits
    ua = (ua & 0x807fffffu) | (ea << 23), ub = (ub & 0x007fffffu) | (eb << 23);
    const float a = __builtin_bit_cast(float, ua), b = __builtin_bit_cast(float, ub);
    bad += !(div_num_ok(a) && div_den_ok(b)) || f2u_(div_by_rcp(a, b, rcp_core(b))) != f2u_(a / b);
    ++n;
  }
  if (blockIdx.x == 0) {
    FastMath fm;
    const float t = (float)threadIdx.x;
    bad255 += f2u_(fm.div255(t)) != f2u_(t / 255.0f);
    float q0, q1, q2;
    fm.div3(0.0f, t, 1.0f, 255.0f, q0, q1, q2); // +0 numerator stays +0
    bad255 += f2u_(q0) != 0u || f2u_(q1) != f2u_(t / 255.0f) || fm.bad;
  }
  atomicAdd(&out[0], n);
  if (bad) atomicAdd(&out[1], bad);
  if (bad255) atomicAdd(&out[2], bad255);
}
void launch_verify_fastdiv(unsigned long long *d_out, hipStream_t s) {
  hipLaunchKernelGGL(k_verify_fastdiv, dim3(8192), dim3(256), 0, s, d_out, 2048u);
}

// Check of len2d_fast (two binary64 Heron steps from the binary32 root) against len2d_f64 (the compiler's correctly rounded binary64
// square root) on pseudo-random pairs: out[0] = pairs, out[1] = results that differ although the flag was clear (must be 0),
// out[2] = flagged pairs among the random ones (must stay rare), out[3] = among the Pythagorean ones, out[4] = among the
// few-significant-bit ones (exact roots and exact ties are common there).  A quarter of
// the pairs each: random mantissas with independent exponents 2^-30 .. 2^40; the same with exponents at most two apart (both
// squares matter); few significant bits (exact and near-exact roots); scaled Pythagorean pairs (m^2 - n^2, 2 m n) whose root
// m^2 + n^2 is an odd number of up to 25 bits — exactly ON a binary32 rounding boundary when it exceeds 2^24 (those must be flagged
// or equal: the reference's tie goes to even).
__global__ void k_verify_fastlen(unsigned long long *out, uint32_t per_thread) {
  unsigned long long bad = 0, flagged = 0, flagged_p = 0, flagged_f = 0, n = 0;
  uint64_t st = mix64(0x5eedull + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x);
  for (uint32_t i = 0; i < per_thread; ++i) {
    st = mix64(st);
    uint32_t ua = (uint32_t)st, ub = (uint32_t)(st >> 32);
    const uint32_t mode = i & 3u;
    float a, b;
    if (mode == 3u) {
      const uint32_t m = 2u + (ua >> 8) % 4094u;           // 2 .. 4095
      uint32_t k = 1u + (ub >> 8) % (m - 1u);              // 1 .. m - 1
      if (((m ^ k) & 1u) == 0u) k = k > 1u ? k - 1u : 2u;  // opposite parity (m = 2: k = 1)
      if (k >= m) k = m - 1u;
      const int e = (int)(ua & 31u) - 16;
      a = __builtin_ldexpf((float)(m * m - k * k), e), b = __builtin_ldexpf((float)(2u * m * k), e); // (both exact)
      if (ub & 1u) a = -a;
    } else {
      const uint32_t ea = 97u + ((ua >> 23) & 0xffu) % 71u;
      uint32_t eb = 97u + ((ub >> 23) & 0xffu) % 71u;
      if (mode == 1u) eb = ea + ((ub >> 23) & 3u) - 1u;
      if (mode == 2u) ua &= ~0x3fffu << ((ub >> 28) & 7u), ub &= ~0xfffu << ((ua >> 28) & 7u);
      a = u2f_((ua & 0x807fffffu) | (ea << 23)), b = u2f_((ub & 0x807fffffu) | (eb << 23));
    }
    bool amb = false;
    uint32_t sb;
    const float x = len2d_fast(a, b, amb, sb), y = len2d_f64(a, b);
    ++n, bad += (!amb && f2u_(x) != f2u_(y)) ? 1 : 0;
    if (amb) (mode == 3u ? flagged_p : mode == 2u ? flagged_f : flagged) += 1;
  }
  atomicAdd(&out[0], n);
  if (bad) atomicAdd(&out[1], bad);
  if (flagged) atomicAdd(&out[2], flagged);
  if (flagged_p) atomicAdd(&out[3], flagged_p);
  if (flagged_f) atomicAdd(&out[4], flagged_f);
}
void launch_verify_fastlen(unsigned long long *d_out4, hipStream_t s) {
  hipLaunchKernelGGL(k_verify_fastlen, dim3(8192), dim3(256), 0, s, d_out4, 2048u);
}

// Exhaustive check of pow_fast against pow_cr: every binary32 x in [2^-40, 1] (+ the operands above 1 the clamped cosines can
// reach by rounding) at exponent p.  out[0] = operands, out[1] = results that differ although pow_fast did NOT flag them (must
// be 0), out[2] = flagged operands whose result is a normal binary32 >= 2^-120 (ambiguous roundings: must stay rare), out[3] =
// flagged operands with smaller results (the band 2^-151 .. 2^-120 is flagged wholesale; cosines that small are rare).
__global__ void k_verify_fastpow(unsigned long long *out, float p) {
  unsigned long long bad = 0, flagged = 0, flagged_small = 0, n = 0;
  const uint32_t lo = 0x2b800000u /* 2^-40 */, hi = 0x3f800010u /* 1 + 16 ulp */;
  for (uint64_t i = lo + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= hi; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (uint32_t)i);
    bool amb = false;
    const float a = pow_fast(x, p, amb), b = pow_cr(x, p);
    ++n, bad += (!amb && f2u_(a) != f2u_(b)) ? 1 : 0;
    if (amb) (b >= 0x1p-120f ? flagged : flagged_small) += 1;
  }
  if (blockIdx.x == 0 && threadIdx.x < 4) { // the special operands: ±0 give +0 unflagged; negative, inf and NaN must be flagged
    const float sp[4] = {0.0f, -0.0f, -1.0f, __builtin_inff()};
    bool amb = false;
    const float a = pow_fast(sp[threadIdx.x], p, amb);
    bad += threadIdx.x < 2 ? (amb || f2u_(a) != 0u) : !amb;
  }
  atomicAdd(&out[0], n);
  if (bad) atomicAdd(&out[1], bad);
  if (flagged) atomicAdd(&out[2], flagged);
  if (flagged_small) atomicAdd(&out[3], flagged_small);
}
void launch_verify_fastpow(unsigned long long *d_out3, float p, hipStream_t s) {
  hipLaunchKernelGGL(k_verify_fastpow, dim3(8192), dim3(256), 0, s, d_out3, p);
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
void launch_vertex(const DrawDesc *draws, uint32_t n_draws, uint32_t max_faces, srz_tri *tris, float *tri_pos, const FrameDesc *frames,
                   BBox *bbox_out, hipStream_t s) {
  if (n_draws == 0 || max_faces == 0) return;
  dim3 grid((max_faces + 255) / 256, n_draws);
  if (grid.x > 1024) grid.x = 1024;
  hipLaunchKernelGGL(k_vertex, grid, dim3(256), 0, s, draws, tris, tri_pos, frames, bbox_out);
}

void launch_chunks(const RenderArgs &a, int n_frames, uint32_t max_tris, hipStream_t s) {
  if (n_frames <= 0 || max_tris == 0) return;
  dim3 grid((max_tris + GROUP_TRIS - 1u) / GROUP_TRIS, n_frames); // one workgroup pass per group
  if (grid.x > 4096) grid.x = 4096;
  hipLaunchKernelGGL(k_chunks, grid, dim3(256), 0, s, a);
}

void launch_setup(const RenderArgs &a, int n_frames, uint32_t max_tris, bool stats, hipStream_t s) {
  if (n_frames <= 0 || max_tris == 0) return;
  dim3 grid((max_tris + GROUP_TRIS - 1u) / GROUP_TRIS, n_frames); // one workgroup pass per group
  if (grid.x > 4096) grid.x = 4096;
  BBox *bb = const_cast<BBox *>(a.bbox);
  if (stats)
    hipLaunchKernelGGL(k_setup<true>, grid, dim3(256), 0, s, a, bb);
  else
    hipLaunchKernelGGL(k_setup<false>, grid, dim3(256), 0, s, a, bb);
}

void launch_bin(const RenderArgs &a, int n_frames, uint32_t max_tris, hipStream_t s) {
  if (n_frames <= 0 || a.n_local_bands == 0) return;
  const uint32_t groups = ((uint32_t)n_frames + 7u) / 8u;
  static const int env = getenv("SRZ_BIN_WAVES") ? atoi(getenv("SRZ_BIN_WAVES")) : 0;
  // (measured: 4 waves best at 5.9 k triangles, 8 at 94 k; a job with fewer band workgroups than CUs is pure latency: 8)
  const int waves = env ? env : ((max_tris <= 32768u && (uint32_t)n_frames * a.n_local_bands > 256u) ? 4 : BIN_MAX_WAVES);
  const size_t lds = sizeof(uint32_t) * (3u * (size_t)a.tiles_x + 64u * waves + 4u + BIN_STAGE + 2u + 128u * waves * bin_keep((uint32_t)waves));
  hipLaunchKernelGGL(k_bin, dim3(groups * 8u * a.n_local_bands), dim3(64 * waves), lds, s, a);
}

// End of a measurement render (one thread; srz_device.h, ClearCtl): file the sample, move to the next grid or decide.
__global__ void k_clear_tune(ClearCtl *c, uint32_t *h_wgs) {
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  if (c->done) return;
  if (c->pos > 0 && c->n[c->cur] < 4u) c->t[c->cur][c->n[c->cur]++] = (float)(now - c->last);
  c->last = now;
  if (++c->pos < (uint32_t)CLEAR_TUNE_BLOCK) return;
  c->pos = 0; // a block has ended
  int next = -1;
  if (c->phase == 0u) {
    // A larger grid that is already more than 5 % behind the best smaller one ends the first pass: the step is the clear, and the grids
    // beyond it would only be worse (config 2: 96 / 128 / 256 workgroups take 0.78 / 0.83 / 1.04 ms) — they are not tried.
    bool worse = false;
    if (c->cur > 0u) {
      float best = 3.0e38f;
      for (uint32_t i = 0; i < c->cur; ++i) best = fminf(best, 0.5f * (c->t[i][0] + c->t[i][1]));
      worse = 0.5f * (c->t[c->cur][0] + c->t[c->cur][1]) > 1.05f * best;
    }
    const int tried = (int)c->cur + 1;
    if (tried < CLEAR_CANDS && !worse) next = tried;
    else { // the first pass is over: who stays?
      // Dropped: a grid that was tried AFTER the best one and is still more than 5 % behind it.  (The first renders after idle, or into
      // buffers touched for the first time, run slower: the first pass favours the later grids, so an earlier one that looks worse
      // stays in, and the mirrored pass settles it.)
      float m[CLEAR_CANDS];
      int ib = 0;
      for (int i = 0; i < tried; ++i) {
        m[i] = 0.5f * (c->t[i][0] + c->t[i][1]);
        if (m[i] < m[ib]) ib = i;
      }
      for (int i = tried; i < CLEAR_CANDS; ++i) m[i] = 0.f;
      uint32_t alive = 0;
      for (int i = 0; i < tried; ++i)
        if (i <= ib || m[i] <= 1.05f * m[ib]) alive |= 1u << i;
      c->alive = alive, c->phase = 1u;
      for (int i = 0; i < CLEAR_CANDS; ++i) c->score[i] = (alive >> i & 1u) ? m[i] : 0.f;
      if (__popc(alive) > 1) next = 31 - __clz((int)alive); // (the mirrored pass starts with the grid in use)
    }
  } else {
    const uint32_t lower = c->alive & ((1u << c->cur) - 1u);
    if (lower) next = 31 - __clz((int)lower);
    else // the mirrored pass is over: medians of four
      for (int i = 0; i < CLEAR_CANDS; ++i)
        if (c->alive >> i & 1u) {
          float *t = c->t[i];
          for (int x = 1; x < 4; ++x)
            for (int y = x; y > 0 && t[y] < t[y - 1]; --y) { const float u = t[y]; t[y] = t[y - 1], t[y - 1] = u; }
          c->score[i] = 0.5f * (t[1] + t[2]);
        }
  }
  if (next >= 0) {
    c->cur = (uint32_t)next, c->wgs = CLEAR_CAND[next];
    return;
  }
  int arg = -1;
  for (int i = 0; i < CLEAR_CANDS; ++i)
    if ((c->alive >> i & 1u) && (arg < 0 || c->score[i] < c->score[arg])) arg = i;
  c->wgs = CLEAR_CAND[arg < 0 ? 0 : arg], c->done = 1u;
  __hip_atomic_store(h_wgs, c->wgs, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); // (mapped host memory: the host launches that grid from now on)
}

void launch_clear_tune(ClearCtl *ctl, uint32_t *h_wgs, hipStream_t s) { hipLaunchKernelGGL(k_clear_tune, dim3(1), dim3(1), 0, s, ctl, h_wgs); }

void launch_clear(const RenderArgs &a, uint32_t max_tiles, bool beside_raster, hipStream_t s, uint32_t wgs) {
  if (max_tiles == 0) return;
  // Beside k_raster/k_shade the clear is THROTTLED by its grid size, so that the stores spread over the whole pipeline instead of starving
  // the rasteriser's loads and the shader's register file.  The best grid depends on what runs beside it (config 2: 96 workgroups, 0.738
  // of the roofline against 0.727 at 80 and 0.681 at 128; config 4: 0.406 at 96, 0.438 at 256), so it is measured per set (srz_device.h,
  // ClearCtl: while that goes on the grid is CLEAR_GRID_MAX and a.clear_wgs_dev says how many of it work).  Work items are planes of bands.
  const uint32_t n_items = a.n_frames * a.n_local_bands * 4u;
  const uint32_t cap = !beside_raster ? 2048u : (wgs ? wgs : 96u);
  (void)max_tiles;
  static const uint32_t env_thr = getenv("SRZ_CLEAR_THREADS") ? (uint32_t)atoi(getenv("SRZ_CLEAR_THREADS")) : 0u; // (A/B: 64 / 128 / 256)
  const uint32_t thr = (beside_raster && env_thr) ? env_thr : 256u;
  // (round 6, same box: the clear's stores as nt / sc1 nt / sc0 sc1 nt / sc1 / plain give config 4 3.10 / 3.22 / 3.26 / 3.29 / 3.29 ms per
  // step and config 5 5.25 / 5.48 / 5.44 / 5.71 / 5.62: nt, as round 3 found on config 2)
  hipLaunchKernelGGL(k_clear, dim3(n_items < cap ? n_items : cap), dim3(thr), 0, s, a);
}

void launch_shade(const RenderArgs &a, uint32_t max_tiles, bool stats, uint32_t fast_mask, bool any_generic, bool approx, hipStream_t s) {
  if (max_tiles == 0) return;
  // One stream: a LARGE grid (a tile or two per workgroup: in-order hand-out, one-tile tail).  Renders interleaved on several
  // streams (a.other_streams): a grid about as large as the machine's resident capacity, the workgroups striding through the
  // lists — measured, 2 x 128 frames of 1024^2 on two streams: 16384 workgroups 1.10 ms per batch, 4096 1.08, 2560 1.07,
  // 1280 1.065 (and on ONE stream 0.73 / 0.78 / — / 0.78 ms for k_shade alone)
  static const uint32_t env_grid = getenv("SRZ_SHADE_GRID") ? (uint32_t)atoi(getenv("SRZ_SHADE_GRID")) : 0u;
  const uint32_t gcap = env_grid ? env_grid : (a.other_streams ? 2048u : 16384u);
  dim3 grid((std::min(max_tiles, gcap) + 7u) & ~7u); // (a multiple of 8: workgroup b serves list b % 8)
  // SRZ_SHADE_LDS_PAD (diagnostic): bytes of unused dynamic LDS per workgroup — 1024 caps k_shade at five workgroups per CU, 6144 at
  // four (26.3 KB static each, 160 KB per CU): how much room the other lane's k_raster waves find on a CU k_shade has filled
  static const uint32_t pad = getenv("SRZ_SHADE_LDS_PAD") ? (uint32_t)atoi(getenv("SRZ_SHADE_LDS_PAD")) : 0u;
  if (stats) { // (counting runs shade every frame with the generic build: force_generic)
    hipLaunchKernelGGL((k_shade<true, 0>), grid, dim3(256), pad, s, a);
    return;
  }
  // one FAST build per (light count, with / without BUMP or DISPLACEMENT batches) some frame of the set has:
  // fast_mask bit NL = plain, bit 8 + NL = with them
  if (approx) { // the tolerance mode's builds (classify_frames sets only the plain bits for the frames they shade)
    if (fast_mask & 2u) hipLaunchKernelGGL((k_shade<false, 1, false, true>), grid, dim3(256), pad, s, a);
    if (fast_mask & 4u) hipLaunchKernelGGL((k_shade<false, 2, false, true>), grid, dim3(256), pad, s, a);
    if (fast_mask & 8u) hipLaunchKernelGGL((k_shade<false, 3, false, true>), grid, dim3(256), pad, s, a);
    if (fast_mask & 16u) hipLaunchKernelGGL((k_shade<false, 4, false, true>), grid, dim3(256), pad, s, a);
    fast_mask = 0;
  }
  if (fast_mask & 2u) hipLaunchKernelGGL((k_shade<false, 1, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 4u) hipLaunchKernelGGL((k_shade<false, 2, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 8u) hipLaunchKernelGGL((k_shade<false, 3, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 16u) hipLaunchKernelGGL((k_shade<false, 4, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x200u) hipLaunchKernelGGL((k_shade<false, 1, true>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x400u) hipLaunchKernelGGL((k_shade<false, 2, true>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x800u) hipLaunchKernelGGL((k_shade<false, 3, true>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x1000u) hipLaunchKernelGGL((k_shade<false, 4, true>), grid, dim3(256), pad, s, a);
  // ... and per light count with a non-integer exponent: bit 16 + NL
  if (fast_mask & 0x20000u) hipLaunchKernelGGL((k_shade<false, -1, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x40000u) hipLaunchKernelGGL((k_shade<false, -2, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x80000u) hipLaunchKernelGGL((k_shade<false, -3, false>), grid, dim3(256), pad, s, a);
  if (fast_mask & 0x100000u) hipLaunchKernelGGL((k_shade<false, -4, false>), grid, dim3(256), pad, s, a);
  // the generic build also serves the tiles the FAST builds hand back: when no frame is generic that is normally
  // nothing, and a small grid does
  dim3 ggrid(any_generic ? grid.x : (grid.x < 128u ? grid.x : 128u));
  hipLaunchKernelGGL((k_shade<false, 0>), ggrid, dim3(256), pad, s, a);
}

bool raster_four_waves(const RenderArgs &a) { return a.n_frames * a.n_local_bands * a.tiles_x <= 4096u; }

void launch_raster(const RenderArgs &a, int n_frames, bool stats, hipStream_t s) {
  if (n_frames <= 0 || a.n_local_bands == 0) return;
  static bool once = false;
  if (!once && getenv("SRZ_DEBUG")) {
    once = true;
    int nb = 0, ns = 0;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_raster<1>, 64, 0);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&ns, (k_shade<false, 2>), 256, 0);
    fprintf(stderr, "[srz] occupancy: k_raster %d waves/CU, k_shade %d WGs/CU\n", nb, ns);
  }
  const uint32_t groups = ((uint32_t)n_frames + 7u) / 8u;
  const uint32_t tiles = groups * 8u * a.n_local_bands * a.tiles_x;
  // a few frames: four waves per tile (the frame is as slow as its heaviest tile); batches: one wave per tile
  if (raster_four_waves(a))
    hipLaunchKernelGGL(k_raster<4>, dim3(tiles), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(k_raster<1>, dim3(tiles), dim3(64), 0, s, a);
  // the ordered rasteriser for whatever k_raster listed (normally nothing: its waves read a zero and leave)
  // (a large grid whenever the ordered rasteriser is known to get every touched tile: counting runs, SRZ_ORDERED_RASTER on the
  // render or on some frame — a.any_ordered, set by the host)
  const uint32_t slow_grid = (stats || a.force_ordered || a.any_ordered) ? (tiles < 4096u ? tiles : 4096u) : (tiles < 256u ? tiles : 256u);
  if (stats)
    hipLaunchKernelGGL(k_raster_slow<true>, dim3(slow_grid), dim3(64), 0, s, a);
  else
    hipLaunchKernelGGL(k_raster_slow<false>, dim3(slow_grid), dim3(64), 0, s, a);
}

void launch_tex_convert(const uint8_t *d_bgr, int w, int h, int row_stride, uint32_t *d_bgrx, hipStream_t s) {
  int n = w * h;
  hipLaunchKernelGGL(k_tex_convert, dim3((n + 255) / 256), dim3(256), 0, s, d_bgr, w, h, row_stride, d_bgrx);
}

} // namespace srz
