/*
 * srz_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A scalar, single-threaded (optionally OpenMP row-band-parallel) restatement, in plain C with exact
 * IEEE-754 binary32 operations, of the triangle rasterization + fragment-shading path of
 * Liupeter01/Software-Rasterizer (all file:line citations are into that repository):
 *
 *   src/Object.cpp:23-31, src/Scene.cpp:263-294,314-335      model / view / projection / NDC matrices
 *   src/Scene.cpp:903-964, src/Tools.cpp:74-76                vertex stage (loadTriangleStream)
 *   src/Triangle.cpp:147-151,243-257                          face normal, bounding box
 *   src/Rasterizer.cpp:183-499                                traversal, coverage, z-test, write-back
 *   src/Tools.cpp:13-24,94-168,228-232                        normal / uv interpolation, normalizedToRGB
 *   src/Shader.cpp:7-12,122-386,510-594, include/shader/Shader.hpp:104-229   fragment shaders
 *   include/loader/TextureLoader.hpp:26-117, src/TextureLoader.cpp:14-31     texel fetch
 *   src/Render.cpp:31-55,61-62                                clear, 8-bit resolve
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / reported baseline — never as the product path.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this path, and its sources
 * cannot be compiled in this image (all dependencies — glm, oneTBB, OpenCV, spdlog, tinyobjloader, simde —
 * are empty un-vendored submodules; the x86 path needs SVML _mm256_pow_ps).  This restatement is pinned
 * only by reading the reference source and by hand-derived known-answer tests (tests/test_oracle_kat.py).
 * Third-party arithmetic (glm matrix functions, OpenCV nearest-texel / convertTo, SSE min/max/cvt
 * semantics) is restated from the libraries' published behaviour.
 *
 * Where the reference uses approximate instructions the oracle uses the exact operation in the same place:
 *   _mm256_rcp_ps(x)  -> 1.0f/x  (then the reference's multiply)     _mm256_pow_ps -> powf
 * and keeps the reference's fused/unfused structure: _mm256_fmadd_ps/_fmsub_ps -> fmaf, plain C++
 * expressions -> separately rounded ops (build with -ffp-contract=off).
 *
 * The reference is two renderers interleaved by column: for each triangle the first
 * 8*floor(bboxW/8) columns of its bounding box go through processFragByAVX2 ("V" semantics below), the
 * remaining <=7 columns through processFragByScalar ("S" semantics).  Both are reproduced per pixel.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/srz.h"

#define ORC_MAX_TEX 64

typedef struct {
  uint8_t *bgr;
  int w, h;
} orc_tex;
static orc_tex g_tex[ORC_MAX_TEX];

/* ------------------------------------------------------------------------------------------------
 * SSE operand-order semantics: max_ps(a,b) = a > b ? a : b (b if unordered)
 * ---------------------------------------------------------------------------------------------- */
static inline float sse_max(float a, float b) { return a > b ? a : b; }
static inline float sse_min(float a, float b) { return a < b ? a : b; }
/* std::max(a,b) = (a < b) ? b : a ; std::clamp(v,lo,hi) = (v < lo) ? lo : (hi < v) ? hi : v */
static inline float std_max(float a, float b) { return (a < b) ? b : a; }
static inline float std_clamp(float v, float lo, float hi) { return (v < lo) ? lo : (hi < v) ? hi : v; }
static inline float fmsubf(float a, float b, float c) { return fmaf(a, b, -c); }
/* _mm256_pow_ps / std::pow(float,float): the exact operation = pow evaluated in binary64, rounded once to binary32 */
static inline float pow_cr(float x, float p) { return (float)pow((double)x, (double)p); }

/* ================================================================================================
 * glm restatement (column-major m[col*4+row]); GLM 0.9.9 expression order
 * ============================================================================================== */
static void m4_identity(float *m) {
  memset(m, 0, 16 * sizeof(float));
  m[0] = m[5] = m[10] = m[15] = 1.0f;
}

/* glm operator*(mat4,mat4): Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3] (left to right) */
void orc_m4_mul(const float *a, const float *b, float *out) {
  float r[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) {
      float t = a[0 * 4 + i] * b[j * 4 + 0];
      t = t + a[1 * 4 + i] * b[j * 4 + 1];
      t = t + a[2 * 4 + i] * b[j * 4 + 2];
      t = t + a[3 * 4 + i] * b[j * 4 + 3];
      r[j * 4 + i] = t;
    }
  memcpy(out, r, sizeof r);
}

/* glm operator*(mat4,vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3) */
void orc_m4_mulv(const float *m, const float *v, float *out) {
  float r[4];
  for (int i = 0; i < 4; ++i) {
    float add0 = m[0 * 4 + i] * v[0] + m[1 * 4 + i] * v[1];
    float add1 = m[2 * 4 + i] * v[2] + m[3 * 4 + i] * v[3];
    r[i] = add0 + add1;
  }
  memcpy(out, r, sizeof r);
}

void orc_m4_transpose(const float *m, float *out) {
  float r[16];
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) r[rr * 4 + c] = m[c * 4 + rr];
  memcpy(out, r, sizeof r);
}

/* glm::inverse(mat4) — cofactor expansion, GLM's compute_inverse<4,4> */
void orc_m4_inverse(const float *mm, float *out) {
#define M(c, r) mm[(c) * 4 + (r)]
  float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
  float c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
  float c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
  float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
  float c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
  float c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
  float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
  float c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
  float c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
  float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
  float c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
  float c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
  float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
  float c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
  float c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
  float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
  float c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
  float c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
  float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
  float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  float v0[4] = {M(1, 0), M(0, 0), M(0, 0), M(0, 0)}, v1[4] = {M(1, 1), M(0, 1), M(0, 1), M(0, 1)};
  float v2[4] = {M(1, 2), M(0, 2), M(0, 2), M(0, 2)}, v3[4] = {M(1, 3), M(0, 3), M(0, 3), M(0, 3)};
  static const float sa[4] = {+1, -1, +1, -1}, sb[4] = {-1, +1, -1, +1};
  float inv[16];
  for (int i = 0; i < 4; ++i) {
    float i0 = v1[i] * f0[i] - v2[i] * f1[i] + v3[i] * f2[i];
    float i1 = v0[i] * f0[i] - v2[i] * f3[i] + v3[i] * f4[i];
    float i2 = v0[i] * f1[i] - v1[i] * f3[i] + v3[i] * f5[i];
    float i3 = v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i];
    inv[0 * 4 + i] = i0 * sa[i];
    inv[1 * 4 + i] = i1 * sb[i];
    inv[2 * 4 + i] = i2 * sa[i];
    inv[3 * 4 + i] = i3 * sb[i];
  }
  float d0 = M(0, 0) * inv[0 * 4 + 0], d1 = M(0, 1) * inv[1 * 4 + 0];
  float d2 = M(0, 2) * inv[2 * 4 + 0], d3 = M(0, 3) * inv[3 * 4 + 0];
  float det = (d0 + d1) + (d2 + d3);
  float ood = 1.0f / det;
  for (int i = 0; i < 16; ++i) out[i] = inv[i] * ood;
#undef M
}

static inline float v3_dot(const float *a, const float *b) {
  float tx = a[0] * b[0], ty = a[1] * b[1], tz = a[2] * b[2];
  return tx + ty + tz; /* glm compute_dot<vec3>: tmp.x + tmp.y + tmp.z */
}
static inline void v3_cross(const float *x, const float *y, float *o) {
  float r0 = x[1] * y[2] - y[1] * x[2];
  float r1 = x[2] * y[0] - y[2] * x[0];
  float r2 = x[0] * y[1] - y[0] * x[1];
  o[0] = r0, o[1] = r1, o[2] = r2;
}
/* glm::normalize(v) = v * inversesqrt(dot(v,v)), inversesqrt(x) = 1/sqrt(x) */
static inline void v3_normalize(const float *v, float *o) {
  float is = 1.0f / sqrtf(v3_dot(v, v));
  o[0] = v[0] * is, o[1] = v[1] * is, o[2] = v[2] * is;
}

/* Object::updateModelMatrix (src/Object.cpp:23-31): M = T * R(radians(angle),axis) * S */
void orc_model_matrix(const float *axis, float angle_deg, const float *t, const float *s, float *out) {
  float T[16], R[16], S[16], TR[16];
  m4_identity(T);
  T[12] = t[0], T[13] = t[1], T[14] = t[2];
  float a = angle_deg * 0.01745329251994329576923690768489f; /* glm::radians */
  float c = cosf(a), sn = sinf(a);
  float ax[3];
  v3_normalize(axis, ax);
  float tmp[3] = {(1.0f - c) * ax[0], (1.0f - c) * ax[1], (1.0f - c) * ax[2]};
  m4_identity(R);
  R[0 * 4 + 0] = c + tmp[0] * ax[0];
  R[0 * 4 + 1] = tmp[0] * ax[1] + sn * ax[2];
  R[0 * 4 + 2] = tmp[0] * ax[2] - sn * ax[1];
  R[1 * 4 + 0] = tmp[1] * ax[0] - sn * ax[2];
  R[1 * 4 + 1] = c + tmp[1] * ax[1];
  R[1 * 4 + 2] = tmp[1] * ax[2] + sn * ax[0];
  R[2 * 4 + 0] = tmp[2] * ax[0] + sn * ax[1];
  R[2 * 4 + 1] = tmp[2] * ax[1] - sn * ax[0];
  R[2 * 4 + 2] = c + tmp[2] * ax[2];
  m4_identity(S);
  S[0] = s[0], S[5] = s[1], S[10] = s[2];
  orc_m4_mul(T, R, TR);
  orc_m4_mul(TR, S, out);
}

/* glm::lookAtLH (src/Scene.cpp:270) */
void orc_look_at_lh(const float *eye, const float *center, const float *up, float *out) {
  float d[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
  float f[3], s[3], u[3], c[3];
  v3_normalize(d, f);
  v3_cross(up, f, c);
  v3_normalize(c, s);
  v3_cross(f, s, u);
  m4_identity(out);
  out[0 * 4 + 0] = s[0], out[1 * 4 + 0] = s[1], out[2 * 4 + 0] = s[2];
  out[0 * 4 + 1] = u[0], out[1 * 4 + 1] = u[1], out[2 * 4 + 1] = u[2];
  out[0 * 4 + 2] = f[0], out[1 * 4 + 2] = f[1], out[2 * 4 + 2] = f[2];
  out[3 * 4 + 0] = -v3_dot(s, eye);
  out[3 * 4 + 1] = -v3_dot(u, eye);
  out[3 * 4 + 2] = -v3_dot(f, eye);
}

/* glm::perspectiveLH_NO (src/Scene.cpp:293) — NB the reference passes fovy=45.0f into this RADIANS api */
void orc_perspective_lh_no(float fovy, float aspect, float zn, float zf, float *out) {
  float th = tanf(fovy / 2.0f);
  memset(out, 0, 16 * sizeof(float));
  out[0 * 4 + 0] = 1.0f / (aspect * th);
  out[1 * 4 + 1] = 1.0f / th;
  out[2 * 4 + 2] = (zf + zn) / (zf - zn);
  out[2 * 4 + 3] = 1.0f;
  out[3 * 4 + 2] = -(2.0f * zf * zn) / (zf - zn);
}

/* Scene::setNDCMatrix (src/Scene.cpp:314-335) */
void orc_ndc_matrix(int width, int height, float *out) {
  float aspect = (float)width / (float)height;
  m4_identity(out);
  out[0 * 4 + 0] = width / 2.0f * aspect;
  out[1 * 4 + 1] = height / 2.0f;
  out[3 * 4 + 0] = width / 2.0f;
  out[3 * 4 + 1] = height / 2.0f;
}

/* ================================================================================================
 * Vertex stage = Scene::loadTriangleStream for one mesh (src/Scene.cpp:917-961)
 *   verts: nV * 8 floats (pos3, nrm3, uv2) ; faces: nF * 3 uint32 ; out: nF srz_tri
 *   zscale/zoffset = (far-near)/2, (far+near)/2 (src/Scene.cpp:279-280)
 * Triangle order = face order (documented deviation from the racy concurrent emplace_back, :956).
 * ============================================================================================== */
void orc_vertex_stage(const float *verts, const uint32_t *faces, uint32_t n_faces, const float *model,
                      const float *view, const float *proj, const float *ndc, float zscale, float zoffset,
                      srz_tri *out) {
  float np[16], npv[16], mvp[16], inv[16], nm[16];
  orc_m4_mul(ndc, proj, np);
  orc_m4_mul(np, view, npv);
  orc_m4_mul(npv, model, mvp); /* NDC_MVP = ndc * P * V * M (:922) */
  orc_m4_inverse(model, inv);
  orc_m4_transpose(inv, nm); /* Normal_M (:923) */
  for (uint32_t f = 0; f < n_faces; ++f) {
    for (int k = 0; k < 3; ++k) {
      const float *v = verts + (size_t)faces[f * 3 + k] * 8;
      float p4[4] = {v[0], v[1], v[2], 1.0f}, n4[4] = {v[3], v[4], v[5], 1.0f}, r[4];
      orc_m4_mulv(mvp, p4, r);
      out[f].pos[k][0] = r[0] / r[3]; /* Tools::to_vec3 (src/Tools.cpp:74-76) */
      out[f].pos[k][1] = r[1] / r[3];
      float z = r[2] / r[3];
      out[f].pos[k][2] = z * zscale + zoffset; /* (:938) */
      orc_m4_mulv(nm, n4, r);                   /* w = 1, then divided by the resulting w (:939) */
      out[f].nrm[k][0] = r[0] / r[3];
      out[f].nrm[k][1] = r[1] / r[3];
      out[f].nrm[k][2] = r[2] / r[3];
      out[f].uv[k][0] = v[6];
      out[f].uv[k][1] = v[7];
    }
  }
}

/* ================================================================================================
 * Per-triangle prologue: bbox (src/Triangle.cpp:243-257) and backface test (src/Rasterizer.cpp:203)
 * ============================================================================================== */
typedef struct {
  long long sx, sy, ex, ey;
} orc_box;

static inline float min3(float a, float b, float c) {
  float m = a; /* std::min({a,b,c}): keeps the first minimum */
  if (b < m) m = b;
  if (c < m) m = c;
  return m;
}
static inline float max3(float a, float b, float c) {
  float m = a;
  if (m < b) m = b;
  if (m < c) m = c;
  return m;
}
static inline long long clampll(long long v, long long lo, long long hi) { return v < lo ? lo : (hi < v ? hi : v); }

/* returns 0 if any coordinate is non-finite (the reference's float->long long cast is UB there;
 * documented deviation: such triangles are dropped by oracle and GPU alike) */
static int tri_box(const srz_tri *t, int W, int H, orc_box *b) {
  for (int k = 0; k < 3; ++k)
    for (int c = 0; c < 3; ++c)
      if (!isfinite(t->pos[k][c])) return 0;
  float mnx = min3(t->pos[0][0], t->pos[1][0], t->pos[2][0]), mxx = max3(t->pos[0][0], t->pos[1][0], t->pos[2][0]);
  float mny = min3(t->pos[0][1], t->pos[1][1], t->pos[2][1]), mxy = max3(t->pos[0][1], t->pos[1][1], t->pos[2][1]);
  /* keep the float->integer cast defined: anything beyond +-2^40 clamps the same way */
  const float BIG = 1099511627776.0f;
  mnx = std_clamp(mnx, -BIG, BIG), mxx = std_clamp(mxx, -BIG, BIG);
  mny = std_clamp(mny, -BIG, BIG), mxy = std_clamp(mxy, -BIG, BIG);
  b->sx = clampll((long long)mnx, 0, W - 1); /* truncation toward zero, then clamp */
  b->sy = clampll((long long)mny, 0, H - 1);
  b->ex = clampll((long long)mxx, 0, W - 1);
  b->ey = clampll((long long)mxy, 0, H - 1);
  return 1;
}

/* glm::dot(normalize(cross(B-A, C-A)), eye) > 0 → culled */
static int tri_culled(const srz_tri *t, const float *eye) {
  float e1[3], e2[3], c[3], n[3];
  for (int i = 0; i < 3; ++i) e1[i] = t->pos[1][i] - t->pos[0][i], e2[i] = t->pos[2][i] - t->pos[0][i];
  v3_cross(e1, e2, c);
  v3_normalize(c, n);
  return v3_dot(n, eye) > 0.0f;
}

/* ================================================================================================
 * Fragment stage — "V" semantics = processFragByAVX2 (src/Rasterizer.cpp:268-407)
 * ============================================================================================== */
typedef struct {
  const srz_frame *fr;
  const srz_batch *batch;
  const orc_tex *tex; /* may be NULL */
  float texw, texh;   /* Shader::width_256 / height_256 (1x1 dummy when no texture) */
} orc_shade_ctx;

/* NormalSIMD::normalized (src/Tools.cpp:13-24) */
static inline void v_normalized(float *x, float *y, float *z) {
  float len = sqrtf(fmaf(*x, *x, fmaf(*y, *y, (*z) * (*z))));
  if (len > 0.0f) {
    float inv = 1.0f / len; /* rcp_ps */
    *x = *x * inv, *y = *y * inv, *z = *z * inv;
  } else {
    *x = *y = *z = 0.0f; /* blendv(zero, …, mask) */
  }
}

/* _mm256_cvtps_epi32: round to nearest even (default MXCSR); out of range -> 0x80000000 */
static inline int32_t cvtps_epi32(float f) {
  if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN;
  return (int32_t)lrintf(f);
}

/* BlinnPhong<__m256> (include/shader/Shader.hpp:104-229), one light */
static void v_blinn_phong(const float n[3], const float ka[3], const float kd[3], const float ks[3],
                          const float cam[3], const srz_light *L, const float P[3], float p, float out[3]) {
  float lx = L->pos[0] - P[0], ly = L->pos[1] - P[1], lz = L->pos[2] - P[2];
  float att = 1.0f / sqrtf(fmaf(lx, lx, ly * ly)); /* rcp(sqrt(x^2+y^2)) (:131-132) */
  float dist[3] = {L->intensity[0] * att, L->intensity[1] * att, L->intensity[2] * att};
  float hx = lx + (cam[0] - P[0]), hy = ly + (cam[1] - P[1]), hz = lz + (cam[2] - P[2]);
  v_normalized(&hx, &hy, &hz);
  float nlx = lx, nly = ly, nlz = lz;
  v_normalized(&nlx, &nly, &nlz);
  float cosA = sse_max(0.0f, fmaf(nlx, n[0], fmaf(nly, n[1], nlz * n[2])));          /* diffuse (:187-192) */
  float cosT = pow_cr(sse_max(0.0f, fmaf(hx, n[0], fmaf(hy, n[1], hz * n[2]))), p);     /* specular (:195-201) */
  for (int c = 0; c < 3; ++c) {
    float kd_dist = dist[c] * kd[c], ks_dist = dist[c] * ks[c];
    out[c] = kd[c] * fmaf(ka[c], L->intensity[c], fmaf(kd_dist, cosA, ks_dist * cosT)); /* (:212-229) */
  }
}

/* Shader::applyFragmentShader SIMD overload + simd_*_fragment_shader_impl (src/Shader.cpp:128-386).
 * n is already interpolated+normalised; (u,v) raw interpolated; P=(x,y,z). out = colour in [0,255] float */
static void v_shade(const orc_shade_ctx *sc, const float P[3], const float n[3], float u, float v, float out[3]) {
  const srz_frame *fr = sc->fr;
  /* prologue (src/Shader.cpp:134-140) — done for every shader type */
  u = u * sc->texw, v = v * sc->texh;
  u = sse_max(0.0f, sse_min(u, sc->texw - 1.0f));
  v = sse_max(0.0f, sse_min(v, sc->texh - 1.0f));
  float col[3] = {1.0f, 1.0f, 1.0f}; /* ColorSIMD() (src/Tools.cpp:26-28) */
  switch (sc->batch->shader) {
  case SRZ_SHADER_NORMAL: /* (src/Shader.cpp:157-174) */
    for (int c = 0; c < 3; ++c) col[c] = (n[c] + 1.0f) * 0.5f;
    break;
  case SRZ_SHADER_TEXTURE:
  case SRZ_SHADER_PHONG: {
    float kd[3] = {1.0f, 1.0f, 1.0f}; /* PHONG: kd = incoming colour = (1,1,1) (src/Shader.cpp:326-328) */
    if (sc->batch->shader == SRZ_SHADER_TEXTURE) {
      /* TextureLoader::getTextureColor<__m256> (include/loader/TextureLoader.hpp:51-101) */
      int32_t xi = cvtps_epi32(u), yi = cvtps_epi32(v);
      const uint8_t *px = sc->tex->bgr + ((size_t)yi * sc->tex->w + xi) * 3;
      const float inv255 = 1.0f / 255.0f; /* rcp_ps(255) */
      kd[0] = (float)px[0] * inv255, kd[1] = (float)px[1] * inv255, kd[2] = (float)px[2] * inv255;
    }
    col[0] = col[1] = col[2] = 0.0f;
    for (uint32_t l = 0; l < fr->n_lights; ++l) {
      float o[3];
      v_blinn_phong(n, fr->ka, kd, fr->ks, fr->eye, &fr->lights[l], P, fr->p, o);
      col[0] = col[0] + o[0], col[1] = col[1] + o[1], col[2] = col[2] + o[2];
    }
    break;
  }
  default: /* DISPLACEMENT / BUMP SIMD versions are empty stubs (src/Shader.cpp:388-444): colour stays (1,1,1) */
    break;
  }
  for (int c = 0; c < 3; ++c) out[c] = sse_min(sse_max(col[c], 0.0f), 1.0f) * 255.0f;
}

/* ================================================================================================
 * Fragment stage — "S" semantics = processFragByScalar (src/Rasterizer.cpp:456-499)
 * ============================================================================================== */
/* TextureLoader::getTextureColor(vec2) (src/TextureLoader.cpp:14-31) */
static void s_texel(const orc_tex *tex, float u, float v, float out[3]) {
  float cu = std_clamp(u, 0.0f, 1.0f), cv = std_clamp(v, 0.0f, 1.0f); /* glm::clamp = min(max(x,lo),hi) — same for non-NaN */
  float fx = cu * (float)tex->w, fy = cv * (float)tex->h;
  int x = (int)fx, y = (int)fy; /* truncation */
  if (x < 0 || x >= tex->w || y < 0 || y >= tex->h) {
    out[0] = out[1] = out[2] = 0.0f; /* u or v == 1.0 → black */
    return;
  }
  const uint8_t *px = tex->bgr + ((size_t)y * tex->w + x) * 3;
  out[0] = px[0] / 255.0f, out[1] = px[1] / 255.0f, out[2] = px[2] / 255.0f;
}

/* Shader::BlinnPhong scalar (src/Shader.cpp:510-543) */
static void s_blinn_phong(const float cam[3], const float P[3], const float nrm_in[3], const float color[3],
                          const srz_light *L, const float ka[3], const float kd[3], const float ks[3], float p,
                          float out[3]) {
  float n[3];
  v3_normalize(nrm_in, n);
  float ld[3] = {L->pos[0] - P[0], L->pos[1] - P[1], L->pos[2] - P[2]};
  /* std::sqrt(std::pow(float,2) + std::pow(float,2)) is evaluated in double, then narrowed (:519-521) */
  double dx = (double)(L->pos[0] - P[0]), dy = (double)(L->pos[1] - P[1]);
  float distanceSquared = (float)sqrt(dx * dx + dy * dy);
  float dist[3] = {L->intensity[0] / distanceSquared, L->intensity[1] / distanceSquared,
                   L->intensity[2] / distanceSquared};
  float nl[3];
  v3_normalize(ld, nl);
  float cosTheta = std_max(0.0f, v3_dot(n, nl));
  float v[3] = {cam[0] - P[0], cam[1] - P[1], cam[2] - P[2]};
  float hv[3] = {ld[0] + v[0], ld[1] + v[1], ld[2] + v[2]}, h[3];
  v3_normalize(hv, h);
  float cosAlpha = std_max(0.0f, v3_dot(n, h));
  float pw = pow_cr(cosAlpha, p);
  for (int c = 0; c < 3; ++c) {
    float La = ka[c] * L->intensity[c];
    float Ld = (cosTheta * kd[c]) * dist[c];
    float Ls = (pw * ks[c]) * dist[c];
    float res = (La + Ld) + Ls;
    out[c] = res * color[c];
  }
}

/* calcBumpMapping / calcDisplacementMapping (src/Shader.cpp:447-507): shared TBN + height-derivative part */
static void s_bump_common(const orc_tex *tex, const float n[3], float u, float v, float kh, float kn,
                          float new_n[3], float *origin_norm) {
  float sq = sqrtf(n[0] * n[0] + n[2] * n[2]);
  float t[3] = {(n[0] * n[1]) / sq, sq, (n[2] * n[1]) / sq};
  float b[3];
  v3_cross(n, t, b);
  float o[3], ud[3], vd[3];
  s_texel(tex, u, v, o);
  float on = sqrtf(v3_dot(o, o)); /* glm::length */
  s_texel(tex, (u + 1.0f) / (float)tex->w, v, ud);
  s_texel(tex, u, (v + 1.0f) / (float)tex->h, vd);
  float dU = kh * kn * (sqrtf(v3_dot(ud, ud)) - on);
  float dV = kh * kn * (sqrtf(v3_dot(vd, vd)) - on);
  float ln[3] = {-dU, -dV, 1.0f};
  /* glm::mat3 TBN(t.x,b.x,n.x, t.y,b.y,n.y, t.z,b.z,n.z): columns (t.x,b.x,n.x),(t.y,b.y,n.y),(t.z,b.z,n.z);
   * mat3*vec3 = col0*v.x + col1*v.y + col2*v.z */
  float r[3] = {t[0] * ln[0] + t[1] * ln[1] + t[2] * ln[2], b[0] * ln[0] + b[1] * ln[1] + b[2] * ln[2],
                n[0] * ln[0] + n[1] * ln[1] + n[2] * ln[2]};
  v3_normalize(r, new_n);
  *origin_norm = on;
}

/* Shader::applyFragmentShader scalar overload + standard_*_impl (src/Shader.cpp:122-126,547-640) followed by
 * Tools::normalizedToRGB (src/Tools.cpp:94-104). nrm is the interpolated+normalised normal. */
/* 0 (default): the reference's result.  1 / 2: probes for tests/test_gpu_approx.py (see the end of s_shade) */
/* test probe of the scalar-tail class (tests/test_gpu_approx.py): compiled into the -O2 checker build only (-DORC_TEST_PROBES,
 * oracle/Makefile); bench.py's cpu_baseline build (-O3) has neither the global nor the two compares per channel in its hot loop,
 * and its orc_debug_s refuses */
#ifdef ORC_TEST_PROBES
static int g_debug_s = 0;
int orc_debug_s(int mode) { g_debug_s = mode; return 0; }
#else
int orc_debug_s(int mode) { return mode == 0 ? 0 : -1; }
#endif

static void s_shade(const orc_shade_ctx *sc, const float P[3], const float nrm[3], float u, float v, float out[3]) {
  const srz_frame *fr = sc->fr;
  float col[3] = {0, 0, 0};
  switch (sc->batch->shader) {
  case SRZ_SHADER_NORMAL: {
    float n[3];
    v3_normalize(nrm, n);
    for (int c = 0; c < 3; ++c) col[c] = (n[c] + 1.0f) / 2.0f;
    break;
  }
  case SRZ_SHADER_TEXTURE:
  case SRZ_SHADER_PHONG:
  case SRZ_SHADER_DISPLACEMENT:
  case SRZ_SHADER_BUMP: {
    float kd[3] = {1.0f, 1.0f, 1.0f}; /* payload.color default (include/shader/Shader.hpp:49) */
    float sp[3] = {P[0], P[1], P[2]}, sn[3] = {nrm[0], nrm[1], nrm[2]};
    if (sc->batch->shader != SRZ_SHADER_PHONG) s_texel(sc->tex, u, v, kd);
    if (sc->batch->shader == SRZ_SHADER_BUMP) {
      float on;
      s_bump_common(sc->tex, nrm, u, v, fr->kh, fr->kn, sn, &on);
    } else if (sc->batch->shader == SRZ_SHADER_DISPLACEMENT) {
      float on, nn[3];
      s_bump_common(sc->tex, nrm, u, v, fr->kh, fr->kn, nn, &on);
      for (int c = 0; c < 3; ++c) sp[c] = P[c] + (fr->kn * nrm[c]) * on; /* position + kn*n*origin_norm */
      sn[0] = nn[0], sn[1] = nn[1], sn[2] = nn[2];
    }
    for (uint32_t l = 0; l < fr->n_lights; ++l) {
      float o[3];
      s_blinn_phong(fr->eye, sp, sn, kd, &fr->lights[l], fr->ka, kd, fr->ks, fr->p, o);
      col[0] = col[0] + o[0], col[1] = col[1] + o[1], col[2] = col[2] + o[2];
    }
    break;
  }
  default:
    break;
  }
  for (int c = 0; c < 3; ++c) {
    float cl = std_clamp(col[c], 0.0f, 1.0f) * 255.0f;
    /* glm::uvec3(float) = float->unsigned truncation; NaN is UB in the reference → 0 here */
    out[c] = (cl == cl) ? (float)(uint32_t)cl : 0.0f;
    /* test probes (orc_debug_s): what the tolerance test needs to state its rule per pixel */
#ifdef ORC_TEST_PROBES
    if (g_debug_s == 1) out[c] = cl;          /* the value in front of the truncation */
    else if (g_debug_s == 2) out[c] = -1.0f;  /* a marker: "this pixel is of the scalar-tail class" */
#endif
  }
}

/* ================================================================================================
 * draw = TraditionalRasterizer::draw for one scene, rows restricted to [row0,row1)
 * ============================================================================================== */
/* prep (optional): per-triangle {keep, box} computed once for the whole frame (used by the OpenMP baseline so that the
 * cull / bbox prologue is not repeated by every row band); indexed by the running triangle number. */
typedef struct {
  int keep;
  orc_box box;
} orc_prep;

static int draw_rows(const srz_frame *fr, float *zb, float *c0, float *c1, float *c2, int row0, int row1,
                     srz_stats *st, uint8_t *owned, const orc_prep *prep) {
  const int W = fr->width, H = fr->height;
  const int unified = (fr->flags & SRZ_UNIFIED) != 0;
  size_t running = 0;
  for (uint32_t bi = 0; bi < fr->n_batches; ++bi) {
    const srz_batch *b = &fr->batches[bi];
    orc_shade_ctx sc;
    sc.fr = fr, sc.batch = b, sc.tex = NULL, sc.texw = 1.0f, sc.texh = 1.0f;
    int needs_tex = b->shader == SRZ_SHADER_TEXTURE || b->shader == SRZ_SHADER_DISPLACEMENT || b->shader == SRZ_SHADER_BUMP;
    if (b->tex_id >= 0 && b->tex_id < ORC_MAX_TEX && g_tex[b->tex_id].bgr) {
      sc.tex = &g_tex[b->tex_id];
      sc.texw = (float)sc.tex->w, sc.texh = (float)sc.tex->h;
    } else if (needs_tex)
      return SRZ_E_TEXTURE;
    for (uint32_t ti = 0; ti < b->n_tris; ++ti) {
      const srz_tri *t = &b->tris[ti];
      orc_box box;
      if (prep) {
        const orc_prep *pp = &prep[running++];
        if (!pp->keep || pp->box.ey < row0 || pp->box.sy >= row1) continue;
      }
      if (st && row0 == 0) st->n_tris++;
      if (!tri_box(t, W, H, &box)) {
        if (st && row0 == 0) st->n_culled++;
        continue;
      }
      if (tri_culled(t, fr->eye)) { /* (src/Rasterizer.cpp:203-205) */
        if (st && row0 == 0) st->n_culled++;
        continue;
      }
      const long long bw = box.ex - box.sx + 1;
      const long long vend = unified ? box.ex + 1 : box.sx + ((bw >> 3) << 3); /* avx2_end (:212-215) */
      const float ax = t->pos[0][0], ay = t->pos[0][1], bx = t->pos[1][0], by = t->pos[1][1];
      const float cx = t->pos[2][0], cy = t->pos[2][1];
      const float z0 = t->pos[0][2], z1 = t->pos[1][2], z2 = t->pos[2][2];
      /* per-triangle constants of the V path (src/Rasterizer.cpp:104-112) */
      const float ABx = bx - ax, ABy = by - ay, ACx = cx - ax, ACy = cy - ay;
      const float v_inv = 1.0f / fmsubf(ABx, ACy, ACx * ABy); /* rcp_ps */
      /* per-triangle constants of the S path (src/Rasterizer.cpp:24-26,53-61) */
      const float BCx = cx - bx, BCy = cy - by, CAx = ax - cx, CAy = ay - cy;
      const float s_area = ABx * ACy - ABy * ACx;
      long long ys = box.sy < row0 ? row0 : box.sy, ye = box.ey >= row1 ? row1 - 1 : box.ey;
      for (long long y = ys; y <= ye; ++y) {
        const float fy = (float)y;
        for (long long x = box.sx; x <= box.ex; ++x) {
          const float fx = (float)x;
          const size_t pos = (size_t)x + (size_t)y * W;
          if (st) st->pixel_tests++;
          float alpha, beta, gamma, z, n[3], u, v, col[3];
          if (x < vend) {
            /* ---------------- V semantics ---------------- */
            float PBx = bx - fx, PBy = by - fy, PCx = cx - fx, PCy = cy - fy, PAx = ax - fx, PAy = ay - fy;
            float aPBC = fmsubf(PBx, PCy, PCx * PBy), aPCA = fmsubf(PCx, PAy, PAx * PCy);
            alpha = aPBC * v_inv, beta = aPCA * v_inv, gamma = 1.0f - (alpha + beta);
            int inside = alpha > 0.0f && alpha < 1.0f && beta > 0.0f && beta < 1.0f && gamma > 0.0f && gamma < 1.0f;
            if (!inside) continue;
            if (st) st->fragments++;
            z = fmaf(alpha, z0, fmaf(beta, z1, gamma * z2));
            if (!(z < zb[pos])) continue; /* strict, ordered (:334) */
            for (int c = 0; c < 3; ++c) n[c] = fmaf(alpha, t->nrm[0][c], fmaf(beta, t->nrm[1][c], gamma * t->nrm[2][c]));
            v_normalized(&n[0], &n[1], &n[2]);
            u = fmaf(alpha, t->uv[0][0], fmaf(beta, t->uv[1][0], gamma * t->uv[2][0]));
            v = fmaf(alpha, t->uv[0][1], fmaf(beta, t->uv[1][1], gamma * t->uv[2][1]));
            float P[3] = {fx, fy, z};
            v_shade(&sc, P, n, u, v, col);
          } else {
            /* ---------------- S semantics ---------------- */
            float APx = fx - ax, APy = fy - ay, BPx = fx - bx, BPy = fy - by, CPx = fx - cx, CPy = fy - cy;
            float e0 = ABx * APy - ABy * APx, e1 = BCx * BPy - BCy * BPx, e2 = CAx * CPy - CAy * CPx;
            int inside = (e0 > 0 && e1 > 0 && e2 > 0) || (e0 < 0 && e1 < 0 && e2 < 0); /* (:39-40) */
            if (!inside) continue;
            if (st) st->fragments++;
            float PAx = ax - fx, PAy = ay - fy, PBx = bx - fx, PBy = by - fy, PCx = cx - fx, PCy = cy - fy;
            float aPBC = PBx * PCy - PBy * PCx, aPCA = PCx * PAy - PCy * PAx;
            alpha = aPBC / s_area, beta = aPCA / s_area, gamma = 1.0f - alpha - beta;
            z = alpha * z0 + beta * z1 + gamma * z2;
            if (z > zb[pos]) continue; /* <= passes, NaN passes (:475) */
            float nn[3];
            for (int c = 0; c < 3; ++c) nn[c] = alpha * t->nrm[0][c] + beta * t->nrm[1][c] + gamma * t->nrm[2][c];
            v3_normalize(nn, n); /* glm::normalize (src/Tools.cpp:115) */
            u = alpha * t->uv[0][0] + beta * t->uv[1][0] + gamma * t->uv[2][0];
            v = alpha * t->uv[0][1] + beta * t->uv[1][1] + gamma * t->uv[2][1];
            float P[3] = {fx, fy, z};
            s_shade(&sc, P, n, u, v, col);
          }
          if (st) st->shaded++;
          if (owned) owned[pos] = (uint8_t)(1 + (needs_tex ? 1 : 0));
          zb[pos] = z;
          c0[pos] = col[0], c1[pos] = col[1], c2[pos] = col[2];
        }
      }
    }
  }
  return SRZ_OK;
}

/* ---- public oracle entry points ---------------------------------------------------------------- */
int orc_texture_set(int tex_id, const uint8_t *bgr, int w, int h, int row_stride) {
  if (tex_id < 0 || tex_id >= ORC_MAX_TEX || !bgr || w <= 0 || h <= 0 || row_stride < 3 * w) return SRZ_E_INVALID;
  free(g_tex[tex_id].bgr);
  g_tex[tex_id].bgr = (uint8_t *)malloc((size_t)w * h * 3);
  if (!g_tex[tex_id].bgr) return SRZ_E_NOMEM;
  for (int y = 0; y < h; ++y) memcpy(g_tex[tex_id].bgr + (size_t)y * w * 3, bgr + (size_t)y * row_stride, (size_t)w * 3);
  g_tex[tex_id].w = w, g_tex[tex_id].h = h;
  return SRZ_OK;
}

/* RenderingPipeline::clear(Color|Depth) (src/Render.cpp:31-55) */
void orc_clear(int W, int H, float *z, float *c0, float *c1, float *c2) {
  for (size_t i = 0; i < (size_t)W * H; ++i) z[i] = INFINITY, c0[i] = c1[i] = c2[i] = 0.0f;
}

/* single-threaded, deterministic */
int orc_draw(int primitive, const srz_frame *fr, float *z, float *c0, float *c1, float *c2, srz_stats *st) {
  if (primitive != SRZ_PRIMITIVE_LINES && primitive != SRZ_PRIMITIVE_TRIANGLES) return SRZ_E_PRIMITIVE;
  if (!fr || !z || !c0 || !c1 || !c2 || fr->width <= 0 || fr->height <= 0) return SRZ_E_INVALID;
  if (st) memset(st, 0, sizeof *st);
  if (fr->flags & SRZ_FUSED_CLEAR) orc_clear(fr->width, fr->height, z, c0, c1, c2);
  uint8_t *owned = st ? (uint8_t *)calloc((size_t)fr->width * fr->height, 1) : NULL;
  int rc = draw_rows(fr, z, c0, c1, c2, 0, fr->height, st, owned, NULL);
  if (owned) {
    for (size_t i = 0; i < (size_t)fr->width * fr->height; ++i) st->visible += owned[i] != 0, st->visible_textured += owned[i] == 2;
    free(owned);
  }
  return rc;
}

/* rows [row0,row1) only — used by the band-sharding tests */
int orc_draw_rows(const srz_frame *fr, float *z, float *c0, float *c1, float *c2, int row0, int row1) {
  if (!fr || !z || !c0 || !c1 || !c2) return SRZ_E_INVALID;
  if (row0 < 0) row0 = 0;
  if (row1 > fr->height) row1 = fr->height;
  if (fr->flags & SRZ_FUSED_CLEAR)
    for (size_t i = (size_t)row0 * fr->width; i < (size_t)row1 * fr->width; ++i) z[i] = INFINITY, c0[i] = c1[i] = c2[i] = 0.0f;
  return draw_rows(fr, z, c0, c1, c2, row0, row1, NULL, NULL, NULL);
}

/* CPU baseline: the same per-pixel code, rows dealt in bands of `band` rows to OpenMP threads (per-pixel
 * results depend only on the per-pixel submission order, which every band preserves). Returns threads used. */
int orc_draw_omp(const srz_frame *fr, float *z, float *c0, float *c1, float *c2, int band, int *threads_used, int num_threads) {
  if (!fr || !z || !c0 || !c1 || !c2 || band <= 0) return SRZ_E_INVALID;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
  const int W = fr->width, H = fr->height, nb = (H + band - 1) / band;
  size_t n_tris = 0;
  for (uint32_t bi = 0; bi < fr->n_batches; ++bi) n_tris += fr->batches[bi].n_tris;
  orc_prep *prep = (orc_prep *)malloc(sizeof(orc_prep) * (n_tris ? n_tris : 1));
  const srz_tri **flat = (const srz_tri **)malloc(sizeof(void *) * (n_tris ? n_tris : 1));
  if (!prep || !flat) {
    free(prep), free(flat);
    return SRZ_E_NOMEM;
  }
  size_t k = 0;
  for (uint32_t bi = 0; bi < fr->n_batches; ++bi)
    for (uint32_t ti = 0; ti < fr->batches[bi].n_tris; ++ti) flat[k++] = &fr->batches[bi].tris[ti];
  int rc = SRZ_OK, nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    nthreads = omp_get_num_threads();
#pragma omp for schedule(static)
#endif
    for (long long i = 0; i < (long long)n_tris; ++i) { /* prologue once per triangle: bbox + backface test */
      prep[i].keep = tri_box(flat[i], W, H, &prep[i].box) && !tri_culled(flat[i], fr->eye);
    }
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (int b = 0; b < nb; ++b) {
      int r0 = b * band, r1 = r0 + band > H ? H : r0 + band;
      if (fr->flags & SRZ_FUSED_CLEAR)
        for (size_t i = (size_t)r0 * W; i < (size_t)r1 * W; ++i) z[i] = INFINITY, c0[i] = c1[i] = c2[i] = 0.0f;
      int r = draw_rows(fr, z, c0, c1, c2, r0, r1, NULL, NULL, prep);
      if (r != SRZ_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
        rc = r;
      }
    }
  }
  free(prep), free(flat);
  if (threads_used) *threads_used = nthreads;
  return rc;
}

/* CPU baseline, throughput form: whole frames dealt to OpenMP threads (each with its own four planes), `n_total` frames
 * taken round-robin from `frames[0..n_unique)`; every frame is clear + draw through orc_draw (serial per frame).  This is
 * the shape that uses every core of a large host: the row-band form above is limited by one frame's parallelism. */
int orc_draw_frames_omp(const srz_frame *const *frames, int n_unique, long long n_total, int num_threads, int *threads_used) {
  if (!frames || n_unique <= 0 || n_total < 0) return SRZ_E_INVALID;
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
  int rc = SRZ_OK, nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
#ifdef _OPENMP
#pragma omp single
    nthreads = omp_get_num_threads();
#endif
    size_t cap = 0;
    float *pl = NULL;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
    for (long long i = 0; i < n_total; ++i) {
      const srz_frame *fr = frames[i % n_unique];
      const size_t px = (size_t)fr->width * (size_t)fr->height;
      if (px > cap) {
        free(pl);
        pl = (float *)malloc(sizeof(float) * 4 * px);
        cap = pl ? px : 0;
      }
      int r = pl ? orc_draw(SRZ_PRIMITIVE_TRIANGLES, fr, pl, pl + px, pl + 2 * px, pl + 3 * px, NULL) : SRZ_E_NOMEM;
      if (r != SRZ_OK) {
#ifdef _OPENMP
#pragma omp critical
#endif
        rc = r;
      }
    }
    free(pl);
  }
  if (threads_used) *threads_used = nthreads;
  return rc;
}

/* display() resolve (src/Render.cpp:61-62): merge planes 0,1,2 -> interleaved, convertTo(CV_8UC3) =
 * saturate_cast<uchar>(cvRound(v)) (round half to even) */
void orc_resolve8(int W, int H, const float *c0, const float *c1, const float *c2, uint8_t *out) {
  const float *pl[3] = {c0, c1, c2};
  for (size_t i = 0; i < (size_t)W * H; ++i)
    for (int c = 0; c < 3; ++c) {
      float v = pl[c][i];
      long r = (v == v && v > -1e9f && v < 1e9f) ? lrintf(v) : (v > 0 ? 255 : 0);
      out[i * 3 + c] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
    }
}
