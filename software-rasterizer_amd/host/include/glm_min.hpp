// glm_min.hpp — the handful of GLM types and functions the raster path of Liupeter01/Software-Rasterizer uses
// (call sites: src/Object.cpp:27-30, src/Scene.cpp:270,293,922-923, src/Triangle.cpp:149-150, src/Tools.cpp:115).
// GLM itself is an un-vendored submodule of the reference and absent from this image, so the host layer carries this
// minimal stand-in under the same names (`glm::vec3`, `glm::mat4`, `glm::lookAtLH`, ...) so that user code written
// against the reference's API compiles unchanged.  Expression order follows GLM 0.9.9 (column-major, m[col][row]) and
// must stay identical to oracle/srz_oracle.c, which the tests compare against bit-for-bit.
// Define SRZ_USE_REAL_GLM to use a real GLM installation instead.
#pragma once
#ifdef SRZ_USE_REAL_GLM
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#else
#include <cmath>
#include <cstddef>

namespace glm {

struct vec2 {
  float x, y;
  constexpr vec2() : x(0), y(0) {}
  constexpr explicit vec2(float s) : x(s), y(s) {}
  constexpr vec2(float x_, float y_) : x(x_), y(y_) {}
  float &operator[](int i) { return (&x)[i]; }
  const float &operator[](int i) const { return (&x)[i]; }
};
struct vec3 {
  float x, y, z;
  constexpr vec3() : x(0), y(0), z(0) {}
  constexpr explicit vec3(float s) : x(s), y(s), z(s) {}
  template <typename A, typename B, typename C> constexpr vec3(A x_, B y_, C z_) : x((float)x_), y((float)y_), z((float)z_) {}
  float &operator[](int i) { return (&x)[i]; }
  const float &operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
  float x, y, z, w;
  constexpr vec4() : x(0), y(0), z(0), w(0) {}
  constexpr explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
  constexpr vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
  constexpr vec4(const vec3 &v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
  float &operator[](int i) { return (&x)[i]; }
  const float &operator[](int i) const { return (&x)[i]; }
};
struct uvec3 {
  unsigned x, y, z;
  constexpr uvec3() : x(0), y(0), z(0) {}
  template <typename A, typename B, typename C> constexpr uvec3(A x_, B y_, C z_) : x((unsigned)x_), y((unsigned)y_), z((unsigned)z_) {}
  unsigned &operator[](int i) { return (&x)[i]; }
  const unsigned &operator[](int i) const { return (&x)[i]; }
};

inline vec2 operator+(const vec2 &a, const vec2 &b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator-(const vec2 &a, const vec2 &b) { return vec2(a.x - b.x, a.y - b.y); }
inline vec2 operator*(float s, const vec2 &a) { return vec2(s * a.x, s * a.y); }
inline vec2 operator*(const vec2 &a, float s) { return vec2(a.x * s, a.y * s); }
inline bool operator==(const vec2 &a, const vec2 &b) { return a.x == b.x && a.y == b.y; }

inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(const vec3 &a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(const vec3 &a, const vec3 &b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(float s, const vec3 &a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(const vec3 &a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator/(const vec3 &a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline bool operator==(const vec3 &a, const vec3 &b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

inline vec4 operator+(const vec4 &a, const vec4 &b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline vec4 operator-(const vec4 &a, const vec4 &b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
inline vec4 operator*(const vec4 &a, const vec4 &b) { return vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
inline vec4 operator*(const vec4 &a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }

// glm compute_dot<vec3>: tmp = a*b; tmp.x + tmp.y + tmp.z
inline float dot(const vec3 &a, const vec3 &b) {
  vec3 t = a * b;
  return t.x + t.y + t.z;
}
inline vec3 cross(const vec3 &x, const vec3 &y) {
  return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
inline vec3 normalize(const vec3 &v) { return v * inversesqrt(dot(v, v)); }
inline float length(const vec3 &v) { return std::sqrt(dot(v, v)); }
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }
inline float asin(float x) { return std::asin(x); }

struct mat4 {
  vec4 c[4]; // columns
  constexpr mat4() : c{vec4(), vec4(), vec4(), vec4()} {}
  constexpr explicit mat4(float d) : c{vec4(d, 0, 0, 0), vec4(0, d, 0, 0), vec4(0, 0, d, 0), vec4(0, 0, 0, d)} {}
  vec4 &operator[](int i) { return c[i]; }
  const vec4 &operator[](int i) const { return c[i]; }
  const float *data() const { return &c[0].x; }
  float *data() { return &c[0].x; }
};
using mat4x4 = mat4;

// operator*(mat4, vec4): (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline vec4 operator*(const mat4 &m, const vec4 &v) {
  vec4 add0 = m[0] * v.x + m[1] * v.y;
  vec4 add1 = m[2] * v.z + m[3] * v.w;
  return add0 + add1;
}
// operator*(mat4, mat4): Result[j] = A0*B[j][0] + A1*B[j][1] + A2*B[j][2] + A3*B[j][3] (left to right)
inline mat4 operator*(const mat4 &a, const mat4 &b) {
  mat4 r;
  for (int j = 0; j < 4; ++j) r[j] = ((a[0] * b[j].x + a[1] * b[j].y) + a[2] * b[j].z) + a[3] * b[j].w;
  return r;
}
inline mat4 transpose(const mat4 &m) {
  mat4 r;
  for (int c = 0; c < 4; ++c)
    for (int rr = 0; rr < 4; ++rr) r[rr][c] = m[c][rr];
  return r;
}
inline mat4 translate(const mat4 &m, const vec3 &v) {
  mat4 r = m;
  r[3] = ((m[0] * v.x + m[1] * v.y) + m[2] * v.z) + m[3];
  return r;
}
inline mat4 scale(const mat4 &m, const vec3 &v) {
  mat4 r;
  r[0] = m[0] * v.x, r[1] = m[1] * v.y, r[2] = m[2] * v.z, r[3] = m[3];
  return r;
}
inline mat4 rotate(const mat4 &m, float angle, const vec3 &v) {
  const float a = angle, c = std::cos(a), s = std::sin(a);
  vec3 axis = normalize(v);
  vec3 temp = (1.0f - c) * axis;
  float R00 = c + temp.x * axis.x, R01 = temp.x * axis.y + s * axis.z, R02 = temp.x * axis.z - s * axis.y;
  float R10 = temp.y * axis.x - s * axis.z, R11 = c + temp.y * axis.y, R12 = temp.y * axis.z + s * axis.x;
  float R20 = temp.z * axis.x + s * axis.y, R21 = temp.z * axis.y - s * axis.x, R22 = c + temp.z * axis.z;
  mat4 r;
  r[0] = (m[0] * R00 + m[1] * R01) + m[2] * R02;
  r[1] = (m[0] * R10 + m[1] * R11) + m[2] * R12;
  r[2] = (m[0] * R20 + m[1] * R21) + m[2] * R22;
  r[3] = m[3];
  return r;
}
inline mat4 lookAtLH(const vec3 &eye, const vec3 &center, const vec3 &up) {
  const vec3 f = normalize(center - eye);
  const vec3 s = normalize(cross(up, f));
  const vec3 u = cross(f, s);
  mat4 r(1.0f);
  r[0][0] = s.x, r[1][0] = s.y, r[2][0] = s.z;
  r[0][1] = u.x, r[1][1] = u.y, r[2][1] = u.z;
  r[0][2] = f.x, r[1][2] = f.y, r[2][2] = f.z;
  r[3][0] = -dot(s, eye), r[3][1] = -dot(u, eye), r[3][2] = -dot(f, eye);
  return r;
}
inline mat4 perspectiveLH_NO(float fovy, float aspect, float zNear, float zFar) {
  const float t = std::tan(fovy / 2.0f);
  mat4 r; // zero
  r[0][0] = 1.0f / (aspect * t);
  r[1][1] = 1.0f / t;
  r[2][2] = (zFar + zNear) / (zFar - zNear);
  r[2][3] = 1.0f;
  r[3][2] = -(2.0f * zFar * zNear) / (zFar - zNear);
  return r;
}
// compute_inverse<4,4>
inline mat4 inverse(const mat4 &m) {
  float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
  float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3], c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
  float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
  float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
  float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
  float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
  float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
  float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2], c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
  float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
  vec4 f0(c00, c00, c02, c03), f1(c04, c04, c06, c07), f2(c08, c08, c10, c11);
  vec4 f3(c12, c12, c14, c15), f4(c16, c16, c18, c19), f5(c20, c20, c22, c23);
  vec4 v0(m[1][0], m[0][0], m[0][0], m[0][0]), v1(m[1][1], m[0][1], m[0][1], m[0][1]);
  vec4 v2(m[1][2], m[0][2], m[0][2], m[0][2]), v3(m[1][3], m[0][3], m[0][3], m[0][3]);
  vec4 i0 = (v1 * f0 - v2 * f1) + v3 * f2, i1 = (v0 * f0 - v2 * f3) + v3 * f4;
  vec4 i2 = (v0 * f1 - v1 * f3) + v3 * f5, i3 = (v0 * f2 - v1 * f4) + v2 * f5;
  vec4 sa(+1, -1, +1, -1), sb(-1, +1, -1, +1);
  mat4 inv;
  inv[0] = i0 * sa, inv[1] = i1 * sb, inv[2] = i2 * sa, inv[3] = i3 * sb;
  vec4 row0(inv[0][0], inv[1][0], inv[2][0], inv[3][0]);
  vec4 d = m[0] * row0;
  float det = (d.x + d.y) + (d.z + d.w);
  float ood = 1.0f / det;
  mat4 r;
  for (int j = 0; j < 4; ++j) r[j] = inv[j] * ood;
  return r;
}

} // namespace glm
#endif
