// TEST INFRASTRUCTURE — "GLSL on the CPU": the subset of GLSL 3.30 that the reference's hot-path shaders
// (Core/Shaders/*.vert|geom|frag|glsl) use, as a C++ vocabulary, so that THOSE SHADER SOURCES can be compiled by g++
// where they lie (oracle/Makefile, target refglsl; oracle/glsl_on_cpu/glsl2cpp.pl does the few token-level rewrites GLSL
// needs, in a pipe) and executed per vertex / per fragment by oracle/ref_glsl_bridge.cpp.  Independent implementation
// of the GLSL semantics; nothing here comes from the reference.
//
// What GLSL leaves to the implementation is SPECIFIED here, identically to the oracle (oracle/efo_common.h):
//   * every operation is the IEEE binary32 operation written in the shader, evaluated left to right, no contraction;
//   * dot = ((x*x' + y*y') + z*z') (+ w*w'); mat * vec = that dot per row; cross as written in the GLSL spec;
//   * normalize(v) = v * (1 / sqrt(dot(v, v))); length = sqrt(dot); inversesqrt = 1 / sqrt;
//   * round() = half away from zero; exp() goes through glsl::exp_hook (libm expf by default; the bridge can install
//     the oracle's IEEE-only polynomial so that exp-dependent outputs can be compared bit for bit);
//   * texture()/textureLod(): NEAREST, CLAMP_TO_EDGE, texel = floor(u * size) evaluated exactly, except that a
//     coordinate within glsl::texel_snap of a texel boundary counts as ON the boundary (belongs to the upper texel) —
//     the shaders build tap coordinates like fl(cx / cols) whose float noise must not pick the neighbour (N4).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <type_traits>

namespace glsl {

typedef unsigned int uint;
template <class S>
using arith = typename std::enable_if<std::is_arithmetic<S>::value, int>::type;

struct vec2; struct vec3; struct vec4;

struct vec2 {
  float x, y;
  vec2() : x(0), y(0) {}
  template <class A, class B, arith<A> = 0, arith<B> = 0> vec2(A a, B b) : x((float)a), y((float)b) {}
  template <class A, arith<A> = 0> explicit vec2(A a) : x((float)a), y((float)a) {}
  explicit vec2(const vec3& v);
  explicit vec2(const vec4& v);
  vec2 xy() const { return *this; }
};
struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  template <class A, class B, class C, arith<A> = 0, arith<B> = 0, arith<C> = 0> vec3(A a, B b, C c) : x((float)a), y((float)b), z((float)c) {}
  template <class A, arith<A> = 0> explicit vec3(A a) : x((float)a), y((float)a), z((float)a) {}
  template <class C, arith<C> = 0> vec3(const vec2& v, C c) : x(v.x), y(v.y), z((float)c) {}
  explicit vec3(const vec4& v);
  vec2 xy() const { return vec2(x, y); }
  vec3 xyz() const { return *this; }
  void set_xyz(const vec3& v) { *this = v; }
};
struct vec4 {
  float x, y, z, w;
  vec4() : x(0), y(0), z(0), w(0) {}
  template <class A, class B, class C, class D, arith<A> = 0, arith<B> = 0, arith<C> = 0, arith<D> = 0>
  vec4(A a, B b, C c, D d) : x((float)a), y((float)b), z((float)c), w((float)d) {}
  template <class A, arith<A> = 0> explicit vec4(A a) : x((float)a), y((float)a), z((float)a), w((float)a) {}
  template <class D, arith<D> = 0> vec4(const vec3& v, D d) : x(v.x), y(v.y), z(v.z), w((float)d) {}
  template <class C, class D, arith<C> = 0, arith<D> = 0> vec4(const vec2& v, C c, D d) : x(v.x), y(v.y), z((float)c), w((float)d) {}
  vec2 xy() const { return vec2(x, y); }
  vec2 zw() const { return vec2(z, w); }
  vec3 xyz() const { return vec3(x, y, z); }
  void set_xyz(const vec3& v) { x = v.x; y = v.y; z = v.z; }
  explicit operator float() const { return x; }   // float(textureLod(...)): first component
};
inline vec2::vec2(const vec3& v) : x(v.x), y(v.y) {}
inline vec2::vec2(const vec4& v) : x(v.x), y(v.y) {}
inline vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}
struct uvec4 {
  uint x, y, z, w;
  explicit operator uint() const { return x; }
  explicit operator float() const { return (float)x; }
  explicit operator int() const { return (int)x; }
};

#define GLSL_VEC_OPS(V, N, ...)                                                                                     \
  inline V operator+(const V& a, const V& b) { V r; const float* p = &a.x; const float* q = &b.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] + q[i]; return r; } \
  inline V operator-(const V& a, const V& b) { V r; const float* p = &a.x; const float* q = &b.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] - q[i]; return r; } \
  inline V operator*(const V& a, const V& b) { V r; const float* p = &a.x; const float* q = &b.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] * q[i]; return r; } \
  inline V operator/(const V& a, const V& b) { V r; const float* p = &a.x; const float* q = &b.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] / q[i]; return r; } \
  inline V operator-(const V& a) { V r; const float* p = &a.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = -p[i]; return r; } \
  template <class S, arith<S> = 0> inline V operator*(const V& a, S s) { V r; const float* p = &a.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] * (float)s; return r; } \
  template <class S, arith<S> = 0> inline V operator*(S s, const V& a) { V r; const float* p = &a.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = (float)s * p[i]; return r; } \
  template <class S, arith<S> = 0> inline V operator/(const V& a, S s) { V r; const float* p = &a.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] / (float)s; return r; } \
  template <class S, arith<S> = 0> inline V operator+(const V& a, S s) { V r; const float* p = &a.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] + (float)s; return r; } \
  template <class S, arith<S> = 0> inline V operator-(const V& a, S s) { V r; const float* p = &a.x; float* o = &r.x; for (int i = 0; i < N; ++i) o[i] = p[i] - (float)s; return r; } \
  inline V& operator+=(V& a, const V& b) { a = a + b; return a; }                                                   \
  inline V& operator-=(V& a, const V& b) { a = a - b; return a; }                                                   \
  template <class S, arith<S> = 0> inline V& operator*=(V& a, S s) { a = a * s; return a; }                          \
  template <class S, arith<S> = 0> inline V& operator/=(V& a, S s) { a = a / s; return a; }
GLSL_VEC_OPS(vec2, 2)
GLSL_VEC_OPS(vec3, 3)
GLSL_VEC_OPS(vec4, 4)
#undef GLSL_VEC_OPS

// ---- scalar built-ins (float unless both arguments are integers) ----
inline float abs(float a) { return fabsf(a); }
inline int abs(int a) { return a < 0 ? -a : a; }
inline float sqrt(float a) { return sqrtf(a); }
inline float inversesqrt(float a) { return 1.0f / sqrtf(a); }
inline float floor(float a) { return floorf(a); }
inline float round(float a) { return roundf(a); }
inline float acos(float a) { return acosf(a); }
// pow(x, 2) is x * x (the exactly rounded square); anything else goes to libm
template <class B, arith<B> = 0> inline float pow(float a, B b) { return (float)b == 2.0f ? a * a : powf(a, (float)b); }
extern float (*exp_hook)(float);
inline float exp(float a) { return exp_hook(a); }
inline float min(float a, float b) { return b < a ? b : a; }   // GLSL: y < x ? y : x
inline float max(float a, float b) { return a < b ? b : a; }   // GLSL: x < y ? y : x
inline int min(int a, int b) { return b < a ? b : a; }
inline int max(int a, int b) { return a < b ? b : a; }
inline uint min(uint a, uint b) { return b < a ? b : a; }
inline uint max(uint a, uint b) { return a < b ? b : a; }
inline float min(int a, float b) { return min((float)a, b); }
inline float min(float a, int b) { return min(a, (float)b); }
inline float max(int a, float b) { return max((float)a, b); }
inline float max(float a, int b) { return max(a, (float)b); }

// ---- geometric built-ins ----
inline float dot(const vec2& a, const vec2& b) { return a.x * b.x + a.y * b.y; }
inline float dot(const vec3& a, const vec3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline float dot(const vec4& a, const vec4& b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
inline vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(const vec2& a) { return sqrtf(dot(a, a)); }
inline float length(const vec3& a) { return sqrtf(dot(a, a)); }
inline vec3 normalize(const vec3& a) { const float rn = 1.0f / sqrtf(dot(a, a)); return vec3(a.x * rn, a.y * rn, a.z * rn); }

// ---- matrices (column major, m[col][row]) ----
struct mat4 {
  float m[4][4];
  mat4() { std::memset(m, 0, sizeof(m)); }
};
struct mat3 {
  float m[3][3];
  mat3() { std::memset(m, 0, sizeof(m)); }
  explicit mat3(const mat4& a) { for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) m[c][r] = a.m[c][r]; }
  mat3(const vec3& c0, const vec3& c1, const vec3& c2) {
    m[0][0] = c0.x; m[0][1] = c0.y; m[0][2] = c0.z; m[1][0] = c1.x; m[1][1] = c1.y; m[1][2] = c1.z; m[2][0] = c2.x; m[2][1] = c2.y; m[2][2] = c2.z;
  }
};
inline vec4 operator*(const mat4& a, const vec4& v) {
  vec4 r;
  float* o = &r.x;
  for (int i = 0; i < 4; ++i) o[i] = ((a.m[0][i] * v.x + a.m[1][i] * v.y) + a.m[2][i] * v.z) + a.m[3][i] * v.w;
  return r;
}
inline vec3 operator*(const mat3& a, const vec3& v) {
  vec3 r;
  float* o = &r.x;
  for (int i = 0; i < 3; ++i) o[i] = (a.m[0][i] * v.x + a.m[1][i] * v.y) + a.m[2][i] * v.z;
  return r;
}
inline mat3 transpose(const mat3& a) { mat3 r; for (int c = 0; c < 3; ++c) for (int k = 0; k < 3; ++k) r.m[c][k] = a.m[k][c]; return r; }
inline mat3 inverse(const mat3& a) {   // adjugate / determinant
  const float (*m)[3] = a.m;
  const float c00 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c01 = m[2][1] * m[0][2] - m[0][1] * m[2][2], c02 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
  const float det = (m[0][0] * c00 + m[1][0] * c01) + m[2][0] * c02;
  const float id = 1.0f / det;
  mat3 r;
  r.m[0][0] = c00 * id; r.m[0][1] = c01 * id; r.m[0][2] = c02 * id;
  r.m[1][0] = (m[2][0] * m[1][2] - m[1][0] * m[2][2]) * id; r.m[1][1] = (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * id; r.m[1][2] = (m[1][0] * m[0][2] - m[0][0] * m[1][2]) * id;
  r.m[2][0] = (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * id; r.m[2][1] = (m[2][0] * m[0][1] - m[0][0] * m[2][1]) * id; r.m[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * id;
  return r;
}

// ---- textures ----
struct Texture {
  enum Format { F32, U8_NORM, U16, U32, I32 };
  const void* data = nullptr;
  int width = 0, height = 0, channels = 1;   // row-major, `channels` interleaved components per texel
  Format format = F32;
};
extern double texel_snap;   // see the header comment
inline int texel_of(float u, int n) {
  const double p = (double)u * (double)n;
  const double r = std::nearbyint(p);
  int i = (int)std::floor(std::fabs(p - r) <= texel_snap ? r : p);
  return i < 0 ? 0 : (i >= n ? n - 1 : i);   // CLAMP_TO_EDGE
}
struct sampler2D { const Texture* t = nullptr; };
struct usampler2D { const Texture* t = nullptr; };
inline vec4 fetch_f(const Texture& t, int ix, int iy) {
  float c[4] = {0, 0, 0, 1};
  const size_t at = ((size_t)iy * t.width + ix) * t.channels;
  for (int k = 0; k < t.channels && k < 4; ++k)
    c[k] = t.format == Texture::F32 ? ((const float*)t.data)[at + k] : (float)((const uint8_t*)t.data)[at + k] / 255.0f;
  return vec4(c[0], c[1], c[2], c[3]);
}
inline uvec4 fetch_u(const Texture& t, int ix, int iy) {
  uint c[4] = {0, 0, 0, 1};
  const size_t at = ((size_t)iy * t.width + ix) * t.channels;
  for (int k = 0; k < t.channels && k < 4; ++k)
    c[k] = t.format == Texture::U16 ? ((const uint16_t*)t.data)[at + k] : ((const uint32_t*)t.data)[at + k];
  return uvec4{c[0], c[1], c[2], c[3]};
}
template <class L, arith<L> = 0> inline vec4 textureLod(const sampler2D& s, const vec2& uv, L) { return fetch_f(*s.t, texel_of(uv.x, s.t->width), texel_of(uv.y, s.t->height)); }
template <class L, arith<L> = 0> inline uvec4 textureLod(const usampler2D& s, const vec2& uv, L) { return fetch_u(*s.t, texel_of(uv.x, s.t->width), texel_of(uv.y, s.t->height)); }
inline vec4 texture(const sampler2D& s, const vec2& uv) { return textureLod(s, uv, 0); }
inline uvec4 texture(const usampler2D& s, const vec2& uv) { return textureLod(s, uv, 0); }
inline vec4 texture2D(const sampler2D& s, const vec2& uv) { return textureLod(s, uv, 0); }

// ---- pipeline variables ----
extern vec4 gl_Position, gl_FragCoord;
extern float gl_PointSize, gl_FragDepth;
extern int gl_VertexID;
struct gl_PerVertex { vec4 gl_Position; };
extern gl_PerVertex gl_in[1];               // geometry shaders with `layout(points) in`
extern bool discard_flag;
extern std::function<void()> emit_hook;   // geometry shaders: EmitVertex() lets the bridge capture the current outputs
inline void EmitVertex() { if (emit_hook) emit_hook(); }
inline void EndPrimitive() {}

}  // namespace glsl
