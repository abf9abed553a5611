// Device-side scalar conventions shared by every kernel of libefusion_hip (gfx950 only).
//
// Arithmetic policy (DESIGN.md §"Numerics"): fp32 where the reference uses float, built with
// -ffp-contract=off; fused multiply-adds appear only where written explicitly (dot, cross, the
// JtJ accumulation).  1/sqrt is 1.0f/sqrtf (HIP's fp32 divide and sqrt are correctly rounded by
// default), exp() of the GLSL passes is ef_expf (IEEE-only polynomial, reproducible on any IEEE machine),
// __float2int_rn is v_rndne + saturating convert (NaN -> 0), like CUDA's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ef_build.hpp"

// Multiply-adds of the tracking kernels appear only through EF_FMA.  The shipped default (ef_build.hpp: reference rounding) splits each
// into an IEEE multiply and an IEEE add, which is what the reference's own CUDA sources compute when compiled without
// contraction — compared bit for bit with the compiled reference / its golden vectors on the GPU (tests/test_gpu_vs_reference.py);
// the opt-in fast build (-DEF_FAST_BUILD) fuses them.
#ifdef EF_NO_FMA
#define EF_FMA(a, b, c) ((a) * (b) + (c))
#else
#define EF_FMA(a, b, c) fmaf((a), (b), (c))
#endif

namespace ef {

struct f3 { float x, y, z; };

__device__ __forceinline__ float qnan() { return __int_as_float(0x7fffffff); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return EF_FMA(a.z, b.z, EF_FMA(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
  return {EF_FMA(a.y, b.z, -(a.z * b.y)), EF_FMA(a.z, b.x, -(a.x * b.z)), EF_FMA(a.x, b.y, -(a.y * b.x))};
}
__device__ __forceinline__ float norm(f3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ f3 normalized(f3 a) {
  const float rn = 1.0f / sqrtf(dot(a, a));
  return {a.x * rn, a.y * rn, a.z * rn};
}
// bgr2IntensityKernel's pixel function, cudafuncs.cu:593 (quirk Q5)
__device__ __forceinline__ uint8_t intensity_of(float c0, float c1, float c2) {
  const int value = (int)(c0 * 0.114f + c1 * 0.299f + c2 * 0.587f);
  return (uint8_t)value;
}
struct m33 { f3 r[3]; };
__device__ __forceinline__ f3 mul(const m33& m, f3 a) { return {dot(m.r[0], a), dot(m.r[1], a), dot(m.r[2], a)}; }
__device__ __forceinline__ m33 m33_load(const float* p) {
  return m33{{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}}};
}
// rigid transform held as row-major 3x4 (rows of [R|t])
struct rt34 { m33 R; f3 t; };
__device__ __forceinline__ rt34 rt34_load16(const float* M) {
  return rt34{{{{M[0], M[1], M[2]}, {M[4], M[5], M[6]}, {M[8], M[9], M[10]}}}, {M[3], M[7], M[11]}};
}
__device__ __forceinline__ f3 xform(const rt34& T, f3 p) { return mul(T.R, p) + T.t; }

// round-half-even to int with CUDA's __float2int_rn edge semantics spelled out (NaN -> 0, saturating);
// a bare (int) cast of NaN / out-of-range is undefined at the IR level, so guard explicitly.
__device__ __forceinline__ int f2i_rn(float x) {
  float r = (x != x) ? 0.0f : rintf(x);
  r = fminf(fmaxf(r, -2147483648.0f), 2147483520.0f);
  return (int)r;
}

// Gates on a Euclidean norm without the square root (round 6).  sqrtf is correctly rounded and monotone, so for a threshold T the set
// {a >= 0 : sqrtf(a) <= T} is an initial segment of the floats: "sqrtf(a) <= T" is "a <= sq_le_max(T)" and "sqrtf(a) < T" is
// "a <= sq_lt_max(T)", bit for bit on every input (NaN fails both forms, +inf passes neither unless T is +inf).  Evaluated once per call on the
// host (glibc's sqrtf is correctly rounded too); an empty set is -1.
__host__ __device__ inline float ef_next_up(float a) { unsigned u; __builtin_memcpy(&u, &a, 4); ++u; __builtin_memcpy(&a, &u, 4); return a; }     // a >= +0, finite
__host__ __device__ inline float ef_next_down(float a) { unsigned u; __builtin_memcpy(&u, &a, 4); --u; __builtin_memcpy(&a, &u, 4); return a; }   // a > +0
__host__ __device__ inline float sq_le_max(float T) {   // largest float a with sqrtf(a) <= T
  if (!(T >= 0.0f)) return -1.0f;
  if (T > 3.4028234664e38f) return T;   // +inf: every a up to +inf
  float a = T * T;
  if (a > 3.4028234664e38f) a = 3.4028234664e38f;
  for (int i = 0; i < 8 && a > 0.0f && !(__builtin_sqrtf(a) <= T); ++i) a = ef_next_down(a);
  for (int i = 0; i < 8 && a < 3.4028234664e38f && __builtin_sqrtf(ef_next_up(a)) <= T; ++i) a = ef_next_up(a);
  return a;
}
__host__ __device__ inline float sq_lt_max(float T) {   // largest float a with sqrtf(a) < T
  if (!(T > 0.0f)) return -1.0f;
  float a = T > 3.4028234664e38f ? 3.4028234664e38f : T * T;
  if (a > 3.4028234664e38f) a = 3.4028234664e38f;
  for (int i = 0; i < 8 && a > 0.0f && !(__builtin_sqrtf(a) < T); ++i) a = ef_next_down(a);
  for (int i = 0; i < 8 && a < 3.4028234664e38f && __builtin_sqrtf(ef_next_up(a)) < T; ++i) a = ef_next_up(a);
  return a;
}

__device__ __forceinline__ float ef_expf(float x) {  // x <= 0
  if (x < -87.0f) return 0.0f;
  const float n = rintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  const float e = fmaf(p, r * r, r) + 1.0f;
  return ldexpf(e, (int)n);
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// 16-byte correspondence record; same layout as the reference's DataTerm (types.cuh:81-86)
struct DataTerm {
  short zero_x, zero_y;
  short one_x, one_y;
  float diff;
  unsigned char valid;
  unsigned char pad[3];
};
static_assert(sizeof(DataTerm) == 16, "DataTerm must be 16 bytes");

// ---------------------------------------------------------------------------------------------
// Block reduction of K per-thread fp32 accumulators to one K-vector, wave64 / LDS staged.
// Layout lds[k*(BLOCK+1)+t] (row stride BLOCK+1 => the column sums below are bank-conflict free).
// Summation order is fixed (deterministic run to run): 32-wide segments in lane order, then segments.
// ---------------------------------------------------------------------------------------------
template <int K, int BLOCK>
__device__ __forceinline__ void block_reduce_store(const float (&acc)[K], float* lds, float* out) {
  static_assert(K <= 32 && BLOCK % 64 == 0, "shape");
  constexpr int STR = BLOCK + 1;
  constexpr int SEG = BLOCK / 32;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < K; ++k) lds[k * STR + t] = acc[k];
  __syncthreads();
  const int v = t & 31, g = t >> 5;
  float s = 0.f;
  if (v < K) {
    const float* row = lds + v * STR + g * 32;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += row[i];
  }
  __syncthreads();
  lds[g * 32 + v] = s;
  __syncthreads();
  if (t < K) {
    float r = 0.f;
#pragma unroll
    for (int gg = 0; gg < SEG; ++gg) r += lds[gg * 32 + t];
    out[t] = r;
  }
}
template <int K, int BLOCK>
constexpr int block_reduce_lds_floats() { return K * (BLOCK + 1); }

// integer pair block reduction (exact, order irrelevant) -> one device-scope atomic per block
template <int BLOCK>
__device__ __forceinline__ void block_reduce_atomic_int2(int a, int b, int* lds, int* out2) {
  const int t = threadIdx.x;
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  if ((t & 63) == 0) { lds[(t >> 6) * 2] = a; lds[(t >> 6) * 2 + 1] = b; }
  __syncthreads();
  if (t == 0) {
    int sa = 0, sb = 0;
    for (int w = 0; w < BLOCK / 64; ++w) { sa += lds[w * 2]; sb += lds[w * 2 + 1]; }
    if (sa) atomicAdd(out2, sa);
    if (sb) atomicAdd(out2 + 1, sb);
  }
}

}  // namespace ef
