// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// Shared scalar conventions of the restatement.  The reference's device arithmetic lives in
// nvcc- and NVIDIA-GLSL-compiled code whose contraction / approximate-intrinsic choices cannot be
// observed here, so the oracle *specifies* them once, and the HIP kernels under
// elasticfusion_amd/csrc restate the same specification independently (everything else — operation order, gates,
// summation trees — is pinned bit for bit against the reference's sources compiled for the CPU, oracle/README.md):
//   * fp32 everywhere the reference uses float; no implicit contraction (-ffp-contract=off);
//     dot / cross / accumulate use explicit fmaf in the order written below;
//   * 1/sqrt is 1.0f / sqrtf(x) (both correctly rounded) where the reference uses rsqrtf /
//     normalize();
//   * exp() in the GLSL passes is efo_expf (Cephes-style, <1 ulp, built only from IEEE ops so
//     CPU and GPU agree bit-for-bit);
//   * __float2int_rn = round-half-even with NaN -> 0 and saturation (CUDA semantics).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

// Fused multiply-adds of the tracking side appear only through EFO_FMA.  The default build fuses them (the
// specification the HIP kernels restate).  -DEFO_NO_FMA builds the same restatement with every EFO_FMA split into an
// IEEE multiply and an IEEE add: that variant must agree BIT FOR BIT with the reference's own sources compiled
// without contraction (oracle/_ref, tests/test_oracle_vs_reference.py), which pins operation order, summation
// tree and every gate of the restatement against the reference; the only thing left to specification is which
// multiply-add pairs are fused.
#ifdef EFO_NO_FMA
#define EFO_FMA(a, b, c) ((a) * (b) + (c))
#else
#define EFO_FMA(a, b, c) fmaf((a), (b), (c))
#endif

namespace efo {

// cpu_baseline leg of bench.py, "all host cores" variant: the embarrassingly parallel loops of the
// restatement (bilateral rows, the 64 independent blocks of each two-stage reduction) are split over efo::threads()
// std::threads.  Every element is computed by the same code in the same order, so results do not depend on the count.
inline int& threads() { static int n = 1; return n; }
template <typename F>
inline void parallel_for(int n, F&& body) {   // body(begin, end)
  const int nt = threads() < n ? threads() : n;
  if (nt <= 1) { body(0, n); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < nt; ++t) pool.emplace_back([&, t] { body((int)((long long)n * t / nt), (int)((long long)n * (t + 1) / nt)); });
  for (auto& th : pool) th.join();
}

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

inline float qnan() {
  uint32_t u = 0x7fffffffu;  // CUDART_NAN_F bit pattern used by the reference (cudafuncs.cu:147)
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline float dot(f3 a, f3 b) { return EFO_FMA(a.z, b.z, EFO_FMA(a.y, b.y, a.x * b.x)); }  // operators.cuh:71
inline f3 cross(f3 a, f3 b) {                                                        // operators.cuh:67
  return {EFO_FMA(a.y, b.z, -(a.z * b.y)), EFO_FMA(a.z, b.x, -(a.x * b.z)), EFO_FMA(a.x, b.y, -(a.y * b.x))};
}
inline float norm(f3 a) { return sqrtf(dot(a, a)); }                                 // operators.cuh:75
inline f3 normalized(f3 a) {                                                         // operators.cuh:79
  float rn = 1.0f / sqrtf(dot(a, a));
  return {a.x * rn, a.y * rn, a.z * rn};
}
struct m33 { f3 r[3]; };  // mat33, types.cuh:67-79 (row-major rows)
inline f3 mul(const m33& m, f3 a) { return {dot(m.r[0], a), dot(m.r[1], a), dot(m.r[2], a)}; }  // operators.cuh:84
inline m33 m33_from(const float* p) { return m33{{{p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}}}; }

inline int f2i_rn(float x) {  // __float2int_rn
  if (std::isnan(x)) return 0;
  if (x >= 2147483648.0f) return std::numeric_limits<int>::max();
  if (x <= -2147483648.0f) return std::numeric_limits<int>::min();
  return (int)nearbyintf(x);
}

// exp for x <= 0 (bilateral weight, surfel confidence). Pure IEEE ops => bit-identical on gfx950.
inline float efo_expf(float x) {
  if (x < -87.0f) return 0.0f;
  float n = nearbyintf(x * 1.44269504088896341f);
  float r = fmaf(n, -0.693359375f, x);
  r = fmaf(n, 2.12194440e-4f, r);
  float p = 1.9875691500e-4f;
  p = fmaf(p, r, 1.3981999507e-3f);
  p = fmaf(p, r, 8.3334519073e-3f);
  p = fmaf(p, r, 4.1665795894e-2f);
  p = fmaf(p, r, 1.6666665459e-1f);
  p = fmaf(p, r, 5.0000001201e-1f);
  float e = fmaf(p, r * r, r) + 1.0f;
  return ldexpf(e, (int)n);
}

// DataTerm, types.cuh:81-86 (16 bytes)
struct DataTerm {
  int16_t zero_x, zero_y;
  int16_t one_x, one_y;
  float diff;
  uint8_t valid;
  uint8_t pad[3];
};
static_assert(sizeof(DataTerm) == 16, "DataTerm layout");

}  // namespace efo
