// TEST INFRASTRUCTURE — the product's sq_le_max / sq_lt_max (elasticfusion_amd/csrc/ef_device.hpp), compiled for the host over the stand-in
// <hip/hip_runtime.h> beside this file: "sqrtf(a) <= T" must be "a <= sq_le_max(T)" and "sqrtf(a) < T" must be "a <= sq_lt_max(T)" for
// every float a (the ICP gates of the two-visits-per-lane rows compare squared norms, tests/test_norm_gates.py).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstring>
#include "ef_device.hpp"
using namespace ef;

static long check_threshold(float T, long& n) {
  long bad = 0;
  const float le = sq_le_max(T), lt = sq_lt_max(T);
  for (int side = 0; side < 2; ++side) {
    float b = side ? lt : le;
    if (!(b >= 0)) b = 0;
    float a = b;
    for (int i = 0; i < 40 && a > 0; ++i) a = ef_next_down(a);
    for (int i = 0; i < 80; ++i) {
      bad += ((std::sqrt(a) <= T) != (a <= le)) + ((std::sqrt(a) < T) != (a <= lt));
      ++n;
      if (a >= 3.4028234664e38f) break;
      a = ef_next_up(a);
    }
  }
  const float special[] = {0.f, INFINITY, NAN, 3.4028234664e38f, 1e-45f, 1.17549435e-38f, 1.0f};
  for (float a : special) { bad += ((std::sqrt(a) <= T) != (a <= le)) + ((std::sqrt(a) < T) != (a <= lt)); ++n; }
  return bad;
}

extern "C" long norm_gates_check(int random_thresholds, long* comparisons, float* le_01, float* lt_sin20) {
  long bad = 0, n = 0;
  const float fixed[] = {0.10f, std::sin(20.f * 3.14159254f / 180.f), 0.f, -0.f, -1.f, INFINITY, NAN, 1e-30f, 1e-20f, 1e-19f, 1e19f, 1.8446743e19f,
                         1.8446744e19f, 2e19f, 3e38f, 3.4028234664e38f, 1e-45f, 1.17549435e-38f, 1.0f, 2.0f, 0.5f};
  for (float T : fixed) bad += check_threshold(T, n);
  uint64_t x = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < random_thresholds; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    const uint32_t u = (uint32_t)(x >> 20) & 0x7fffffffu;   // every non-negative bit pattern: normals, subnormals, inf, NaNs
    float T;
    std::memcpy(&T, &u, 4);
    bad += check_threshold(T, n);
  }
  *comparisons = n;
  *le_01 = sq_le_max(0.10f);
  *lt_sin20 = sq_lt_max(std::sin(20.f * 3.14159254f / 180.f));
  return bad;
}
