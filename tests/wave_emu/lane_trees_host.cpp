// TEST INFRASTRUCTURE — the fast order's wavefront trees (lane_tree64, lane_trees: ef_track_fast.inc; their cross-lane helpers: ef_track_kernels.hip),
// cut out of the product's sources by tests/test_lane_trees_emulation.py (LANE_TREES_SOURCE) and executed by 64 host threads, one per lane,
// that meet at every cross-lane operation.  The cross-lane builtins are emulated as the product's comments (and the GPU suite's bit-exact
// results) say they behave:
//   mov_dpp quad_perm            lane i reads lane (i & ~3) + perm[i & 3]
//   update_dpp row_shl:n         lane i reads lane i + n of its 16-lane row, 0 beyond the row (bound_ctrl)
//   permlane16_swap(a, a)[1]     rows 0 / 2 read rows 1 / 3 ([0]: rows 1 / 3 read rows 0 / 2)
//   permlane32_swap(a, a)[1]     lanes 0..31 read lanes 32..63 ([0]: lanes 32..63 read lanes 0..31)
//   ds_swizzle, bit mode         lane i reads lane ((i & and) | or) ^ xor of its 32-lane half
// What this pins is the LOGIC of lane_trees — that splitting the accumulators between the lanes of a pair, level by level, forms for every
// accumulator exactly the adjacent-pair tree lane_tree64 forms (same operands per addition, possibly swapped) — on the CPU, for good.
#include <barrier>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define __device__
#define __host__
#define __forceinline__ inline
struct uint3e { unsigned x, y, z; };
namespace emu {
struct Wave { std::barrier<>* barrier; unsigned* slots; int lane; };
inline thread_local Wave wave;
inline thread_local uint3e tid;
template <typename F> inline unsigned gather(unsigned v, F src_of) {   // every lane publishes v, then reads the lane src_of(lane) (< 0: zero)
  Wave& W = wave;
  W.slots[W.lane] = v;
  W.barrier->arrive_and_wait();
  const int s = src_of(W.lane);
  const unsigned r = s < 0 ? 0u : W.slots[s & 63];
  W.barrier->arrive_and_wait();
  return r;
}
}  // namespace emu
#define threadIdx (emu::tid)
inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
inline float __uint_as_float(unsigned v) { float f; std::memcpy(&f, &v, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned v; std::memcpy(&v, &f, 4); return v; }
inline int __builtin_amdgcn_mov_dpp(int v, int ctrl, int row_mask, int bank_mask, bool) {
  if (ctrl < 0 || ctrl > 0xFF || row_mask != 0xF || bank_mask != 0xF) std::abort();
  return (int)emu::gather((unsigned)v, [ctrl](int l) { return (l & ~3) + ((ctrl >> (2 * (l & 3))) & 3); });
}
inline int __builtin_amdgcn_update_dpp(int, int v, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  if (ctrl < 0x101 || ctrl > 0x10F || row_mask != 0xF || bank_mask != 0xF || !bound_ctrl) std::abort();
  const int n = ctrl - 0x100;
  return (int)emu::gather((unsigned)v, [n](int l) { return ((l & 15) + n < 16) ? l + n : -1; });
}
struct uint2e { unsigned v[2]; unsigned operator[](int i) const { return v[i]; } };
inline uint2e __builtin_amdgcn_permlane16_swap(unsigned a, unsigned b, bool, bool) {
  if (a != b) std::abort();   // the product only calls it with both operands equal
  uint2e r;
  r.v[0] = emu::gather(a, [](int l) { return (l & 16) ? l - 16 : l; });
  r.v[1] = emu::gather(a, [](int l) { return (l & 16) ? l : l + 16; });
  return r;
}
inline uint2e __builtin_amdgcn_permlane32_swap(unsigned a, unsigned b, bool, bool) {
  if (a != b) std::abort();
  uint2e r;
  r.v[0] = emu::gather(a, [](int l) { return (l & 32) ? l - 32 : l; });
  r.v[1] = emu::gather(a, [](int l) { return (l & 32) ? l : l + 32; });
  return r;
}
inline int __builtin_amdgcn_ds_swizzle(int v, int pattern) {
  if (pattern & 0x8000) std::abort();   // bit mode only
  const int am = pattern & 31, om = (pattern >> 5) & 31, xm = (pattern >> 10) & 31;
  return (int)emu::gather((unsigned)v, [=](int l) { return (l & 32) | ((((l & 31) & am) | om) ^ xm); });
}
#define EF_FMA(a, b, c) fmaf((a), (b), (c))
constexpr int SE3_ACCS = 29, SO3_ACCS = 11;
#include LANE_TREES_SOURCE

// in: [64 lanes][na] floats.  out_tree64[a] = lane 0's lane_tree64 of accumulator a; out_trees[a] = what lane_trees_store leaves for a
extern "C" void run_lane_trees(const float* in, int na, float* out_tree64, float* out_trees) {
  std::barrier<> bar(64);
  std::vector<unsigned> slots(64, 0u);
  std::vector<std::thread> th;
  for (int l = 0; l < 64; ++l)
    th.emplace_back([&, l] {
      emu::wave = emu::Wave{&bar, slots.data(), l};
      emu::tid = uint3e{(unsigned)l, 0, 0};
      for (int a = 0; a < na; ++a) {
        const float x = lane_tree64(in[l * na + a]);
        if (l == 0) out_tree64[a] = x;
      }
      if (na == SE3_ACCS) {
        float acc[SE3_ACCS];
        for (int a = 0; a < na; ++a) acc[a] = in[l * na + a];
        lane_trees_store<SE3_ACCS>(acc, out_trees);
      } else {
        float acc[SO3_ACCS];
        for (int a = 0; a < na; ++a) acc[a] = in[l * na + a];
        lane_trees_store<SO3_ACCS>(acc, out_trees);
      }
    });
  for (auto& t : th) t.join();
}
