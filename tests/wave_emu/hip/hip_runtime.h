// TEST INFRASTRUCTURE — a stand-in for <hip/hip_runtime.h> that lets g++ compile selected device code of the product for the HOST, so
// that kernels built on cross-lane operations can be executed without a GPU (tests/test_wave_emulation.py).  Covers what the extracted
// code uses and nothing more: the execution-space keywords, the thread indices, a few intrinsics, and DPP quad permutes — each lane of a
// quad runs on its own host thread and meets the others at every cross-lane operation (emu::lane_barrier).
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct float4 { float x, y, z, w; };
struct uchar4 { unsigned char x, y, z, w; };
struct uint3e { unsigned x, y, z; };
namespace emu {
struct Lane {
  uint3e tid, bid, bdim, gdim;
  int quad_lane = 0;                 // position in the quad (lane & 3)
  int* quad_slots = nullptr;         // 4 ints shared by the quad
  std::barrier<>* quad_barrier = nullptr;
};
inline thread_local Lane lane;
}  // namespace emu
#define threadIdx (emu::lane.tid)
#define blockIdx (emu::lane.bid)
#define blockDim (emu::lane.bdim)
#define gridDim (emu::lane.gdim)

inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; std::memcpy(&v, &f, 4); return v; }
using std::isnan;

// v_mov_b32 with a DPP quad_perm control (dpp_ctrl 0x00..0xFF: two bits per destination lane of the quad = its source lane); row/bank
// masks 0xF and bound_ctrl as the product uses them
inline int __builtin_amdgcn_mov_dpp(int v, int dpp_ctrl, int row_mask, int bank_mask, bool) {
  emu::Lane& L = emu::lane;
  if (dpp_ctrl < 0 || dpp_ctrl > 0xFF || row_mask != 0xF || bank_mask != 0xF || !L.quad_barrier) std::abort();
  L.quad_slots[L.quad_lane] = v;
  L.quad_barrier->arrive_and_wait();
  const int r = L.quad_slots[(dpp_ctrl >> (2 * L.quad_lane)) & 3];
  L.quad_barrier->arrive_and_wait();
  return r;
}

// names the included product headers mention in code this harness never runs
typedef void* hipStream_t;
typedef void* hipEvent_t;
inline void __syncthreads() { std::abort(); }
inline int __shfl_down(int, int, int) { std::abort(); }
inline float __shfl_down(float, int, int) { std::abort(); }
inline unsigned __shfl_down(unsigned, int, int) { std::abort(); }   // (model_maps_use_fill's tally: never taken here, tally_image is null)
#ifndef __shared__
#define __shared__ static
#endif
inline int atomicAdd(int*, int) { std::abort(); }

// ---- a whole wavefront: 64 host threads that meet at every cross-lane operation (emu::wave) ----
namespace emu {
struct Wave {
  std::barrier<>* barrier = nullptr;     // 64 participants
  unsigned long long* slots = nullptr;   // 64 x 8 bytes
  int lane = 0;
};
inline thread_local Wave wave;
inline unsigned long long exchange(unsigned long long v, int src) {
  Wave& W = wave;
  if (!W.barrier) std::abort();
  W.slots[W.lane] = v;
  W.barrier->arrive_and_wait();
  const unsigned long long r = W.slots[src & 63];
  W.barrier->arrive_and_wait();
  return r;
}
}  // namespace emu
inline double __shfl(double v, int src, int width) { if (width != 64) std::abort(); unsigned long long b; std::memcpy(&b, &v, 8); b = emu::exchange(b, src); std::memcpy(&v, &b, 8); return v; }
inline int __shfl(int v, int src, int width) { if (width != 64) std::abort(); return (int)(unsigned)emu::exchange((unsigned)v, src); }
inline float __shfl(float v, int src, int width) { return __int_as_float(__shfl(__float_as_int(v), src, width)); }
inline int __builtin_amdgcn_readlane(int v, int src) { return (int)(unsigned)emu::exchange((unsigned)v, src); }   // src is wave-uniform
inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(unsigned)emu::exchange((unsigned)v, 0); }
inline void __builtin_amdgcn_wave_barrier() { emu::wave.barrier->arrive_and_wait(); }
#define __builtin_amdgcn_fence(...) ((void)0)   /* the barrier of the emulation orders the lanes' memory accesses */
inline int __double2hiint(double d) { unsigned long long b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline int __double2loint(double d) { unsigned long long b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffffu); }
inline double __hiloint2double(int hi, int lo) { const unsigned long long b = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo; double d; std::memcpy(&d, &b, 8); return d; }
