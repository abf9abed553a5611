// TEST INFRASTRUCTURE — "CUDA on the CPU": the subset of the CUDA programming model that the reference's
// Core/Cuda/{reduce,cudafuncs}.cu + containers/ use, emulated on the host so that THOSE SOURCE FILES can be
// compiled with g++ where they lie (/root/reference) and executed without a GPU (oracle/Makefile, target _ref).
// Nothing here is derived from the reference or from the CUDA headers: it is an independent implementation of the
// documented semantics:
//   * a kernel launch runs every thread of every block; the threads of one block are coroutines scheduled round
//     robin on one host thread, so __syncthreads() and __shfl_down_sync() have their real meaning (block barrier;
//     lock-step exchange inside a 32-lane warp);  warpSize == 32 as on every NVIDIA GPU;
//   * blocks run one after another (so `static __shared__` storage may simply be static);
//   * device memory is host memory; cudaMallocPitch pads rows to 512 bytes like a real device would;
//   * texture objects over cudaArrays: unnormalised coordinates, point filtering, clamp addressing — the only mode
//     the reference sets up (cudafuncs.cu:57-72);
//   * arithmetic: every operation is the IEEE-754 binary32 operation the source spells out (the build uses
//     -ffp-contract=off, i.e. no fused multiply-add is introduced), rsqrtf(x) = 1/sqrtf(x), __float2int_rn =
//     round-half-even with CUDA's NaN -> 0 and saturation.  nvcc's own contraction choices cannot be observed
//     here; see oracle/README.md for what that means for the comparison with the oracle.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

#if !defined(__CUDACC__) && !defined(EFR_HOST_TU)
#define __CUDACC__ 1   // the reference's headers hide their Eigen-dependent host helpers behind this (types.cuh:58);
#endif                 // EFR_HOST_TU: a HOST translation unit of the reference (RGBDOdometry.cpp), compiled against host_on_cpu/
#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __inline__ inline
#define __shared__
#define __constant__

// ---- vector types ------------------------------------------------------------------------------
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct alignas(16) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct alignas(16) int4 { int x, y, z, w; };
struct short2 { short x, y; };
struct uchar3 { unsigned char x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct ushort2 { unsigned short x, y; };
struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline short2 make_short2(short x, short y) { return short2{x, y}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }

// ---- built-in variables ------------------------------------------------------------------------
extern uint3 threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 32;

// ---- device intrinsics / math ------------------------------------------------------------------
using std::isnan;
using std::isinf;
static inline int __float2int_rn(float x) {
  if (x != x) return 0;
  if (x >= 2147483648.0f) return 2147483647;
  if (x <= -2147483648.0f) return (-2147483647 - 1);
  return (int)nearbyintf(x);   // default rounding mode: to nearest, ties to even
}
static inline int __float2int_rz(float x) { return (int)x; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }

namespace cpucuda {
void syncthreads();
uint64_t shfl_down(uint64_t bits, unsigned delta, unsigned width);
void launch(dim3 grid, dim3 block, const std::function<void()>& thread_body);
}  // namespace cpucuda
static inline void __syncthreads() { cpucuda::syncthreads(); }
template <typename T>
static inline T __shfl_down_sync(unsigned /*mask: the reference always passes the full warp*/, T v, unsigned delta, int width = 32) {
  static_assert(sizeof(T) <= 8, "shuffle of a type wider than 64 bits");
  uint64_t b = 0;
  std::memcpy(&b, &v, sizeof(T));
  b = cpucuda::shfl_down(b, delta, (unsigned)width);
  std::memcpy(&v, &b, sizeof(T));
  return v;
}
// kernel<<<grid, block>>>(args);  is rewritten by oracle/Makefile (one perl substitution, on the fly, nothing is
// written to disk) into  CPUCUDA_LAUNCH(grid, block, kernel(args));
#define CPUCUDA_LAUNCH(grid, block, ...) ::cpucuda::launch((grid), (block), [&]() { __VA_ARGS__; })

// ---- runtime API -------------------------------------------------------------------------------
typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorMemoryAllocation = 2, cudaErrorInvalidValue = 11 };
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
static inline const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "cpucuda error"; }
static inline cudaError_t cudaGetLastError() { return cudaSuccess; }
static inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
static inline cudaError_t cudaMalloc(void** p, size_t n) {
  *p = nullptr;
  if (posix_memalign(p, 512, n ? n : 1) != 0) return cudaErrorMemoryAllocation;
  std::memset(*p, 0, n ? n : 1);   // real device memory is uninitialised; zeros make the stale planes of quirk Q3 deterministic
  return cudaSuccess;
}
template <typename T>
static inline cudaError_t cudaMalloc(T** p, size_t n) { return cudaMalloc((void**)p, n); }
static inline cudaError_t cudaMallocPitch(void** p, size_t* pitch, size_t width_bytes, size_t height) {
  *pitch = (width_bytes + 511) / 512 * 512;
  return cudaMalloc(p, *pitch * (height ? height : 1));
}
static inline cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
static inline cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind) { std::memcpy(dst, src, n); return cudaSuccess; }
static inline cudaError_t cudaMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, cudaMemcpyKind) {
  for (size_t r = 0; r < height; ++r) std::memcpy((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
  return cudaSuccess;
}
template <typename T>
static inline cudaError_t cudaMemcpyToSymbol(T& symbol, const void* src, size_t n) { std::memcpy((void*)&symbol, src, n); return cudaSuccess; }

// ---- arrays + texture objects (point filter, clamp, unnormalised, element read: cudafuncs.cu:57-72) ----
struct cudaArray { void* data; int width, height; size_t elem_bytes; };
typedef cudaArray* cudaArray_t;
typedef unsigned long long cudaTextureObject_t;
enum cudaResourceType { cudaResourceTypeArray = 0 };
enum cudaTextureFilterMode { cudaFilterModePoint = 0, cudaFilterModeLinear = 1 };
enum cudaTextureAddressMode { cudaAddressModeWrap = 0, cudaAddressModeClamp = 1 };
enum cudaTextureReadMode { cudaReadModeElementType = 0 };
struct cudaResourceDesc { cudaResourceType resType; struct { struct { cudaArray_t array; } array; } res; };
struct cudaTextureDesc { cudaTextureAddressMode addressMode[3]; cudaTextureFilterMode filterMode; cudaTextureReadMode readMode; int normalizedCoords; };
static inline cudaError_t cudaCreateTextureObject(cudaTextureObject_t* obj, const cudaResourceDesc* res, const cudaTextureDesc* tex, const void*) {
  if (tex->normalizedCoords || tex->filterMode != cudaFilterModePoint || tex->addressMode[0] != cudaAddressModeClamp ||
      tex->addressMode[1] != cudaAddressModeClamp)
    return cudaErrorInvalidValue;   // only the mode the reference uses is implemented
  *obj = (cudaTextureObject_t)(uintptr_t)res->res.array.array;
  return cudaSuccess;
}
static inline cudaError_t cudaDestroyTextureObject(cudaTextureObject_t) { return cudaSuccess; }
template <typename T>
static inline T tex2D(cudaTextureObject_t obj, float x, float y) {
  const cudaArray* a = (const cudaArray*)(uintptr_t)obj;
  int ix = (int)floorf(x), iy = (int)floorf(y);
  ix = ix < 0 ? 0 : (ix >= a->width ? a->width - 1 : ix);
  iy = iy < 0 ? 0 : (iy >= a->height ? a->height - 1 : iy);
  T v;
  std::memcpy(&v, (const char*)a->data + ((size_t)iy * a->width + ix) * a->elem_bytes, sizeof(T));
  return v;
}
