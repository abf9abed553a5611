// TEST INFRASTRUCTURE — the CUDA/OpenGL interop calls of Core/Utils/RGBDOdometry.cpp:121-257 on the CPU: a "graphics
// resource" is a cudaArray over host memory (the texture's texels), mapping is a no-op.
#pragma once
#include "cuda_runtime.h"
struct cudaGraphicsResource { cudaArray array; };
typedef cudaGraphicsResource* cudaGraphicsResource_t;
enum { cudaGraphicsRegisterFlagsReadOnly = 1 };
static inline cudaError_t cudaGraphicsMapResources(int, cudaGraphicsResource**, void* = nullptr) { return cudaSuccess; }
static inline cudaError_t cudaGraphicsUnmapResources(int, cudaGraphicsResource**, void* = nullptr) { return cudaSuccess; }
static inline cudaError_t cudaGraphicsSubResourceGetMappedArray(cudaArray_t* a, cudaGraphicsResource* r, unsigned, unsigned) {
  *a = &r->array;
  return cudaSuccess;
}
static inline cudaError_t cudaMemcpy2DFromArray(void* dst, size_t dpitch, cudaArray_t src, size_t wOffset, size_t hOffset, size_t width,
                                                size_t height, cudaMemcpyKind) {
  const size_t spitch = (size_t)src->width * src->elem_bytes;
  for (size_t y = 0; y < height; ++y)
    std::memcpy((char*)dst + y * dpitch, (const char*)src->data + (y + hOffset) * spitch + wOffset, width);
  return cudaSuccess;
}
