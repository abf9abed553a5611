// HIP kernels (gfx950, wave64) for the tracking pyramid and the Gauss-Newton reductions.
// Reference behaviour: Core/Cuda/cudafuncs.cu (image operators), Core/Cuda/reduce.cu (icpStep,
// computeRgbResidual, rgbStep, so3Step), Core/Utils/RGBDOdometry.cpp (driver).  The CUDA versions are
// warp32 / 64x256-thread grid-stride / host-in-the-loop.  Here the fp32 sums keep the reference's summation
// ORDER (so results are bit-identical to the reference's tree, see "Reference-order fp32 reductions" below) but
// not its schedule: a quad of lanes per virtual thread, Jacobian rows transposed inside the quad and accumulated as
// rank-1 updates on the matrix pipe (v_mfma_f32_4x4x1, one fmaf per element), lane swaps for the trees; the 6x6 solve
// + SE(3) update run in a one-workgroup device kernel so the 19 iterations are enqueued back to back with no host
// round trip.
#include "ef_device.hpp"
#include "ef_linalg_dev.hpp"
#include "ef_solve_dev.hpp"
#include <stddef.h>
#include <mutex>
#include <hip/hip_ext.h>
#include "ef_track.hpp"

using namespace ef;

namespace eft {

namespace {

constexpr int TILE_X = 64, TILE_Y = 4;  // 256-thread image tile: one wave per row segment, coalesced in x

inline dim3 tile_grid(int cols, int rows) { return dim3((cols + TILE_X - 1) / TILE_X, (rows + TILE_Y - 1) / TILE_Y); }
inline dim3 tile_block() { return dim3(TILE_X, TILE_Y); }

// ------------------------------------------------------------------------------------------
// image operators
// ------------------------------------------------------------------------------------------

// pyrDownGaussKernel, cudafuncs.cu:75-109.  Round 5: all 25 taps are LOADED first (clamped addresses, one round trip) and the clipped
// loops of the reference become predicated additions in the same order — the loops with their run-time bounds kept the compiler from
// unrolling, and 25 dependent load -> use round trips (~12 us for a 320 x 240 image) were the whole cost of the pyramid kernels.
__device__ __forceinline__ void pyr_down_u16_px(const uint16_t* __restrict__ src, int scols, int srows, uint16_t* __restrict__ dst, int x, int y) {
  const int dcols = scols / 2;
  const int D = 5;
  const float sigma_color = 30.f;
  int vals[D * D];
#pragma unroll
  for (int yi = -D / 2; yi <= D / 2; ++yi)
#pragma unroll
    for (int xi = -D / 2; xi <= D / 2; ++xi)
      vals[(yi + D / 2) * D + xi + D / 2] = src[min(max(2 * y + yi, 0), srows - 1) * scols + min(max(2 * x + xi, 0), scols - 1)];
  const int center = vals[(D / 2) * D + D / 2];
  float sum = 0, wall = 0;
#pragma unroll
  for (int yi = -D / 2; yi <= D / 2; ++yi)
#pragma unroll
    for (int xi = -D / 2; xi <= D / 2; ++xi) {
      // the reference's bounds: max(0, 2 x - 2) - 2 x <= xi < min(scols, 2 x + 3) - 2 x (and the same in y)
      const bool in = 2 * x + xi >= 0 && 2 * x + xi < scols && 2 * y + yi >= 0 && 2 * y + yi < srows;
      const int val = vals[(yi + D / 2) * D + xi + D / 2];
      if (in && abs(val - center) < 3 * sigma_color) {
        const int ax = abs(xi), ay = abs(yi);
        const float wx = ax == 0 ? 0.375f : (ax == 1 ? 0.25f : 0.0625f);
        const float wy = ay == 0 ? 0.375f : (ay == 1 ? 0.25f : 0.0625f);
        sum += (float)val * wx * wy;
        wall += wx * wy;
      }
    }
  dst[y * dcols + x] = (uint16_t) static_cast<int>(sum / wall);
}
__global__ void k_pyr_down_u16(const uint16_t* __restrict__ src, int scols, int srows, uint16_t* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= scols / 2 || y >= srows / 2) return;
  pyr_down_u16_px(src, scols, srows, dst, x, y);
}

// computeVmapKernel, cudafuncs.cu:123-149
__device__ __forceinline__ bool vmap_point(int d, int u, int v, float fx_inv, float fy_inv, float cx, float cy, float cutoff, f3& p) {
  const float z = d / 1000.f;
  if (z != 0 && z < cutoff) {
    p = {z * (u - cx) * fx_inv, z * (v - cy) * fy_inv, z};
    return true;
  }
  return false;
}
__global__ void k_create_vmap(const uint16_t* __restrict__ depth, int cols, int rows, float fx_inv, float fy_inv, float cx,
                              float cy, float cutoff, float* __restrict__ vmap) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y * blockDim.y + threadIdx.y;
  if (u >= cols || v >= rows) return;
  f3 p;
  if (vmap_point(depth[v * cols + u], u, v, fx_inv, fy_inv, cx, cy, cutoff, p)) {
    vmap[v * cols + u] = p.x;
    vmap[(v + rows) * cols + u] = p.y;
    vmap[(v + 2 * rows) * cols + u] = p.z;
  } else {
    vmap[v * cols + u] = qnan();
  }
}
// computeNmapKernel, cudafuncs.cu:170-204
__global__ void k_create_nmap(const float* __restrict__ vmap, int cols, int rows, float* __restrict__ nmap) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x, v = blockIdx.y * blockDim.y + threadIdx.y;
  if (u >= cols || v >= rows) return;
  if (u == cols - 1 || v == rows - 1) { nmap[v * cols + u] = qnan(); return; }
  const float x00 = vmap[v * cols + u], x01 = vmap[v * cols + u + 1], x10 = vmap[(v + 1) * cols + u];
  if (!isnan(x00) && !isnan(x01) && !isnan(x10)) {
    const f3 v00{x00, vmap[(v + rows) * cols + u], vmap[(v + 2 * rows) * cols + u]};
    const f3 v01{x01, vmap[(v + rows) * cols + u + 1], vmap[(v + 2 * rows) * cols + u + 1]};
    const f3 v10{x10, vmap[(v + 1 + rows) * cols + u], vmap[(v + 1 + 2 * rows) * cols + u]};
    const f3 r = normalized(cross(v01 - v00, v10 - v00));
    nmap[v * cols + u] = r.x;
    nmap[(v + rows) * cols + u] = r.y;
    nmap[(v + 2 * rows) * cols + u] = r.z;
  } else {
    nmap[v * cols + u] = qnan();
  }
}

// Fused createVMap + createNMap for all three levels in ONE launch (blockIdx.z = level): the normal is
// evaluated straight from the three depth samples with the same operations createVMap would have used,
// so the planar maps come out bit-identical to the two-kernel path without re-reading the vertex map.
struct VNLevels {
  const uint16_t* depth[NUM_PYRS];
  float* vmap[NUM_PYRS];
  float* nmap[NUM_PYRS];
  int cols[NUM_PYRS], rows[NUM_PYRS];
  Intr k[NUM_PYRS];
  float cutoff;
};
__device__ __forceinline__ void vmap_nmap_px(const VNLevels& L, int l, int u, int v) {
  const int cols = L.cols[l], rows = L.rows[l];
  if (u >= cols || v >= rows) return;
  const uint16_t* __restrict__ depth = L.depth[l];
  float* __restrict__ vmap = L.vmap[l];
  float* __restrict__ nmap = L.nmap[l];
  const float fx_inv = 1.f / L.k[l].fx, fy_inv = 1.f / L.k[l].fy, cx = L.k[l].cx, cy = L.k[l].cy;
  f3 v00, v01, v10;
  const bool ok00 = vmap_point(depth[v * cols + u], u, v, fx_inv, fy_inv, cx, cy, L.cutoff, v00);
  if (ok00) {
    vmap[v * cols + u] = v00.x;
    vmap[(v + rows) * cols + u] = v00.y;
    vmap[(v + 2 * rows) * cols + u] = v00.z;
  } else {
    vmap[v * cols + u] = qnan();
  }
  if (u == cols - 1 || v == rows - 1) { nmap[v * cols + u] = qnan(); return; }
  const bool ok01 = vmap_point(depth[v * cols + u + 1], u + 1, v, fx_inv, fy_inv, cx, cy, L.cutoff, v01);
  const bool ok10 = vmap_point(depth[(v + 1) * cols + u], u, v + 1, fx_inv, fy_inv, cx, cy, L.cutoff, v10);
  if (ok00 && ok01 && ok10) {
    const f3 r = normalized(cross(v01 - v00, v10 - v00));
    nmap[v * cols + u] = r.x;
    nmap[(v + rows) * cols + u] = r.y;
    nmap[(v + 2 * rows) * cols + u] = r.z;
  } else {
    nmap[v * cols + u] = qnan();
  }
}
__global__ void k_vmap_nmap_levels(const VNLevels L) {
  vmap_nmap_px(L, blockIdx.z, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y);
}

// tranformMapsKernel, cudafuncs.cu:221-270 (R, t read from device memory)
__global__ void k_transform_maps(const float* __restrict__ vsrc, const float* __restrict__ nsrc, int cols, int rows,
                                 const float* __restrict__ Rp, const float* __restrict__ tp, float* vdst, float* ndst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  const m33 R = m33_load(Rp);
  const f3 t{tp[0], tp[1], tp[2]};
  float ox = qnan();
  const float vx = vsrc[y * cols + x];
  if (!isnan(vx)) {
    const f3 vs{vx, vsrc[(y + rows) * cols + x], vsrc[(y + 2 * rows) * cols + x]};
    const f3 vd = mul(R, vs) + t;
    vdst[(y + rows) * cols + x] = vd.y;
    vdst[(y + 2 * rows) * cols + x] = vd.z;
    ox = vd.x;
  }
  vdst[y * cols + x] = ox;
  float onx = qnan();
  const float nx = nsrc[y * cols + x];
  if (!isnan(nx)) {
    const f3 ns{nx, nsrc[(y + rows) * cols + x], nsrc[(y + 2 * rows) * cols + x]};
    const f3 nd = mul(R, ns);
    ndst[(y + rows) * cols + x] = nd.y;
    ndst[(y + 2 * rows) * cols + x] = nd.z;
    onx = nd.x;
  }
  ndst[y * cols + x] = onx;
}

// copyMapsKernelTex, cudafuncs.cu:295-350
__global__ void k_copy_maps(const float4* __restrict__ vtex, const float4* __restrict__ ntex, int cols, int rows,
                            float4* __restrict__ vmaps_tmp, float* __restrict__ vmap, float* __restrict__ nmap) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  const float4 vs = vtex[y * cols + x];
  const float4 ns = ntex[y * cols + x];
  vmaps_tmp[y * cols + x] = vs;
  f3 vd{qnan(), qnan(), qnan()}, nd{qnan(), qnan(), qnan()};
  if (!(vs.z == 0)) {
    vd = {vs.x, vs.y, vs.z};
    nd = {ns.x, ns.y, ns.z};
  }
  vmap[y * cols + x] = vd.x;
  vmap[(y + rows) * cols + x] = vd.y;
  vmap[(y + 2 * rows) * cols + x] = vd.z;
  nmap[y * cols + x] = nd.x;
  nmap[(y + rows) * cols + x] = nd.y;
  nmap[(y + 2 * rows) * cols + x] = nd.z;
}

// resizeMapKernel<normalize>, cudafuncs.cu:413-465
template <bool NORMALIZE>
__global__ void k_resize_map(const float* __restrict__ in, int scols, int srows, float* __restrict__ out) {
  const int dcols = scols / 2, drows = srows / 2;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= dcols || y >= drows) return;
  const int xs = x * 2, ys = y * 2;
  const float x00 = in[ys * scols + xs], x01 = in[ys * scols + xs + 1], x10 = in[(ys + 1) * scols + xs], x11 = in[(ys + 1) * scols + xs + 1];
  if (isnan(x00) || isnan(x01) || isnan(x10) || isnan(x11)) { out[y * dcols + x] = qnan(); return; }
  f3 n;
  n.x = (x00 + x01 + x10 + x11) / 4;
  const float* py = in + (size_t)srows * scols;
  const float* pz = in + (size_t)2 * srows * scols;
  n.y = (py[ys * scols + xs] + py[ys * scols + xs + 1] + py[(ys + 1) * scols + xs] + py[(ys + 1) * scols + xs + 1]) / 4;
  n.z = (pz[ys * scols + xs] + pz[ys * scols + xs + 1] + pz[(ys + 1) * scols + xs] + pz[(ys + 1) * scols + xs + 1]) / 4;
  if (NORMALIZE) n = normalized(n);
  out[y * dcols + x] = n.x;
  out[(y + drows) * dcols + x] = n.y;
  out[(y + 2 * drows) * dcols + x] = n.z;
}

__device__ __forceinline__ float gauss25(int idx) {  // {1 4 6 4 1} (x) {1 4 6 4 1}, cudafuncs.cu:498-499
  const int r = idx / 5, c = idx - r * 5;
  const float wr = r == 2 ? 6.f : ((r == 1 || r == 3) ? 4.f : 1.f);
  const float wc = c == 2 ? 6.f : ((c == 1 || c == 3) ? 4.f : 1.f);
  return wr * wc;
}

// {1 4 6 4 1}[r] as gauss25 indexes it (r counted from the END of the clipped window: quirk Q7), r in 0..4
__device__ __forceinline__ float gauss5(int r) { return r == 2 ? 6.f : ((r == 1 || r == 3) ? 4.f : 1.f); }
// pyrDownKernelGaussF, cudafuncs.cu:383-411 (quirk Q7 kept).  Round 5: taps loaded first, predicated additions in the reference's order (see
// pyr_down_u16_px); the weight of window position (oy, ox) is gauss25((ty - cy - 1) * 5 + (tx - cx - 1)) as before.
__device__ __forceinline__ void pyr_down_gauss_f_px(const float* __restrict__ src, int scols, int srows, float* __restrict__ dst, int x, int y) {
  const int dcols = scols / 2, D = 5;
  const int tx = min(2 * x - D / 2 + D, scols - 1), ty = min(2 * y - D / 2 + D, srows - 1);
  float vals[D * D];
#pragma unroll
  for (int oy = 0; oy < D; ++oy)
#pragma unroll
    for (int ox = 0; ox < D; ++ox)
      vals[oy * D + ox] = src[min(max(2 * y - D / 2 + oy, 0), srows - 1) * scols + min(max(2 * x - D / 2 + ox, 0), scols - 1)];
  float sum = 0;
  int count = 0;
#pragma unroll
  for (int oy = 0; oy < D; ++oy)
#pragma unroll
    for (int ox = 0; ox < D; ++ox) {
      const int cy = 2 * y - D / 2 + oy, cx = 2 * x - D / 2 + ox;
      const float s = vals[oy * D + ox];
      if (cy >= 0 && cy < ty && cx >= 0 && cx < tx && !isnan(s)) {
        const float g = gauss5(ty - cy - 1) * gauss5(tx - cx - 1);
        sum += s * g;
        count = (int)((float)count + g);
      }
    }
  dst[y * dcols + x] = (float)(sum / (float)count);
}
__global__ void k_pyr_down_gauss_f(const float* __restrict__ src, int scols, int srows, float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= scols / 2 || y >= srows / 2) return;
  pyr_down_gauss_f_px(src, scols, srows, dst, x, y);
}

// pyrDownKernelIntensityGauss, cudafuncs.cu:512-542 (taps loaded first, as above)
__device__ __forceinline__ void pyr_down_uchar_gauss_px(const uint8_t* __restrict__ src, int scols, int srows, uint8_t* __restrict__ dst, int x, int y) {
  const int dcols = scols / 2, D = 5;
  const int tx = min(2 * x - D / 2 + D, scols - 1), ty = min(2 * y - D / 2 + D, srows - 1);
  int vals[D * D];
#pragma unroll
  for (int oy = 0; oy < D; ++oy)
#pragma unroll
    for (int ox = 0; ox < D; ++ox)
      vals[oy * D + ox] = src[min(max(2 * y - D / 2 + oy, 0), srows - 1) * scols + min(max(2 * x - D / 2 + ox, 0), scols - 1)];
  float sum = 0;
  int count = 0;
#pragma unroll
  for (int oy = 0; oy < D; ++oy)
#pragma unroll
    for (int ox = 0; ox < D; ++ox) {
      const int cy = 2 * y - D / 2 + oy, cx = 2 * x - D / 2 + ox;
      const int sv = vals[oy * D + ox];
      if (cy >= 0 && cy < ty && cx >= 0 && cx < tx && sv > 0) {
        const float g = gauss5(ty - cy - 1) * gauss5(tx - cx - 1);
        sum += (float)sv * g;
        count = (int)((float)count + g);
      }
    }
  const float q = sum / (float)count;
  const int iv = (q != q) ? 0 : (int)fminf(fmaxf(q, 0.0f), 255.0f);
  dst[y * dcols + x] = (uint8_t)iv;
}
__global__ void k_pyr_down_uchar_gauss(const uint8_t* __restrict__ src, int scols, int srows, uint8_t* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= scols / 2 || y >= srows / 2) return;
  pyr_down_uchar_gauss_px(src, scols, srows, dst, x, y);
}
// One launch for one pyramid step of SEVERAL images of the same size (blockIdx.z = job): the frame's u16 depth, the
// model's f32 depth and both intensity images are each only a 5-9 us, launch-bound kernel on their own.
struct PyrJobs {
  const void* src[4];
  void* dst[4];
  int type[4];   // 0: u16 (pyrDownGaussKernel)  1: f32 (pyrDownKernelGaussF)  2: u8 (pyrDownKernelIntensityGauss)
  int scols, srows;
};
__global__ void k_pyr_down_multi(const PyrJobs J) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= J.scols / 2 || y >= J.srows / 2) return;
  const int j = blockIdx.z;
  if (J.type[j] == 0) pyr_down_u16_px((const uint16_t*)J.src[j], J.scols, J.srows, (uint16_t*)J.dst[j], x, y);
  else if (J.type[j] == 1) pyr_down_gauss_f_px((const float*)J.src[j], J.scols, J.srows, (float*)J.dst[j], x, y);
  else pyr_down_uchar_gauss_px((const uint8_t*)J.src[j], J.scols, J.srows, (uint8_t*)J.dst[j], x, y);
}
// verticesToDepthKernel, cudafuncs.cu:564-574
__global__ void k_vertices_to_depth(const float4* __restrict__ vmaps_tmp, int cols, int rows, float cutOff, float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  const float z = vmaps_tmp[y * cols + x].z;
  dst[y * cols + x] = (z > cutOff || z <= 0) ? qnan() : z;
}

// bgr2IntensityKernel, cudafuncs.cu:584-596; CH = 4 (RGBA8 texel) or 3 (packed RGB as uploaded)
template <int CH>
__global__ void k_bgr_to_intensity(const uint8_t* __restrict__ src, int n, uint8_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* s = src + (size_t)i * CH;
  dst[i] = intensity_of((float)s[0], (float)s[1], (float)s[2]);
}

// applyKernel, cudafuncs.cu:612-637 (quirk Q6 kept: the kernel index k counts down over the taps the clipped loops VISIT, so it shifts
// at the image border).  Round 5: the nine taps come preloaded (v[dj + 1][di + 1], anything where the tap is outside the image) and the
// loops become predicated additions in the reference's order; interior pixels see compile-time weights.
__device__ __forceinline__ float sobel_gsx(int k) {   // gsx[k] of cudafuncs.cu:612-637
  const float a = 0.52201f, b = 0.79451f;
  return k == 8 ? -a : k == 7 ? 0.00000f : k == 6 ? a : k == 5 ? -b : k == 4 ? -0.00000f : k == 3 ? b : k == 2 ? -a : k == 1 ? 0.00000f : a;
}
__device__ __forceinline__ float sobel_gsy(int k) {
  const float a = 0.52201f, b = 0.79451f;
  return k == 8 ? -a : k == 7 ? -b : k == 6 ? -a : (k >= 3) ? 0.00000f : k == 2 ? a : k == 1 ? b : a;
}
__device__ __forceinline__ void sobel_taps(const float (&v)[3][3], int cols, int rows, int x, int y, int16_t& dx, int16_t& dy) {
  const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
  const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
  float dxVal = 0, dyVal = 0;
  if (x >= 1 && y >= 1 && x <= cols - 2 && y <= rows - 2) {
    int k = 8;
#pragma unroll
    for (int dj = 0; dj < 3; ++dj)
#pragma unroll
      for (int di = 0; di < 3; ++di) {
        dxVal += v[dj][di] * gsx[k];
        dyVal += v[dj][di] * gsy[k];
        --k;
      }
  } else {
    const int jlo = max(y - 1, 0), ilo = max(x - 1, 0), ni = min(x + 1, cols - 1) - ilo + 1;
#pragma unroll
    for (int dj = 0; dj < 3; ++dj)
#pragma unroll
      for (int di = 0; di < 3; ++di) {
        const int j = y - 1 + dj, i = x - 1 + di;
        if (j >= 0 && j <= rows - 1 && i >= 0 && i <= cols - 1) {
          const int k = 8 - ((j - jlo) * ni + (i - ilo));
          dxVal += v[dj][di] * sobel_gsx(k);
          dyVal += v[dj][di] * sobel_gsy(k);
        }
      }
  }
  dx = (int16_t)(int)dxVal;
  dy = (int16_t)(int)dyVal;
}
__device__ __forceinline__ void sobel_px(const uint8_t* __restrict__ src, int cols, int rows, int x, int y, int16_t* dx, int16_t* dy) {
  float v[3][3];
#pragma unroll
  for (int dj = 0; dj < 3; ++dj)
#pragma unroll
    for (int di = 0; di < 3; ++di) v[dj][di] = (float)src[min(max(y - 1 + dj, 0), rows - 1) * cols + min(max(x - 1 + di, 0), cols - 1)];
  int16_t ox, oy;
  sobel_taps(v, cols, rows, x, y, ox, oy);
  dx[y * cols + x] = ox;
  dy[y * cols + x] = oy;
}
__global__ void k_sobel(const uint8_t* __restrict__ src, int cols, int rows, int16_t* __restrict__ dx, int16_t* __restrict__ dy) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  sobel_px(src, cols, rows, x, y, dx, dy);
}
// 4-byte photometric correspondence (frame tier): bit31 valid | (diff + 255) << 22 | v0 << 11 | u0.
// DataTerm.one is the pixel itself, DataTerm.diff an integer in [-255, 255] (difference of two u8 intensities).
__device__ __forceinline__ uint32_t pack_corres(int u0, int v0, int idiff) {
  return 0x80000000u | ((uint32_t)(idiff + 255) << 22) | ((uint32_t)v0 << 11) | (uint32_t)u0;
}
// Sobel for all three levels in one launch + the iteration-invariant half of the photometric gate.
// residualKernel (reduce.cu:631-667) re-tests, on every one of the 19 iterations, conditions that only depend on
// the frame: image-border limits, the 4x4 "no zero pixel" window on nextImage (quirk Q10), the gradient-magnitude
// threshold and !isnan(nextDepth).  They are evaluated here once per frame into a 1-byte mask; the packed
// correspondence of every pixel is reset to "invalid" and only masked-in pixels are revisited by the iterations.
struct SobelLevels {
  const uint8_t* src[NUM_PYRS];
  int16_t* dx[NUM_PYRS];
  int16_t* dy[NUM_PYRS];
  const float* nextDepth[NUM_PYRS];
  uint8_t* mask[NUM_PYRS];
  uint32_t* corres[NUM_PYRS];
  float minScale[NUM_PYRS];
  int cols[NUM_PYRS], rows[NUM_PYRS];
};
__device__ __forceinline__ void sobel_mask_px(const SobelLevels& L, int l, int x, int y) {
  const int cols = L.cols[l], rows = L.rows[l];
  if (x >= cols || y >= rows) return;
  const uint8_t* __restrict__ img = L.src[l];
  const int k = y * cols + x;
  // the 4 x 4 window (y - 2 .. y + 1) x (x - 2 .. x + 1) covers the Sobel taps and the "no zero pixel" window: 16 loads, one round trip
  int w[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) w[a][b] = img[min(max(y - 2 + a, 0), rows - 1) * cols + min(max(x - 2 + b, 0), cols - 1)];
  const float nd = L.nextDepth[l][k];
  float v[3][3];
#pragma unroll
  for (int dj = 0; dj < 3; ++dj)
#pragma unroll
    for (int di = 0; di < 3; ++di) v[dj][di] = (float)w[dj + 1][di + 1];
  int16_t sx, sy;
  sobel_taps(v, cols, rows, x, y, sx, sy);
  L.dx[l][k] = sx;
  L.dy[l][k] = sy;
  bool ok = (x < cols - 5 && y < rows - 1);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {   // u in [max(y - 2, 0), min(y + 2, rows)), v in [max(x - 2, 0), min(x + 2, cols))
      const int u = y - 2 + a, vv = x - 2 + b;
      if (u >= 0 && u < rows && vv >= 0 && vv < cols) ok = ok && (w[a][b] > 0);
    }
  if (ok) {
    const int valx = sx, valy = sy;
    const float mTwo = (float)((valx * valx) + (valy * valy));
    ok = mTwo >= L.minScale[l] && !isnan(nd);
  }
  L.mask[l][k] = ok ? 1 : 0;
  L.corres[l][k] = 0u;
}
__global__ void k_sobel_levels(const SobelLevels L) {
  sobel_mask_px(L, blockIdx.z, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.y * blockDim.y + threadIdx.y);
}
// k_vmap_nmap_levels and k_sobel_levels as ONE launch (round 5: they only share their inputs' producers, and every launch boundary of the frame
// script is 2-6 us depending on the box): blockIdx.z = 0..2 vertex / normal maps of level z, 3..5 Sobel + photometric gates of level z - 3
__global__ void k_vn_sobel_levels(const VNLevels V, const SobelLevels S) {
  const int z = blockIdx.z, x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (z < NUM_PYRS) vmap_nmap_px(V, z, x, y);
  else sobel_mask_px(S, z - NUM_PYRS, x, y);
}

// One STAGE of the frame's pyramids as one launch (round 6; A/B build "pyrstages", not the default: see build_pyramids): the pyramid step level
// l -> l + 1 of up to four images AND the vertex / normal maps and the Sobel + photometric gates of level l — which read level l only, i.e. what
// the previous stage (or, for l = 0, the input launch) wrote.  blockIdx.z: 0 .. n_pyr - 1 pyramid jobs, then the maps, then the Sobel.
struct StageJobs {
  PyrJobs J;
  int n_pyr;
  VNLevels V;
  SobelLevels S;
  int level;
};
__global__ void k_pyramid_stage(const StageJobs A) {
  const int z = blockIdx.z, x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (z < A.n_pyr) {
    if (x >= A.J.scols / 2 || y >= A.J.srows / 2) return;
    if (A.J.type[z] == 0) pyr_down_u16_px((const uint16_t*)A.J.src[z], A.J.scols, A.J.srows, (uint16_t*)A.J.dst[z], x, y);
    else if (A.J.type[z] == 1) pyr_down_gauss_f_px((const float*)A.J.src[z], A.J.scols, A.J.srows, (float*)A.J.dst[z], x, y);
    else pyr_down_uchar_gauss_px((const uint8_t*)A.J.src[z], A.J.scols, A.J.srows, (uint8_t*)A.J.dst[z], x, y);
  } else if (z == A.n_pyr) {
    vmap_nmap_px(A.V, A.level, x, y);
  } else {
    sobel_mask_px(A.S, A.level, x, y);
  }
}

// projectPointsKernel, cudafuncs.cu:670-688
__device__ __forceinline__ f3 project_point(int x, int y, float z, float invFx, float invFy, float cx, float cy) {
  return {(float)((x - cx) * z * invFx), (float)((y - cy) * z * invFy), z};
}
__global__ void k_project_points(const float* __restrict__ depth, int cols, int rows, float invFx, float invFy, float cx,
                                 float cy, float* __restrict__ cloud) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= cols || y >= rows) return;
  const f3 p = project_point(x, y, depth[y * cols + x], invFx, invFy, cx, cy);
  float* c = cloud + (size_t)(y * cols + x) * 3;
  c[0] = p.x; c[1] = p.y; c[2] = p.z;
}

// ------------------------------------------------------------------------------------------
// Model-side pyramids in one launch.  Each thread owns a 4x4 block of level-0 pixels and emits
// level 0 (16 px), level 1 (2x2) and level 2 (1 px): copyMaps -> resizeVMap/NMap x2 -> tranformMaps x3
// (RGBDOdometry.cpp:171-210) plus verticesToDepth for level 0 (:217).  Resizing happens on the
// camera-frame values and the rigid transform is applied per level afterwards, exactly the
// reference's order, so every output equals the multi-kernel path bit for bit; the intermediate
// camera-frame pyramids never touch HBM.
// ------------------------------------------------------------------------------------------
// !denseEnough(): float(sum) / float(rows * cols) > 0.75f over the (W/20)x(H/20) samples (ElasticFusion.cpp:256-268,304-305)
__device__ __forceinline__ bool use_fill_in(const TrackState* __restrict__ st) {
  return !((float)st->dense_count / (float)st->dense_samples > 0.75f);
}
struct ModelMapsArgs {
  const float4* pred_vertex;
  const float4* pred_normal;
  const float4* fill_vertex;
  const float4* fill_normal;
  float* vmap[NUM_PYRS];
  float* nmap[NUM_PYRS];
  float* depth0;
  int cols, rows;
  float maxDepthRGB;
  bool camera_frame;   // initICP(predictedVertices, predictedNormals): copyMaps + resize only, no tranformMaps
  // optional (round 6): the model's level-0 intensity image rides along (k_intensity_both's model half: the same texels are in flight here)
  const uint8_t* pred_image;
  const uint8_t* fill_image;
  bool force_fill_image;
  uint8_t* last0;
  // optional (round 6): denseEnough()'s tally (ElasticFusion.cpp:256-268) taken HERE, by every workgroup for itself, from the predicted image's
  // (W / 20) x (H / 20) sample texels — instead of one atomicAdd per sample from the prediction's resolve pass: 768 (640 x 480) / 3 072
  // (1280 x 960) device-scope atomics on ONE word, which the memory side retires one after the other (≈ 12 ns each: 9 / 37 us, the length of
  // that launch).  Null: TrackState::dense_count stands (a prediction that was not to be counted, the operator tier).
  const uint8_t* tally_image;
  unsigned* tally_out;   // TrackState::dense_count: written by the launch's first model-map workgroup for whoever reads it later
};
// the workgroup's decision "tracking reads the fill-in maps" (uniform); called by all 256 threads of a model-map workgroup before any of them leaves
__device__ __forceinline__ bool model_maps_use_fill(const ModelMapsArgs& A, const TrackState* __restrict__ st, bool first_wg) {
  if (!A.tally_image) return use_fill_in(st);
  __shared__ unsigned tally_s[4];
  const int t = (int)(threadIdx.y * blockDim.x + threadIdx.x);
  const int dc = A.cols / 20, dr = A.rows / 20;
  unsigned cnt = 0;
  for (int i = t; i < dc * dr; i += 256) {   // Resize::image: dest (a, b) <- source texel (20 a + 10, 20 b + 10) (k_dense_count, ef_map_kernels.hip)
    const int b = i / dc, a = i - b * dc;
    const uchar4 px = ((const uchar4*)A.tally_image)[(20 * b + 10) * A.cols + (20 * a + 10)];
    cnt += (px.x > 0 && px.y > 0 && px.z > 0) ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((t & 63) == 0) tally_s[t >> 6] = cnt;
  __syncthreads();
  const unsigned total = tally_s[0] + tally_s[1] + tally_s[2] + tally_s[3];
  if (first_wg && t == 0) *A.tally_out = total;
  return !((float)total / (float)st->dense_samples > 0.75f);
}
// ALL_PLANES: level 0 (copyMaps NaNs x, y and z of an empty texel); the resized levels only get the
// x-plane NaN that resizeMapKernel / tranformMapsKernel write (quirk Q3: y/z planes keep stale data).
template <bool ALL_PLANES>
__device__ __forceinline__ void store_planar(float* m, int cols, int rows, int x, int y, bool valid, f3 v) {
  if (valid) {
    m[(y + rows) * cols + x] = v.y;
    m[(y + 2 * rows) * cols + x] = v.z;
    m[y * cols + x] = v.x;
  } else {
    m[y * cols + x] = qnan();
    if (ALL_PLANES) {
      m[(y + rows) * cols + x] = qnan();
      m[(y + 2 * rows) * cols + x] = qnan();
    }
  }
}
// Four lanes share a 4x4 block: lane j of the quad owns COLUMN j of it, so that every load and every level-0 store of a wavefront
// covers 64 consecutive pixels of one row (1 KB of float4s per load instruction, four times as many workgroups as one thread per
// block gave: 300 instead of 75 at 640x480).  The 2x2 boxes of level 1 need the neighbouring column (quad_perm xor 1), the one of
// level 2 the neighbouring pair (xor 2); both lanes of a pair / all lanes of the quad evaluate the same expressions on the same
// operands in the reference's order ((a + b) + c) + d, one of them stores.
__device__ __forceinline__ float qx1(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float qx2(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ f3 qx1(f3 v) { return f3{qx1(v.x), qx1(v.y), qx1(v.z)}; }
__device__ __forceinline__ f3 qx2(f3 v) { return f3{qx2(v.x), qx2(v.y), qx2(v.z)}; }
__device__ __forceinline__ f3 box4(f3 a, f3 b, f3 c, f3 d) {
  return f3{(a.x + b.x + c.x + d.x) / 4, (a.y + b.y + c.y + d.y) / 4, (a.z + b.z + c.z + d.z) / 4};
}
// (gx: the lane's pixel column; by: its row of 4 x 4 blocks — k_model_maps takes them from its own grid, k_frame_inputs from its share of a joint one)
__device__ __forceinline__ void model_maps_lane(const ModelMapsArgs& A, const TrackState* __restrict__ st, int gx, int by, bool fill) {
  const int cols = A.cols, rows = A.rows;
  if (gx >= cols || by * 4 >= rows) return;   // cols is a multiple of 4: a quad is in or out as a whole
  const int j = gx & 3, bx = gx >> 2;
  const bool even = (j & 1) == 0, leftpair = j < 2;
  const float4* __restrict__ vsrc = fill ? A.fill_vertex : A.pred_vertex;
  const float4* __restrict__ nsrc = fill ? A.fill_normal : A.pred_normal;
  const m33 R = m33_load(st->R_wc_f);
  const f3 t{st->t_wc_f[0], st->t_wc_f[1], st->t_wc_f[2]};
  const bool cf = A.camera_frame;
  const int c1 = cols / 2, r1 = rows / 2, c2 = cols / 4, r2 = rows / 4;
  float4 vs[4], ns[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    vs[r] = vsrc[(by * 4 + r) * cols + gx];
    ns[r] = nsrc[(by * 4 + r) * cols + gx];
  }
  if (A.last0) {   // populateRGBDData(model)'s level-0 intensity (k_model_intensity), same choice of source image
    const uchar4* __restrict__ isrc = (const uchar4*)((A.force_fill_image || fill) ? A.fill_image : A.pred_image);
    uchar4 cs[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[r] = isrc[(by * 4 + r) * cols + gx];
#pragma unroll
    for (int r = 0; r < 4; ++r) A.last0[(by * 4 + r) * cols + gx] = intensity_of((float)cs[r].x, (float)cs[r].y, (float)cs[r].z);
  }
  f3 v0[4], n0[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int y = by * 4 + r;
    const bool ok = !(vs[r].z == 0);
    v0[r] = {vs[r].x, vs[r].y, vs[r].z};
    n0[r] = {ns[r].x, ns[r].y, ns[r].z};
    A.depth0[y * cols + gx] = (vs[r].z > A.maxDepthRGB || vs[r].z <= 0) ? qnan() : vs[r].z;
    // level 0: copyMaps NaNs all planes where z == 0, transform propagates NaN via the x-plane
    const bool nok = ok && !isnan(ns[r].x);
    if (cf) {   // copyMaps alone: the NaN test of tranformMaps is not applied, an empty texel NaNs all planes
      store_planar<true>(A.vmap[0], cols, rows, gx, y, ok, v0[r]);
      store_planar<true>(A.nmap[0], cols, rows, gx, y, ok, n0[r]);
    } else {
      store_planar<true>(A.vmap[0], cols, rows, gx, y, ok && !isnan(vs[r].x), mul(R, v0[r]) + t);
      store_planar<true>(A.nmap[0], cols, rows, gx, y, nok, mul(R, n0[r]));
    }
  }
  // level 1: 2x2 box of the camera-frame level-0 maps (x-plane NaN test only); this lane's pair of columns, both box rows
  f3 v1[2], n1[2];
  bool v1ok[2], n1ok[2];
#pragma unroll
  for (int qy = 0; qy < 2; ++qy) {
    const f3 pv0 = qx1(v0[2 * qy]), pv1 = qx1(v0[2 * qy + 1]), pn0 = qx1(n0[2 * qy]), pn1 = qx1(n0[2 * qy + 1]);
    // [sy][sx] of the one-thread-per-block form: [0][0], [0][1], [1][0], [1][1]
    const f3 va = even ? v0[2 * qy] : pv0, vb = even ? pv0 : v0[2 * qy], vc = even ? v0[2 * qy + 1] : pv1, vd = even ? pv1 : v0[2 * qy + 1];
    const f3 na_ = even ? n0[2 * qy] : pn0, nb = even ? pn0 : n0[2 * qy], nc = even ? n0[2 * qy + 1] : pn1, nd = even ? pn1 : n0[2 * qy + 1];
    const bool oka = !(va.z == 0), okb = !(vb.z == 0), okc = !(vc.z == 0), okd = !(vd.z == 0);
    const bool vok = oka && okb && okc && okd && !isnan(va.x) && !isnan(vb.x) && !isnan(vc.x) && !isnan(vd.x);
    const bool nok = oka && okb && okc && okd && !isnan(na_.x) && !isnan(nb.x) && !isnan(nc.x) && !isnan(nd.x);
    const f3 vavg = box4(va, vb, vc, vd);
    const f3 navg = normalized(box4(na_, nb, nc, nd));
    v1[qy] = vavg; n1[qy] = navg;
    // a valid-flagged average can still be NaN in x (NaN y/z never matter: only x is tested downstream)
    v1ok[qy] = vok; n1ok[qy] = nok;
    if (even) {
      const int x1 = bx * 2 + (j >> 1), y1 = by * 2 + qy;
      if (cf) {   // resizeMapKernel alone: x-plane NaN where a source x is NaN, else the three averages as they come
        store_planar<false>(A.vmap[1], c1, r1, x1, y1, vok, vavg);
        store_planar<false>(A.nmap[1], c1, r1, x1, y1, nok, navg);
      } else {
        store_planar<false>(A.vmap[1], c1, r1, x1, y1, vok && !isnan(vavg.x), mul(R, vavg) + t);
        store_planar<false>(A.nmap[1], c1, r1, x1, y1, nok && !isnan(navg.x), mul(R, navg));
      }
    }
  }
  // level 2: [qy][qx] = this pair's and the other pair's level-1 values
  const f3 ov0 = qx2(v1[0]), ov1 = qx2(v1[1]), on0 = qx2(n1[0]), on1 = qx2(n1[1]);
  const bool ovok0 = qx2(v1ok[0] ? 1.0f : 0.0f) != 0.0f, ovok1 = qx2(v1ok[1] ? 1.0f : 0.0f) != 0.0f;
  const bool onok0 = qx2(n1ok[0] ? 1.0f : 0.0f) != 0.0f, onok1 = qx2(n1ok[1] ? 1.0f : 0.0f) != 0.0f;
  const f3 v00 = leftpair ? v1[0] : ov0, v01 = leftpair ? ov0 : v1[0], v10 = leftpair ? v1[1] : ov1, v11 = leftpair ? ov1 : v1[1];
  const f3 n00 = leftpair ? n1[0] : on0, n01 = leftpair ? on0 : n1[0], n10 = leftpair ? n1[1] : on1, n11 = leftpair ? on1 : n1[1];
  const bool vok2 = v1ok[0] && ovok0 && v1ok[1] && ovok1 && !isnan(v00.x) && !isnan(v01.x) && !isnan(v10.x) && !isnan(v11.x);
  const bool nok2 = n1ok[0] && onok0 && n1ok[1] && onok1 && !isnan(n00.x) && !isnan(n01.x) && !isnan(n10.x) && !isnan(n11.x);
  const f3 va = box4(v00, v01, v10, v11);
  const f3 na = normalized(box4(n00, n01, n10, n11));
  if (j == 0) {
    if (cf) {
      store_planar<false>(A.vmap[2], c2, r2, bx, by, vok2, va);
      store_planar<false>(A.nmap[2], c2, r2, bx, by, nok2, na);
    } else {
      store_planar<false>(A.vmap[2], c2, r2, bx, by, vok2 && !isnan(va.x), mul(R, va) + t);
      store_planar<false>(A.nmap[2], c2, r2, bx, by, nok2 && !isnan(na.x), mul(R, na));
    }
  }
}
__global__ void __launch_bounds__(256) k_model_maps(const ModelMapsArgs A, const TrackState* __restrict__ st) {
  const bool fill = model_maps_use_fill(A, st, blockIdx.x == 0 && blockIdx.y == 0);
  model_maps_lane(A, st, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(blockIdx.y * blockDim.y + threadIdx.y), fill);
}
// ------------------------------------------------------------------------------------------
// per-pixel Jacobian rows
// ------------------------------------------------------------------------------------------
struct IcpView {
  const float* vmap_curr;
  const float* nmap_curr;
  const float* vmap_g_prev;
  const float* nmap_g_prev;
  int cols, rows;
  Intr k;
  float distThres, angleThres;
  float dist2Max, sine2Max;   // sq_le_max(distThres), sq_lt_max(angleThres): the two gates on squared norms (ef_device.hpp)
};
struct IcpPose { m33 Rcurr; f3 tcurr; m33 Rprev_inv; f3 tprev; };
// The pose as 24 wave-uniform scalars (every caller's pose is the same in all lanes: kernel arguments or the workgroup's state).  Besides keeping
// it in scalar registers, the read-first-lane results are values, not loads: the two-visit functions below broadcast pose components into
// two-component vectors, and a broadcast of a LOADED scalar is rewritten by the compiler into a one-element vector load from the pose object,
// which then can no longer be promoted to registers (it ended up in scratch memory, 160 bytes per lane).
__device__ __forceinline__ float uniform_f(float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); }
__device__ __forceinline__ f3 uniform_f3(f3 a) { return f3{uniform_f(a.x), uniform_f(a.y), uniform_f(a.z)}; }
__device__ __forceinline__ m33 uniform_m33(const m33& m) { return m33{{uniform_f3(m.r[0]), uniform_f3(m.r[1]), uniform_f3(m.r[2])}}; }
__device__ __forceinline__ IcpPose icp_pose_uniform(const IcpPose& P) {
  return IcpPose{uniform_m33(P.Rcurr), uniform_f3(P.tcurr), uniform_m33(P.Rprev_inv), uniform_f3(P.tprev)};
}

// search() + getProducts(), reduce.cu:228-309: fills row[7]; returns found
__device__ __forceinline__ bool icp_row(const IcpView& V, const IcpPose& P, int x, int y, float (&row)[7]) {
  const int cols = V.cols, rows = V.rows;
  const int plane = cols * rows, idx = y * cols + x;
  const f3 vcurr{V.vmap_curr[idx], V.vmap_curr[idx + plane], V.vmap_curr[idx + 2 * plane]};
  const f3 vcurr_g = mul(P.Rcurr, vcurr) + P.tcurr;
  const f3 vcurr_cp = mul(P.Rprev_inv, vcurr_g - P.tprev);
  const int ux = f2i_rn(vcurr_cp.x * V.k.fx / vcurr_cp.z + V.k.cx);
  const int uy = f2i_rn(vcurr_cp.y * V.k.fy / vcurr_cp.z + V.k.cy);
  if (ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0) return false;
  const int pidx = uy * cols + ux;
  const f3 vprev_g{V.vmap_g_prev[pidx], V.vmap_g_prev[pidx + plane], V.vmap_g_prev[pidx + 2 * plane]};
  const f3 ncurr{V.nmap_curr[idx], V.nmap_curr[idx + plane], V.nmap_curr[idx + 2 * plane]};
  const f3 ncurr_g = mul(P.Rcurr, ncurr);
  const f3 nprev_g{V.nmap_g_prev[pidx], V.nmap_g_prev[pidx + plane], V.nmap_g_prev[pidx + 2 * plane]};
  const float dist = norm(vprev_g - vcurr_g);
  const float sine = norm(cross(ncurr_g, nprev_g));
  if (!(sine < V.angleThres && dist <= V.distThres && !isnan(ncurr.x) && !isnan(nprev_g.x))) return false;
  const f3 s_cp = mul(P.Rprev_inv, vcurr_g - P.tprev);
  const f3 d_cp = mul(P.Rprev_inv, vprev_g - P.tprev);
  const f3 n_cp = mul(P.Rprev_inv, nprev_g);
  const f3 c = cross(s_cp, n_cp);
  row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
  row[3] = c.x; row[4] = c.y; row[5] = c.z;
  row[6] = dot(n_cp, s_cp - d_cp);
  return true;
}

struct RgbView {
  const void* corres;         // PACKED: uint32 per pixel (frame tier); else DataTerm (operator tier, types.cuh:81-86)
  const float* lastDepth;     // level depth of the model image; the cloud is evaluated on the fly
  const float* cloud;         // or an explicit float3 cloud (operator tier); one of the two is null
  const int16_t* dIdx;
  const int16_t* dIdy;
  int cols, rows;
  Intr k;
  float sobelScale;
};
// RGBReduction::getProducts, reduce.cu:420-476
template <bool PACKED>
__device__ __forceinline__ bool rgb_row(const RgbView& V, float sigma, int i, float (&row)[7]) {
  int zx, zy, oi;
  float diff;
  if (PACKED) {
    const uint32_t c = ((const uint32_t*)V.corres)[i];
    if (!(c & 0x80000000u)) return false;
    zx = (int)(c & 0x7FFu); zy = (int)((c >> 11) & 0x7FFu);
    diff = (float)((int)((c >> 22) & 0x1FFu) - 255);
    oi = i;
  } else {
    const DataTerm c = ((const DataTerm*)V.corres)[i];
    if (!c.valid) return false;
    zx = c.zero_x; zy = c.zero_y;
    diff = c.diff;
    oi = c.one_y * V.cols + c.one_x;
  }
  float w = sigma + fabsf(diff);
  w = w > 1.19209290E-07F ? 1.0f / w : 1.0f;
  if (sigma == -1) w = 1;
  row[6] = -w * diff;
  f3 p;
  const int zi = zy * V.cols + zx;
  if (V.cloud) {
    p = {V.cloud[(size_t)zi * 3], V.cloud[(size_t)zi * 3 + 1], V.cloud[(size_t)zi * 3 + 2]};
  } else {  // projectPointsKernel folded in: same operations, same bits, no 12 B/px cloud in HBM
    p = project_point(zx, zy, V.lastDepth[zi], 1.0f / V.k.fx, 1.0f / V.k.fy, V.k.cx, V.k.cy);
  }
  const float invz = (float)(1.0 / (double)p.z);
  const float dI_dx_val = w * V.sobelScale * V.dIdx[oi];
  const float dI_dy_val = w * V.sobelScale * V.dIdy[oi];
  const float v0 = dI_dx_val * V.k.fx * invz;
  const float v1 = dI_dy_val * V.k.fy * invz;
  const float v2 = -(v0 * p.x + v1 * p.y) * invz;
  row[0] = v0; row[1] = v1; row[2] = v2;
  row[3] = -p.z * v1 + p.y * v2;
  row[4] = p.z * v0 - p.x * v2;
  row[5] = -p.y * v0 + p.x * v1;
  return true;
}

struct ResidualView {
  const int16_t* dIdx;
  const int16_t* dIdy;
  const float* lastDepth;
  const float* nextDepth;
  const uint8_t* lastImage;
  const uint8_t* nextImage;
  DataTerm* corres;
  int cols, rows;
  float minScale, maxDepthDelta;
};
// RGBResidual::getProducts, reduce.cu:631-701; returns {valid, diff^2 as int}
__device__ __forceinline__ void residual_px(const ResidualView& V, const m33& K, const f3& kt, int k, int& cnt, int& sq) {
  const int cols = V.cols, rows = V.rows;
  const int i = k / cols, j0 = k - i * cols;
  DataTerm corres;
  corres.zero_x = corres.zero_y = corres.one_x = corres.one_y = 0;
  corres.diff = 0.f;
  corres.valid = 0;
  corres.pad[0] = corres.pad[1] = corres.pad[2] = 0;
  if (j0 < cols - 5 && i < rows - 1) {
    bool valid = true;
    for (int u = max(i - 2, 0); u < min(i + 2, rows); ++u)
      for (int v = max(j0 - 2, 0); v < min(j0 + 2, cols); ++v) valid = valid && (V.nextImage[u * cols + v] > 0);
    if (valid) {
      const int valx = V.dIdx[k], valy = V.dIdy[k];
      const float mTwo = (float)((valx * valx) + (valy * valy));
      if (mTwo >= V.minScale) {
        const int y = i, x = j0;
        const float d1 = V.nextDepth[k];
        if (!isnan(d1)) {
          const float transformed_d1 = (float)(d1 * (K.r[2].x * x + K.r[2].y * y + K.r[2].z) + kt.z);
          const int u0 = f2i_rn((d1 * (K.r[0].x * x + K.r[0].y * y + K.r[0].z) + kt.x) / transformed_d1);
          const int v0 = f2i_rn((d1 * (K.r[1].x * x + K.r[1].y * y + K.r[1].z) + kt.y) / transformed_d1);
          if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
            const float d0 = V.lastDepth[v0 * cols + u0];
            const uint8_t li = V.lastImage[v0 * cols + u0];
            if (d0 > 0 && fabsf(transformed_d1 - d0) <= V.maxDepthDelta && li != 0) {
              corres.zero_x = (short)u0; corres.zero_y = (short)v0;
              corres.one_x = (short)x; corres.one_y = (short)y;
              corres.diff = (float)V.nextImage[k] - (float)li;
              corres.valid = 1;
              cnt += 1;
              sq += (int)(corres.diff * corres.diff);
            }
          }
        }
      }
    }
  }
  V.corres[k] = corres;
}

// SO3Reduction::getProducts, reduce.cu:820-897
__device__ __forceinline__ void so3_gradient(const uint8_t* __restrict__ img, int cols, int x, int y, float& gx, float& gy) {
  const float actu = (float)img[y * cols + x];
  float back = (float)img[y * cols + x - 1], fore = (float)img[y * cols + x + 1];
  gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
  back = (float)img[(y - 1) * cols + x];
  fore = (float)img[(y + 1) * cols + x];
  gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
}
__device__ __forceinline__ bool so3_row(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage, int cols,
                                        int rows, const m33& IB, const m33& KI, const m33& KR, int k, float (&row)[4]) {
  const int y = k / cols, x = k - y * cols;
  const f3 unwarped{(float)x, (float)y, 1.0f};
  const f3 warped = mul(IB, unwarped);
  const int wx = f2i_rn(warped.x / warped.z), wy = f2i_rn(warped.y / warped.z);
  const bool found = (wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1);
  if (!found) return false;  // zero row
  float gnx, gny, glx, gly;
  so3_gradient(nextImage, cols, wx, wy, gnx, gny);
  so3_gradient(lastImage, cols, x, y, glx, gly);
  const float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
  const f3 point = mul(KI, unwarped);
  const float z2 = point.z * point.z;
  const float a = KR.r[0].x, b = KR.r[0].y, c = KR.r[0].z;
  const float d = KR.r[1].x, e = KR.r[1].y, f = KR.r[1].z;
  const float g = KR.r[2].x, h = KR.r[2].y, ii = KR.r[2].z;
  const f3 left{((point.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
                ((point.z * (e * gy + b * gx)) - (gy * h * y) - (gx * h * x)) / z2,
                ((point.z * (f * gy + c * gx)) - (gy * ii * y) - (gx * ii * x)) / z2};
  const f3 jr = cross(left, point);
  row[0] = jr.x; row[1] = jr.y; row[2] = jr.z;
  row[3] = -((float)nextImage[wy * cols + wx] - (float)lastImage[y * cols + x]);
  return true;
}

// ------------------------------------------------------------------------------------------
// reduction kernels
// ------------------------------------------------------------------------------------------

// K6a: photometric correspondence search.  Integer sums go straight to two device-scope atomics
// (exact, order-free), replacing reduceSum(int2) + cudaMalloc/cudaFree per call (reduce.cu:774-783).
// Operator tier: every gate of residualKernel evaluated per call, 16-byte DataTerm out (the reference's layout).
__global__ void __launch_bounds__(REDUCE_BLOCK) k_rgb_residual_op(const ResidualView V, const float* __restrict__ krkinv,
                                                                   const float* __restrict__ ktp, int* sums) {
  __shared__ int lds[2 * REDUCE_BLOCK / 64];
  const m33 K = m33_load(krkinv);
  const f3 kt{ktp[0], ktp[1], ktp[2]};
  const int N = V.cols * V.rows;
  int cnt = 0, sq = 0;
  const int k = blockIdx.x * REDUCE_BLOCK + threadIdx.x;
  if (k < N) residual_px(V, K, kt, k, cnt, sq);
  block_reduce_atomic_int2<REDUCE_BLOCK>(cnt, sq, lds, sums);
}
// Frame tier: the iteration-invariant gates come from the per-frame mask (k_sobel_levels); only masked-in pixels
// do the warp + gathers and rewrite their 4-byte packed correspondence.
struct ResidualPackedView {
  const uint8_t* mask;
  const float* lastDepth;
  const float* nextDepth;     // the same buffer as lastDepth in frame-to-model tracking (quirk Q1)
  const uint8_t* lastImage;
  const uint8_t* nextImage;
  uint32_t* corres;
  int cols, rows;
  float maxDepthDelta;
};
// One Gauss-Newton iteration's FIRST kernel: the update step of the previous iteration (head) + this iteration's
// correspondence search (body).
//   head  Every workgroup evaluates the update redundantly — reduceSum over the pair partials k_se3_accum left (59 KB, L2-resident
//         after the first touch per XCD), the 6x6 solve, the SE(3) update, the next K R K^-1 / K t (ef_solve_dev.hpp, one wavefront) —
//         so that no workgroup waits for another one: the only exchange between workgroups is the kernel boundary, which is
//         cheaper on this part than any in-launch hand-over (MI355X_MICROARCH.md "boundary" vs "handoff-flag" rows).  Workgroup 0
//         publishes the result into the other GNState buffer and the statistics into the TrackState.  The body's pixel-addressed
//         loads are issued before the head, so they are in flight while it runs.
//   body  residualKernel (reduce.cu:603-787) on the packed correspondences, as before.
struct StepArgs {
  bool has_head;              // false for the first iteration of a call: K R K^-1 / K t come from prev as they are
  bool has_body;              // false when the photometric term is off (the launch is one workgroup: head only)
  bool icp, rgb, rgbOnly;     // of the update step (RGBDOdometry.cpp:266-267)
  float icpWeight;
  Intr knext;                 // intrinsics of THIS iteration's level (for the head's K R K^-1)
  bool level_changes;         // this iteration runs at another level than the one whose update the head evaluates
  int it;                     // index of the iteration within the call (tag of the fused launch's record)
  int ng = 0;                 // fast order: group partials per accumulator the head's tree looks at
};
template <bool SPLIT_TAIL = false>
__device__ __forceinline__ void solve_step_wave(TrackState* st, const GNState* prev, GNState* next, bool publish, const float* sums,
                                                const StepArgs& A, efs::SolveScratch& S, const efs::SolvePrefetch& PF, bool stats);
template <int BLOCK, bool COHERENT = false>
__device__ __forceinline__ void pair_partials_tree(const float* __restrict__ pairs, bool icp, bool rgb, float* sums_s);
// reduceSum over what the accumulation launch left -> sums_s[term * SE3_ACCS + acc]: the fast order's 256-leaf tree over `ng` group
// partials (ef_track_fast.inc), or the reference's 8-warp / 64-block trees over the pair partials (pair_partials_tree)
template <int BLOCK>
__device__ __forceinline__ void head_sums(const float* __restrict__ pairs, int ng, bool icp, bool rgb, float* sums_s);

template <int PPT>
__global__ void __launch_bounds__(REDUCE_BLOCK) k_track_step(const ResidualPackedView V, TrackState* st, const GNState* __restrict__ prev,
                                                              GNState* next, const float* __restrict__ pairs,
                                                              const int* __restrict__ slots_prev, int* slots_out, const StepArgs A) {
  __shared__ int lds[2 * REDUCE_BLOCK / 64];
  __shared__ efs::SolveScratch S;
  __shared__ float sums_s[2 * SE3_ACCS];
  const int N = V.cols * V.rows, cols = V.cols, rows = V.rows;
  const int t = threadIdx.x;
  const int base = blockIdx.x * REDUCE_BLOCK * PPT + t;
  // stage 1 of the body: everything addressed by the pixel itself, branch-free (an out-of-range lane reads pixel N-1 and is masked
  // out), issued before the head so that one memory round trip covers all of it
  uint8_t m[PPT], ni[PPT];
  float d1s[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = base + j * REDUCE_BLOCK, q = (A.has_body && k < N) ? k : (A.has_body ? N - 1 : 0);
    m[j] = A.has_body ? V.mask[q] : (uint8_t)0;
    d1s[j] = A.has_body ? V.nextDepth[q] : 0.f;  // mask guarantees !isnan(d1)
    ni[j] = A.has_body ? V.nextImage[q] : (uint8_t)0;
    if (k >= N) m[j] = 0;
  }
  m33 K;
  f3 kt;
  int skip;
  if (A.has_head) {
    // head + body in one launch (ef_set_fused_step): ONLY workgroup 0 evaluates the update; it hands the thirteen words the search needs
    // to the other workgroups of the launch as tagged granules, which they poll with their pixel loads already in flight.  (Workgroups
    // are dispatched in order, so workgroup 0 is resident whenever another one waits; the spin is bounded all the same.)
    const unsigned tag = (st->call_seq << 6) | (unsigned)(A.it + 1);
    if (!A.has_body || blockIdx.x == 0) {
      efs::SolvePrefetch PF{};
      if (t < 64) PF = efs::solve_prefetch(st, prev, slots_prev);
      head_sums<REDUCE_BLOCK>(pairs, A.ng, A.icp, A.rgb, sums_s);
      __syncthreads();
      if (t < 64) solve_step_wave(st, prev, next, blockIdx.x == 0, sums_s, A, S, PF, blockIdx.x == 0);
      __syncthreads();
      if (A.has_body && t < 13) {
        const unsigned bits = t < 9 ? __float_as_uint(S.krkinv[t]) : (t < 12 ? __float_as_uint(S.kt[t - 9]) : (unsigned)S.broken);
        __hip_atomic_store(&st->step_rec[t], ((unsigned long long)tag << 32) | bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if (t < 13) {
        unsigned long long g = __hip_atomic_load(&st->step_rec[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spin = 0; (unsigned)(g >> 32) != tag; ++spin) {
          if (spin >= (1 << 20)) { st->step_timeout = 1u; break; }
          __builtin_amdgcn_s_sleep(1);
          g = __hip_atomic_load(&st->step_rec[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (t < 9) S.krkinv[t] = __uint_as_float((unsigned)g);
        else if (t < 12) S.kt[t - 9] = __uint_as_float((unsigned)g);
        else S.broken = (int)(unsigned)g;
      }
      __syncthreads();
    }
    K = m33_load(S.krkinv);
    kt = f3{S.kt[0], S.kt[1], S.kt[2]};
    skip = S.broken;
  } else {
    K = m33_load(prev->krkinv);
    kt = f3{prev->kt[0], prev->kt[1], prev->kt[2]};
    skip = prev->rgb_broken;
  }
  if (!A.has_body || skip) return;
  // stage 2: the warped pixel's gathers of all PPT pixels in flight together
  int cnt = 0, sq = 0;
  int gi[PPT], u0s[PPT], v0s[PPT];
  float td1[PPT], d0s[PPT];
  int lis[PPT];
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    const int k = base + j * REDUCE_BLOCK;
    const int y = k / cols, x = k - y * cols;
    const float d1 = d1s[j];
    td1[j] = (float)(d1 * (K.r[2].x * x + K.r[2].y * y + K.r[2].z) + kt.z);
    u0s[j] = f2i_rn((d1 * (K.r[0].x * x + K.r[0].y * y + K.r[0].z) + kt.x) / td1[j]);
    v0s[j] = f2i_rn((d1 * (K.r[1].x * x + K.r[1].y * y + K.r[1].z) + kt.y) / td1[j]);
    gi[j] = (m[j] && u0s[j] >= 0 && v0s[j] >= 0 && u0s[j] < cols && v0s[j] < rows) ? v0s[j] * cols + u0s[j] : -1;
    d0s[j] = 0.f;
    lis[j] = 0;
    if (gi[j] >= 0) { d0s[j] = V.lastDepth[gi[j]]; lis[j] = V.lastImage[gi[j]]; }
  }
#pragma unroll
  for (int j = 0; j < PPT; ++j) {
    if (!m[j]) continue;
    const int k = base + j * REDUCE_BLOCK;
    uint32_t packed = 0u;
    if (gi[j] >= 0 && d0s[j] > 0 && fabsf(td1[j] - d0s[j]) <= V.maxDepthDelta && lis[j] != 0) {
      const int idiff = (int)ni[j] - lis[j];   // exact: float(next) - float(last) of two u8
      packed = pack_corres(u0s[j], v0s[j], idiff);
      cnt += 1;
      sq += idiff * idiff;                      // (int)(diff*diff), exact for |diff| <= 255
    }
    V.corres[k] = packed;
  }
  block_reduce_atomic_int2<REDUCE_BLOCK>(cnt, sq, lds, slots_out + (blockIdx.x % RGB_SLOTS) * 16);
}

// {count, sum diff^2} = sum over the RGB_SLOTS slots; call with the whole first wave (threadIdx.x < 64) converged
__device__ __forceinline__ void sum_rgb_slots(const int* __restrict__ slots, int& cnt, int& sq) {
  static_assert(RGB_SLOTS == 64, "one slot per lane");
  const int lane = threadIdx.x & 63;
  int a = slots[lane * 16], b = slots[lane * 16 + 1];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64);
    b += __shfl_down(b, off, 64);
  }
  cnt = __shfl(a, 0, 64);
  sq = __shfl(b, 0, 64);
}
// sigma follows RGBDOdometry.cpp:442 (quirk Q2)
__device__ __forceinline__ float sigma_from_sums(int sigma, int rgbSize, bool rgbOnly) {
  if (rgbOnly) return -1.0f;
  const int arg = ((float)sigma / rgbSize == 0) ? 1 : rgbSize;
  return (float)sqrt((double)arg);
}

// ------------------------------------------------------------------------------------------
// Reference-order fp32 reductions.
//
// The reference sums with a fixed tree: <<<64,256>>> grid-stride threads (virtual thread g owns pixels g, g+16384,
// ... summed in that order with one FMA per product), warp32 shuffle tree, shared[32] + warp-0 tree per block,
// reduceSum<<<1,1024>>> over the 64 block partials (reduce.cu:313-317, :57-140).  fp32 addition is not associative,
// and the tracker is a chaotic feedback loop (pose -> association -> map -> next pose), so any other order drifts
// away from the reference frame by frame.  These kernels keep the reference's order but not its schedule:
//   phase A  one workgroup per virtual WARP (512 of them): every pixel-visit of that warp's 32 virtual threads is an
//            independent task spread over the whole workgroup (the per-pixel Jacobian rows — all the memory traffic
//            and nearly all the arithmetic); rows are staged in LDS as [pass][component][lane];
//   phase B  256 threads walk the rows in pass order: (lane, term, part) owns <= 8 of the 29 accumulators of one
//            virtual thread and applies the same FMA chain the reference thread would, then the warp32 tree runs as
//            five width-32 shuffles.  One partial vector per virtual warp, acc-major, goes to HBM;
//   final    the 8-warp and 64-block trees are shuffles of width 8 and 32 in the consumer (solve) kernel.
// Adding an exact 0.0f (zero rows of rejected pixels, zero-padded lanes of the reference's shared[32]) never changes
// a sum, so only the non-trivial additions are performed.
// ------------------------------------------------------------------------------------------
constexpr int ROW_STRIDE = 8 * 32;   // floats per pass in LDS (SO(3) kernel): 8 components x 32 lanes

// ------------------------------------------------------------------------------------------
// The normal equations of one icpStep / rgbStep, in the reference's summation order, with no LDS and no barrier
// in the main loop.
//
// Quad layout.  A wavefront covers HALF a virtual warp: lane = 4 v + j, v = 0..15 the virtual thread, j = 0..3 the lane's
// slot in its quad.  The quad of virtual thread g works through g's pixel visits (passes k = 0, 1, 2, ... = pixels g,
// g + 16384, ...) four at a time:
//   phase A  one lane per pixel visit computes the Jacobian row of pass 4 s + j (loads split by data dependence exactly as before:
//            stage 1 = everything the pixel itself addresses, stage 2 = the gathers behind the projective association), for CH
//            steps s at once so that 5 x (6 + 6) loads per lane are in flight.  This phase runs in a LOAD layout (lane = 16 j + v:
//            a quarter-wave covers 16 consecutive pixels of one pass) and the finished rows move to the quad layout through the
//            LDS crossbar (ds_bpermute), see accum_quads;
//   phase B  the four rows of a step are transposed inside the quad (two DPP butterfly stages: lane i ends up with
//            component i of every pass) and accumulated as rank-1 updates A += r r^T in pass order by the matrix pipe:
//            v_mfma_f32_4x4x1_16b_f32 is sixteen independent 4x4 outer products, one per quad, each output element
//            one fmaf(a_i, b_j, c_ij) (bit for bit an f32 FMA, subnormals kept: cdna_hip_programming.md "FP32-input
//            MFMA") -- three of them per pass cover the 28 products + the inlier count of JtJJtrSE3 (types.cuh:98-143):
//            lo x lo, lo x hi, hi x hi with lo = (r0..r3), hi = (r4, r5, r6, found).  Duplicate (i > j) and unused
//            outputs are simply not stored.  The VALU sees only the 32 transposition moves per step; the 12 MFMAs run
//            beside the other wave's phase A.  (EF_NO_FMA builds, where every product is rounded before it is added,
//            do the same outer products with quad broadcasts on the VALU.)
//   tree     warpReduceSum (reduce.cu:57-95): offset 16 is the other half's wave (one LDS exchange, the only barrier of the
//            kernel), offsets 8..1 are lane shuffles by 32..4.
// A workgroup is two virtual warps x {ICP, RGB} x two halves = 8 waves; waves i and i + 4 share a SIMD, so every SIMD
// hosts one ICP wave (memory + ALU heavy) and one RGB wave (light).  256 workgroups = one per CU, one dispatch round.
// The two virtual warps of a workgroup are warps m and m + 4 of one reference block, i.e. a pair that the first level of
// blockReduceSum's 8-warp tree adds: that addition is done here too, and the kernel leaves ONE partial per accumulator
// and pair (256 x 58 floats, plain stores, no hand-over between workgroups: the consumer is the next kernel).
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ACC_NW = 2;   // virtual warps per workgroup

__device__ __forceinline__ float quad_xor1(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float quad_xor2(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
template <int Q>
__device__ __forceinline__ float quad_bcast(float x) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), Q * 0x55, 0xF, 0xF, true)); }
// shfl_down by 32 / 16 / 8 / 4 lanes for the lanes the warp32 tree needs (the lower half, the even rows, the lower lanes of a row):
// two half-wave / row swaps of gfx950 and two in-row DPP shifts -- one VALU instruction each, no trip through the LDS crossbar
__device__ __forceinline__ float down32(float x) {   // valid in lanes 0..31
  const unsigned u = __float_as_uint(x);
  return __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[1]);
}
__device__ __forceinline__ float down16(float x) {   // valid in rows 0 and 2 (lanes 0..15, 32..47)
  const unsigned u = __float_as_uint(x);
  return __uint_as_float(__builtin_amdgcn_permlane16_swap(u, u, false, false)[1]);
}
template <int N>
__device__ __forceinline__ float row_down(float x) {   // lane i of a 16-lane row reads lane i + N of the same row (row_shl:N)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x100 + N, 0xF, 0xF, true));
}
// 4x4 transpose across (lane-in-quad, register): afterwards m[q] of lane i holds what m[i] of lane q held
__device__ __forceinline__ void quad_transpose(float (&m)[4], int j) {
  const bool o1 = (j & 1) != 0, o2 = (j & 2) != 0;
  const float a0 = quad_xor1(m[0]), a1 = quad_xor1(m[1]), a2 = quad_xor1(m[2]), a3 = quad_xor1(m[3]);
  const float s0 = o1 ? a1 : m[0], s1 = o1 ? m[1] : a0, s2 = o1 ? a3 : m[2], s3 = o1 ? m[3] : a2;
  const float b0 = quad_xor2(s0), b1 = quad_xor2(s1), b2 = quad_xor2(s2), b3 = quad_xor2(s3);
  m[0] = o2 ? b2 : s0; m[1] = o2 ? b3 : s1; m[2] = o2 ? s2 : b0; m[3] = o2 ? s3 : b1;
}
// c[i] (lane j of the quad) = fma(a of lane i, b of lane j, c[i]): sixteen 4x4 rank-1 updates per wavefront
__device__ __forceinline__ f32x4 quad_outer(float a, float b, f32x4 c) {
#ifdef EF_NO_FMA
  c.x = EF_FMA(quad_bcast<0>(a), b, c.x);
  c.y = EF_FMA(quad_bcast<1>(a), b, c.y);
  c.z = EF_FMA(quad_bcast<2>(a), b, c.z);
  c.w = EF_FMA(quad_bcast<3>(a), b, c.w);
  return c;
#else
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
#endif
}
// Phase B of one step for the reference-rounding build (round 6).  The DPP formulation above — 8 ds_bpermute, two quad transposes (32 moves /
// selects), then 48 broadcast multiplies + 48 adds per lane — is 136 VALU instructions per step, a third of an ICP step and more than half of a
// photometric one, and the normal-equation kernels are VALU-ISSUE bound (round 6: with every pixel-addressed load of an iteration removed the
// persistent launch was no faster, profiles/r06b_*).  Here the rows cross lanes through LDS instead: the lane that computed the row of pass
// 4 s + jl of virtual thread vl (load layout, lane = 16 jl + vl) writes its 8 values once (two 16-byte stores); lane 4 v + j then reads, for each of
// the step's passes q, the WHOLE row of (v, q) (two 16-byte broadcast reads: the four lanes of a quad read the same address) plus its own two
// components, and the rank-1 updates are plain register arithmetic — c0[i] += R[i] * R[j], c1[i] += R[i] * R[4 + j], c2[i] += R[4 + i] * R[4 + j] —
// which the compiler emits as PACKED f32 multiplies and adds (v_pk_mul_f32 / v_pk_add_f32: two products per instruction, each product rounded, then
// added: the same two roundings as v_mul + v_add).  48 VALU instructions per step instead of 136, 18 LDS instructions (their own issue port).
// Element (i, j) of every block receives the same products in the same (pass) order as quad_outer gives it: bit-identical sums.
// One 2 KB buffer per wavefront (LDS operations of a wavefront execute in order: no barrier, a compiler fence only).
__device__ __forceinline__ float* quad_rowbuf() {
  __shared__ float4 rb[8][128];   // [wavefront of the workgroup (<= 512 threads)][lo4 of lane 0..63 | hi4 of lane 0..63]
  return (float*)rb[(threadIdx.x >> 6) & 7];
}
__device__ __forceinline__ void quad_rows_accumulate(const float (&rows)[8], int s, int K, f32x4 (&c)[3]) {
#ifdef EF_NO_FMA
  const int lane = threadIdx.x & 63, j = lane & 3, v = lane >> 2;
  float* wb = quad_rowbuf();
  f32x4* w4 = (f32x4*)wb;
  w4[lane] = f32x4{rows[0], rows[1], rows[2], rows[3]};
  w4[64 + lane] = f32x4{rows[4], rows[5], rows[6], rows[7]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // Passes in order: the chain of every accumulator is the reference thread's.  No test for 4 s + q < K (the DPP formulation skips the passes
  // beyond the level's last): the row of a visit that does not exist is eight exact +0, its products are +0, and a running sum that started
  // at +0 is never -0, so x + (+0) == x bit for bit — while a uniform branch per pass makes the compiler copy all twelve accumulators at every
  // join (measured in the ISA: 12 v_mov per pass, more than the arithmetic they guard).
  (void)K;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int W = 16 * q + v;   // the lane that computed pass 4 s + q of virtual thread v
    const f32x4 lo4 = w4[W], hi4 = w4[64 + W];
    const float lo = wb[4 * W + j], hi = wb[256 + 4 * W + j];
    c[0] = c[0] + lo4 * lo;
    c[1] = c[1] + lo4 * hi;
    c[2] = c[2] + hi4 * hi;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (the next step's stores stay behind these reads)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#else
  const int lane = threadIdx.x & 63, j = lane & 3;
  const int gather_from = (16 * j + (lane >> 2)) * 4;
  float r[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) r[q] = __int_as_float(__builtin_amdgcn_ds_bpermute(gather_from, __float_as_int(rows[q])));
  float lo[4] = {r[0], r[1], r[2], r[3]};
  float hi[4] = {r[4], r[5], r[6], r[7]};
  quad_transpose(lo, j);
  quad_transpose(hi, j);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (4 * s + q < K) {
      c[0] = quad_outer(lo[q], lo[q], c[0]);
      c[1] = quad_outer(lo[q], hi[q], c[1]);
      c[2] = quad_outer(hi[q], hi[q], c[2]);
    }
  }
#endif
}
// (which outer product, register i, lane-in-quad j) -> JtJJtrSE3 member index, or -1 for duplicates / unused outputs.
// kind 0: lo x lo, 1: lo x hi, 2: hi x hi with lo = row[0..3], hi = (row[4], row[5], row[6], found)
__device__ __forceinline__ int quad_member(int kind, int i, int j) {
  if (kind == 0) return i <= j ? efs::se3_member_of(i, j) : -1;
  if (kind == 1) return j < 3 ? efs::se3_member_of(i, 4 + j) : -1;
  if (i > j) return -1;
  if (j == 3) return i == 3 ? 28 : -1;            // found^2 = found: the inlier count
  if (i == 2) return 27;                          // row[6]^2: the residual
  return efs::se3_member_of(4 + i, 4 + j);
}

struct Se3Inputs {            // device pointers: the Gauss-Newton state the accumulation reads
  const float* Rcurr;         // 9
  const float* tcurr;         // 3
  const float* Rprev_inv;     // 9
  const float* tprev;         // 3
  const int* rgb_slots;       // residual-pass sums (RGB_SLOTS x 16 ints), or null => sigma_fixed
  const int* broken;          // rgbOnly early-exit flag, or null
  float sigma_fixed;
  bool rgbOnly;
  // Workgroup b runs on XCD b % 8 and every XCD has its own L2.  With the swizzle on, the workgroups of one XCD own a
  // CONTIGUOUS range of virtual warps (= of pixels in every pass), so the lines of the model maps that neighbouring virtual
  // warps share through the projective association (shifted 128-byte segments) are fetched once per XCD instead of once per
  // workgroup, and the four workgroups of a reference block sit behind one L2; where a partial sum lands does not change.
  bool xcd_swizzle = true;
  int* slots_zero = nullptr;  // residual-pass slots to re-zero for the next iteration (frame tier), or null
};
struct Se3Out {
  float* pairs;               // [term present in the launch, ICP first][SE3_ACCS][64 blocks][4 pairs]: sum of virtual warps w and w + 4
};
__device__ __forceinline__ void solve_step_wave(TrackState* st, const float* sums, bool broken, bool icp, bool rgb, bool rgbOnly,
                                                float icpWeight, Intr knext, bool level_changes, efs::SolveScratch& S,
                                                const efs::SolvePrefetch& PF);
// In-launch hand-off of partial sums between workgroups (which may sit on different XCDs, each with its own L2):
// payload goes out with agent-scope (write-through) stores, the writer drains them (s_waitcnt vmcnt(0)) before its
// workgroup takes a ticket with a relaxed atomic, and the last arriver reads with agent-scope loads.  No
// __threadfence(): on this part a release fence writes back the whole L2 and ~2000 of them per launch cost 200+ us
// (MI355X_MICROARCH.md, "handoff-flag": sc1 payload -> vmcnt(0) -> flag).
__device__ __forceinline__ float coherent_load(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void coherent_store(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned take_ticket(unsigned* p) { return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Phase A of one pixel-visit, split by data dependence so that a thread has TWO memory round trips in flight at most
// instead of six: stage 1 = everything addressed by the pixel itself (current vertex + normal, packed correspondence,
// image gradients), issued before the pose / sigma are even known; stage 2 = the gathers addressed through the
// projective association (model vertex + normal) and through the photometric correspondence (model depth).  The
// arithmetic is icp_row / rgb_row operation for operation.
struct VisitLoads {
  f3 vcurr, ncurr;
  uint32_t corr;
  int gx, gy;
  bool inb;
};
template <bool HAS_ICP, bool HAS_RGB>
__device__ __forceinline__ VisitLoads visit_stage1(const IcpView& IV, const RgbView& RV, int p, int N) {
  VisitLoads L;
  L.inb = p < N;
  // branch-free: an out-of-range visit (tail of the last pass) reads pixel N-1 and is discarded through `inb`, so the
  // compiler has no control-flow join at which it would have to wait for the loads
  const int q = L.inb ? p : N - 1;
  L.vcurr = L.ncurr = f3{0.f, 0.f, 0.f};
  L.corr = 0u;
  L.gx = L.gy = 0;
  if (HAS_ICP) {
    const int plane = IV.cols * IV.rows;
    L.vcurr = f3{IV.vmap_curr[q], IV.vmap_curr[q + plane], IV.vmap_curr[q + 2 * plane]};
    L.ncurr = f3{IV.nmap_curr[q], IV.nmap_curr[q + plane], IV.nmap_curr[q + 2 * plane]};
  }
  if (HAS_RGB) {
    L.corr = ((const uint32_t*)RV.corres)[q];
    L.gx = RV.dIdx[q];
    L.gy = RV.dIdy[q];
  }
  return L;
}
// stage 2a: addresses of the dependent gathers + the gathers themselves (issued, not waited for)
struct VisitGathers {
  f3 vcurr_g, vprev_g, nprev_g;
  float d0;
  int pidx, zi, zx, zy;
};
template <bool HAS_ICP, bool HAS_RGB>
__device__ __forceinline__ VisitGathers visit_stage2a(const IcpView& IV, const RgbView& RV, const IcpPose& P, const VisitLoads& L) {
  VisitGathers G;
  G.vcurr_g = G.vprev_g = G.nprev_g = f3{0.f, 0.f, 0.f};
  G.d0 = 0.f;
  G.pidx = G.zi = -1;
  G.zx = G.zy = 0;
  if (HAS_ICP && L.inb) {
    G.vcurr_g = mul(P.Rcurr, L.vcurr) + P.tcurr;
    const f3 vcurr_cp = mul(P.Rprev_inv, G.vcurr_g - P.tprev);
    const int ux = f2i_rn(vcurr_cp.x * IV.k.fx / vcurr_cp.z + IV.k.cx);
    const int uy = f2i_rn(vcurr_cp.y * IV.k.fy / vcurr_cp.z + IV.k.cy);
    if (!(ux < 0 || uy < 0 || ux >= IV.cols || uy >= IV.rows || vcurr_cp.z < 0)) G.pidx = uy * IV.cols + ux;
  }
  if (HAS_RGB && L.inb && (L.corr & 0x80000000u)) {
    G.zx = (int)(L.corr & 0x7FFu); G.zy = (int)((L.corr >> 11) & 0x7FFu);
    G.zi = G.zy * RV.cols + G.zx;
  }
  if (G.pidx >= 0) {
    const int plane = IV.cols * IV.rows;
    G.vprev_g = f3{IV.vmap_g_prev[G.pidx], IV.vmap_g_prev[G.pidx + plane], IV.vmap_g_prev[G.pidx + 2 * plane]};
    G.nprev_g = f3{IV.nmap_g_prev[G.pidx], IV.nmap_g_prev[G.pidx + plane], IV.nmap_g_prev[G.pidx + 2 * plane]};
  }
  if (G.zi >= 0) G.d0 = RV.lastDepth[G.zi];
  return G;
}
// stage 2b: the arithmetic of icp_row (reduce.cu:241-309) and rgb_row<PACKED> (reduce.cu:420-476) on the gathered values
template <bool HAS_ICP, bool HAS_RGB>
__device__ __forceinline__ void visit_stage2b(const IcpView& IV, const RgbView& RV, const IcpPose& P, float sigma, const VisitLoads& L,
                                              const VisitGathers& G, float (&irow)[7], float& ifound, float (&grow)[7], float& gfound) {
  ifound = gfound = 0.f;
  if (HAS_ICP && G.pidx >= 0) {
    const f3 ncurr_g = mul(P.Rcurr, L.ncurr);
    const float dist = norm(G.vprev_g - G.vcurr_g);
    const float sine = norm(cross(ncurr_g, G.nprev_g));
    if (sine < IV.angleThres && dist <= IV.distThres && !isnan(L.ncurr.x) && !isnan(G.nprev_g.x)) {
      const f3 s_cp = mul(P.Rprev_inv, G.vcurr_g - P.tprev);
      const f3 d_cp = mul(P.Rprev_inv, G.vprev_g - P.tprev);
      const f3 n_cp = mul(P.Rprev_inv, G.nprev_g);
      const f3 c = cross(s_cp, n_cp);
      irow[0] = n_cp.x; irow[1] = n_cp.y; irow[2] = n_cp.z;
      irow[3] = c.x; irow[4] = c.y; irow[5] = c.z;
      irow[6] = dot(n_cp, s_cp - d_cp);
      ifound = 1.f;
    }
  }
  if (HAS_RGB && G.zi >= 0) {
    const float diff = (float)((int)((L.corr >> 22) & 0x1FFu) - 255);
    float w = sigma + fabsf(diff);
    w = w > 1.19209290E-07F ? 1.0f / w : 1.0f;
    if (sigma == -1) w = 1;
    grow[6] = -w * diff;
    const f3 p = project_point(G.zx, G.zy, G.d0, 1.0f / RV.k.fx, 1.0f / RV.k.fy, RV.k.cx, RV.k.cy);
    const float invz = (float)(1.0 / (double)p.z);
    const float dI_dx_val = w * RV.sobelScale * L.gx;
    const float dI_dy_val = w * RV.sobelScale * L.gy;
    const float v0 = dI_dx_val * RV.k.fx * invz;
    const float v1 = dI_dy_val * RV.k.fy * invz;
    const float v2 = -(v0 * p.x + v1 * p.y) * invz;
    grow[0] = v0; grow[1] = v1; grow[2] = v2;
    grow[3] = -p.z * v1 + p.y * v2;
    grow[4] = p.z * v0 - p.x * v2;
    grow[5] = -p.y * v0 + p.x * v1;
    gfound = 1.f;
  }
}

// ------------------------------------------------------------------------------------------
// TWO visits per lane (round 6; the reference-rounding build).  The normal-equation kernels are bound by VALU issue, and the ISA of one
// visit_stage2a + 2b showed why: ~325 VALU instructions per ICP row for ~210 floating-point operations — 80 v_mov (the compiler pairing scalars
// for its own packed arithmetic and un-pairing them again), 34 of 64-bit address arithmetic, two IEEE square roots whose results are only ever
// compared with a threshold, a branch with its exec-mask bookkeeping around every gate.  Here a lane evaluates the rows of two visits (steps
// 2 u and 2 u + 1 of its virtual thread) at once, every value a two-component vector: the products and sums below compile to v_pk_mul_f32 /
// v_pk_add_f32 (each component one IEEE multiply or add — exactly the scalar instruction's rounding), the gates are branch-free selects, the two
// norms are compared as squares (IcpView::dist2Max / sine2Max: ef_device.hpp, sq_le_max — the same verdict on every input).  133 VALU
// instructions per row.  Operation for operation icp_row / rgb_row: same operands, same order, same roundings; bit-identical rows.
// ------------------------------------------------------------------------------------------
#if defined(EF_NO_FMA) && !defined(EF_NO_VISIT_PAIRS)   // (-DEF_NO_VISIT_PAIRS: the A/B build, build.VARIANTS["nopairs"])
#define EF_VISIT_PAIRS 1
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct p3 { f32x2 x, y, z; };   // a point / vector of each of the two visits
__device__ __forceinline__ f32x2 both(float a) { return f32x2{a, a}; }
__device__ __forceinline__ p3 operator-(p3 a, p3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ p3 operator-(p3 a, f3 b) { return {a.x - both(b.x), a.y - both(b.y), a.z - both(b.z)}; }
__device__ __forceinline__ p3 operator+(p3 a, f3 b) { return {a.x + both(b.x), a.y + both(b.y), a.z + both(b.z)}; }
// ef_device.hpp's dot / cross / mul with EF_FMA(a, b, c) = a * b + c
__device__ __forceinline__ f32x2 dot(f3 a, p3 b) { return both(a.z) * b.z + (both(a.y) * b.y + both(a.x) * b.x); }
__device__ __forceinline__ f32x2 dot(p3 a, p3 b) { return a.z * b.z + (a.y * b.y + a.x * b.x); }
__device__ __forceinline__ p3 cross(p3 a, p3 b) { return {a.y * b.z + (-(a.z * b.y)), a.z * b.x + (-(a.x * b.z)), a.x * b.y + (-(a.y * b.x))}; }
__device__ __forceinline__ p3 mul(const m33& m, p3 a) { return {dot(m.r[0], a), dot(m.r[1], a), dot(m.r[2], a)}; }

// Loads go through buffer instructions: one VGPR byte offset per visit serves all six planes of a pixel (the plane stride rides in the
// instruction's scalar offset), where a global_load needs a 64-bit address per plane — two to three VALU instructions each.
typedef __amdgpu_buffer_rsrc_t bufrsrc;
__device__ __forceinline__ bufrsrc buf_of(const void* base, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ float buf_f32(bufrsrc r, unsigned voff, unsigned soff) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0)); }
__device__ __forceinline__ uint32_t buf_u32(bufrsrc r, unsigned voff) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0); }
__device__ __forceinline__ int buf_i16(bufrsrc r, unsigned voff) { return (int)(int16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)voff, 0, 0); }
__device__ __forceinline__ p3 buf_p3(bufrsrc r, unsigned a4, unsigned b4, unsigned plane4) {   // planar float[3][plane] at byte offsets a4, b4
  return p3{{buf_f32(r, a4, 0u), buf_f32(r, b4, 0u)}, {buf_f32(r, a4, plane4), buf_f32(r, b4, plane4)}, {buf_f32(r, a4, 2u * plane4), buf_f32(r, b4, 2u * plane4)}};
}
struct IcpLoads2 { p3 vcurr, ncurr; bool inb[2]; };
struct IcpGathers2 { p3 vprev_g, nprev_g; int pidx[2]; };
// visit_stage1's ICP half for the visits of pixels pa, pb (N: the visit does not exist)
__device__ __forceinline__ IcpLoads2 icp2_stage1(const IcpView& IV, int pa, int pb, int N) {
  IcpLoads2 L;
  L.inb[0] = pa < N; L.inb[1] = pb < N;
  const unsigned plane4 = (unsigned)(IV.cols * IV.rows) * 4u;
  const unsigned a4 = (unsigned)(L.inb[0] ? pa : N - 1) * 4u, b4 = (unsigned)(L.inb[1] ? pb : N - 1) * 4u;
  L.vcurr = buf_p3(buf_of(IV.vmap_curr, 3u * plane4), a4, b4, plane4);
  L.ncurr = buf_p3(buf_of(IV.nmap_curr, 3u * plane4), a4, b4, plane4);
  return L;
}
// the pose-dependent head of both stages: vcurr_g and s_cp (= visit_stage2a's vcurr_cp = visit_stage2b's s_cp: one expression)
__device__ __forceinline__ void icp2_transform(const IcpPose& P, const IcpLoads2& L, p3& vcurr_g, p3& s_cp) {
  vcurr_g = mul(P.Rcurr, L.vcurr) + P.tcurr;
  s_cp = mul(P.Rprev_inv, vcurr_g - P.tprev);
}
// visit_stage2a's ICP half: the projective association and its gathers (a visit without one reads texel 0 and is discarded through pidx)
__device__ __forceinline__ IcpGathers2 icp2_stage2a(const IcpView& IV, const IcpLoads2& L, const p3& s_cp) {
  IcpGathers2 G;
  const f32x2 px = s_cp.x * both(IV.k.fx) / s_cp.z + both(IV.k.cx);
  const f32x2 py = s_cp.y * both(IV.k.fy) / s_cp.z + both(IV.k.cy);
  unsigned g[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int ux = f2i_rn(px[e]), uy = f2i_rn(py[e]);
    const bool ok = L.inb[e] && !(ux < 0 || uy < 0 || ux >= IV.cols || uy >= IV.rows || s_cp.z[e] < 0);
    G.pidx[e] = ok ? uy * IV.cols + ux : -1;
    g[e] = ok ? (unsigned)(uy * IV.cols + ux) : 0u;
  }
  const unsigned plane4 = (unsigned)(IV.cols * IV.rows) * 4u;
  G.vprev_g = buf_p3(buf_of(IV.vmap_g_prev, 3u * plane4), g[0] * 4u, g[1] * 4u, plane4);
  G.nprev_g = buf_p3(buf_of(IV.nmap_g_prev, 3u * plane4), g[0] * 4u, g[1] * 4u, plane4);
  return G;
}
// visit_stage2b's ICP half: rows[e] = {row[0..6], found} of visit e
__device__ __forceinline__ void icp2_stage2b(const IcpView& IV, const IcpPose& P, const IcpLoads2& L, const IcpGathers2& G, const p3& vcurr_g, const p3& s_cp,
                                             float (&ra)[8], float (&rb)[8]) {
  const p3 ncurr_g = mul(P.Rcurr, L.ncurr);
  const p3 dv = G.vprev_g - vcurr_g;
  const f32x2 dist2 = dot(dv, dv);             // norm(vprev_g - vcurr_g)^2 before the root
  const p3 cr = cross(ncurr_g, G.nprev_g);
  const f32x2 sine2 = dot(cr, cr);             // norm(cross(ncurr_g, nprev_g))^2 before the root
  const p3 d_cp = mul(P.Rprev_inv, G.vprev_g - P.tprev);
  const p3 n_cp = mul(P.Rprev_inv, G.nprev_g);
  const p3 c = cross(s_cp, n_cp);
  const f32x2 r6 = dot(n_cp, s_cp - d_cp);
  bool f[2];
#pragma unroll
  for (int e = 0; e < 2; ++e)   // sine < angleThres && dist <= distThres && !isnan(ncurr.x) && !isnan(nprev_g.x), reduce.cu:263-266
    f[e] = G.pidx[e] >= 0 && sine2[e] <= IV.sine2Max && dist2[e] <= IV.dist2Max && !isnan(L.ncurr.x[e]) && !isnan(G.nprev_g.x[e]);
  ra[0] = f[0] ? n_cp.x[0] : 0.f; ra[1] = f[0] ? n_cp.y[0] : 0.f; ra[2] = f[0] ? n_cp.z[0] : 0.f;
  ra[3] = f[0] ? c.x[0] : 0.f; ra[4] = f[0] ? c.y[0] : 0.f; ra[5] = f[0] ? c.z[0] : 0.f;
  ra[6] = f[0] ? r6[0] : 0.f; ra[7] = f[0] ? 1.f : 0.f;
  rb[0] = f[1] ? n_cp.x[1] : 0.f; rb[1] = f[1] ? n_cp.y[1] : 0.f; rb[2] = f[1] ? n_cp.z[1] : 0.f;
  rb[3] = f[1] ? c.x[1] : 0.f; rb[4] = f[1] ? c.y[1] : 0.f; rb[5] = f[1] ? c.z[1] : 0.f;
  rb[6] = f[1] ? r6[1] : 0.f; rb[7] = f[1] ? 1.f : 0.f;
}
// visit_stage2b's photometric half (rgb_row<true>, reduce.cu:420-476) for two visits: corr = the packed correspondence (valid: bit 31), gx / gy the
// Sobel gradients at the pixel, d0 the model depth at the correspondence
__device__ __forceinline__ void rgb2_rows(const RgbView& RV, float sigma, const uint32_t (&corr)[2], const int (&gx)[2], const int (&gy)[2],
                                          const float (&d0)[2], const bool (&valid)[2], float (&ra)[8], float (&rb)[8]) {
  const f32x2 diff{(float)((int)((corr[0] >> 22) & 0x1FFu) - 255), (float)((int)((corr[1] >> 22) & 0x1FFu) - 255)};
  const f32x2 ws = both(sigma) + f32x2{fabsf(diff[0]), fabsf(diff[1])};
  f32x2 w{ws[0] > 1.19209290E-07F ? 1.0f / ws[0] : 1.0f, ws[1] > 1.19209290E-07F ? 1.0f / ws[1] : 1.0f};
  if (sigma == -1) w = both(1.0f);
  const f32x2 r6 = -w * diff;
  // project_point(zx, zy, d0, 1 / fx, 1 / fy, cx, cy), cudafuncs.cu:670-688
  const f32x2 zx{(float)(int)(corr[0] & 0x7FFu), (float)(int)(corr[1] & 0x7FFu)}, zy{(float)(int)((corr[0] >> 11) & 0x7FFu), (float)(int)((corr[1] >> 11) & 0x7FFu)};
  const float invFx = 1.0f / RV.k.fx, invFy = 1.0f / RV.k.fy;
  const f32x2 pz{d0[0], d0[1]};
  const f32x2 px = (zx - both(RV.k.cx)) * pz * both(invFx), py = (zy - both(RV.k.cy)) * pz * both(invFy);
  const f32x2 invz{(float)(1.0 / (double)pz[0]), (float)(1.0 / (double)pz[1])};
  const f32x2 dI_dx_val = w * both(RV.sobelScale) * f32x2{(float)gx[0], (float)gx[1]};
  const f32x2 dI_dy_val = w * both(RV.sobelScale) * f32x2{(float)gy[0], (float)gy[1]};
  const f32x2 v0 = dI_dx_val * both(RV.k.fx) * invz;
  const f32x2 v1 = dI_dy_val * both(RV.k.fy) * invz;
  const f32x2 v2 = -(v0 * px + v1 * py) * invz;
  const f32x2 r3 = -pz * v1 + py * v2, r4 = pz * v0 - px * v2, r5 = -py * v0 + px * v1;
  ra[0] = valid[0] ? v0[0] : 0.f; ra[1] = valid[0] ? v1[0] : 0.f; ra[2] = valid[0] ? v2[0] : 0.f;
  ra[3] = valid[0] ? r3[0] : 0.f; ra[4] = valid[0] ? r4[0] : 0.f; ra[5] = valid[0] ? r5[0] : 0.f;
  ra[6] = valid[0] ? r6[0] : 0.f; ra[7] = valid[0] ? 1.f : 0.f;
  rb[0] = valid[1] ? v0[1] : 0.f; rb[1] = valid[1] ? v1[1] : 0.f; rb[2] = valid[1] ? v2[1] : 0.f;
  rb[3] = valid[1] ? r3[1] : 0.f; rb[4] = valid[1] ? r4[1] : 0.f; rb[5] = valid[1] ? r5[1] : 0.f;
  rb[6] = valid[1] ? r6[1] : 0.f; rb[7] = valid[1] ? 1.f : 0.f;
}
#endif   // visit pairs

// developer instrumentation (-DEF_ACCUM_CLOCKS, tools/accum_clocks.py): wall_clock64() (100 MHz) stamps of the first ICP wavefront of
// every workgroup of the last level-0 launch
#ifdef EF_ACCUM_CLOCKS
__device__ unsigned long long g_accum_stamps[8][VWARPS / ACC_NW];
#define EF_ASTAMP(i)                                                                                                        \
  do {                                                                                                                      \
    if (N > 8 * VTHREADS && threadIdx.x == 0) g_accum_stamps[i][blockIdx.x] = wall_clock64();                               \
  } while (0)
extern "C" int ef_debug_accum_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_accum_stamps), sizeof(unsigned long long) * 8 * (VWARPS / ACC_NW)) == hipSuccess ? 0 : -1;
}
#else
#define EF_ASTAMP(i) do { } while (0)
#endif

// One wavefront = one half of virtual warp W for ONE term: all passes of its 16 virtual threads, accumulated into c[0..2]
// (lo x lo, lo x hi, hi x hi; register i of lane 4 v + j holds element (i, j) of virtual thread v's 4x4 block).
// slot_a / slot_b: this lane's residual-pass slot (count, sum diff^2) when sigma comes from the slots.
template <int CH, bool ICP, bool PACKED>
__device__ __forceinline__ void accum_quads(const IcpView& IV, const RgbView& RV, const Se3Inputs& in, int gbase, int N, int K,
                                            int slot_a, int slot_b, bool with_slots, f32x4 (&c)[3]) {
  const int S = (K + 3) >> 2;
  const int lane = threadIdx.x & 63;
  // Phase A runs in the LOAD layout, lane = 16 jl + vl: a quarter-wave (16 lanes) covers 16 consecutive pixels of ONE pass, so
  // every planar load touches one 64-byte segment per quarter-wave (four 16-byte pieces of four different rows in the quad
  // layout: 4x the tag look-ups of the CU's one address pipe).  The rows then move to the quad layout (phase B: lane = 4 v + j) through
  // LDS (quad_rows_accumulate; the fast build: ds_bpermute, lane 4 v + j takes lane 16 j + v's, 8 moves per step).
  const int jl = lane >> 4, g = gbase + (lane & 15);
  IcpPose P;
  if (ICP) {
    P.Rcurr = m33_load(in.Rcurr);
    P.tcurr = {in.tcurr[0], in.tcurr[1], in.tcurr[2]};
    P.Rprev_inv = m33_load(in.Rprev_inv);
    P.tprev = {in.tprev[0], in.tprev[1], in.tprev[2]};
  }
  __builtin_amdgcn_sched_barrier(0);   // keep the (scalar) pose loads ahead of the vector loads below: they overlap
#ifdef EF_VISIT_PAIRS
  if constexpr ((ICP || PACKED) && CH % 2 == 0) {   // two visits per lane: steps 2 u, 2 u + 1 of a chunk as one packed evaluation
    constexpr int NP = CH / 2;
    // DEEP: two rounds in flight instead of one.  Built and measured at 1280 x 960 (five rounds of CH = 4): k_se3_accum 22.9 us against 21.5 with
    // one (profiles/r07c_1280x960_kernel_stats*.csv) — a third set of loads queues behind the others in the CU's one address pipe (the same
    // finding as CH = 10 in round 3) and costs a register copy per value and round.  Off; the code stays for the A/B (-DEF_DEEP_PIPE).
#ifdef EF_DEEP_PIPE
    constexpr bool DEEP = CH <= 4;
#else
    constexpr bool DEEP = false;
#endif
    auto pixel = [&](int s) { const int k = 4 * s + jl; return (s < S && k < K) ? k * VTHREADS + g : N; };
    float sigma = in.sigma_fixed;
    // A round = CH steps = NP packed evaluations; the next round's loads are issued before this round's outer products.
    if constexpr (ICP) {
      const IcpPose Pu = icp_pose_uniform(P);
      IcpLoads2 L[NP], Ln[DEEP ? NP : 1];
      IcpGathers2 G[NP];
      p3 vg[NP], scp[NP];
#pragma unroll
      for (int u = 0; u < NP; ++u) L[u] = icp2_stage1(IV, pixel(2 * u), pixel(2 * u + 1), N);
#pragma unroll
      for (int u = 0; u < NP; ++u) {
        icp2_transform(Pu, L[u], vg[u], scp[u]);
        G[u] = icp2_stage2a(IV, L[u], scp[u]);
      }
      EF_ASTAMP(1);
      if constexpr (DEEP) {
#pragma unroll
        for (int u = 0; u < NP; ++u) Ln[u] = L[u];
        if (CH < S) {   // uniform
#pragma unroll
          for (int u = 0; u < NP; ++u) Ln[u] = icp2_stage1(IV, pixel(CH + 2 * u), pixel(CH + 2 * u + 1), N);
        }
      }
#pragma unroll 1
      for (int s0 = 0; s0 < S; s0 += CH) {
        if (!DEEP && s0 > 0) {   // (one round deep: this round's gathers behind its loads, issued before the previous round's outer products)
#pragma unroll
          for (int u = 0; u < NP; ++u) {
            icp2_transform(Pu, L[u], vg[u], scp[u]);
            G[u] = icp2_stage2a(IV, L[u], scp[u]);
          }
        }
        float rows[CH][8];
#pragma unroll
        for (int u = 0; u < NP; ++u) icp2_stage2b(IV, Pu, L[u], G[u], vg[u], scp[u], rows[2 * u], rows[2 * u + 1]);
        if (s0 == 0) EF_ASTAMP(2);
        if (s0 + CH < S) {   // uniform
          if constexpr (DEEP) {   // the next round's gathers, the round after's loads
#pragma unroll
            for (int u = 0; u < NP; ++u) {
              L[u] = Ln[u];
              icp2_transform(Pu, L[u], vg[u], scp[u]);
              G[u] = icp2_stage2a(IV, L[u], scp[u]);
            }
            if (s0 + 2 * CH < S) {
#pragma unroll
              for (int u = 0; u < NP; ++u) Ln[u] = icp2_stage1(IV, pixel(s0 + 2 * CH + 2 * u), pixel(s0 + 2 * CH + 2 * u + 1), N);
            }
          } else {
#pragma unroll
            for (int u = 0; u < NP; ++u) L[u] = icp2_stage1(IV, pixel(s0 + CH + 2 * u), pixel(s0 + CH + 2 * u + 1), N);
          }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (s0 + u < S) quad_rows_accumulate(rows[u], s0 + u, K, c);   // uniform; phase B
        }
      }
    } else {
      const bufrsrc rc = buf_of(RV.corres, (unsigned)N * 4u), rdx = buf_of(RV.dIdx, (unsigned)N * 2u), rdy = buf_of(RV.dIdy, (unsigned)N * 2u),
                    rd0 = buf_of(RV.lastDepth, (unsigned)N * 4u);
      struct Rgb2 { uint32_t corr[2]; int gx[2], gy[2]; bool inb[2]; };
      auto load = [&](int s, auto& X) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
          const int pa = pixel(s + 2 * u), pb = pixel(s + 2 * u + 1);
          X[u].inb[0] = pa < N; X[u].inb[1] = pb < N;
          const unsigned qa = (unsigned)(X[u].inb[0] ? pa : N - 1), qb = (unsigned)(X[u].inb[1] ? pb : N - 1);
          X[u].corr[0] = buf_u32(rc, qa * 4u); X[u].corr[1] = buf_u32(rc, qb * 4u);
          X[u].gx[0] = buf_i16(rdx, qa * 2u); X[u].gx[1] = buf_i16(rdx, qb * 2u);
          X[u].gy[0] = buf_i16(rdy, qa * 2u); X[u].gy[1] = buf_i16(rdy, qb * 2u);
        }
      };
      bool valid[NP][2];
      float d0[NP][2];
      auto gather = [&](const auto& X) {   // visit_stage2a's photometric half: the model depth behind the correspondence
#pragma unroll
        for (int u = 0; u < NP; ++u) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            valid[u][e] = X[u].inb[e] && (X[u].corr[e] & 0x80000000u);
            const unsigned zi = valid[u][e] ? ((X[u].corr[e] >> 11) & 0x7FFu) * (unsigned)RV.cols + (X[u].corr[e] & 0x7FFu) : 0u;
            d0[u][e] = buf_f32(rd0, zi * 4u, 0u);
          }
        }
      };
      Rgb2 X[NP], Xn[DEEP ? NP : 1];
      load(0, X);
      gather(X);
      if constexpr (DEEP) {
#pragma unroll
        for (int u = 0; u < NP; ++u) Xn[u] = X[u];
        if (CH < S) load(CH, Xn);   // uniform
      }
      if (with_slots) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
          slot_a += __shfl_down(slot_a, off, 64);
          slot_b += __shfl_down(slot_b, off, 64);
        }
        sigma = sigma_from_sums(__shfl(slot_b, 0, 64), __shfl(slot_a, 0, 64), in.rgbOnly);
      }
#pragma unroll 1
      for (int s0 = 0; s0 < S; s0 += CH) {
        if (!DEEP && s0 > 0) gather(X);
        float rows[CH][8];
#pragma unroll
        for (int u = 0; u < NP; ++u) rgb2_rows(RV, sigma, X[u].corr, X[u].gx, X[u].gy, d0[u], valid[u], rows[2 * u], rows[2 * u + 1]);
        if (s0 + CH < S) {   // uniform
          if constexpr (DEEP) {   // the next round's gathers, the round after's loads
#pragma unroll
            for (int u = 0; u < NP; ++u) X[u] = Xn[u];
            gather(X);
            if (s0 + 2 * CH < S) load(s0 + 2 * CH, Xn);
          } else {
            load(s0 + CH, X);
          }
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
          if (s0 + u < S) quad_rows_accumulate(rows[u], s0 + u, K, c);   // uniform; phase B
        }
      }
    }
    return;
  }
#endif
  VisitLoads L0[CH];
  if (ICP || PACKED) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int k = 4 * u + jl;
      L0[u] = visit_stage1<ICP, !ICP>(IV, RV, k < K ? k * VTHREADS + g : N, N);
    }
  }
  float sigma = in.sigma_fixed;
  if (!ICP && with_slots) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      slot_a += __shfl_down(slot_a, off, 64);
      slot_b += __shfl_down(slot_b, off, 64);
    }
    sigma = sigma_from_sums(__shfl(slot_b, 0, 64), __shfl(slot_a, 0, 64), in.rgbOnly);
  }
  for (int s0 = 0; s0 < S; s0 += CH) {
    float rows[CH][8];
    if (ICP || PACKED) {
      VisitLoads L[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int k = 4 * (s0 + u) + jl;
        L[u] = s0 == 0 ? L0[u] : visit_stage1<ICP, !ICP>(IV, RV, k < K ? k * VTHREADS + g : N, N);
      }
      VisitGathers G[CH];
#pragma unroll
      for (int u = 0; u < CH; ++u) G[u] = visit_stage2a<ICP, !ICP>(IV, RV, P, L[u]);
      if (ICP && s0 == 0) EF_ASTAMP(1);   // stage-1 data consumed, gathers issued
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        float irow[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, grow[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ifound, gfound;
        visit_stage2b<ICP, !ICP>(IV, RV, P, sigma, L[u], G[u], irow, ifound, grow, gfound);
#pragma unroll
        for (int q = 0; q < 7; ++q) rows[u][q] = ICP ? irow[q] : grow[q];
        rows[u][7] = ICP ? ifound : gfound;
      }
      if (ICP && s0 == 0) EF_ASTAMP(2);   // gathers consumed, rows computed
    } else {   // operator tier's photometric term: 16-byte DataTerm + explicit point cloud (types.cuh:81-86)
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int k = 4 * (s0 + u) + jl;
        const int p = k < K ? k * VTHREADS + g : N;
        float row[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const bool found = p < N && rgb_row<false>(RV, sigma, p, row);
#pragma unroll
        for (int q = 0; q < 7; ++q) rows[u][q] = row[q];
        rows[u][7] = found ? 1.f : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      if (s0 + u < S) quad_rows_accumulate(rows[u], s0 + u, K, c);   // uniform; phase B
    }
  }
}

// accum_quads for BOTH halves of one virtual warp in ONE wavefront, for the small levels (K <= 8 passes: at most two steps per half):
// the persistent kernel (k_track_small) gives a wavefront the 32 virtual threads of a warp, so that 8 wavefronts (2 per SIMD, 256
// registers each) cover what 16 cover in k_se3_accum.  All four (half, step) visits have their loads in flight together; each half
// accumulates into its own c[half][0..2] in pass order, so every accumulator sees the chain of additions accum_quads gives it.
// `before_rows(slot_a, slot_b)`: called once every load and gather is issued and before anything needs sigma — the photometric wavefronts of
// the persistent kernel wait THERE for the correspondence search's global count, with their memory round trips already under way.
struct NoWait { __device__ __forceinline__ void operator()(int&, int&) const {} };
template <bool ICP, typename Hook = NoWait>
__device__ __forceinline__ void accum_quads_halves(const IcpView& IV, const RgbView& RV, const Se3Inputs& in, int wbase, int N, int K,
                                                   int slot_a, int slot_b, bool with_slots, f32x4 (&c)[2][3], Hook before_rows = Hook()) {
  constexpr int SU = 2, CH = 2 * SU;
  const int S = (K + 3) >> 2;   // <= SU
  const int lane = threadIdx.x & 63;
  const int j = lane & 3;
  const int jl = lane >> 4, vl = lane & 15;                 // load layout, see accum_quads
  const int gather_from = (16 * j + (lane >> 2)) * 4;
  IcpPose P;
  if (ICP) {
    P.Rcurr = m33_load(in.Rcurr);
    P.tcurr = {in.tcurr[0], in.tcurr[1], in.tcurr[2]};
    P.Rprev_inv = m33_load(in.Rprev_inv);
    P.tprev = {in.tprev[0], in.tprev[1], in.tprev[2]};
  }
  VisitLoads L[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int hf = u / SU, sp = u % SU, k = 4 * sp + jl;
    L[u] = visit_stage1<ICP, !ICP>(IV, RV, (sp < S && k < K) ? k * VTHREADS + wbase + 16 * hf + vl : N, N);
  }
  VisitGathers G[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) G[u] = visit_stage2a<ICP, !ICP>(IV, RV, P, L[u]);
  before_rows(slot_a, slot_b);
  float sigma = in.sigma_fixed;
  if (!ICP && with_slots) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      slot_a += __shfl_down(slot_a, off, 64);
      slot_b += __shfl_down(slot_b, off, 64);
    }
    sigma = sigma_from_sums(__shfl(slot_b, 0, 64), __shfl(slot_a, 0, 64), in.rgbOnly);
  }
  float rows[CH][8];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    float irow[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, grow[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ifound, gfound;
    visit_stage2b<ICP, !ICP>(IV, RV, P, sigma, L[u], G[u], irow, ifound, grow, gfound);
#pragma unroll
    for (int q = 0; q < 7; ++q) rows[u][q] = ICP ? irow[q] : grow[q];
    rows[u][7] = ICP ? ifound : gfound;
  }
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int hf = u / SU, sp = u % SU;
    if (sp < S) {   // uniform
#pragma unroll
      for (int q = 0; q < 8; ++q) rows[u][q] = __int_as_float(__builtin_amdgcn_ds_bpermute(gather_from, __float_as_int(rows[u][q])));
      float lo[4] = {rows[u][0], rows[u][1], rows[u][2], rows[u][3]};
      float hi[4] = {rows[u][4], rows[u][5], rows[u][6], rows[u][7]};
      quad_transpose(lo, j);
      quad_transpose(hi, j);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (4 * sp + q < K) {   // uniform; passes in order
          c[hf][0] = quad_outer(lo[q], lo[q], c[hf][0]);
          c[hf][1] = quad_outer(lo[q], hi[q], c[hf][1]);
          c[hf][2] = quad_outer(hi[q], hi[q], c[hf][2]);
        }
      }
    }
  }
}

template <int CH, bool HAS_ICP, bool HAS_RGB, bool PACKED>
__global__ void __launch_bounds__(64 * 2 * ACC_NW * ((HAS_ICP && HAS_RGB) ? 2 : 1))
k_se3_accum(const IcpView IV, const RgbView RV, const Se3Inputs in, const Se3Out out) {
  constexpr int NT = (HAS_ICP && HAS_RGB) ? 2 : 1;
  __shared__ float xch[ACC_NW][NT][12][64];
  __shared__ float pair[NT][12][4];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int v = lane >> 2, j = lane & 3;
  const int wg = in.xcd_swizzle ? (int)((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const int tix = wave / (2 * ACC_NW);                       // 0: the launch's first term
  const bool rgb_wave = HAS_RGB && (!HAS_ICP || tix == 1);
  const int rem = wave % (2 * ACC_NW), wl = rem >> 1, half = rem & 1;
  const int W = 8 * (wg >> 2) + (wg & 3) + 4 * wl;   // workgroup 4 b + m owns virtual warps m and m + 4 of reference block b
  const int gbase = W * 32 + half * 16;   // first of this wavefront's 16 virtual threads
  const int cols = HAS_ICP ? IV.cols : RV.cols, nrows = HAS_ICP ? IV.rows : RV.rows;
  const int N = cols * nrows;
  const int K = (N + VTHREADS - 1) / VTHREADS;
  const int broken = in.broken ? *in.broken : 0;
  const bool with_slots = HAS_RGB && in.rgb_slots;
  int slot_a = 0, slot_b = 0;
  if (rgb_wave && with_slots) { slot_a = in.rgb_slots[lane * 16]; slot_b = in.rgb_slots[lane * 16 + 1]; }
  // re-arm the residual-pass slots the NEXT iteration's correspondence search adds into (nobody reads them during this launch)
  if (in.slots_zero && blockIdx.x == 0 && t < RGB_SLOTS) { in.slots_zero[t * 16] = 0; in.slots_zero[t * 16 + 1] = 0; }
  if (broken) return;  // rgbOnly "break": the level is over (the update step does the bookkeeping)
  EF_ASTAMP(0);
  f32x4 c[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) c[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (HAS_ICP && !rgb_wave) accum_quads<CH, true, PACKED>(IV, RV, in, gbase, N, K, 0, 0, false, c);
  if (HAS_RGB && rgb_wave) accum_quads<CH, false, PACKED>(IV, RV, in, gbase, N, K, slot_a, slot_b, with_slots, c);
  EF_ASTAMP(3);   // transposes + outer products issued
  // warpReduceSum, reduce.cu:57-95: val += shfl_down(val, offset) for offset = 16 (the other half's wave), 8, 4, 2, 1
  if (half == 1) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      xch[wl][tix][q * 4 + 0][lane] = c[q].x; xch[wl][tix][q * 4 + 1][lane] = c[q].y;
      xch[wl][tix][q * 4 + 2][lane] = c[q].z; xch[wl][tix][q * 4 + 3][lane] = c[q].w;
    }
  }
  __syncthreads();
  EF_ASTAMP(4);   // every wavefront of the workgroup has finished its passes
  float r[12];
  if (half == 0) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      r[q * 4 + 0] = c[q].x + xch[wl][tix][q * 4 + 0][lane]; r[q * 4 + 1] = c[q].y + xch[wl][tix][q * 4 + 1][lane];
      r[q * 4 + 2] = c[q].z + xch[wl][tix][q * 4 + 2][lane]; r[q * 4 + 3] = c[q].w + xch[wl][tix][q * 4 + 3][lane];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {   // offsets 8, 4, 2, 1 in virtual threads = 32, 16, 8, 4 lanes
      r[i] += down32(r[i]);
      r[i] += down16(r[i]);
      r[i] += row_down<8>(r[i]);
      r[i] += row_down<4>(r[i]);
    }
    // blockReduceSum's second stage (reduce.cu:97-117), first level: the 8 warp sums of a reference block sit in lanes 0..7 of its
    // warp 0 (the other 24 lanes hold exact zeros) and shfl_down(offset 4) adds warp w + 4 to warp w: the workgroup holds exactly
    // such a pair, the upper one hands its sums over
    if (wl == 1 && v == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) pair[tix][i][j] = r[i];
    }
  }
  __syncthreads();
  if (half == 0 && wl == 0 && v == 0) {
    float* dst = out.pairs + (size_t)tix * SE3_ACCS * SE3_PAIRS + wg;   // wg = 4 * reference block + pair
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int a = quad_member(q, i, j);
        if (a >= 0) dst[(size_t)a * SE3_PAIRS] = r[q * 4 + i] + pair[tix][q * 4 + i][j];
      }
  }
  EF_ASTAMP(5);
}

#ifdef EF_FAST_ORDER
#include "ef_track_fast.inc"
#endif

// The rest of the reference tree over the 512 virtual-warp partials of `na` accumulators (acc-major):
//   blockReduceSum's second stage: lanes 0..7 of warp 0 hold the 8 warp sums, the other 24 lanes hold 0.0f
//     => shfl_down tree of width 8 (offsets 4, 2, 1);
//   reduceSum<<<1,1024>>>: threads 0..63 hold the 64 block partials => two warp32 trees, then shared[0] + shared[1].
// bs: na*64 floats of LDS, out: na floats of LDS.  COHERENT: read the partials with agent-scope atomic loads (used
// by the last-workgroup-done pattern, where the partials were written by other workgroups of the same launch).
// All loads of a thread are issued before the first shuffle (NA is a compile-time count): one memory round trip, not NA.
template <int BLOCK, int NA, bool COHERENT>
__device__ __forceinline__ void final_tree(const float* partials, float* bs, float* out) {
  static_assert((NA * VWARPS) % 64 == 0 && BLOCK % 64 == 0, "whole waves");
  constexpr int TOTAL = NA * VWARPS, PASSES = (TOTAL + BLOCK - 1) / BLOCK;   // the last pass may be ragged (whole waves are in or out)
  const int t = threadIdx.x;
  float v[PASSES];
#pragma unroll
  for (int j = 0; j < PASSES; ++j) {
    const int idx = t + j * BLOCK, q = idx < TOTAL ? idx : 0;
    v[j] = COHERENT ? __hip_atomic_load(partials + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : partials[q];
  }
#pragma unroll
  for (int j = 0; j < PASSES; ++j) {
    const int idx = t + j * BLOCK;
    float x = v[j];
    x += __shfl_down(x, 4, 8);
    x += __shfl_down(x, 2, 8);
    x += __shfl_down(x, 1, 8);
    if (idx < TOTAL && (idx & 7) == 0) bs[idx >> 3] = x;   // [acc][block]
  }
  __syncthreads();
  for (int idx = t; idx < NA * 64; idx += BLOCK) {
    float x = bs[idx];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) x += __shfl_down(x, off, 32);
    const float w1 = __shfl(x, 32, 64);
    if ((idx & 63) == 0) out[idx >> 6] = x + w1;
  }
  __syncthreads();
}

// unpack 29 floats -> symmetric A[36], b[6] (reduce.cu:385-400)
template <typename T>
__host__ __device__ inline void unpack29(const float* h, T* A, T* b) {
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const float value = h[shift++];
      if (j == 6) b[i] = (T)value;
      else A[j * 6 + i] = A[i * 6 + j] = (T)value;
    }
}

// K R K^-1 and K t for the coming iteration (RGBDOdometry.cpp:395-417)
__device__ inline void compute_krk(const double* resultRt, Intr k, float* krkinv, float* kt) {
  double Rt[16];
  efl::m4_affine_inverse(resultRt, Rt);
  double R[9], K[9] = {k.fx, 0, k.cx, 0, k.fy, k.cy, 0, 0, 1}, Kinv[9], KR[9], KRK[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r * 3 + c] = Rt[r * 4 + c];
  efl::m3_inverse<double>(K, Kinv);
  efl::m3_mul(K, R, KR);
  efl::m3_mul(KR, Kinv, KRK);
  for (int i = 0; i < 9; ++i) krkinv[i] = (float)KRK[i];
  const double tv[3] = {Rt[3], Rt[7], Rt[11]};
  double Kt[3];
  efl::m3_mulv(K, tv, Kt);
  for (int i = 0; i < 3; ++i) kt[i] = (float)Kt[i];
}

// float matrices of one SO(3) iteration: homography K R K^-1, K^-1, K R (RGBDOdometry.cpp:309-316)
__device__ inline void so3_matrices(const double* resultR, Intr k, float* mats27) {
  double K[9] = {k.fx, 0, k.cx, 0, k.fy, k.cy, 0, 0, 1}, Kinv[9], KR[9], H[9];
  efl::m3_inverse<double>(K, Kinv);
  efl::m3_mul(K, resultR, KR);
  efl::m3_mul(KR, Kinv, H);
  for (int i = 0; i < 9; ++i) { mats27[i] = (float)H[i]; mats27[9 + i] = (float)Kinv[i]; mats27[18 + i] = (float)KR[i]; }
}

// first kernel of getIncrementalTransformation: Rprev/tprev/Rcurr/tcurr (RGBDOdometry.cpp:266-273,375-377), the
// SO(3) loop state or — without SO(3) — the first level's K R K^-1; re-arms the denseEnough() tally (its consumers,
// the model-side pyramids, have run by now).
__global__ void k_track_begin(TrackState* st, bool so3, Intr kso3, Intr kfirst) {
  if (threadIdx.x != 0) return;
  double R[9];
  efl::quat_to_mat<double>(st->q, R);
  GNState& g = st->gn[0];   // track() starts in buffer 0
  for (int i = 0; i < 9; ++i) st->Rprev[i] = g.Rcurr[i] = (float)R[i];
  for (int i = 0; i < 3; ++i) st->tprev[i] = g.tcurr[i] = (float)st->t[i];
  efl::m3_inverse<float>(st->Rprev, st->Rprev_inv);
  for (int i = 0; i < 4; ++i) st->q_prev[i] = st->q[i];
  for (int i = 0; i < 3; ++i) st->t_prev[i] = st->t[i];
  efl::m4_identity(g.resultRt);
  for (int i = 0; i < RGB_SLOTS; ++i) st->rgb_slots[0][i][0] = st->rgb_slots[0][i][1] = st->rgb_slots[1][i][0] = st->rgb_slots[1][i][1] = 0;
  g.rgb_broken = 0;
  g.lastRGBErrorLevel = 3.402823466e+38f;
  st->call_seq += 1u;
  st->so3_iterations = 0;
  st->dbg_clock[11] = ~0ull; st->dbg_clock[12] = 0;
  st->so3_ticket = 0;
  st->dense_count = 0;
  if (so3) {
    efl::m3_identity(st->so3_resultR);
    efl::m3_identity(st->so3_lastResultR);
    for (int i = 0; i < 9; ++i) st->so3_R_lr[i] = (i % 4 == 0) ? 1.f : 0.f;
    st->so3_lastError = 3.402823466e+38f / 2;
    st->so3_lastCount = 3.402823466e+38f / 2;
    st->so3_done = 0;
    so3_matrices(st->so3_resultR, kso3, st->so3_mats);
  } else {
    st->so3_done = 1;
    compute_krk(g.resultRt, kfirst, g.krkinv, g.kt);
  }
}

constexpr int SOLVE_BLOCK = 512;
// K6c: the Gauss-Newton update (RGBDOdometry.cpp:440-551 + OdometryProvider.h:73-96) on one wavefront, one matrix
// element per lane (ef_solve_dev.hpp); evaluated at the head of k_track_step / k_track_end by every workgroup.
// bookkeeping around the update (rgbOnly early exit, statistics); all lanes of wave 0 take the same path.  S receives what the
// calling workgroup needs (krkinv, kt, Rcurr, tcurr, broken); with `publish` the whole GNState goes to `next` and the
// statistics to st.
template <bool SPLIT_TAIL>
__device__ __forceinline__ void solve_step_wave(TrackState* st, const GNState* prev, GNState* next, bool publish, const float* sums,
                                                const StepArgs& A, efs::SolveScratch& S, const efs::SolvePrefetch& PF, bool stats) {
  const int lane = threadIdx.x & 63;
  int sigma = PF.slot_b, rgbSize = PF.slot_a;   // {count, sum diff^2} slots of the residual pass, prefetched
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    rgbSize += __shfl_down(rgbSize, off, 64);
    sigma += __shfl_down(sigma, off, 64);
  }
  rgbSize = __shfl(rgbSize, 0, 64);
  sigma = __shfl(sigma, 0, 64);
  efs::solve_prefetch_publish(PF, S);
  const float lastLevelErr = PF.lastRGBErrorLevel;
  const bool broken = PF.broken != 0;
  const float rgbError = (float)(sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));
  const bool brk = !broken && A.rgbOnly && rgbError > lastLevelErr;   // "break": skip the rest of this level
  if (broken || brk) {
    // nothing of the pose changes; at a level change the flag clears and K R K^-1 / K t are re-evaluated for the new level
    if (lane == 0) {
      S.tail_pending = 0;
      S.broken = A.level_changes ? 0 : 1;
      if (A.level_changes) {
        compute_krk(prev->resultRt, A.knext, S.krkinv, S.kt);
      } else {
        for (int i = 0; i < 9; ++i) S.krkinv[i] = prev->krkinv[i];
        for (int i = 0; i < 3; ++i) S.kt[i] = prev->kt[i];
      }
      for (int i = 0; i < 9; ++i) S.Rcurr[i] = prev->Rcurr[i];
      for (int i = 0; i < 3; ++i) S.tcurr[i] = prev->tcurr[i];
      if (publish) {
        for (int i = 0; i < 9; ++i) { next->Rcurr[i] = S.Rcurr[i]; next->krkinv[i] = S.krkinv[i]; }
        for (int i = 0; i < 3; ++i) { next->tcurr[i] = S.tcurr[i]; next->kt[i] = S.kt[i]; }
        for (int i = 0; i < 16; ++i) next->resultRt[i] = prev->resultRt[i];
        next->rgb_broken = S.broken;
        next->lastRGBErrorLevel = A.level_changes ? 3.402823466e+38f : lastLevelErr;
      }
    }
    efs::wave_sync();
    return;
  }
  if (lane == 0) {
    S.broken = 0;
    if (publish) {
      next->rgb_broken = 0;
      next->lastRGBErrorLevel = A.level_changes ? 3.402823466e+38f : rgbError;
    }
    if (stats) {
      st->lastRGBError = rgbError;
      st->lastRGBCount = (float)rgbSize;
      if (A.icp) {
        st->lastICPError = sqrtf(sums[27]) / sums[28];
        st->lastICPCount = sums[28];
      } else {
        // RGBDOdometry.cpp:492-493 evaluates sqrt(residual[0]) / residual[1] on an UNINITIALISED residual[] when the ICP term is
        // off; the specification (oracle) zero-initialises it: NaN error, zero count
        st->lastICPError = __int_as_float(0x7fc00000);
        st->lastICPCount = 0.f;
      }
    }
  }
  EF_STAMP(st, 3);
  efs::SolveInputs in{A.icp, A.rgb, A.rgbOnly, A.icpWeight, A.knext, A.level_changes};
  efs::gauss_newton_update_wave<SPLIT_TAIL>(st, next, publish, sums, in, S, stats);
}
// reduceSum over what k_se3_accum leaves: pairs[term][acc][block][pair] -> the rest of blockReduceSum's 8-warp tree (offsets 2, 1
// over the four pair sums of a block: (p0 + p2) + (p1 + p3)), then reduceSum<<<1,1024>>> over the 64 block partials (two warp32
// trees + one add).  sums_s[term * SE3_ACCS + acc]; call with the whole workgroup (BLOCK threads), follow with __syncthreads().
// One 16-byte load per (acc, block), all of a thread's loads in flight together, one wavefront per accumulator.
// COHERENT: the partials were written by other workgroups of the SAME launch (agent-scope stores, drained, then a grid barrier): read
// them with agent-scope loads, which no CU's L1 serves (two 8-byte loads per float4)
// pair_partials_tree for the persistent kernel's own iterations: slot 0 of a block holds p0 + p2 and slot 1 holds p1 + p3 already (each
// workgroup owns pairs h and h + 2 of a reference block and adds them before it publishes), so the gather is one 8-byte agent-scope
// load per (accumulator, block) — half the bytes and half the publishing stores — and x = s02 + s13 is the same (p0 + p2) + (p1 + p3).
template <int BLOCK>
__device__ __forceinline__ void half_partials_tree(const float* __restrict__ pairs, bool icp, bool rgb, float* sums_s) {
  static_assert(BLOCK % 64 == 0, "one wavefront per accumulator");
  const int t = threadIdx.x;
  const int na = (icp ? SE3_ACCS : 0) + (rgb ? SE3_ACCS : 0);
  constexpr int PASSES = (2 * SE3_ACCS * 64 + BLOCK - 1) / BLOCK;
  unsigned long long v[PASSES];
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int idx = t + q * BLOCK;
    v[q] = idx < na * 64 ? __hip_atomic_load((const unsigned long long*)pairs + (size_t)idx * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
  }
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int idx = t + q * BLOCK;
    float x = __uint_as_float((unsigned)v[q]) + __uint_as_float((unsigned)(v[q] >> 32));
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) x += __shfl_down(x, off, 32);
    const float w1 = __shfl(x, 32, 64);
    if (idx < na * 64 && (idx & 63) == 0) sums_s[(icp ? 0 : SE3_ACCS) + (idx >> 6)] = x + w1;
  }
}
template <int BLOCK, bool COHERENT>
__device__ __forceinline__ void pair_partials_tree(const float* __restrict__ pairs, bool icp, bool rgb, float* sums_s) {
  static_assert(BLOCK % 64 == 0, "one wavefront per accumulator");
  const int t = threadIdx.x;
  const int na = (icp ? SE3_ACCS : 0) + (rgb ? SE3_ACCS : 0);
  constexpr int PASSES = (2 * SE3_ACCS * 64 + BLOCK - 1) / BLOCK;
  float4 v[PASSES];
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int idx = t + q * BLOCK;
    if (COHERENT) {
      unsigned long long lo = 0ull, hi = 0ull;
      if (idx < na * 64) {
        const unsigned long long* src = (const unsigned long long*)pairs + (size_t)idx * 2;
        lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      v[q] = make_float4(__uint_as_float((unsigned)lo), __uint_as_float((unsigned)(lo >> 32)), __uint_as_float((unsigned)hi),
                         __uint_as_float((unsigned)(hi >> 32)));
    } else {
      v[q] = idx < na * 64 ? ((const float4*)pairs)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int q = 0; q < PASSES; ++q) {
    const int idx = t + q * BLOCK;
    float x = (v[q].x + v[q].z) + (v[q].y + v[q].w);
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) x += __shfl_down(x, off, 32);
    const float w1 = __shfl(x, 32, 64);
    if (idx < na * 64 && (idx & 63) == 0) sums_s[(icp ? 0 : SE3_ACCS) + (idx >> 6)] = x + w1;
  }
}

template <int BLOCK>
__device__ __forceinline__ void head_sums(const float* __restrict__ pairs, int ng, bool icp, bool rgb, float* sums_s) {
#ifdef EF_FAST_ORDER
  fast_tree<BLOCK, false>(pairs, ng, (icp ? SE3_ACCS : 0) + (rgb ? SE3_ACCS : 0), sums_s + (icp ? 0 : SE3_ACCS));
#else
  if (ng == 1) {   // the persistent launch's reducers left the TOTALS in column 0 of every accumulator (k_track_ref)
    const int na = (icp ? SE3_ACCS : 0) + (rgb ? SE3_ACCS : 0);
    if ((int)threadIdx.x < na) sums_s[(icp ? 0 : SE3_ACCS) + threadIdx.x] = pairs[(size_t)threadIdx.x * 256];
  } else {
    pair_partials_tree<BLOCK>(pairs, icp, rgb, sums_s);
  }
#endif
}

// tail of getIncrementalTransformation (RGBDOdometry.cpp:555-570) + velocity weighting
// (ElasticFusion.cpp:369-383) + the float matrices the map kernels consume.
__device__ inline void publish_pose(TrackState* st) {
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = st->q[i];
  for (int i = 0; i < 3; ++i) T.t[i] = st->t[i];
  efl::se3_inverse_matrix_f(T, st->T_cw);
  efl::se3_castf_matrix(T, st->pose_f);
  double R[9];
  efl::quat_to_mat<double>(T.q, R);
  for (int i = 0; i < 9; ++i) st->R_wc_f[i] = (float)R[i];
  for (int i = 0; i < 3; ++i) st->t_wc_f[i] = (float)T.t[i];
}
__device__ inline void compute_weighting(TrackState* st, float weightMultiplier) {
  efl::SE3 T, Tp;
  for (int i = 0; i < 4; ++i) { T.q[i] = st->q[i]; Tp.q[i] = st->q_prev[i]; }
  for (int i = 0; i < 3; ++i) { T.t[i] = st->t[i]; Tp.t[i] = st->t_prev[i]; }
  const efl::SE3 Tcp = efl::se3_mul(efl::se3_inverse(T), Tp);
  const double tn = sqrt(Tcp.t[0] * Tcp.t[0] + Tcp.t[1] * Tcp.t[1] + Tcp.t[2] * Tcp.t[2]);
  const double ln = efl::se3_log_norm(Tcp);
  float weighting = (float)(tn > ln ? tn : ln);
  const float largest = 0.01f, minWeight = 0.5f;
  if (weighting > largest) weighting = largest;
  const float w = 1.0f - (weighting / largest);
  st->weighting = (w > minWeight ? w : minWeight) * weightMultiplier;
}
// t_T_wc.push_back(T_wc), ElasticFusion.cpp:588: one 4x4 double matrix per frame in a device-resident log
__device__ inline void log_pose(const TrackState* st, double* traj, int slot) {
  if (!traj) return;
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = st->q[i];
  for (int i = 0; i < 3; ++i) T.t[i] = st->t[i];
  efl::se3_matrix(T, traj + (size_t)slot * 16);
}
// Last kernel of getIncrementalTransformation: the update step of the LAST iteration (same head as k_track_step; with no iteration at
// all the pose is prev's), then the tail on one lane: 0.3 m guard, SVD re-orthonormalisation (RGBDOdometry.cpp:555-570),
// velocity weighting (ElasticFusion.cpp:369-383), the float matrices of the map passes, the trajectory log.
__device__ __forceinline__ void track_end_tail(TrackState* st, const float* Rc_in, const float* tc_in, bool rgb, float weightMultiplier, double* traj, int slot);
__global__ void __launch_bounds__(REDUCE_BLOCK) k_track_end(TrackState* st, const GNState* __restrict__ prev, GNState* next,
                                                             const float* __restrict__ pairs, const int* __restrict__ slots_prev, const StepArgs A,
                                                             bool rgb, float weightMultiplier, double* traj, int slot, const unsigned* abort_word,
                                                             unsigned* abort_report) {
  __shared__ efs::SolveScratch S;
  __shared__ float sums_s[2 * SE3_ACCS];
  const int t = threadIdx.x;
  if (A.has_head) {
    efs::SolvePrefetch PF{};
    if (t < 64) PF = efs::solve_prefetch(st, prev, slots_prev);
    head_sums<REDUCE_BLOCK>(pairs, A.ng, A.icp, A.rgb, sums_s);
    __syncthreads();
    if (t < 64) solve_step_wave(st, prev, next, true, sums_s, A, S, PF, true);
    __syncthreads();
  } else if (t == 0) {
    for (int i = 0; i < 9; ++i) S.Rcurr[i] = prev->Rcurr[i];
    for (int i = 0; i < 3; ++i) S.tcurr[i] = prev->tcurr[i];
  }
  if (t != 0) return;
  track_end_tail(st, S.Rcurr, S.tcurr, rgb, weightMultiplier, traj, slot);
  // the sticky "a persistent launch gave up waiting" flag of this tracker instance, handed to the HOST (a word of mapped pinned memory the
  // next ef_process_frame looks at without synchronising): a protocol failure is reported where the front end calls, not only by ef_synchronize
  if (abort_report) {
    const unsigned v = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v) __hip_atomic_store(abort_report, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
// (one lane) 0.3 m guard, SVD re-orthonormalisation, velocity weighting, the float matrices of the map passes, the trajectory log
__device__ __forceinline__ void track_end_tail(TrackState* st, const float* Rc_in, const float* tc_in, bool rgb, float weightMultiplier, double* traj, int slot) {
  float Rcurr[9], tcurr[3];
  for (int i = 0; i < 9; ++i) Rcurr[i] = Rc_in[i];
  for (int i = 0; i < 3; ++i) tcurr[i] = tc_in[i];
  if (rgb) {
    const float d0 = tcurr[0] - st->tprev[0], d1 = tcurr[1] - st->tprev[1], d2 = tcurr[2] - st->tprev[2];
    if ((double)sqrtf(d0 * d0 + d1 * d1 + d2 * d2) > 0.3) {
      for (int i = 0; i < 9; ++i) Rcurr[i] = st->Rprev[i];
      for (int i = 0; i < 3; ++i) tcurr[i] = st->tprev[i];
    }
  }
  double Rc[9], Rp[9];
  for (int i = 0; i < 9; ++i) Rc[i] = (double)Rcurr[i];
  efl::polar3(Rc, Rp);
  efl::SE3 T;
  efl::se3_set_rotation(T, Rp);
  for (int i = 0; i < 4; ++i) st->q[i] = T.q[i];
  for (int i = 0; i < 3; ++i) st->t[i] = (double)tcurr[i];
  publish_pose(st);
  compute_weighting(st, weightMultiplier);
  log_pose(st, traj, slot);
}
// pose injected by the caller (in_T_wc != 0, ElasticFusion.cpp:367-369)
__global__ void k_pose_injected(TrackState* st, efl::SE3 T, bool save_prev, float weightMultiplier, bool with_weighting, double* traj,
                                int slot) {
  if (threadIdx.x != 0) return;
  if (save_prev) {
    for (int i = 0; i < 4; ++i) st->q_prev[i] = st->q[i];
    for (int i = 0; i < 3; ++i) st->t_prev[i] = st->t[i];
  }
  for (int i = 0; i < 4; ++i) st->q[i] = T.q[i];
  for (int i = 0; i < 3; ++i) st->t[i] = T.t[i];
  publish_pose(st);
  if (with_weighting) compute_weighting(st, weightMultiplier);
  st->dense_count = 0;
  log_pose(st, traj, slot);
}
// model-to-model tracking of the local loop closure (ElasticFusion.cpp:469-471): T_wc_est starts as a copy of T_wc_curr
__global__ void k_copy_pose(TrackState* dst, const TrackState* src) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 4; ++i) dst->q[i] = src->q[i];
  for (int i = 0; i < 3; ++i) dst->t[i] = src->t[i];
  publish_pose(dst);
}
// T_wc_curr = T_wc_est (ElasticFusion.cpp:525) after an accepted deformation; the logged pose of this frame follows (:588)
__global__ void k_adopt_pose(TrackState* st, const TrackState* est, double* traj, int slot) {
  if (threadIdx.x != 0) return;
  for (int i = 0; i < 4; ++i) st->q[i] = est->q[i];
  for (int i = 0; i < 3; ++i) st->t[i] = est->t[i];
  publish_pose(st);
  log_pose(st, traj, slot);
}
// Resize::vertex + Resize::time (Resize.cpp:85-159) for the constraint grid: texel (20a+10, 20b+10) of the ACTIVE vertex map
// and of the INACTIVE time map -> out[(a * ch + b)] = {x, y, z, time} in the order ElasticFusion.cpp:488-489 walks them
__global__ void k_sample_constraints(const float4* __restrict__ vertex, const uint16_t* __restrict__ old_time, int cols, int cw, int ch,
                                     int step, float4* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cw * ch) return;
  const int a = i / ch, b = i - a * ch;
  const int texel = (b * step + step / 2) * cols + (a * step + step / 2);
  const float4 v = vertex[texel];
  out[i] = make_float4(v.x, v.y, v.z, (float)old_time[texel]);
}
__global__ void k_log_pose(const TrackState* st, double* traj, int slot) {
  if (threadIdx.x == 0) log_pose(st, traj, slot);
}

// ------------------------------------------------------------------------------------------
// SO(3) pre-alignment, RGBDOdometry.cpp:284-369.  One launch per iteration, SO3_WPB virtual warps per workgroup
// (level 2 is small: 160x120 = 1.2 passes of the 16384 virtual threads), reference-order sums as above; the LAST
// workgroup to finish runs the rest of the tree, the 3x3 float LDL^T, Rodrigues, the convergence / divergence tests
// and the next iteration's matrices, so an iteration is a single kernel with no host round trip.  Launches after
// convergence return immediately.
// ------------------------------------------------------------------------------------------
constexpr int SO3_WPB = 4, SO3_BLOCK = 256, SO3_KC = 8;
__device__ __forceinline__ void so3_chains(const float* __restrict__ R, int l, int kend, float (&acc)[SO3_ACCS]) {
  for (int k = 0; k < kend; ++k) {
    const float* r = R + k * ROW_STRIDE + l;
    const float r0 = r[0], r1 = r[32], r2 = r[64], r3 = r[96], f = r[128];
    acc[0] = EF_FMA(r0, r0, acc[0]); acc[1] = EF_FMA(r0, r1, acc[1]); acc[2] = EF_FMA(r0, r2, acc[2]); acc[3] = EF_FMA(r0, r3, acc[3]);
    acc[4] = EF_FMA(r1, r1, acc[4]); acc[5] = EF_FMA(r1, r2, acc[5]); acc[6] = EF_FMA(r1, r3, acc[6]);
    acc[7] = EF_FMA(r2, r2, acc[7]); acc[8] = EF_FMA(r2, r3, acc[8]);
    acc[9] = EF_FMA(r3, r3, acc[9]);
    acc[10] += f;
  }
}
// phase A + B of so3Step for the virtual warps of this workgroup; leaves partials[acc * VWARPS + warp]
template <int BLOCK = SO3_BLOCK>
__device__ __forceinline__ void so3_accumulate(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage, int cols,
                                               int rows, const m33& IB, const m33& KI, const m33& KR, float* lds_rows,
                                               float* __restrict__ partials) {
  const int t = threadIdx.x, N = cols * rows;
  const int K = (N + VTHREADS - 1) / VTHREADS;
  const int l = t & 31, w = t >> 5;   // phase-B identity (t < 32 * SO3_WPB)
  const int W = blockIdx.x * SO3_WPB + w;
  const int g = W * 32 + l;
  const int nk = g < N ? (N - g + VTHREADS - 1) / VTHREADS : 0;
  float acc[SO3_ACCS];
#pragma unroll
  for (int i = 0; i < SO3_ACCS; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += SO3_KC) {
    const int kc = min(SO3_KC, K - k0);
    for (int s = t; s < SO3_WPB * kc * 32; s += BLOCK) {
      const int sl = s & 31, q = s >> 5, k = q % kc, sw = q / kc;
      const int p = (k0 + k) * VTHREADS + (blockIdx.x * SO3_WPB + sw) * 32 + sl;
      float row[4] = {0.f, 0.f, 0.f, 0.f};
      float found = 0.f;
      if (p < N && so3_row(lastImage, nextImage, cols, rows, IB, KI, KR, p, row)) found = 1.f;
      float* r = lds_rows + (sw * SO3_KC + k) * ROW_STRIDE + sl;
      r[0] = row[0]; r[32] = row[1]; r[64] = row[2]; r[96] = row[3]; r[128] = found;
    }
    __syncthreads();
    if (t < 32 * SO3_WPB) so3_chains(lds_rows + w * SO3_KC * ROW_STRIDE, l, min(kc, nk - k0), acc);
    __syncthreads();
  }
  if (t < 32 * SO3_WPB) {
#pragma unroll
    for (int i = 0; i < SO3_ACCS; ++i)
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) acc[i] += __shfl_down(acc[i], off, 32);
    if (l == 0) {
#pragma unroll
      for (int i = 0; i < SO3_ACCS; ++i) coherent_store(partials + (size_t)i * VWARPS + W, acc[i]);
      drain_stores();
    }
  }
}

// so3_accumulate for the persistent kernel: workgroup 2 b + h owns virtual warps 8 b + {h, h + 4, h + 2, h + 6} (pt_vwarp), so that the
// first two levels of blockReduceSum's 8-warp tree — (x_h + x_{h+4}) + (x_{h+2} + x_{h+6}) — are added here, and ONE value per
// accumulator and workgroup is published: partials[(acc * 64 + b) * 2 + h], 5.6 KB for the whole grid instead of 22 KB.
template <int BLOCK>
__device__ __forceinline__ void so3_accumulate_halves(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage, int cols, int rows,
                                                      const m33& IB, const m33& KI, const m33& KR, float* lds_rows, float* wsum /* [4][SO3_ACCS] */,
                                                      float* __restrict__ partials) {
  const int t = threadIdx.x, N = cols * rows, wg = blockIdx.x;
  const int K = (N + VTHREADS - 1) / VTHREADS;
  const int l = t & 31, w = t >> 5;   // phase-B identity (t < 32 * SO3_WPB)
  const int W = (8 * (wg >> 1) + (wg & 1)) + 2 * ((w & 3) >> 1) + 4 * (w & 1);   // pt_vwarp(wg, w)
  const int g = W * 32 + l;
  const int nk = g < N ? (N - g + VTHREADS - 1) / VTHREADS : 0;
  float acc[SO3_ACCS];
#pragma unroll
  for (int i = 0; i < SO3_ACCS; ++i) acc[i] = 0.f;
  for (int k0 = 0; k0 < K; k0 += SO3_KC) {
    const int kc = min(SO3_KC, K - k0);
    for (int s = t; s < SO3_WPB * kc * 32; s += BLOCK) {
      const int sl = s & 31, q = s >> 5, k = q % kc, sw = q / kc;
      const int Ws = (8 * (wg >> 1) + (wg & 1)) + 2 * (sw >> 1) + 4 * (sw & 1);
      const int p = (k0 + k) * VTHREADS + Ws * 32 + sl;
      float row[4] = {0.f, 0.f, 0.f, 0.f};
      float found = 0.f;
      if (p < N && so3_row(lastImage, nextImage, cols, rows, IB, KI, KR, p, row)) found = 1.f;
      float* r = lds_rows + (sw * SO3_KC + k) * ROW_STRIDE + sl;
      r[0] = row[0]; r[32] = row[1]; r[64] = row[2]; r[96] = row[3]; r[128] = found;
    }
    __syncthreads();
    if (t < 32 * SO3_WPB) so3_chains(lds_rows + w * SO3_KC * ROW_STRIDE, l, min(kc, nk - k0), acc);
    __syncthreads();
  }
  if (t < 32 * SO3_WPB) {
#pragma unroll
    for (int i = 0; i < SO3_ACCS; ++i)
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) acc[i] += __shfl_down(acc[i], off, 32);
    if (l == 0) {
#pragma unroll
      for (int i = 0; i < SO3_ACCS; ++i) wsum[w * SO3_ACCS + i] = acc[i];
    }
  }
  __syncthreads();
  if (t < SO3_ACCS) {
    const float z = (wsum[t] + wsum[SO3_ACCS + t]) + (wsum[2 * SO3_ACCS + t] + wsum[3 * SO3_ACCS + t]);
    coherent_store(partials + ((size_t)t * 64 + (wg >> 1)) * 2 + (wg & 1), z);
    drain_stores();
  }
}
// ... and the rest of the tree over those: z_0 + z_1 per block (8-byte agent-scope loads), then reduceSum's 64-block tree as final_tree
template <int BLOCK, int NA>
__device__ __forceinline__ void final_tree_halves(const float* partials, float* bs, float* out) {
  const int t = threadIdx.x;
  for (int idx = t; idx < NA * 64; idx += BLOCK) {
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)partials + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bs[idx] = __uint_as_float((unsigned)v) + __uint_as_float((unsigned)(v >> 32));
  }
  __syncthreads();
  for (int idx = t; idx < NA * 64; idx += BLOCK) {
    float x = bs[idx];
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) x += __shfl_down(x, off, 32);
    const float w1 = __shfl(x, 32, 64);
    if ((idx & 63) == 0) out[idx >> 6] = x + w1;
  }
  __syncthreads();
}

__global__ void __launch_bounds__(SO3_BLOCK) k_so3_iteration(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage,
                                                              int cols, int rows, Intr k, Intr kfirst, int it, TrackState* st,
                                                              float* __restrict__ partials) {
  __shared__ float lds_rows[SO3_WPB * SO3_KC * ROW_STRIDE];
  __shared__ float red[SO3_ACCS];
  __shared__ int is_last;
  if (st->so3_done) return;
  const int t = threadIdx.x;
  // state the update step will need, fetched by thread 0 of EVERY workgroup while the rows are accumulated (which
  // workgroup arrives last is not known yet): a dependent global load costs ~1 us on the serial tail otherwise
  float p_lastError = 0.f, p_lastCount = 0.f, p_Rlr[9];
  double p_resR[9];
  if (t == 0) {
    p_lastError = st->so3_lastError;
    p_lastCount = st->so3_lastCount;
#pragma unroll
    for (int i = 0; i < 9; ++i) { p_Rlr[i] = st->so3_R_lr[i]; p_resR[i] = st->so3_resultR[i]; }
  }
  {
    const m33 IB = m33_load(st->so3_mats), KI = m33_load(st->so3_mats + 9), KR = m33_load(st->so3_mats + 18);
#ifdef EF_FAST_ORDER
    so3_accumulate_fast<true>(lastImage, nextImage, cols, rows, IB, KI, KR, lds_rows, partials);   // one group per workgroup; agent-scope stores, drained
#else
    so3_accumulate(lastImage, nextImage, cols, rows, IB, KI, KR, lds_rows, partials);
#endif
  }
  // last-workgroup-done: our partials are drained to the coherence point, take a ticket
  __syncthreads();
  if (t == 0) is_last = (take_ticket(&st->so3_ticket) == gridDim.x - 1);
  __syncthreads();
  if (!is_last) return;
#ifdef EF_FAST_ORDER
  fast_tree<SO3_BLOCK, true>(partials, fast_plan(cols * rows).NG, SO3_ACCS, red);
  __syncthreads();
#else
  final_tree<SO3_BLOCK, SO3_ACCS, true>(partials, lds_rows /* reused: 11*64 floats */, red);
#endif
  if (t != 0) return;
  st->so3_ticket = 0;
  st->so3_iterations = it + 1;
  float jtj[9], jtr[3];
  int shift = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 4; ++j) {
      const float value = red[shift++];
      if (j == 3) jtr[i] = value;
      else jtj[j * 3 + i] = jtj[i * 3 + j] = value;
    }
  float err = sqrtf(red[9]) / red[10];
  float cnt = red[10];
  bool done = false;
  double resR[9];   // so3_resultR as the rest of this step sees it
#pragma unroll
  for (int i = 0; i < 9; ++i) resR[i] = p_resR[i];
  if (err < p_lastError && p_lastCount == cnt) {
    done = true;
  } else if ((double)err > (double)p_lastError + 0.001) {
    err = p_lastError; cnt = p_lastCount;
    for (int i = 0; i < 9; ++i) { resR[i] = st->so3_lastResultR[i]; st->so3_resultR[i] = resR[i]; }
    done = true;
  } else {
    st->so3_lastError = err; st->so3_lastCount = cnt;
    for (int i = 0; i < 9; ++i) st->so3_lastResultR[i] = p_resR[i];
    float delta[3];
    efl::ldlt_solve<float, 3>(jtj, jtr, delta);
    const double dv[3] = {(double)delta[0], (double)delta[1], (double)delta[2]};
    double ru[9];
    efl::rodrigues(dv, ru);
    float ruf[9], nR[9];
    for (int i = 0; i < 9; ++i) ruf[i] = (float)ru[i];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        float s = 0;
        for (int kk = 0; kk < 3; ++kk) s += ruf[r * 3 + kk] * p_Rlr[kk * 3 + c];
        nR[r * 3 + c] = s;
      }
    for (int i = 0; i < 9; ++i) { st->so3_R_lr[i] = nR[i]; resR[i] = (double)nR[i]; st->so3_resultR[i] = resR[i]; }
  }
  st->lastSO3Error = err;
  st->lastSO3Count = cnt;
  if (done || it == 9) {
    // resultRt.topLeftCorner(3,3) = resultR (RGBDOdometry.cpp:381-388) and the first level's K R K^-1, K t
    // the rest of resultRt is the identity k_track_begin wrote (nothing touches it before the first SE(3) iteration)
    double Rt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int x = 0; x < 3; ++x)
      for (int y = 0; y < 3; ++y) { Rt[x * 4 + y] = resR[x * 3 + y]; st->gn[0].resultRt[x * 4 + y] = resR[x * 3 + y]; }
    compute_krk(Rt, kfirst, st->gn[0].krkinv, st->gn[0].kt);
    st->so3_done = 1;
  } else {
    so3_matrices(resR, k, st->so3_mats);
  }
}

// ---- operator-tier single-shot reductions ----
__global__ void __launch_bounds__(SO3_BLOCK) k_so3_op(const uint8_t* __restrict__ lastImage, const uint8_t* __restrict__ nextImage,
                                                       int cols, int rows, const So3Args a, float* __restrict__ partials) {
  __shared__ float lds_rows[SO3_WPB * SO3_KC * ROW_STRIDE];
  const m33 IB = m33_load(a.imageBasis), KI = m33_load(a.kinv), KR = m33_load(a.krlr);
#ifdef EF_FAST_ORDER
  so3_accumulate_fast<false>(lastImage, nextImage, cols, rows, IB, KI, KR, lds_rows, partials);
#else
  so3_accumulate(lastImage, nextImage, cols, rows, IB, KI, KR, lds_rows, partials);
#endif
}
__global__ void __launch_bounds__(SOLVE_BLOCK) k_final_tree_op(const float* __restrict__ partials, int na, int ng, float* __restrict__ out) {
  __shared__ float bs[SE3_ACCS * 64];
  __shared__ float sums[2 * SE3_ACCS];
#ifdef EF_FAST_ORDER
  fast_tree<SOLVE_BLOCK, false>(partials, ng, na, sums);
  __syncthreads();
  if ((int)threadIdx.x < na) out[threadIdx.x] = sums[threadIdx.x];
  return;
#endif
  if (na == SE3_ACCS) {   // k_se3_accum's pair partials (one term)
    pair_partials_tree<SOLVE_BLOCK>(partials, true, false, sums);
    __syncthreads();
  } else {
    final_tree<SOLVE_BLOCK, SO3_ACCS, false>(partials, bs, sums);
  }
  if ((int)threadIdx.x < na) out[threadIdx.x] = sums[threadIdx.x];
}


// ------------------------------------------------------------------------------------------
// The SMALL pyramid levels in ONE launch (round 3): k_track_begin, the <= 10 SO(3) iterations and every Gauss-Newton iteration of the
// levels with at most PT_MAX_PIXELS pixels (levels 2 and 1 at 640x480: 4 + 5 iterations) — 1 + 10 + 27 dependent launches of the
// launch-per-step script — as one persistent kernel of PT_WGS co-resident workgroups.
//
//   * Every workgroup keeps its OWN copy of the Gauss-Newton state in LDS and evaluates every update step redundantly (the same
//     wave-parallel solve, ef_solve_dev.hpp): nothing is broadcast, the only traffic between workgroups is what the reference
//     reduces globally — the pair partials of the normal equations (all-gathered: 59 KB, read by every workgroup) and the two
//     integers of the correspondence search.
//   * Hand-over = agent-scope (write-through) stores, drained, then ONE counter barrier; readers use agent-scope loads, which no
//     CU's L1 serves.  Placement-independent: nothing assumes which XCD a workgroup runs on (HIP promises no mapping).  Payloads
//     alternate between two regions by iteration parity, so a workgroup that runs ahead never overwrites what a slow one still reads.
//   * Two barriers per SE(3) iteration: A after the correspondence search (the photometric rows need the global count: sigma,
//     quirk Q2) — its latency is hidden behind the ICP wavefronts, which do not depend on it — and B after the accumulation.
//     One barrier per SO(3) iteration; the loop leaves at convergence (no post-convergence no-op launches).
//   * Summation order, arithmetic and device functions are those of the per-step kernels (accum_quads, pair_partials_tree,
//     solve_step_wave, so3_accumulate, final_tree), so every result is bit-identical to the launch-per-step script
//     (tests/test_gpu_frame.py, test_gpu_steady.py run both).
//   * Every spin is bounded (PT_SPIN polls); a barrier that times out (the grid was not co-resident: only possible when other work
//     occupies the chip's wave slots for ever) raises PtSync::abort and the launch runs to its end without waiting any more; later
//     launches on that tracker instance return at once and ef_synchronize reports EF_EHIP.
// Work split: workgroup w = 2 b + h owns pairs h and h + 2 of reference block b, i.e. virtual warps 8 b + {h, h + 4, h + 2, h + 6}
// (SE(3)), and virtual warps 4 w .. 4 w + 3 (SO(3), as k_so3_iteration).  8 wavefronts = 2 pair tasks x {ICP, RGB} x 2 virtual
// warps, each wavefront working through BOTH halves of its warp (accum_quads_halves; two wavefronts per SIMD leave each 256 registers:
// with 16 wavefronts the update step spilled); the correspondence search runs two pixels per thread over exactly the pixels the
// workgroup's own RGB wavefronts visit afterwards, so the packed correspondences never cross a workgroup.
// ------------------------------------------------------------------------------------------
constexpr int PT_SPIN = 1 << 20;
struct PtSync {                   // behind the partial regions (Pyramid::partials + 2 * PARTIAL_FLOATS), zero-filled at allocation
  unsigned count, pad0[31];       // arrivals of the barrier in flight (its atomics do not share a cache line with the pollers' word)
  unsigned gen, pad1[31];         // generations completed, monotonic across launches
  unsigned abort, pad2[31];       // sticky: a wait timed out
  int wg_sums[2][PT_WGS][2];      // {count, sum diff^2} of each workgroup's share of a correspondence search, by iteration parity
  unsigned long long clk[24];     // developer instrumentation (-DEF_STAGE_CLOCKS builds): 10 ns ticks per phase, summed over launches
};
// -DEF_STAGE_CLOCKS: workgroup 0 adds the time since its last stamp to PtSync::clk[i] (thread 0: the phases of an ICP wavefront; PT_CLK2:
// lane 0 of the polling RGB wavefront).  tools/small_clocks.py reads them through ef_debug_small_clocks.
#ifdef EF_STAGE_CLOCKS
#define PT_CLK(i) do { if (wg == 0 && t == 0) { const unsigned long long now_ = wall_clock64(); Y->clk[i] += now_ - clk_last; clk_last = now_; } } while (0)
#define PT_CLK2(i) do { if (wg == 0 && lane == 0) { const unsigned long long now_ = wall_clock64(); Y->clk[i] += now_ - clk_last2; clk_last2 = now_; } } while (0)
#define PT_COUNT(i) do { if (wg == 0 && t == 0) Y->clk[i] += 1; } while (0)
#else
#define PT_CLK(i) do { } while (0)
#define PT_CLK2(i) do { } while (0)
#define PT_COUNT(i) do { } while (0)
#endif
static_assert(sizeof(PtSync) <= PT_SYNC_FLOATS * sizeof(float), "PtSync fits its reservation");
struct PtLevel {
  const float* vmap_curr; const float* nmap_curr; const float* vmap_g_prev; const float* nmap_g_prev;
  const uint8_t* mask; const float* lastDepth; const float* nextDepth; const uint8_t* lastImage; const uint8_t* nextImage;
  uint32_t* corres; const int16_t* dIdx; const int16_t* dIdy;
  int cols, rows;
  Intr k;
};
struct PtArgs {
  PtLevel L[NUM_PYRS];
  int n_iter;
  unsigned levels;                // level of iteration i in bits 2 i, 2 i + 1 (a kernel argument indexed at run time would live in scratch)
  bool so3;
  const uint8_t* so3_last; const uint8_t* so3_next;
  int so3_cols, so3_rows;
  Intr kso3, kfirst;
  float icpWeight, distThres, angleThres;
  float dist2Max, sine2Max;       // IcpView's gates on squared norms
  float* partials;                // two regions of PARTIAL_FLOATS + the PtSync
  int out_cur;                    // TrackState::gn buffer the launches that follow read
};
struct So3Loop {                  // k_so3_iteration's state, per workgroup in LDS
  double resR[9], lastResR[9];
  float R_lr[9], mats[27];
  float lastError, lastCount, err, cnt;
  int done, iterations;
};
__device__ __forceinline__ unsigned pt_load(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int pt_loadi(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// one thread per workgroup, after the workgroup's agent-scope stores were drained and a __syncthreads()
__device__ __forceinline__ void pt_arrive(PtSync* Y) {
  __atomic_signal_fence(__ATOMIC_RELEASE);   // (compiler only: the payload stores were drained with s_waitcnt vmcnt(0) before this call)
  const unsigned old = __hip_atomic_fetch_add(&Y->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old == PT_WGS - 1) {   // last arriver: re-arm the counter, then open the generation
    __hip_atomic_store(&Y->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    drain_stores();
    __hip_atomic_fetch_add(&Y->gen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// one lane; returns true when the wait was abandoned (time-out here or in another workgroup)
__device__ __forceinline__ bool pt_wait(PtSync* Y, unsigned target, bool dead) {
  if (dead) return true;
  for (int i = 0; i < PT_SPIN; ++i) {
    // (relaxed poll + agent-scope payload loads behind it, which no L1 serves: an ACQUIRE load here costs a cache invalidate per poll,
    // MI355X_MICROARCH.md "polling with acquire loads: 2-3x slower per hop"; the signal fence keeps the COMPILER from moving a payload
    // load above the poll, the hardware returns a wavefront's loads in order)
    if ((int)(pt_load(&Y->gen) - target) >= 0) { __atomic_signal_fence(__ATOMIC_ACQUIRE); return false; }
    if ((i & 255) == 255 && pt_load(&Y->abort)) return true;
    __builtin_amdgcn_s_sleep(1);
  }
  __hip_atomic_store(&Y->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
__device__ __forceinline__ int pt_level_of(const PtArgs& A, int it) { return (int)((A.levels >> (2 * it)) & 3u); }
// The level descriptors are read from the kernel-argument segment AS MEMORY (scalar loads at a run-time offset), one level at a time:
// taken from the by-value argument they would all have to sit in SGPRs at once (3 x 30: spilled), or, indexed at run time, be copied
// into a private array (scratch).
typedef const __attribute__((address_space(4))) PtArgs* PtArgsK;
__device__ __forceinline__ PtLevel pt_load_level(PtArgsK Ak, int lvl) {
  PtLevel L;
  L.vmap_curr = Ak->L[lvl].vmap_curr; L.nmap_curr = Ak->L[lvl].nmap_curr;
  L.vmap_g_prev = Ak->L[lvl].vmap_g_prev; L.nmap_g_prev = Ak->L[lvl].nmap_g_prev;
  L.mask = Ak->L[lvl].mask; L.lastDepth = Ak->L[lvl].lastDepth; L.nextDepth = Ak->L[lvl].nextDepth;
  L.lastImage = Ak->L[lvl].lastImage; L.nextImage = Ak->L[lvl].nextImage;
  L.corres = Ak->L[lvl].corres; L.dIdx = Ak->L[lvl].dIdx; L.dIdy = Ak->L[lvl].dIdy;
  L.cols = Ak->L[lvl].cols; L.rows = Ak->L[lvl].rows;
  L.k.fx = Ak->L[lvl].k.fx; L.k.fy = Ak->L[lvl].k.fy; L.k.cx = Ak->L[lvl].k.cx; L.k.cy = Ak->L[lvl].k.cy;
  return L;
}
// virtual warp of slot 0..3 of workgroup wg: slot = 2 * task + wl; task 0 / 1 = pairs h / h + 2 of reference block b, wl = upper warp of the pair
__device__ __forceinline__ int pt_vwarp(int wg, int slot) {
  const int b = wg >> 1, h = wg & 1;
  return 8 * b + h + 2 * (slot >> 1) + 4 * (slot & 1);
}
// the update of one SO(3) iteration (k_so3_iteration's tail) on the workgroup's own state; returns true when the loop is over
__device__ __forceinline__ bool so3_update(So3Loop& Z, const float* red, int it, Intr k, Intr kfirst, GNState& g0) {
  float jtj[9], jtr[3];
  int shift = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = i; j < 4; ++j) {
      const float value = red[shift++];
      if (j == 3) jtr[i] = value;
      else jtj[j * 3 + i] = jtj[i * 3 + j] = value;
    }
  float err = sqrtf(red[9]) / red[10];
  float cnt = red[10];
  bool done = false;
  double resR[9];
  for (int i = 0; i < 9; ++i) resR[i] = Z.resR[i];
  if (err < Z.lastError && Z.lastCount == cnt) {
    done = true;
  } else if ((double)err > (double)Z.lastError + 0.001) {
    err = Z.lastError; cnt = Z.lastCount;
    for (int i = 0; i < 9; ++i) { resR[i] = Z.lastResR[i]; Z.resR[i] = resR[i]; }
    done = true;
  } else {
    Z.lastError = err; Z.lastCount = cnt;
    for (int i = 0; i < 9; ++i) Z.lastResR[i] = resR[i];
    float delta[3];
    efl::ldlt_solve<float, 3>(jtj, jtr, delta);
    const double dv[3] = {(double)delta[0], (double)delta[1], (double)delta[2]};
    double ru[9];
    efl::rodrigues(dv, ru);
    float ruf[9], nR[9], Rlr[9];
    for (int i = 0; i < 9; ++i) { ruf[i] = (float)ru[i]; Rlr[i] = Z.R_lr[i]; }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        float s = 0;
        for (int kk = 0; kk < 3; ++kk) s += ruf[r * 3 + kk] * Rlr[kk * 3 + c];
        nR[r * 3 + c] = s;
      }
    for (int i = 0; i < 9; ++i) { Z.R_lr[i] = nR[i]; resR[i] = (double)nR[i]; Z.resR[i] = resR[i]; }
  }
  Z.err = err;
  Z.cnt = cnt;
  Z.iterations = it + 1;
  if (done || it == 9) {
    double Rt[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int x = 0; x < 3; ++x)
      for (int y = 0; y < 3; ++y) { Rt[x * 4 + y] = resR[x * 3 + y]; g0.resultRt[x * 4 + y] = resR[x * 3 + y]; }
    compute_krk(Rt, kfirst, g0.krkinv, g0.kt);
    return true;
  }
  so3_matrices(resR, k, Z.mats);
  return false;
}

template <bool HAS_ICP, bool HAS_RGB>
__global__ void __launch_bounds__(PT_BLOCK) k_track_small(const PtArgs A, TrackState* st) {
  constexpr int NT = (HAS_ICP && HAS_RGB) ? 2 : 1;
  static_assert(PT_BLOCK == 64 * 2 * 2 * ACC_NW && PT_MAX_PIXELS == 2 * PT_BLOCK * PT_WGS && PT_WGS * 128 == VTHREADS,
                "8 wavefronts: 2 tasks x 2 terms x 2 warps; 128 virtual threads per workgroup; <= 2 pixels per thread in the search");
  __shared__ GNState gs[2];
  __shared__ efs::SolveScratch S;
  __shared__ float sums_s[2 * SE3_ACCS];
  __shared__ float bg[24];   // Rprev[9] | tprev[3] | Rprev_inv[9]: constants of the call (k_track_begin)
  __shared__ float lds_rows[SO3_WPB * SO3_KC * ROW_STRIDE];
  __shared__ float pairx[2 * 2 * 12 * 4];   // [task][term][12][4]
  __shared__ float tsum[2 * 12 * 4];        // [term][12][4]: task 1's pair sums on their way to task 0's storing lanes
  __shared__ float wsum[SO3_WPB * SO3_ACCS];
  __shared__ float red[SO3_ACCS];
  __shared__ So3Loop Z;
  __shared__ int ired[2 * PT_BLOCK / 64];
  __shared__ unsigned sync_s[2];
  __shared__ int flag_s;
  __shared__ unsigned seen_a[2];   // barrier A relay: {generation the workgroup's polling wavefront has seen complete, wait abandoned}
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wg = blockIdx.x;
  PtSync* Y = (PtSync*)(A.partials + 2 * PARTIAL_FLOATS);
#ifdef EF_STAGE_CLOCKS
  unsigned long long clk_last = wall_clock64(), clk_last2 = clk_last;
  const unsigned long long clk_start = clk_last;
#endif
  // ---- k_track_begin, by every workgroup for itself; workgroup 0 also leaves the global side of it ----
  if (t == 0) {
    sync_s[0] = pt_load(&Y->gen) + 1u;   // the generation that ends this launch's first barrier
    sync_s[1] = pt_load(&Y->abort);
    seen_a[0] = sync_s[0] - 1u;
    seen_a[1] = 0u;
    double R[9];
    efl::quat_to_mat<double>(st->q, R);
    GNState& g = gs[0];
    for (int i = 0; i < 9; ++i) bg[i] = g.Rcurr[i] = (float)R[i];
    for (int i = 0; i < 3; ++i) bg[9 + i] = g.tcurr[i] = (float)st->t[i];
    efl::m3_inverse<float>(bg, bg + 12);
    efl::m4_identity(g.resultRt);
    g.rgb_broken = 0;
    g.lastRGBErrorLevel = 3.402823466e+38f;
    if (!A.so3) compute_krk(g.resultRt, A.kfirst, g.krkinv, g.kt);
    if (wg == 0) {
      for (int i = 0; i < 9; ++i) { st->Rprev[i] = bg[i]; st->Rprev_inv[i] = bg[12 + i]; }
      for (int i = 0; i < 3; ++i) st->tprev[i] = bg[9 + i];
      for (int i = 0; i < 4; ++i) st->q_prev[i] = st->q[i];
      for (int i = 0; i < 3; ++i) st->t_prev[i] = st->t[i];
      for (int i = 0; i < RGB_SLOTS; ++i) st->rgb_slots[0][i][0] = st->rgb_slots[0][i][1] = st->rgb_slots[1][i][0] = st->rgb_slots[1][i][1] = 0;
      st->call_seq += 1u;
      st->so3_iterations = 0;
      st->so3_ticket = 0;
      st->so3_done = 1;
      st->dense_count = 0;
    }
  }
  __syncthreads();
  PT_CLK(0);   // begin
  if (sync_s[1]) return;   // an earlier launch of this tracker instance timed out in a barrier
  unsigned gen_next = sync_s[0];
  bool dead = false;
  // ---- SO(3) pre-alignment, RGBDOdometry.cpp:284-369 ----
  if (A.so3) {
    if (t == 0) {
      efl::m3_identity(Z.resR);
      efl::m3_identity(Z.lastResR);
      for (int i = 0; i < 9; ++i) Z.R_lr[i] = (i % 4 == 0) ? 1.f : 0.f;
      Z.lastError = 3.402823466e+38f / 2;
      Z.lastCount = 3.402823466e+38f / 2;
      Z.done = 0;
      so3_matrices(Z.resR, A.kso3, Z.mats);
    }
    __syncthreads();
    for (int it = 0; it < 10; ++it) {
      float* region = A.partials + (size_t)(it & 1) * PARTIAL_FLOATS;
      {
        const m33 IB = m33_load(Z.mats), KI = m33_load(Z.mats + 9), KR = m33_load(Z.mats + 18);
        so3_accumulate_halves<PT_BLOCK>(A.so3_last, A.so3_next, A.so3_cols, A.so3_rows, IB, KI, KR, lds_rows, wsum, region);   // agent-scope stores, drained
      }
      __syncthreads();
      PT_CLK(1);   // SO(3): rows, chains, publish
      if (t == 0) {
        pt_arrive(Y);
        flag_s = pt_wait(Y, gen_next, dead) ? 1 : 0;
      }
      __syncthreads();
      PT_CLK(2);   // SO(3): barrier
      dead = flag_s != 0;
      ++gen_next;
      final_tree_halves<PT_BLOCK, SO3_ACCS>(region, lds_rows, red);
      PT_CLK(3);   // SO(3): gather + tree
      if (t == 0) Z.done = so3_update(Z, red, it, A.kso3, A.kfirst, gs[0]) ? 1 : 0;
      __syncthreads();
      PT_CLK(4);   // SO(3): update
      PT_COUNT(20);
      if (Z.done) break;   // every workgroup computes the same bits, so every workgroup leaves in the same iteration
    }
    if (wg == 0 && t == 0) {
      st->lastSO3Error = Z.err;
      st->lastSO3Count = Z.cnt;
      st->so3_iterations = Z.iterations;
    }
    if (!HAS_RGB) {
      // ICP only (icpWeight >= 100) after an SO(3) loop: no barrier A separates the last SO(3) iteration's gather from the first SE(3)
      // iteration's publish, and the two use overlapping floats of the same region when the parities match: a fast workgroup could
      // overwrite sums a slow one is still reading.  One more barrier closes the window.
      if (t == 0) {
        pt_arrive(Y);
        flag_s = pt_wait(Y, gen_next, dead) ? 1 : 0;
      }
      __syncthreads();
      dead = dead || flag_s != 0;
      ++gen_next;
    }
  }
  // ---- the Gauss-Newton iterations of the small levels, RGBDOdometry.cpp:371-553 ----
  // Two barriers per iteration: A after the correspondence search (its latency hidden behind the ICP wavefronts), B after the
  // accumulation.  (Round 3 also built the exchange on tagged 8-byte granules without any barrier — every reader polling exactly the
  // granules it needs: bit-identical, and 2.5 % SLOWER end to end, profiles/r03d_ab_granules_vs_barriers.log: 7424 granule loads per
  // workgroup cost more than one barrier + 3712 plain 8-byte loads.  Dropped.)
  int cur = 0;
  for (int it = 0; it < A.n_iter; ++it) {
    const int lvl = pt_level_of(A, it);
    const PtLevel Lv = pt_load_level((PtArgsK)__builtin_amdgcn_kernarg_segment_ptr(), lvl);   // PtArgs is the kernel's FIRST argument
    const int cols = Lv.cols, rows = Lv.rows, N = cols * rows;
    const int K = (N + VTHREADS - 1) / VTHREADS;
    const bool last = it == A.n_iter - 1;
    float* region = A.partials + (size_t)((A.n_iter - 1 - it) & 1) * PARTIAL_FLOATS;   // the last iteration's partials land in region 0
    if (it > 0) {
      // head: the update step of iteration it - 1 (k_track_step's head), by every workgroup on its own state
      StepArgs H{true, false, HAS_ICP, HAS_RGB, false, A.icpWeight, Lv.k, lvl != pt_level_of(A, it - 1), it};
      efs::SolvePrefetch PF{};
      if (t < 64) {
        const GNState& g = gs[cur];
        PF.rt = g.resultRt[lane & 15];
        PF.pose = bg[lane < 12 ? lane : 0];
        PF.slot_a = PF.slot_b = 0;
        if (HAS_RGB) {
          const int (*ws)[2] = Y->wg_sums[(it - 1) & 1];
          PF.slot_a = pt_loadi(&ws[lane][0]) + pt_loadi(&ws[lane + 64][0]);
          PF.slot_b = pt_loadi(&ws[lane][1]) + pt_loadi(&ws[lane + 64][1]);
        }
        PF.lastRGBErrorLevel = g.lastRGBErrorLevel;
        PF.broken = g.rgb_broken;
      }
      half_partials_tree<PT_BLOCK>(A.partials + (size_t)((A.n_iter - it) & 1) * PARTIAL_FLOATS, HAS_ICP, HAS_RGB, sums_s);
      __syncthreads();
      PT_CLK(5);   // SE(3) head: gather + trees
      if (t < 64) solve_step_wave(st, &gs[cur], &gs[cur ^ 1], true, sums_s, H, S, PF, wg == 0);
      __syncthreads();
      PT_CLK(6);   // SE(3) head: solve
      cur ^= 1;
    }
    const GNState& G = gs[cur];
    // correspondence search (k_track_step's body, two pixels per thread, their loads in flight together) over exactly the pixels this
    // workgroup's RGB wavefronts visit
    if (HAS_RGB) {
      const m33 Km = m33_load(G.krkinv);
      const f3 kt{G.kt[0], G.kt[1], G.kt[2]};
      int qs[2];
      uint8_t m[2];
      float d1s[2];
      int nis[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int idx = t + e * PT_BLOCK, kpass = idx >> 7, vt = idx & 127;
        const int q = kpass * VTHREADS + pt_vwarp(wg, vt >> 5) * 32 + (vt & 31);
        const bool in = kpass < K && q < N;
        const int qq = in ? q : N - 1;
        qs[e] = q;
        m[e] = in ? Lv.mask[qq] : (uint8_t)0;
        d1s[e] = Lv.nextDepth[qq];
        nis[e] = Lv.nextImage[qq];
      }
      int gi[2], u0s[2], v0s[2], lis[2];
      float td1[2], d0s[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int q = qs[e];
        const int y = q / cols, x = q - y * cols;
        const float d1 = d1s[e];
        td1[e] = (float)(d1 * (Km.r[2].x * x + Km.r[2].y * y + Km.r[2].z) + kt.z);
        u0s[e] = f2i_rn((d1 * (Km.r[0].x * x + Km.r[0].y * y + Km.r[0].z) + kt.x) / td1[e]);
        v0s[e] = f2i_rn((d1 * (Km.r[1].x * x + Km.r[1].y * y + Km.r[1].z) + kt.y) / td1[e]);
        gi[e] = (m[e] && u0s[e] >= 0 && v0s[e] >= 0 && u0s[e] < cols && v0s[e] < rows) ? v0s[e] * cols + u0s[e] : -1;
        d0s[e] = 0.f;
        lis[e] = 0;
        if (gi[e] >= 0) { d0s[e] = Lv.lastDepth[gi[e]]; lis[e] = Lv.lastImage[gi[e]]; }
      }
      int cnt = 0, sq = 0;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (!m[e]) continue;
        uint32_t packed = 0u;
        if (gi[e] >= 0 && d0s[e] > 0 && fabsf(td1[e] - d0s[e]) <= 0.07f /* maxDepthDeltaRGB, RGBDOdometry.cpp:41 */ && lis[e] != 0) {
          const int idiff = nis[e] - lis[e];
          packed = pack_corres(u0s[e], v0s[e], idiff);
          cnt += 1;
          sq += idiff * idiff;
        }
        Lv.corres[qs[e]] = packed;
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off, 64);
        sq += __shfl_down(sq, off, 64);
      }
      if (lane == 0) { ired[wave * 2] = cnt; ired[wave * 2 + 1] = sq; }
      __syncthreads();   // also: the workgroup's packed correspondences are visible to its RGB wavefronts
      if (t == 0) {
        int sa = 0, sb = 0;
        for (int w = 0; w < PT_BLOCK / 64; ++w) { sa += ired[w * 2]; sb += ired[w * 2 + 1]; }
        int (*ws)[2] = Y->wg_sums[it & 1];
        __hip_atomic_store(&ws[wg][0], sa, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ws[wg][1], sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        drain_stores();
        pt_arrive(Y);   // barrier A: arrive now, the RGB wavefronts wait below
      }
      PT_CLK(7);   // SE(3): correspondence search + arrive A
    }
    // normal equations of the workgroup's two pair tasks: one wavefront per (task, term, warp of the pair), both halves of the warp
    const int task = wave / (2 * NT), w4 = wave % (2 * NT);
    const bool active = task < 2;
    const int tix = w4 >> 1, wl = w4 & 1;
    const bool rgb_wave = HAS_RGB && (!HAS_ICP || tix == 1);
    const int v = lane >> 2, j = lane & 3;
    f32x4 c[2][3];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int q = 0; q < 3; ++q) c[hf][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (active) {
      const int wbase = pt_vwarp(wg, 2 * task + wl) * 32;
      const IcpView IV{Lv.vmap_curr, Lv.nmap_curr, Lv.vmap_g_prev, Lv.nmap_g_prev, cols, rows, Lv.k, A.distThres, A.angleThres, A.dist2Max, A.sine2Max};
      const RgbView RV{Lv.corres, Lv.lastDepth, nullptr, Lv.dIdx, Lv.dIdy, cols, rows, Lv.k, 1.0f / 8.0f};
      Se3Inputs in{G.Rcurr, G.tcurr, bg + 12, bg + 9, nullptr, nullptr, 0.f, false};
      if (HAS_ICP && !rgb_wave) accum_quads_halves<true>(IV, RV, in, wbase, N, K, 0, 0, false, c);
      if (HAS_ICP && !rgb_wave) PT_CLK(8);   // SE(3): ICP accumulation (wavefront 0)
      if (HAS_RGB && rgb_wave) {
#ifdef EF_STAGE_CLOCKS
        if (task == 0 && wl == 0 && lane == 0 && wg == 0) clk_last2 = wall_clock64();
#endif
        // barrier A (the search's global {count, sum diff^2}: sigma, quirk Q2): ONE wavefront of the workgroup polls the global word, the
        // other RGB wavefronts watch its relay in LDS; all of them wait with their loads and gathers already issued (the hook)
        const bool leader = task == 0 && wl == 0;
        bool gone = dead;
        const int (*ws)[2] = Y->wg_sums[it & 1];
        auto wait_totals = [&](int& slot_a, int& slot_b) {
          if (lane == 0) {
            if (leader) {
              gone = pt_wait(Y, gen_next, gone);
              if (gone) __hip_atomic_store(&seen_a[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              __hip_atomic_store(&seen_a[0], gen_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
              int spins = 0;
              while ((int)(__hip_atomic_load(&seen_a[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - gen_next) < 0 && ++spins < 8 * PT_SPIN)
                __builtin_amdgcn_s_sleep(1);
              gone = gone || spins >= 8 * PT_SPIN || __hip_atomic_load(&seen_a[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;
            }
          }
          gone = __shfl((int)gone, 0, 64) != 0;
          if (leader) PT_CLK2(12);   // SE(3): wait for barrier A (polling RGB wavefront; its own loads are in flight)
          slot_a = pt_loadi(&ws[lane][0]) + pt_loadi(&ws[lane + 64][0]);
          slot_b = pt_loadi(&ws[lane][1]) + pt_loadi(&ws[lane + 64][1]);
        };
        accum_quads_halves<false>(IV, RV, in, wbase, N, K, 0, 0, true, c, wait_totals);
        dead = dead || gone;
        if (leader) PT_CLK2(13);   // SE(3): RGB rows + outer products after A
      }
    }
    if (HAS_RGB) ++gen_next;
    // warpReduceSum (offset 16 = the other half, here in the same wavefront) + the first level of blockReduceSum's 8-warp tree
    float r[12];
    float* pw = pairx + (size_t)((active ? task : 0) * 2 + tix) * 12 * 4;
    if (active) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        r[q * 4 + 0] = c[0][q].x + c[1][q].x; r[q * 4 + 1] = c[0][q].y + c[1][q].y;
        r[q * 4 + 2] = c[0][q].z + c[1][q].z; r[q * 4 + 3] = c[0][q].w + c[1][q].w;
      }
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        r[i] += down32(r[i]);
        r[i] += down16(r[i]);
        r[i] += row_down<8>(r[i]);
        r[i] += row_down<4>(r[i]);
      }
      if (wl == 1 && v == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) pw[i * 4 + j] = r[i];
      }
    }
    __syncthreads();
    // second level of the 8-warp tree inside the workgroup: pair h (task 0) + pair h + 2 (task 1): s_h = p_h + p_{h+2}, so that a reader's
    // s_0 + s_1 is blockReduceSum's (p0 + p2) + (p1 + p3)
    if (active && wl == 0 && v == 0) {
#pragma unroll
      for (int i = 0; i < 12; ++i) r[i] = r[i] + pw[i * 4 + j];
      if (task == 1) {
#pragma unroll
        for (int i = 0; i < 12; ++i) tsum[(tix * 12 + i) * 4 + j] = r[i];
      }
    }
    __syncthreads();
    if (active && task == 0 && wl == 0 && v == 0) {
      // slot h of the block receives s_h; the launch's LAST iteration, whose partials a per-step kernel reads as four pair slots, also
      // zeroes slot h + 2 ((s + 0) + (s' + 0) = s + s' to the bit)
      float* dst = region + (size_t)tix * SE3_ACCS * SE3_PAIRS + 4 * (wg >> 1) + (wg & 1);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int a = quad_member(q, i, j);
          if (a >= 0) {
            coherent_store(dst + (size_t)a * SE3_PAIRS, r[q * 4 + i] + tsum[(tix * 12 + q * 4 + i) * 4 + j]);
            if (last) coherent_store(dst + (size_t)a * SE3_PAIRS + 2, 0.f);
          }
        }
      drain_stores();
    }
    __syncthreads();
    PT_CLK(9);   // SE(3): join, trees, publish
    if (t == 0) {   // barrier B: every pair partial of this iteration is out
      pt_arrive(Y);
      flag_s = pt_wait(Y, gen_next, dead) ? 1 : 0;
    }
    __syncthreads();
    PT_CLK(10);   // SE(3): barrier B
    PT_COUNT(21);
    dead = dead || flag_s != 0;
    ++gen_next;
  }
  // ---- what the launches that follow read: the Gauss-Newton state and the last search's sums (update of iteration n_iter - 1
  //      at the head of the next k_track_step / k_track_end) ----
  if (wg == 0 && t == 0) {
    const GNState& g = gs[cur];
    GNState& o = st->gn[A.out_cur];
    for (int i = 0; i < 9; ++i) { o.Rcurr[i] = g.Rcurr[i]; o.krkinv[i] = g.krkinv[i]; }
    for (int i = 0; i < 3; ++i) { o.tcurr[i] = g.tcurr[i]; o.kt[i] = g.kt[i]; }
    for (int i = 0; i < 16; ++i) o.resultRt[i] = g.resultRt[i];
    o.lastRGBErrorLevel = g.lastRGBErrorLevel;
    o.rgb_broken = g.rgb_broken;
    if (HAS_RGB && A.n_iter > 0) {   // the last search's totals, where the next update step's head looks for them
      const int set = (A.n_iter - 1) & 1;
      int sa = 0, sb = 0;
      for (int w = 0; w < PT_WGS; ++w) { sa += pt_loadi(&Y->wg_sums[set][w][0]); sb += pt_loadi(&Y->wg_sums[set][w][1]); }
      st->rgb_slots[set][0][0] = sa;
      st->rgb_slots[set][0][1] = sb;
    }
#ifdef EF_STAGE_CLOCKS
    Y->clk[11] += wall_clock64() - clk_start;   // whole launch
    Y->clk[22] += 1;
#endif
  }
}

// ------------------------------------------------------------------------------------------
// The two INDEPENDENT launches at the head of a tracked frame as one (round 6): the depth pre-processing of the new frame (k_preprocess:
// LDS look-ups, 2.3 wavefronts per SIMD, hardly any HBM traffic) and the model-side maps of the tracker (k_model_maps: 20 MB streamed at
// 2.6 TB/s) read nothing of each other's — one grid carries the tiles of both, the second kind filling the issue slots and the memory pipe the
// first leaves idle, and a launch boundary goes.  Same device functions as the two kernels: same results.
// ------------------------------------------------------------------------------------------
namespace pre {
#include "ef_preprocess.inc"
}
struct PreArgs {
  const uint16_t* raw;
  int cols, rows;
  unsigned maxv;
  const float* table;
  uint16_t* filtered;
  float* metric;
  float* metric_filtered;
  const uint8_t* rgb3;
  uint8_t* next0;
  uint8_t* rgb_keep;
  int tiles_x, tiles;   // the filter's tiles: the first `tiles` workgroups of the grid, tiles_x per row
  int mm_x;             // the model maps' workgroups behind them: mm_x per row (k_model_maps' own grid, row-major)
};
__global__ void __launch_bounds__(256) k_frame_inputs(const PreArgs P, const ModelMapsArgs A, const TrackState* __restrict__ st) {
  // (the model maps' workgroups dispatched IN FRONT of the filter's tiles instead of behind them: measured, 2077 against 2091 frames/s — profiles/r08p_*)
  const int b = (int)blockIdx.x;
  if (b < P.tiles) {   // (uniform per workgroup)
    pre::preprocess_tile<true>(b % P.tiles_x, b / P.tiles_x, P.raw, P.cols, P.rows, P.maxv, P.table, P.filtered, P.metric, P.metric_filtered, P.rgb3,
                               P.next0, P.rgb_keep);
  } else {
    const int m = b - P.tiles, mx = m % P.mm_x, my = m / P.mm_x;
    const bool fill = model_maps_use_fill(A, st, m == 0);
    model_maps_lane(A, st, mx * 64 + (int)threadIdx.x, my * 4 + (int)threadIdx.y, fill);
  }
}

#include "ef_track_exchange.inc"
#ifdef EF_FAST_ORDER
#include "ef_track_fast_persistent.inc"
#else
#include "ef_track_ref_persistent.inc"
#endif

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void pyr_down_u16(const uint16_t* src, int scols, int srows, uint16_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_down_u16, tile_grid(scols / 2, srows / 2), tile_block(), 0, s, src, scols, srows, dst);
}
void create_vmap(const uint16_t* depth, int cols, int rows, Intr k, float cutoff, float* vmap, hipStream_t s) {
  hipLaunchKernelGGL(k_create_vmap, tile_grid(cols, rows), tile_block(), 0, s, depth, cols, rows, 1.f / k.fx, 1.f / k.fy, k.cx, k.cy, cutoff, vmap);
}
void create_nmap(const float* vmap, int cols, int rows, float* nmap, hipStream_t s) {
  hipLaunchKernelGGL(k_create_nmap, tile_grid(cols, rows), tile_block(), 0, s, vmap, cols, rows, nmap);
}
void transform_maps(const float* vsrc, const float* nsrc, int cols, int rows, const float* R9_dev, const float* t3_dev, float* vdst,
                    float* ndst, hipStream_t s) {
  hipLaunchKernelGGL(k_transform_maps, tile_grid(cols, rows), tile_block(), 0, s, vsrc, nsrc, cols, rows, R9_dev, t3_dev, vdst, ndst);
}
void copy_maps(const float* vtex, const float* ntex, int cols, int rows, float* vmaps_tmp, float* vmap, float* nmap, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_maps, tile_grid(cols, rows), tile_block(), 0, s, (const float4*)vtex, (const float4*)ntex, cols, rows,
                     (float4*)vmaps_tmp, vmap, nmap);
}
void resize_map(const float* in, int scols, int srows, float* out, bool normalize, hipStream_t s) {
  if (normalize) hipLaunchKernelGGL(k_resize_map<true>, tile_grid(scols / 2, srows / 2), tile_block(), 0, s, in, scols, srows, out);
  else hipLaunchKernelGGL(k_resize_map<false>, tile_grid(scols / 2, srows / 2), tile_block(), 0, s, in, scols, srows, out);
}
void pyr_down_gauss_f(const float* src, int scols, int srows, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_down_gauss_f, tile_grid(scols / 2, srows / 2), tile_block(), 0, s, src, scols, srows, dst);
}
void pyr_down_uchar_gauss(const uint8_t* src, int scols, int srows, uint8_t* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_pyr_down_uchar_gauss, tile_grid(scols / 2, srows / 2), tile_block(), 0, s, src, scols, srows, dst);
}
void vertices_to_depth(const float* vmaps_tmp, int cols, int rows, float cutoff, float* dst, hipStream_t s) {
  hipLaunchKernelGGL(k_vertices_to_depth, tile_grid(cols, rows), tile_block(), 0, s, (const float4*)vmaps_tmp, cols, rows, cutoff, dst);
}
void bgr_to_intensity(const uint8_t* src, int channels, int cols, int rows, uint8_t* dst, hipStream_t s) {
  const int n = cols * rows;
  if (channels == 4) hipLaunchKernelGGL(k_bgr_to_intensity<4>, dim3(ceil_div(n, 256)), dim3(256), 0, s, src, n, dst);
  else hipLaunchKernelGGL(k_bgr_to_intensity<3>, dim3(ceil_div(n, 256)), dim3(256), 0, s, src, n, dst);
}
void derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy, hipStream_t s) {
  hipLaunchKernelGGL(k_sobel, tile_grid(cols, rows), tile_block(), 0, s, src, cols, rows, dx, dy);
}
void project_to_point_cloud(const float* depth, int cols, int rows, Intr k, float* cloud, hipStream_t s) {
  hipLaunchKernelGGL(k_project_points, tile_grid(cols, rows), tile_block(), 0, s, depth, cols, rows, 1.0f / k.fx, 1.0f / k.fy, k.cx, k.cy, cloud);
}

namespace {
// one normal-equation accumulation launch (either tier); start / stop: optional events that receive the kernel's own begin / end
// timestamps (hipExtLaunchKernelGGL: what rocprofv3 --kernel-trace reports as the dispatch's duration)
template <bool HAS_ICP, bool HAS_RGB, bool PACKED>
void launch_accum(const IcpView& IV, const RgbView& RV, const Se3Inputs& in, int N, const Se3Out& out, hipStream_t s, hipEvent_t start = nullptr,
                  hipEvent_t stop = nullptr) {
#ifdef EF_FAST_ORDER
  {   // one workgroup per group of four tasks, four wavefronts per term (ef_track_fast.inc)
    const FastPlan fp = fast_plan(N);
    const dim3 fgrid(in.xcd_swizzle ? 8 * fast_groups_per_xcd(fp.NG) : fp.NG), fblock(256 * ((HAS_ICP && HAS_RGB) ? 2 : 1));
    hipExtLaunchKernelGGL((k_se3_accum_fast<HAS_ICP, HAS_RGB, PACKED>), fgrid, fblock, 0, s, start, stop, 0, IV, RV, in, out.pairs);
  }
#else
  constexpr int BLOCK = 64 * 2 * ACC_NW * ((HAS_ICP && HAS_RGB) ? 2 : 1);
  const dim3 grid(VWARPS / ACC_NW), block(BLOCK);
  // CH = steps (of four passes) a wavefront has in flight per round: 640x480 has 19 passes = 5 steps = ONE round of CH = 5; 1280x960 has
  // 75 passes = 19 steps = four dependent rounds.  CH = 10 there (two rounds, 231 registers: the launch runs two wavefronts per SIMD
  // anyway) was measured SLOWER: 28.4 vs 25.1 us (profiles/r03f_ab_fused_step_and_ch10.log) — twice the loads per round queue behind each
  // other in the CU's one address pipe for longer than the two saved round trips.
#ifdef EF_VISIT_PAIRS
  // (round 6: two visits per lane — a round is CH / 2 packed evaluations; 640x480: 5 steps = one round of CH = 6, the sixth step empty;
  // 1280x960: 19 steps = five rounds of CH = 4)
  if (HAS_ICP || PACKED) {
    const int S = ((N + VTHREADS - 1) / VTHREADS + 3) >> 2;
    if (S <= 2) hipExtLaunchKernelGGL((k_se3_accum<2, HAS_ICP, HAS_RGB, PACKED>), grid, block, 0, s, start, stop, 0, IV, RV, in, out);
    else if (S % 6 == 0 || S % 6 == 5 || !(S % 4 == 0 || S % 4 == 3)) hipExtLaunchKernelGGL((k_se3_accum<6, HAS_ICP, HAS_RGB, PACKED>), grid, block, 0, s, start, stop, 0, IV, RV, in, out);
    else hipExtLaunchKernelGGL((k_se3_accum<4, HAS_ICP, HAS_RGB, PACKED>), grid, block, 0, s, start, stop, 0, IV, RV, in, out);
    return;
  }
#endif
  if (N > 8 * VTHREADS) hipExtLaunchKernelGGL((k_se3_accum<5, HAS_ICP, HAS_RGB, PACKED>), grid, block, 0, s, start, stop, 0, IV, RV, in, out);
  else hipExtLaunchKernelGGL((k_se3_accum<2, HAS_ICP, HAS_RGB, PACKED>), grid, block, 0, s, start, stop, 0, IV, RV, in, out);
#endif
}
}  // namespace

namespace {
// fast order: groups of a level / workgroups of a so3Step launch (one per group, XCD-contiguous); the reference order ignores the former
inline int op_groups(int N) {
#ifdef EF_FAST_ORDER
  return fast_plan(N).NG;
#else
  (void)N;
  return 0;
#endif
}
inline int so3_grid(int N) {
#ifdef EF_FAST_ORDER
  return 8 * fast_groups_per_xcd(fast_plan(N).NG);
#else
  (void)N;
  return VWARPS / SO3_WPB;
#endif
}
}  // namespace

void icp_step_op(const IcpArgs& a, const float* vmap_curr, const float* nmap_curr, const float* vmap_g_prev,
                 const float* nmap_g_prev, int cols, int rows, float* scratch, float* out29_dev, hipStream_t s) {
  IcpView V{vmap_curr, nmap_curr, vmap_g_prev, nmap_g_prev, cols, rows, a.k, a.distThres, a.angleThres, sq_le_max(a.distThres), sq_lt_max(a.angleThres)};
  RgbView RV{};
  float* pose = scratch + SE3_ACCS * VWARPS;   // 24 floats of parameters behind the partials
  float h[24];
  for (int i = 0; i < 9; ++i) { h[i] = a.Rcurr[i]; h[12 + i] = a.Rprev_inv[i]; }
  for (int i = 0; i < 3; ++i) { h[9 + i] = a.tcurr[i]; h[21 + i] = a.tprev[i]; }
  (void)hipMemcpyAsync(pose, h, sizeof(h), hipMemcpyHostToDevice, s);
  (void)hipStreamSynchronize(s);  // h is a stack buffer
  Se3Inputs in{pose, pose + 9, pose + 12, pose + 21, nullptr, nullptr, 0.f, false};
  launch_accum<true, false, false>(V, RV, in, cols * rows, Se3Out{scratch}, s);
  hipLaunchKernelGGL(k_final_tree_op, dim3(1), dim3(SOLVE_BLOCK), 0, s, (const float*)scratch, SE3_ACCS, op_groups(cols * rows), out29_dev);
}
void rgb_residual_op(const RgbResidualArgs& a, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                     const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, void* corres, int cols,
                     int rows, int* out2_dev, hipStream_t s) {
  ResidualView V{dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, (DataTerm*)corres, cols, rows, a.minScale, a.maxDepthDelta};
  float* params;
  (void)hipMalloc((void**)&params, 12 * sizeof(float));
  float h[12];
  for (int i = 0; i < 9; ++i) h[i] = a.krkinv[i];
  for (int i = 0; i < 3; ++i) h[9 + i] = a.kt[i];
  (void)hipMemcpyAsync(params, h, sizeof(h), hipMemcpyHostToDevice, s);
  (void)hipMemsetAsync(out2_dev, 0, 2 * sizeof(int), s);
  hipLaunchKernelGGL(k_rgb_residual_op, dim3(ceil_div(cols * rows, REDUCE_BLOCK)), dim3(REDUCE_BLOCK), 0, s, V, (const float*)params,
                     (const float*)(params + 9), out2_dev);
  (void)hipStreamSynchronize(s);
  (void)hipFree(params);
}
void rgb_step_op(const void* corres, float sigma, const float* cloud, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                 float sobelScale, int cols, int rows, float* scratch, float* out29_dev, hipStream_t s) {
  IcpView IV{};
  RgbView V{corres, nullptr, cloud, dIdx, dIdy, cols, rows, Intr{fx, fy, 0, 0}, sobelScale};
  Se3Inputs in{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, sigma, false};
  launch_accum<false, true, false>(IV, V, in, cols * rows, Se3Out{scratch}, s);
  hipLaunchKernelGGL(k_final_tree_op, dim3(1), dim3(SOLVE_BLOCK), 0, s, (const float*)scratch, SE3_ACCS, op_groups(cols * rows), out29_dev);
}
void so3_step_op(const So3Args& a, const uint8_t* lastImage, const uint8_t* nextImage, int cols, int rows, float* scratch, float* out11_dev,
                 hipStream_t s) {
  hipLaunchKernelGGL(k_so3_op, dim3(so3_grid(cols * rows)), dim3(SO3_BLOCK), 0, s, lastImage, nextImage, cols, rows, a, scratch);
  hipLaunchKernelGGL(k_final_tree_op, dim3(1), dim3(SOLVE_BLOCK), 0, s, (const float*)scratch, SO3_ACCS, op_groups(cols * rows), out11_dev);
}

// ---- frame tier ----
void init_icp(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, hipStream_t s) {
  // level 0 of depth_tmp is the filtered depth image itself (no copy: cudaMemcpy2DFromArray of
  // RGBDOdometry.cpp:126-134 existed only to cross the GL/CUDA boundary)
  for (int i = 1; i < NUM_PYRS; ++i)
    pyr_down_u16(i == 1 ? depth_filtered : p.depth_tmp[i - 1], p.W(i - 1), p.H(i - 1), p.depth_tmp[i], s);
  VNLevels L;
  for (int i = 0; i < NUM_PYRS; ++i) {
    L.depth[i] = i == 0 ? depth_filtered : p.depth_tmp[i];
    L.vmap[i] = p.vmap_curr[i];
    L.nmap[i] = p.nmap_curr[i];
    L.cols[i] = p.W(i);
    L.rows[i] = p.H(i);
    L.k[i] = intr_level(k, i);
  }
  L.cutoff = cutoff;
  dim3 g = tile_grid(p.W(0), p.H(0));
  g.z = NUM_PYRS;
  hipLaunchKernelGGL(k_vmap_nmap_levels, g, tile_block(), 0, s, L);
}

// k_model_maps: one lane per level-0 column of a 4-row band (a quad of lanes per 4x4 block)
static inline dim3 model_maps_grid(const Pyramid& p) {
  return dim3(ceil_div(p.W(0), 64), ceil_div(p.H(0) / 4, 4));
}
void init_icp_model(const Pyramid& p, const float* pred_vertex, const float* pred_normal, const float* fill_vertex,
                    const float* fill_normal, const TrackState* st, float maxDepthRGB, hipStream_t s, const uint8_t* pred_image_rgba,
                    const uint8_t* fill_image_rgba, bool frameToFrameRGB, const FramePreprocess* with, bool tally) {
  ModelMapsArgs A{};
  A.pred_image = pred_image_rgba; A.fill_image = fill_image_rgba; A.force_fill_image = frameToFrameRGB;
  A.tally_image = (tally && pred_image_rgba) ? pred_image_rgba : nullptr;
  A.tally_out = const_cast<unsigned*>(&st->dense_count);
  A.last0 = pred_image_rgba ? p.lastImage[0] : nullptr;
  A.pred_vertex = (const float4*)pred_vertex; A.pred_normal = (const float4*)pred_normal;
  A.fill_vertex = (const float4*)fill_vertex; A.fill_normal = (const float4*)fill_normal;
  for (int i = 0; i < NUM_PYRS; ++i) { A.vmap[i] = p.vmap_g_prev[i]; A.nmap[i] = p.nmap_g_prev[i]; }
  A.depth0 = p.lastDepth[0];
  A.cols = p.W(0); A.rows = p.H(0);
  A.maxDepthRGB = maxDepthRGB;
  A.camera_frame = false;
  const dim3 mg = model_maps_grid(p);
  if (with) {   // the frame's depth pre-processing rides on this launch (k_frame_inputs)
    PreArgs P{};
    P.raw = with->raw; P.cols = p.W(0); P.rows = p.H(0); P.maxv = (unsigned)(with->maxD * 1000.0f); P.table = with->table;
    P.filtered = with->filtered; P.metric = with->metric; P.metric_filtered = with->metric_filtered;
    P.rgb3 = with->rgb3; P.next0 = p.nextImage[0]; P.rgb_keep = with->rgb_keep;
    P.tiles_x = (P.cols + pre::PRE_TW - 1) / pre::PRE_TW;
    P.tiles = P.tiles_x * ((P.rows + pre::PRE_TH - 1) / pre::PRE_TH);
    P.mm_x = (int)mg.x;
    hipLaunchKernelGGL(k_frame_inputs, dim3(P.tiles + (int)(mg.x * mg.y)), dim3(64, 4), 0, s, P, A, st);
    return;
  }
  hipLaunchKernelGGL(k_model_maps, mg, dim3(64, 4), 0, s, A, st);
}

namespace {
// model intensity L0 comes from the predicted image unless the device-side flag says fill-in
__global__ void k_model_intensity(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ fill, bool force_fill,
                                  const TrackState* __restrict__ st, int n, uint8_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint8_t* src = (force_fill || use_fill_in(st)) ? fill : pred;
  const uchar4 c = ((const uchar4*)src)[i];
  dst[i] = intensity_of((float)c.x, (float)c.y, (float)c.z);
}
}  // namespace

void init_rgb_model(const Pyramid& p, const uint8_t* pred_image_rgba, const uint8_t* fill_image_rgba, bool frameToFrameRGB,
                    const TrackState* st, hipStream_t s) {
  const int n = p.W(0) * p.H(0);
  // populateRGBDData(model): depth L0 already written by init_icp_model
  for (int i = 0; i + 1 < NUM_PYRS; ++i) pyr_down_gauss_f(p.lastDepth[i], p.W(i), p.H(i), p.lastDepth[i + 1], s);
  hipLaunchKernelGGL(k_model_intensity, dim3(ceil_div(n, 256)), dim3(256), 0, s, pred_image_rgba, fill_image_rgba, frameToFrameRGB, st, n,
                     p.lastImage[0]);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) pyr_down_uchar_gauss(p.lastImage[i], p.W(i), p.H(i), p.lastImage[i + 1], s);
}
void init_icp_maps(const Pyramid& p, const float* vertex, const float* normal, const uint8_t* image_rgba, const TrackState* st,
                   float maxDepthRGB, hipStream_t s) {
  ModelMapsArgs A{};
  A.pred_vertex = A.fill_vertex = (const float4*)vertex;
  A.pred_normal = A.fill_normal = (const float4*)normal;
  for (int i = 0; i < NUM_PYRS; ++i) { A.vmap[i] = p.vmap_curr[i]; A.nmap[i] = p.nmap_curr[i]; }
  A.depth0 = p.nextDepth[0];
  A.cols = p.W(0); A.rows = p.H(0);
  A.maxDepthRGB = maxDepthRGB;
  A.camera_frame = true;
  hipLaunchKernelGGL(k_model_maps, model_maps_grid(p), dim3(64, 4), 0, s, A, st);
  const int n = p.W(0) * p.H(0);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) pyr_down_gauss_f(p.nextDepth[i], p.W(i), p.H(i), p.nextDepth[i + 1], s);
  hipLaunchKernelGGL(k_model_intensity, dim3(ceil_div(n, 256)), dim3(256), 0, s, image_rgba, image_rgba, true, st, n, p.nextImage[0]);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) pyr_down_uchar_gauss(p.nextImage[i], p.W(i), p.H(i), p.nextImage[i + 1], s);
}
namespace {
// level-0 intensity of two RGBA images in one launch: blockIdx.y 0 = the model image ("last"), 1 = the current side's ("next")
__global__ void k_rgba_intensity_pair(const uint8_t* __restrict__ last_rgba, const uint8_t* __restrict__ next_rgba, int n, uint8_t* __restrict__ last0,
                                      uint8_t* __restrict__ next0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uchar4 c = ((const uchar4*)(blockIdx.y == 0 ? last_rgba : next_rgba))[i];
  (blockIdx.y == 0 ? last0 : next0)[i] = intensity_of((float)c.x, (float)c.y, (float)c.z);
}
}  // namespace
// Model-to-model tracking (the local loop closure's second tracker, ElasticFusion.cpp:463-467): init_icp_model + init_rgb_model on one
// predicted view and init_icp_maps on the other, as FIVE launches instead of twelve — the two k_model_maps, one launch for both level-0
// intensity images, one k_pyr_down_multi per pyramid step over {model depth, current depth, model intensity, current intensity}.  The two
// sides share no buffer (pyr2 / pyr3 keep nextDepth apart from lastDepth), the per-pixel functions are the ones the separate launches
// call: same results (rocprofv3, closed-loop bench: the separate launches were 4.3 + 4.3 + 2.1 per frame, ~76 us).
void init_model_pair(const Pyramid& p, const float* model_vertex, const float* model_normal, const uint8_t* model_image_rgba, const float* cur_vertex,
                     const float* cur_normal, const uint8_t* cur_image_rgba, const TrackState* st, float maxDepthRGB, hipStream_t s) {
  ModelMapsArgs A{};
  A.pred_vertex = A.fill_vertex = (const float4*)model_vertex;
  A.pred_normal = A.fill_normal = (const float4*)model_normal;
  for (int i = 0; i < NUM_PYRS; ++i) { A.vmap[i] = p.vmap_g_prev[i]; A.nmap[i] = p.nmap_g_prev[i]; }
  A.depth0 = p.lastDepth[0];
  A.cols = p.W(0); A.rows = p.H(0);
  A.maxDepthRGB = maxDepthRGB;
  A.camera_frame = false;
  hipLaunchKernelGGL(k_model_maps, model_maps_grid(p), dim3(64, 4), 0, s, A, st);
  A.pred_vertex = A.fill_vertex = (const float4*)cur_vertex;
  A.pred_normal = A.fill_normal = (const float4*)cur_normal;
  for (int i = 0; i < NUM_PYRS; ++i) { A.vmap[i] = p.vmap_curr[i]; A.nmap[i] = p.nmap_curr[i]; }
  A.depth0 = p.nextDepth[0];
  A.camera_frame = true;
  hipLaunchKernelGGL(k_model_maps, model_maps_grid(p), dim3(64, 4), 0, s, A, st);
  const int n = p.W(0) * p.H(0);
  hipLaunchKernelGGL(k_rgba_intensity_pair, dim3(ceil_div(n, 256), 2), dim3(256), 0, s, model_image_rgba, cur_image_rgba, n, p.lastImage[0], p.nextImage[0]);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) {
    PyrJobs J;
    J.src[0] = p.lastDepth[i]; J.dst[0] = p.lastDepth[i + 1]; J.type[0] = 1;
    J.src[1] = p.nextDepth[i]; J.dst[1] = p.nextDepth[i + 1]; J.type[1] = 1;
    J.src[2] = p.lastImage[i]; J.dst[2] = p.lastImage[i + 1]; J.type[2] = 2;
    J.src[3] = p.nextImage[i]; J.dst[3] = p.nextImage[i + 1]; J.type[3] = 2;
    J.scols = p.W(i); J.srows = p.H(i);
    dim3 g = tile_grid(p.W(i + 1), p.H(i + 1));
    g.z = 4;
    hipLaunchKernelGGL(k_pyr_down_multi, g, tile_block(), 0, s, J);
  }
}
void init_rgb_frame(const Pyramid& p, const uint8_t* rgb3, hipStream_t s) {
  // populateRGBDData(frame): nextDepth == lastDepth (Q1), only the intensity pyramid is new
  bgr_to_intensity(rgb3, 3, p.W(0), p.H(0), p.nextImage[0], s);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) pyr_down_uchar_gauss(p.nextImage[i], p.W(i), p.H(i), p.nextImage[i + 1], s);
}
namespace {
// both level-0 intensity images in one launch: blockIdx.y 0 = the camera frame ("next"), 1 = the model image ("last")
__global__ void k_intensity_both(const uint8_t* __restrict__ rgb3, const uint8_t* __restrict__ pred, const uint8_t* __restrict__ fill,
                                 bool force_fill, const TrackState* __restrict__ st, int n, uint8_t* __restrict__ next0,
                                 uint8_t* __restrict__ last0, uint8_t* __restrict__ rgb_keep, int first_half) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (blockIdx.y + first_half == 0) {   // (first_half + gridDim.y halves: both in the single-stream script, one each in the two-stream script)
    const uint8_t* sp = rgb3 + (size_t)i * 3;
    const uint8_t r = sp[0], g = sp[1], b = sp[2];
    next0[i] = intensity_of((float)r, (float)g, (float)b);
    if (rgb_keep) { rgb_keep[(size_t)i * 3] = r; rgb_keep[(size_t)i * 3 + 1] = g; rgb_keep[(size_t)i * 3 + 2] = b; }   // the context's copy of the frame
  } else {
    const uint8_t* src = (force_fill || use_fill_in(st)) ? fill : pred;
    const uchar4 c = ((const uchar4*)src)[i];
    last0[i] = intensity_of((float)c.x, (float)c.y, (float)c.z);
  }
}
}  // namespace
// initICP's depth pyramid + both halves of populateRGBDData in THREE launches instead of twelve (single-stream frame
// script): level-0 intensities, then one k_pyr_down_multi per pyramid step over {frame depth u16, model depth f32,
// model intensity, frame intensity}; then the per-level vertex/normal maps.  Same per-pixel functions, same results.
static SobelLevels sobel_levels_of(const Pyramid& p);
static void build_vn_levels(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, bool with_sobel, hipStream_t s);
void build_pyramids(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, const uint8_t* pred_image_rgba,
                    const uint8_t* fill_image_rgba, bool frameToFrameRGB, const uint8_t* rgb3, const TrackState* st, hipStream_t s,
                    uint8_t* rgb_keep, bool with_sobel) {
  const int n = p.W(0) * p.H(0);
  // (rgb3 == null: both level-0 intensity images are already there — the frame's from the pre-processing launch, the model's from k_model_maps)
  if (rgb3) hipLaunchKernelGGL(k_intensity_both, dim3(ceil_div(n, 256), 2), dim3(256), 0, s, rgb3, pred_image_rgba, fill_image_rgba, frameToFrameRGB, st,
                     n, p.nextImage[0], p.lastImage[0], rgb_keep, 0);
  // (both pyramid steps as ONE launch — a workgroup computing the 36 x 36 level-1 pixels its 16 x 16 level-2 tile reads, then the tile — was
  // built and measured in round 6: 25.2 us against 8.0 + 5.4 for the two launches: six dependent 25-tap trips per thread; dropped,
  // profiles/r06q_kernel_stats_fused_pyramid_steps.csv)
  // (one launch per pyramid STAGE — the step level l -> l + 1 together with the vertex / normal maps and the Sobel of level l, which read level l
  // only: k_pyramid_stage — was built and measured in round 6: 1934 against 1939 frames/s same box, 3 x 18.7 us against 2 x 13.5 + 25.5 at
  // 1280 x 960, profiles/r07d_ab_pyramid_stages.log: the level-0 maps no longer wait for two launches, but the two small stages cost what the one
  // joint launch of all levels cost; off, kept as the A/B build "pyrstages")
#ifdef EF_PYR_STAGES
  {
    StageJobs A{};
    for (int i = 0; i < NUM_PYRS; ++i) {
      A.V.depth[i] = i == 0 ? depth_filtered : p.depth_tmp[i];
      A.V.vmap[i] = p.vmap_curr[i]; A.V.nmap[i] = p.nmap_curr[i];
      A.V.cols[i] = p.W(i); A.V.rows[i] = p.H(i);
      A.V.k[i] = intr_level(k, i);
    }
    A.V.cutoff = cutoff;
    A.S = sobel_levels_of(p);
    for (int i = 0; i < NUM_PYRS; ++i) {   // stage i: level i -> i + 1 (not behind the last level), maps + Sobel of level i
      A.level = i;
      A.n_pyr = 0;
      if (i + 1 < NUM_PYRS) {
        A.J.src[0] = i == 0 ? (const void*)depth_filtered : (const void*)p.depth_tmp[i]; A.J.dst[0] = p.depth_tmp[i + 1]; A.J.type[0] = 0;
        A.J.src[1] = p.lastDepth[i]; A.J.dst[1] = p.lastDepth[i + 1]; A.J.type[1] = 1;
        A.J.src[2] = p.lastImage[i]; A.J.dst[2] = p.lastImage[i + 1]; A.J.type[2] = 2;
        A.J.src[3] = p.nextImage[i]; A.J.dst[3] = p.nextImage[i + 1]; A.J.type[3] = 2;
        A.J.scols = p.W(i); A.J.srows = p.H(i);
        A.n_pyr = 4;
      }
      dim3 g = tile_grid(p.W(i), p.H(i));
      g.z = A.n_pyr + 1 + (with_sobel ? 1 : 0);
      hipLaunchKernelGGL(k_pyramid_stage, g, tile_block(), 0, s, A);
    }
    return;
  }
#endif
  for (int i = 0; i + 1 < NUM_PYRS; ++i) {
    PyrJobs J;
    J.src[0] = i == 0 ? (const void*)depth_filtered : (const void*)p.depth_tmp[i]; J.dst[0] = p.depth_tmp[i + 1]; J.type[0] = 0;
    J.src[1] = p.lastDepth[i]; J.dst[1] = p.lastDepth[i + 1]; J.type[1] = 1;
    J.src[2] = p.lastImage[i]; J.dst[2] = p.lastImage[i + 1]; J.type[2] = 2;
    J.src[3] = p.nextImage[i]; J.dst[3] = p.nextImage[i + 1]; J.type[3] = 2;
    J.scols = p.W(i); J.srows = p.H(i);
    dim3 g = tile_grid(p.W(i + 1), p.H(i + 1));
    g.z = 4;
    hipLaunchKernelGGL(k_pyr_down_multi, g, tile_block(), 0, s, J);
  }
  build_vn_levels(p, depth_filtered, k, cutoff, with_sobel, s);
}
// The two-stream frame script (ef_set_input_overlap; round 6): build_pyramids split by what each kernel READS, so that the half that needs
// nothing but the new frame runs on the input stream beside the previous frame's fusion and prediction.  Same per-pixel functions, same
// results; four launches each instead of round 5's six and seven.
//   frame side:  level-0 intensity of the frame (+ the context's copy of the RGB image), two pyramid steps over {frame depth u16, frame
//                intensity}, the vertex / normal maps of all levels
//   model side:  (behind init_icp_model) level-0 intensity of the model image, two pyramid steps over {model depth f32, model intensity};
//                the Sobel / photometric-gate launch (init_rgb_sobel) reads both sides (the gate tests the MODEL's depth: quirk Q1) and
//                follows the join
void build_pyramids_frame_side(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, const uint8_t* rgb3, hipStream_t s,
                               uint8_t* rgb_keep) {
  const int n = p.W(0) * p.H(0);
  if (rgb3)   // (null: the pre-processing launch wrote the frame's level-0 intensity)
    hipLaunchKernelGGL(k_intensity_both, dim3(ceil_div(n, 256), 1), dim3(256), 0, s, rgb3, (const uint8_t*)nullptr, (const uint8_t*)nullptr, false,
                       (const TrackState*)nullptr, n, p.nextImage[0], p.lastImage[0], rgb_keep, 0);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) {
    PyrJobs J{};
    J.src[0] = i == 0 ? (const void*)depth_filtered : (const void*)p.depth_tmp[i]; J.dst[0] = p.depth_tmp[i + 1]; J.type[0] = 0;
    J.src[1] = p.nextImage[i]; J.dst[1] = p.nextImage[i + 1]; J.type[1] = 2;
    J.scols = p.W(i); J.srows = p.H(i);
    dim3 g = tile_grid(p.W(i + 1), p.H(i + 1));
    g.z = 2;
    hipLaunchKernelGGL(k_pyr_down_multi, g, tile_block(), 0, s, J);
  }
  build_vn_levels(p, depth_filtered, k, cutoff, false, s);
}
void build_pyramids_model_side(const Pyramid& p, const uint8_t* pred_image_rgba, const uint8_t* fill_image_rgba, bool frameToFrameRGB,
                               const TrackState* st, hipStream_t s) {
  const int n = p.W(0) * p.H(0);
  if (pred_image_rgba)   // (null: k_model_maps wrote the model's level-0 intensity)
    hipLaunchKernelGGL(k_intensity_both, dim3(ceil_div(n, 256), 1), dim3(256), 0, s, (const uint8_t*)nullptr, pred_image_rgba, fill_image_rgba,
                       frameToFrameRGB, st, n, p.nextImage[0], p.lastImage[0], (uint8_t*)nullptr, 1);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) {
    PyrJobs J{};
    J.src[0] = p.lastDepth[i]; J.dst[0] = p.lastDepth[i + 1]; J.type[0] = 1;
    J.src[1] = p.lastImage[i]; J.dst[1] = p.lastImage[i + 1]; J.type[1] = 2;
    J.scols = p.W(i); J.srows = p.H(i);
    dim3 g = tile_grid(p.W(i + 1), p.H(i + 1));
    g.z = 2;
    hipLaunchKernelGGL(k_pyr_down_multi, g, tile_block(), 0, s, J);
  }
}
static void build_vn_levels(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, bool with_sobel, hipStream_t s) {
  VNLevels L;
  for (int i = 0; i < NUM_PYRS; ++i) {
    L.depth[i] = i == 0 ? depth_filtered : p.depth_tmp[i];
    L.vmap[i] = p.vmap_curr[i];
    L.nmap[i] = p.nmap_curr[i];
    L.cols[i] = p.W(i);
    L.rows[i] = p.H(i);
    L.k[i] = intr_level(k, i);
  }
  L.cutoff = cutoff;
  dim3 g = tile_grid(p.W(0), p.H(0));
  if (with_sobel) {   // init_rgb_sobel's launch rides along
    g.z = 2 * NUM_PYRS;
    hipLaunchKernelGGL(k_vn_sobel_levels, g, tile_block(), 0, s, L, sobel_levels_of(p));
    return;
  }
  g.z = NUM_PYRS;
  hipLaunchKernelGGL(k_vmap_nmap_levels, g, tile_block(), 0, s, L);
}
void init_rgb_sobel(const Pyramid& p, hipStream_t s) {
  dim3 g = tile_grid(p.W(0), p.H(0));
  g.z = NUM_PYRS;
  hipLaunchKernelGGL(k_sobel_levels, g, tile_block(), 0, s, sobel_levels_of(p));
}
static SobelLevels sobel_levels_of(const Pyramid& p) {
  SobelLevels L;
  const float minGrad[NUM_PYRS] = {5, 3, 1};  // RGBDOdometry.cpp:112-114
  const float sobelScale = 1.0f / 8.0f;       // RGBDOdometry.cpp:39-40
  for (int i = 0; i < NUM_PYRS; ++i) {
    L.src[i] = p.nextImage[i]; L.dx[i] = p.dIdx[i]; L.dy[i] = p.dIdy[i];
    L.nextDepth[i] = p.nextDepth[i]; L.mask[i] = p.rgbMask[i]; L.corres[i] = p.corres[i];
    L.minScale[i] = (float)(pow((double)minGrad[i], 2.0) / pow((double)sobelScale, 2.0));  // RGBDOdometry.cpp:425
    L.cols[i] = p.W(i); L.rows[i] = p.H(i);
  }
  return L;
}

void init_first_rgb(const Pyramid& p, const uint8_t* rgb3, hipStream_t s) {
  bgr_to_intensity(rgb3, 3, p.W(0), p.H(0), p.lastNextImage[0], s);
  for (int i = 0; i + 1 < NUM_PYRS; ++i) pyr_down_uchar_gauss(p.lastNextImage[i], p.W(i), p.H(i), p.lastNextImage[i + 1], s);
}

namespace {
template <int PPT>
void launch_step(const Pyramid& p, TrackState* st, int level, int cur, int slots_out, const StepArgs& A, hipStream_t s) {
  const int cols = p.W(level), rows = p.H(level), N = cols * rows;
  ResidualPackedView RV{p.rgbMask[level], p.lastDepth[level], p.nextDepth[level], p.lastImage[level], p.nextImage[level], p.corres[level], cols, rows,
                        0.07f /* maxDepthDeltaRGB, RGBDOdometry.cpp:41 */};
  const int grid = A.has_body ? ceil_div(N, REDUCE_BLOCK * PPT) : 1;
  hipLaunchKernelGGL(k_track_step<PPT>, dim3(grid), dim3(REDUCE_BLOCK), 0, s, RV, st, (const GNState*)&st->gn[cur], &st->gn[cur ^ 1],
                     (const float*)p.partials, (const int*)&st->rgb_slots[slots_out ^ 1][0][0], &st->rgb_slots[slots_out][0][0], A);
}
// one Gauss-Newton iteration = two launches: k_track_step (update of the previous iteration + correspondence search) and
// k_se3_accum (normal equations).  `it` = index of the iteration within the call; cur = GNState buffer to read.
// Returns the buffer the NEXT step reads.
int launch_iteration(const Pyramid& p, TrackState* st, int level, Intr kl, const TrackParams& tp, bool icp, bool rgb, int it, int cur,
                     int prev_level, hipStream_t s, KernelProbe* probe) {
  const int cols = p.W(level), rows = p.H(level), N = cols * rows;
  const bool sample = probe && level == 0 && probe->used < probe->capacity;
  const int sp = it & 1;
  const bool level_changes = level != prev_level;
  StepArgs A{it > 0, rgb, icp, rgb, tp.rgbOnly, tp.icpWeight, kl, level_changes, it, op_groups(p.W(prev_level) * p.H(prev_level))};
  // The update step as its own ONE-workgroup launch (head only), then the correspondence search alone (body only): three launches
  // per iteration.  Evaluating the update redundantly at the head of every workgroup of the correspondence kernel instead (two
  // launches) was built and measured in round 2: 888 vs 1322 frames/s; dropped from the source in round 3.
  if (A.has_head && A.has_body && tp.fused_step && !tp.rgbOnly) {
    // two launches per iteration: the update step by workgroup 0 of the search launch, handed to the others in-launch (see k_track_step)
    if (N >= 256 * 1024) launch_step<2>(p, st, level, cur, sp, A, s);
    else launch_step<1>(p, st, level, cur, sp, A, s);
    cur ^= 1;
  } else {
  if (A.has_head) {
    StepArgs H = A;
    H.has_body = false;
    launch_step<1>(p, st, level, cur, sp, H, s);
    cur ^= 1;
  }
  if (A.has_body) {
    StepArgs B = A;
    B.has_head = false;
    // (four times fewer, four times fatter wavefronts — PPT 8 / 4 — measured 4.5 % slower end to end: DESIGN.md 6)
    if (N >= 256 * 1024) launch_step<2>(p, st, level, cur, sp, B, s);
    else launch_step<1>(p, st, level, cur, sp, B, s);
  }
  }
  const GNState* g = &st->gn[cur];
  IcpView IV{p.vmap_curr[level], p.nmap_curr[level], p.vmap_g_prev[level], p.nmap_g_prev[level], cols, rows, kl, tp.distThres, tp.angleThres, sq_le_max(tp.distThres), sq_lt_max(tp.angleThres)};
  RgbView GV{p.corres[level], p.lastDepth[level], nullptr, p.dIdx[level], p.dIdy[level], cols, rows, kl, 1.0f / 8.0f};
  Se3Inputs in{g->Rcurr, g->tcurr, st->Rprev_inv, st->tprev, &st->rgb_slots[sp][0][0], &g->rgb_broken, 0.f, tp.rgbOnly};
  in.slots_zero = &st->rgb_slots[sp ^ 1][0][0];
  const Se3Out out{p.partials};
  hipEvent_t e0 = sample ? probe->start[probe->used] : nullptr, e1 = sample ? probe->stop[probe->used] : nullptr;
  if (sample) probe->used++;
  if (icp && rgb) launch_accum<true, true, true>(IV, GV, in, N, out, s, e0, e1);
  else if (icp) launch_accum<true, false, true>(IV, GV, in, N, out, s, e0, e1);
  else launch_accum<false, true, true>(IV, GV, in, N, out, s, e0, e1);
  return cur;
}
}  // namespace

namespace {
std::mutex g_chain_mu;                 // the per-device chain of persistent launches (see track())
hipStream_t g_chain_last[64] = {};
bool g_chain_has[64] = {};
hipEvent_t g_chain_ev[64] = {};
inline bool last_valid_other(int dev, hipStream_t s) { return g_chain_has[dev] && g_chain_last[dev] != s; }
}  // namespace
void persistent_chain_forget(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_chain_mu);
  for (int d = 0; d < 64; ++d)
    if (g_chain_has[d] && g_chain_last[d] == s) { g_chain_has[d] = false; g_chain_last[d] = nullptr; }
}

// Which script a call runs decides what the exchange areas (Pyramid::partials) hold: 1 = the granules of a 256-workgroup persistent launch, 0 = the
// per-step kernels' plain partials / k_track_small's barrier words.  A context that changes scripts gets the areas cleared (tags must never match
// by accident; the other script's words start from zeros) — and the sticky words they hold, the abort flag and the fallback count, carried over
// on the host first (a clear used to erase them silently): one synchronisation per switch, which happens when an option is toggled or
// a sampled frame leaves a graph-replaying context's script, never in a steady replay.  Under a CALLER's capture nothing can be read back: the
// clear is recorded (the caller's graph then runs the launch-per-step script, which keeps no sticky word).
static void switch_script(Pyramid& p, int target, hipStream_t s, bool capturing) {
  if (p.last_mode == target) return;
  if (p.partials) {
    if (!capturing) {
      if (tracker_aborted(p, s) > 0) p.sticky_abort = 1u;
      const int f = tracker_fallbacks(p, s);
      if (f > 0) p.fallbacks_base = (unsigned)f;   // (tracker_fallbacks already includes the earlier base)
    }
    (void)hipMemsetAsync(p.partials, 0, sizeof(float) * PARTIAL_ALLOC_FLOATS, s);
  }
  p.last_mode = target;
}
static int script_of(const TrackParams& tp, bool capturing, int n_total) {
#ifdef EF_FAST_ORDER
  const int pmode = (tp.persistent && !capturing) ? 1 : 0;
#else
  const int pmode = capturing ? 0 : tp.persistent;
#endif
  return (pmode == 1 && !tp.rgbOnly && n_total <= FT_MAX_ITER) ? 1 : 0;
}
void track_prepare(Pyramid& p, const TrackParams& tp, hipStream_t s) {
  const int n_total = (tp.fastOdom ? 3 : 10) + (tp.pyramid ? 9 : 0);
  switch_script(p, script_of(tp, false, n_total), s, false);
}
TrackTail track(Pyramid& p, TrackState* st, Intr k, const TrackParams& tp, hipStream_t s, KernelProbe* probe, KernelProbe* probe_all) {
  const bool icp = !tp.rgbOnly && tp.icpWeight > 0;       // RGBDOdometry.cpp:266-267
  const bool rgb = tp.rgbOnly || tp.icpWeight < 100;
  int iterations[NUM_PYRS];
  iterations[0] = tp.fastOdom ? 3 : 10;  // RGBDOdometry.cpp:371-373
  iterations[1] = tp.pyramid ? 5 : 0;
  iterations[2] = tp.pyramid ? 4 : 0;
  int first_level = 0;
  for (int i = NUM_PYRS - 1; i >= 0; --i)
    if (iterations[i] > 0) { first_level = i; break; }
  const int so3_level = 2;
  // (a caller capturing `s` into a graph of its own gets the launch-per-step script: the persistent launches take a fresh epoch per launch
  // and wait for the device's chain event, neither of which a replayed graph can carry)
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(s, &capture) == hipSuccess && capture != hipStreamCaptureStatusNone;
#ifdef EF_FAST_ORDER
  const int pmode = (tp.persistent && !capturing) ? 1 : 0;
#else
  const int pmode = capturing ? 0 : tp.persistent;
#endif
  // The persistent launch takes k_track_begin, the SO(3) loop and the leading iterations whose level fits (coarse to fine: once a
  // level is too large, it and everything after it run one launch per step).  rgbOnly keeps the per-step script (its per-level
  // "break" bookkeeping is not in the persistent kernel).
  int n_small = 0;
  PtArgs PA{};
  if (pmode == 2 && !tp.rgbOnly) {   // (reference-order builds only: the fast order has no launch of the small levels)
    bool fits = true;
    for (int i = NUM_PYRS - 1; i >= 0 && fits; --i) {
      if (iterations[i] == 0) continue;
      fits = p.W(i) * p.H(i) <= PT_MAX_PIXELS && n_small + iterations[i] <= PT_MAX_ITER;
      for (int j = 0; fits && j < iterations[i]; ++j) PA.levels |= (unsigned)i << (2 * n_small++);
    }
  }
  int it = 0, cur = 0, prev_level = first_level;
  // The persistent launch of 256 co-resident workgroups takes the WHOLE call — k_track_begin, the SO(3) loop, every iteration of every level
  // (fast order: k_track_fast, ef_track_fast_persistent.inc; reference order: k_track_ref, ef_track_ref_persistent.inc) — or, with `persistent`
  // off / rgbOnly (whose per-level "break" bookkeeping is not in the kernels), nothing: then every step is its own launch, in the same order
  // of additions.  (Reference-order builds: persistent == 2 asks for round 3's launch of the small levels, k_track_small, below.)
  const int n_total = iterations[0] + iterations[1] + iterations[2];
  if (pmode == 1 && !tp.rgbOnly && n_total <= FT_MAX_ITER) {
    switch_script(p, 1, s, capturing);   // another script of this instance may have left anything in the exchange areas: tags must never match by accident
    FtArgs FA{};
    for (int i = 0; i < NUM_PYRS; ++i)
      FA.L[i] = PtLevel{p.vmap_curr[i], p.nmap_curr[i], p.vmap_g_prev[i], p.nmap_g_prev[i], p.rgbMask[i], p.lastDepth[i], p.nextDepth[i],
                        p.lastImage[i], p.nextImage[i], p.corres[i], p.dIdx[i], p.dIdy[i], p.W(i), p.H(i), intr_level(k, i)};
    int n = 0;
    for (int i = NUM_PYRS - 1; i >= 0; --i)
      for (int j = 0; j < iterations[i]; ++j) FA.levels |= (unsigned long long)i << (2 * n++);
    FA.n_iter = n;
    FA.so3 = tp.so3;
    FA.so3_last = p.lastNextImage[so3_level];
    FA.so3_next = p.nextImage[so3_level];
    FA.so3_cols = p.W(so3_level);
    FA.so3_rows = p.H(so3_level);
    FA.kso3 = intr_level(k, so3_level);
    FA.kfirst = intr_level(k, first_level);
    FA.icpWeight = tp.icpWeight;
    FA.distThres = tp.distThres;
    FA.angleThres = tp.angleThres;
    FA.dist2Max = sq_le_max(tp.distThres);
    FA.sine2Max = sq_lt_max(tp.angleThres);
    FA.partials = p.partials;
    if (p.epoch > 0xFFFFFFFFu - 2u * FT_EPOCHS) p.epoch = 1;   // (a slot keeps a 2^32-launch-old tag only if nobody wrote it since)
    FA.epoch = p.epoch;
    p.epoch += FT_EPOCHS;
    FA.out_cur = 0;
    FA.empty_model_flag = tp.so3 ? nullptr : tp.empty_model_flag;   // (the SO(3) loop looks at the frame-side images only: not covered by the shortcut)
    FA.empty_model_value = tp.empty_model_value;
#ifndef EF_FAST_ORDER
    // level-resident pixel data (ef_track_ref_persistent.inc, rt_res_mode): the largest need of a level whose slots fit — both sides, or the
    // ICP side alone (1280 x 960 level 0) — under what the chip gives one workgroup beside the kernel's static LDS
    unsigned res_bytes = 0;
    if (!tp.no_resident && rt_res_budget() > 0) {
      for (int i = 0; i < NUM_PYRS; ++i) {
        if (iterations[i] == 0) continue;
        const unsigned S = (unsigned)(((p.W(i) * p.H(i) + VTHREADS - 1) / VTHREADS + 3) >> 2);
        const unsigned n2 = S * ((icp ? RT_ICP_STEP_BYTES : 0) + (rgb ? RT_RGB_STEP_BYTES : 0)), n1 = icp ? S * RT_ICP_STEP_BYTES : 0u;
        const unsigned need = n2 <= rt_res_budget() ? n2 : (n1 <= rt_res_budget() ? n1 : 0u);
        res_bytes = need > res_bytes ? need : res_bytes;
      }
    }
    FA.res_bytes = res_bytes;
#endif
    const bool sample_all = probe_all && probe_all->used < probe_all->capacity;
    hipEvent_t e0 = sample_all ? probe_all->start[probe_all->used] : nullptr, e1 = sample_all ? probe_all->stop[probe_all->used] : nullptr;
    if (sample_all) probe_all->used++;
    {
      // The launch needs its 256 workgroups resident TOGETHER, one per CU.  Two such launches in flight at once (two contexts of this
      // process, each on its own stream) could each hold a part of the chip and wait for the rest for ever, so the persistent launches of
      // a device are chained, in the order the host enqueues them, through one event per device: a launch first waits (on the GPU, not on
      // the host) for the previous one, whatever stream that ran on.  On one stream this is a no-op.  Ordinary kernels of other streams
      // only delay it (they end without waiting for anybody); another PROCESS's persistent kernels are outside this chain: bounded spins,
      // FtSync::abort, tracker_aborted.
      // Round 6: the chain costs nothing while ONE stream launches them (the common case: an event record between two kernels of a stream is a
      // barrier packet, measured as a 6 us bubble behind every tracker launch in profiles/r06f_timeline_single_stream.txt) — the dependency is
      // created by the launch that finds ANOTHER stream's launch before it: it records the event on that stream now (everything enqueued there
      // so far, the persistent launch included) and makes its own stream wait for it.  persistent_chain_forget() drops a stream that is destroyed.
      std::lock_guard<std::mutex> lk(g_chain_mu);
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (dev >= 0 && dev < 64) {   // (a device index beyond the table: no chain, the bounded spins remain)
        hipStream_t& last = g_chain_last[dev];
        hipEvent_t& ev = g_chain_ev[dev];
        if (last_valid_other(dev, s)) {
          if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
          if (ev && hipEventRecord(ev, last) == hipSuccess) (void)hipStreamWaitEvent(s, ev, 0);
          (void)hipGetLastError();
        }
        last = s;
        g_chain_has[dev] = true;
      }
#ifdef EF_FAST_ORDER
      if (icp && rgb) hipExtLaunchKernelGGL((k_track_fast<true, true>), dim3(FT_WGS), dim3(FT_BLOCK), 0, s, e0, e1, 0, FA, st);
      else if (icp) hipExtLaunchKernelGGL((k_track_fast<true, false>), dim3(FT_WGS), dim3(FT_BLOCK), 0, s, e0, e1, 0, FA, st);
      else hipExtLaunchKernelGGL((k_track_fast<false, true>), dim3(FT_WGS), dim3(FT_BLOCK), 0, s, e0, e1, 0, FA, st);
#else
      if (icp && rgb) hipExtLaunchKernelGGL((k_track_ref<true, true>), dim3(FT_WGS), dim3(FT_BLOCK), res_bytes, s, e0, e1, 0, FA, st);
      else if (icp) hipExtLaunchKernelGGL((k_track_ref<true, false>), dim3(FT_WGS), dim3(FT_BLOCK), res_bytes, s, e0, e1, 0, FA, st);
      else hipExtLaunchKernelGGL((k_track_ref<false, true>), dim3(FT_WGS), dim3(FT_BLOCK), res_bytes, s, e0, e1, 0, FA, st);
#endif
    }
    // (returns at once unless the launch above found part of the chip taken: see its admission step)
#ifdef EF_FAST_ORDER
    if (icp && rgb) hipLaunchKernelGGL((k_track_serial<true, true>), dim3(1), dim3(REDUCE_BLOCK), 0, s, p.partials, st, FA.epoch);
    else if (icp) hipLaunchKernelGGL((k_track_serial<true, false>), dim3(1), dim3(REDUCE_BLOCK), 0, s, p.partials, st, FA.epoch);
    else hipLaunchKernelGGL((k_track_serial<false, true>), dim3(1), dim3(REDUCE_BLOCK), 0, s, p.partials, st, FA.epoch);
#endif
    track_swap(p, tp);
    TrackTail tail{0, (n - 1) & 1, n > 0, icp, rgb, tp.rgbOnly, tp.icpWeight, intr_level(k, 0), p.partials + FT_P_OFF};
    tail.ng = 1;   // the reducers of the last iteration left the TOTALS in column 0
#ifndef EF_FAST_ORDER
    tail.merged_end = true;   // (the one-workgroup fallback rides at the head of track_end's launch: k_track_ref_end)
    tail.epoch = FA.epoch;
    tail.partials = p.partials;
#endif
    return tail;
  }
  switch_script(p, 0, s, capturing);   // (the per-step kernels' plain partials and k_track_small's barrier words start from zeros)
#ifdef EF_FAST_ORDER
  n_small = 0;
  hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, s, st, tp.so3, intr_level(k, so3_level), intr_level(k, first_level));
  if (tp.so3) {
    for (int i = 0; i < 10; ++i)
      hipLaunchKernelGGL(k_so3_iteration, dim3(so3_grid(p.W(so3_level) * p.H(so3_level))), dim3(SO3_BLOCK), 0, s,
                         (const uint8_t*)p.lastNextImage[so3_level], (const uint8_t*)p.nextImage[so3_level], p.W(so3_level), p.H(so3_level),
                         intr_level(k, so3_level), intr_level(k, first_level), i, st, p.partials);
  }
#else
  if (pmode == 2 && !tp.rgbOnly && (n_small > 0 || tp.so3)) {   // (pmode, not tp.persistent: a capturing caller gets the launch-per-step script)
    for (int i = 0; i < NUM_PYRS; ++i)
      PA.L[i] = PtLevel{p.vmap_curr[i], p.nmap_curr[i], p.vmap_g_prev[i], p.nmap_g_prev[i], p.rgbMask[i], p.lastDepth[i], p.nextDepth[i],
                        p.lastImage[i], p.nextImage[i], p.corres[i], p.dIdx[i], p.dIdy[i], p.W(i), p.H(i), intr_level(k, i)};
    PA.n_iter = n_small;
    PA.so3 = tp.so3;
    PA.so3_last = p.lastNextImage[so3_level];
    PA.so3_next = p.nextImage[so3_level];
    PA.so3_cols = p.W(so3_level);
    PA.so3_rows = p.H(so3_level);
    PA.kso3 = intr_level(k, so3_level);
    PA.kfirst = intr_level(k, first_level);
    PA.icpWeight = tp.icpWeight;
    PA.distThres = tp.distThres;
    PA.angleThres = tp.angleThres;
    PA.dist2Max = sq_le_max(tp.distThres);
    PA.sine2Max = sq_lt_max(tp.angleThres);
    PA.partials = p.partials;
    PA.out_cur = 0;
    if (icp && rgb) hipLaunchKernelGGL((k_track_small<true, true>), dim3(PT_WGS), dim3(PT_BLOCK), 0, s, PA, st);
    else if (icp) hipLaunchKernelGGL((k_track_small<true, false>), dim3(PT_WGS), dim3(PT_BLOCK), 0, s, PA, st);
    else hipLaunchKernelGGL((k_track_small<false, true>), dim3(PT_WGS), dim3(PT_BLOCK), 0, s, PA, st);
    it = n_small;
    if (n_small > 0) prev_level = (int)((PA.levels >> (2 * (n_small - 1))) & 3u);
  } else {
    n_small = 0;
    hipLaunchKernelGGL(k_track_begin, dim3(1), dim3(64), 0, s, st, tp.so3, intr_level(k, so3_level), intr_level(k, first_level));
    if (tp.so3) {
      for (int i = 0; i < 10; ++i)
        hipLaunchKernelGGL(k_so3_iteration, dim3(VWARPS / SO3_WPB), dim3(SO3_BLOCK), 0, s, (const uint8_t*)p.lastNextImage[so3_level],
                           (const uint8_t*)p.nextImage[so3_level], p.W(so3_level), p.H(so3_level), intr_level(k, so3_level),
                           intr_level(k, first_level), i, st, p.partials);
    }
  }
#endif
  int done = 0;
  for (int i = NUM_PYRS - 1; i >= 0; --i) {
    const Intr kl = intr_level(k, i);
    for (int j = 0; j < iterations[i]; ++j) {
      if (done++ < n_small) continue;   // ran inside the persistent launch
      cur = launch_iteration(p, st, i, kl, tp, icp, rgb, it, cur, prev_level, s, probe);
      prev_level = i;
      ++it;
    }
  }
  track_swap(p, tp);
  // the last iteration's update is evaluated at the head of k_track_end (track_end below)
  TrackTail tail{cur, (it - 1) & 1, it > 0, icp, rgb, tp.rgbOnly, tp.icpWeight, intr_level(k, 0), p.partials};
  tail.ng = op_groups(p.W(prev_level) * p.H(prev_level));
  return tail;
}
// host-side tail of getIncrementalTransformation: the frame's intensity pyramid becomes the SO(3) reference of the next
// (RGBDOdometry.cpp:284-288 swaps lastNextImage / nextImage); separate so that a replayed hipGraph can do it without launching
// developer instrumentation: the -DEF_STAGE_CLOCKS sums of k_track_small (24 x u64, 10 ns ticks; zeros in a normal build), read and reset
int tracker_small_clocks(const Pyramid& p, unsigned long long* out32, hipStream_t s) {
  if (!p.partials) return -1;
  unsigned long long* src = (unsigned long long*)((char*)(p.partials + FT_SY_OFF) + offsetof(FtSync, clk));
  int n = 32;
#ifndef EF_FAST_ORDER
  if (p.last_mode != 1) { src = (unsigned long long*)((char*)(p.partials + 2 * PARTIAL_FLOATS) + offsetof(PtSync, clk)); n = 24; }   // k_track_small
#endif
  for (int i = 0; i < 32; ++i) out32[i] = 0ull;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  if (hipMemcpy(out32, src, n * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return hipMemset(src, 0, n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
int tracker_aborted(const Pyramid& p, hipStream_t s) {
  if (p.sticky_abort) return 1;   // (carried over a script switch)
  if (!p.partials) return 0;
#ifndef EF_FAST_ORDER
  if (p.last_mode != 1) {   // round 3's launch of the small levels (k_track_small) keeps its flag in PtSync
    PtSync h;
    if (hipMemcpyAsync(&h, p.partials + 2 * PARTIAL_FLOATS, offsetof(PtSync, wg_sums), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
    if (hipStreamSynchronize(s) != hipSuccess) return -1;
    return h.abort != 0 ? 1 : 0;
  }
#endif
  unsigned flag = 0;
  if (hipMemcpyAsync(&flag, (const char*)(p.partials + FT_SY_OFF) + offsetof(FtSync, abort), sizeof(flag), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  return flag != 0 ? 1 : 0;
}
int tracker_fallbacks(const Pyramid& p, hipStream_t s) {
  if (!p.partials || p.last_mode != 1) return (int)p.fallbacks_base;   // (what earlier persistent launches of this instance counted)
  unsigned n = 0;
  if (hipMemcpyAsync(&n, (const char*)(p.partials + FT_SY_OFF) + offsetof(FtSync, fallbacks), sizeof(n), hipMemcpyDeviceToHost, s) != hipSuccess) return -1;
  if (hipStreamSynchronize(s) != hipSuccess) return -1;
  return (int)(n + p.fallbacks_base);
}
void track_swap(Pyramid& p, const TrackParams& tp) {
  if (tp.so3)
    for (int i = 0; i < NUM_PYRS; ++i) { uint8_t* tmp = p.lastNextImage[i]; p.lastNextImage[i] = p.nextImage[i]; p.nextImage[i] = tmp; }
}

// exported for the context: finishing kernels
void track_end(TrackState* st, const TrackTail& u, bool rgb, float weightMultiplier, double* traj, int slot, hipStream_t s, const unsigned* abort_word,
               unsigned* abort_report) {
  const StepArgs A{u.has_head, false, u.icp, u.rgb, u.rgbOnly, u.icpWeight, u.k0, true, 63, u.ng};
#ifndef EF_FAST_ORDER
  if (u.merged_end) {
    unsigned* report = abort_word ? abort_report : nullptr;
    if (u.icp && u.rgb) hipLaunchKernelGGL((k_track_ref_end<true, true>), dim3(1), dim3(FT_BLOCK), 0, s, u.partials, st, u.epoch, A, u.slots, rgb, weightMultiplier, traj, slot, report);
    else if (u.icp) hipLaunchKernelGGL((k_track_ref_end<true, false>), dim3(1), dim3(FT_BLOCK), 0, s, u.partials, st, u.epoch, A, u.slots, rgb, weightMultiplier, traj, slot, report);
    else hipLaunchKernelGGL((k_track_ref_end<false, true>), dim3(1), dim3(FT_BLOCK), 0, s, u.partials, st, u.epoch, A, u.slots, rgb, weightMultiplier, traj, slot, report);
    return;
  }
#endif
  hipLaunchKernelGGL(k_track_end, dim3(1), dim3(REDUCE_BLOCK), 0, s, st, (const GNState*)&st->gn[u.cur], &st->gn[u.cur ^ 1], u.pairs,
                     (const int*)&st->rgb_slots[u.slots & 1][0][0], A, rgb, weightMultiplier, traj, slot, abort_word, abort_word ? abort_report : nullptr);
}
// device address of the sticky abort word of this tracker instance's persistent launches (FtSync::abort; round 3's k_track_small: PtSync::abort)
unsigned* tracker_abort_word(const Pyramid& p) {
  if (!p.partials) return nullptr;
#ifndef EF_FAST_ORDER
  if (p.last_mode != 1) return (unsigned*)((char*)(p.partials + 2 * PARTIAL_FLOATS) + offsetof(PtSync, abort));
#endif
  return (unsigned*)((char*)(p.partials + FT_SY_OFF) + offsetof(FtSync, abort));
}
void pose_injected(TrackState* st, const double* T_wc16, bool save_prev, float weightMultiplier, bool with_weighting, double* traj, int slot,
                   hipStream_t s) {
  hipLaunchKernelGGL(k_pose_injected, dim3(1), dim3(64), 0, s, st, efl::se3_from_matrix(T_wc16), save_prev, weightMultiplier, with_weighting,
                     traj, slot);
}
void pose_restored(TrackState* st, const double* q4, const double* t3, hipStream_t s) {
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = q4[i];
  for (int i = 0; i < 3; ++i) T.t[i] = t3[i];
  hipLaunchKernelGGL(k_pose_injected, dim3(1), dim3(64), 0, s, st, T, false, 1.0f, false, (double*)nullptr, 0);
}
void copy_pose(TrackState* dst, const TrackState* src, hipStream_t s) { hipLaunchKernelGGL(k_copy_pose, dim3(1), dim3(64), 0, s, dst, src); }
void adopt_pose(TrackState* st, const TrackState* est, double* traj, int slot, hipStream_t s) {
  hipLaunchKernelGGL(k_adopt_pose, dim3(1), dim3(64), 0, s, st, est, traj, slot);
}
void sample_constraints(const float* vertex4, const uint16_t* old_time, int cols, int rows, int step, float* out4, hipStream_t s) {
  const int cw = cols / step, ch = rows / step;
  hipLaunchKernelGGL(k_sample_constraints, dim3(ceil_div(cw * ch, 256)), dim3(256), 0, s, (const float4*)vertex4, old_time, cols, cw, ch, step,
                     (float4*)out4);
}
void log_pose(const TrackState* st, double* traj, int slot, hipStream_t s) { hipLaunchKernelGGL(k_log_pose, dim3(1), dim3(64), 0, s, st, traj, slot); }

}  // namespace eft
