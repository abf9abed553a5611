// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// CPU restatement of the reference's CUDA tracking operators:
//   Core/Cuda/cudafuncs.cu  (pyramids, vertex/normal maps, Sobel, back-projection)
//   Core/Cuda/reduce.cu     (icpStep, computeRgbResidual, rgbStep, so3Step + two-stage reductions)
// Each function cites the lines it follows.  Image layouts are the reference's: planar maps are
// float[3*rows][cols] (x rows, then y rows, then z rows); quirks Q1-Q14 of SURVEY.md §8a are kept.
// parity PINNED: built with -DEFO_NO_FMA this file reproduces, bit for bit, the reference's own Core/Cuda sources compiled
// for the CPU (oracle/_ref/libefr_cuda.so, oracle/cuda_on_cpu/) and the golden vectors made from them (tests/golden/).
#include "efo_common.h"
#include "efo_api.h"
#include <algorithm>
#include <cstdlib>
#include <vector>

using namespace efo;

extern "C" {

// pyrDownGaussKernel, cudafuncs.cu:75-109 (sigma_color = 30, cudafuncs.cu:117)
void efo_pyr_down_u16(const uint16_t* src, int scols, int srows, uint16_t* dst) {
  const int dcols = scols / 2, drows = srows / 2;
  const float sigma_color = 30.f;
  const float weights[3] = {0.375f, 0.25f, 0.0625f};
  const int D = 5;
  for (int y = 0; y < drows; ++y)
    for (int x = 0; x < dcols; ++x) {
      int center = src[(2 * y) * scols + 2 * x];
      int x_mi = std::max(0, 2 * x - D / 2) - 2 * x;
      int y_mi = std::max(0, 2 * y - D / 2) - 2 * y;
      int x_ma = std::min(scols, 2 * x - D / 2 + D) - 2 * x;
      int y_ma = std::min(srows, 2 * y - D / 2 + D) - 2 * y;
      float sum = 0, wall = 0;
      for (int yi = y_mi; yi < y_ma; ++yi)
        for (int xi = x_mi; xi < x_ma; ++xi) {
          int val = src[(2 * y + yi) * scols + 2 * x + xi];
          if (std::abs(val - center) < 3 * sigma_color) {
            sum += val * weights[std::abs(xi)] * weights[std::abs(yi)];  // ((float)val*wx)*wy, no contraction
            wall += weights[std::abs(xi)] * weights[std::abs(yi)];
          }
        }
      dst[y * dcols + x] = (uint16_t)static_cast<int>(sum / wall);
    }
}

// computeVmapKernel, cudafuncs.cu:123-149.  Q3: only the x-plane gets NaN for invalid pixels.
void efo_create_vmap(const uint16_t* depth, int cols, int rows, float fx, float fy, float cx, float cy,
                     float depthCutoff, float* vmap) {
  const float fx_inv = 1.f / fx, fy_inv = 1.f / fy;  // createVMap, cudafuncs.cu:166
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u) {
      float z = depth[v * cols + u] / 1000.f;
      if (z != 0 && z < depthCutoff) {
        vmap[v * cols + u] = z * (u - cx) * fx_inv;
        vmap[(v + rows) * cols + u] = z * (v - cy) * fy_inv;
        vmap[(v + 2 * rows) * cols + u] = z;
      } else {
        vmap[v * cols + u] = qnan();
      }
    }
}

// computeNmapKernel, cudafuncs.cu:170-204
void efo_create_nmap(const float* vmap, int cols, int rows, float* nmap) {
  auto V = [&](int plane, int y, int x) { return vmap[(y + plane * rows) * cols + x]; };
  for (int v = 0; v < rows; ++v)
    for (int u = 0; u < cols; ++u) {
      if (u == cols - 1 || v == rows - 1) { nmap[v * cols + u] = qnan(); continue; }
      float x00 = V(0, v, u), x01 = V(0, v, u + 1), x10 = V(0, v + 1, u);
      if (!std::isnan(x00) && !std::isnan(x01) && !std::isnan(x10)) {
        f3 v00{x00, V(1, v, u), V(2, v, u)};
        f3 v01{x01, V(1, v, u + 1), V(2, v, u + 1)};
        f3 v10{x10, V(1, v + 1, u), V(2, v + 1, u)};
        f3 r = normalized(cross(v01 - v00, v10 - v00));
        nmap[v * cols + u] = r.x;
        nmap[(v + rows) * cols + u] = r.y;
        nmap[(v + 2 * rows) * cols + u] = r.z;
      } else {
        nmap[v * cols + u] = qnan();
      }
    }
}

// tranformMapsKernel, cudafuncs.cu:221-270 (in place, as RGBDOdometry.cpp:199-207 calls it)
void efo_transform_maps(float* vmap, float* nmap, int cols, int rows, const float* R, const float* t) {
  m33 Rm = m33_from(R);
  f3 tv{t[0], t[1], t[2]};
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float vx = vmap[y * cols + x];
      float outx = qnan();
      if (!std::isnan(vx)) {
        f3 vs{vx, vmap[(y + rows) * cols + x], vmap[(y + 2 * rows) * cols + x]};
        f3 vd = mul(Rm, vs) + tv;
        vmap[(y + rows) * cols + x] = vd.y;
        vmap[(y + 2 * rows) * cols + x] = vd.z;
        outx = vd.x;
      }
      vmap[y * cols + x] = outx;
      float nx = nmap[y * cols + x];
      float outnx = qnan();
      if (!std::isnan(nx)) {
        f3 ns{nx, nmap[(y + rows) * cols + x], nmap[(y + 2 * rows) * cols + x]};
        f3 nd = mul(Rm, ns);
        nmap[(y + rows) * cols + x] = nd.y;
        nmap[(y + 2 * rows) * cols + x] = nd.z;
        outnx = nd.x;
      }
      nmap[y * cols + x] = outnx;
    }
}

// copyMapsKernelTex, cudafuncs.cu:295-350: float4 vertex/normal images -> vmaps_tmp (AoS copy of the
// vertex image) + planar maps with z==0 -> NaN in all three planes.
void efo_copy_maps(const float* vtex, const float* ntex, int cols, int rows, float* vmaps_tmp, float* vmap,
                   float* nmap) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const float* vs = vtex + (size_t)(y * cols + x) * 4;
      const float* ns = ntex + (size_t)(y * cols + x) * 4;
      for (int c = 0; c < 4; ++c) vmaps_tmp[(size_t)y * cols * 4 + x * 4 + c] = vs[c];
      f3 vd{qnan(), qnan(), qnan()}, nd{qnan(), qnan(), qnan()};
      if (!(vs[2] == 0)) {
        vd = {vs[0], vs[1], vs[2]};
        nd = {ns[0], ns[1], ns[2]};
      }
      vmap[y * cols + x] = vd.x;
      vmap[(y + rows) * cols + x] = vd.y;
      vmap[(y + 2 * rows) * cols + x] = vd.z;
      nmap[y * cols + x] = nd.x;
      nmap[(y + rows) * cols + x] = nd.y;
      nmap[(y + 2 * rows) * cols + x] = nd.z;
    }
}

// resizeMapKernel<normalize>, cudafuncs.cu:413-465.  Only the x-plane is written for NaN cells.
void efo_resize_map(const float* in, int scols, int srows, float* out, int normalize) {
  const int dcols = scols / 2, drows = srows / 2;
  auto I = [&](int plane, int y, int x) { return in[(y + plane * srows) * scols + x]; };
  for (int y = 0; y < drows; ++y)
    for (int x = 0; x < dcols; ++x) {
      int xs = x * 2, ys = y * 2;
      float x00 = I(0, ys, xs), x01 = I(0, ys, xs + 1), x10 = I(0, ys + 1, xs), x11 = I(0, ys + 1, xs + 1);
      if (std::isnan(x00) || std::isnan(x01) || std::isnan(x10) || std::isnan(x11)) {
        out[y * dcols + x] = qnan();
        continue;
      }
      f3 n;
      n.x = (x00 + x01 + x10 + x11) / 4;
      n.y = (I(1, ys, xs) + I(1, ys, xs + 1) + I(1, ys + 1, xs) + I(1, ys + 1, xs + 1)) / 4;
      n.z = (I(2, ys, xs) + I(2, ys, xs + 1) + I(2, ys + 1, xs) + I(2, ys + 1, xs + 1)) / 4;
      if (normalize) n = normalized(n);
      out[y * dcols + x] = n.x;
      out[(y + drows) * dcols + x] = n.y;
      out[(y + 2 * drows) * dcols + x] = n.z;
    }
}

static const float kGauss25[25] = {1, 4, 6, 4, 1, 4, 16, 24, 16, 4, 6, 24, 36, 24, 6, 4, 16, 24, 16, 4, 1, 4, 6, 4, 1};

// pyrDownKernelGaussF, cudafuncs.cu:383-411.  Q7: window [2x-2, min(2x+3, W-1)), mirrored index,
// int count += float, 0/0 -> NaN when every tap is NaN.
void efo_pyr_down_gauss_f(const float* src, int scols, int srows, float* dst) {
  const int dcols = scols / 2, drows = srows / 2, D = 5;
  for (int y = 0; y < drows; ++y)
    for (int x = 0; x < dcols; ++x) {
      int tx = std::min(2 * x - D / 2 + D, scols - 1);
      int ty = std::min(2 * y - D / 2 + D, srows - 1);
      float sum = 0;
      int count = 0;
      for (int cy = std::max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = std::max(0, 2 * x - D / 2); cx < tx; ++cx) {
          float s = src[cy * scols + cx];
          if (!std::isnan(s)) {
            float g = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
            sum += s * g;
            count = (int)((float)count + g);  // "count += gaussKernel[...]" with int count
          }
        }
      dst[y * dcols + x] = (float)(sum / (float)count);
    }
}

// pyrDownKernelIntensityGauss, cudafuncs.cu:512-542 (zeros ignored; float -> u8 truncation)
void efo_pyr_down_uchar_gauss(const uint8_t* src, int scols, int srows, uint8_t* dst) {
  const int dcols = scols / 2, drows = srows / 2, D = 5;
  for (int y = 0; y < drows; ++y)
    for (int x = 0; x < dcols; ++x) {
      int tx = std::min(2 * x - D / 2 + D, scols - 1);
      int ty = std::min(2 * y - D / 2 + D, srows - 1);
      float sum = 0;
      int count = 0;
      for (int cy = std::max(0, 2 * y - D / 2); cy < ty; ++cy)
        for (int cx = std::max(0, 2 * x - D / 2); cx < tx; ++cx) {
          uint8_t s = src[cy * scols + cx];
          if (s > 0) {
            float g = kGauss25[(ty - cy - 1) * 5 + (tx - cx - 1)];
            sum += s * g;
            count = (int)((float)count + g);
          }
        }
      float q = sum / (float)count;  // count==0 -> NaN/inf; CUDA float->u8 conversion of NaN gives 0
      int iv = std::isnan(q) ? 0 : (int)std::min(std::max(q, 0.0f), 255.0f);
      dst[y * dcols + x] = (uint8_t)iv;
    }
}

// verticesToDepthKernel, cudafuncs.cu:564-574
void efo_vertices_to_depth(const float* vmaps_tmp, int cols, int rows, float cutOff, float* dst) {
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float z = vmaps_tmp[(size_t)y * cols * 4 + x * 4 + 2];
      dst[y * cols + x] = (z > cutOff || z <= 0) ? qnan() : z;
    }
}

// bgr2IntensityKernel, cudafuncs.cu:584-596 on the RGBA8 texel as stored (Q5)
void efo_bgr_to_intensity(const uint8_t* rgba, int cols, int rows, uint8_t* dst) {
  for (int i = 0; i < cols * rows; ++i) {
    const uint8_t* s = rgba + (size_t)i * 4;
    int value = (int)((float)s[0] * 0.114f + (float)s[1] * 0.299f + (float)s[2] * 0.587f);
    dst[i] = (uint8_t)value;
  }
}

// applyKernel + coefficients, cudafuncs.cu:612-668.  Q6: tap index walks 8 -> 0 over the clipped window.
void efo_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy) {
  const float gsx[9] = {0.52201f, 0.00000f, -0.52201f, 0.79451f, -0.00000f, -0.79451f, 0.52201f, 0.00000f, -0.52201f};
  const float gsy[9] = {0.52201f, 0.79451f, 0.52201f, 0.00000f, 0.00000f, 0.00000f, -0.52201f, -0.79451f, -0.52201f};
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float dxVal = 0, dyVal = 0;
      int k = 8;
      for (int j = std::max(y - 1, 0); j <= std::min(y + 1, rows - 1); ++j)
        for (int i = std::max(x - 1, 0); i <= std::min(x + 1, cols - 1); ++i) {
          float s = (float)src[j * cols + i];
          dxVal += s * gsx[k];
          dyVal += s * gsy[k];
          --k;
        }
      dx[y * cols + x] = (int16_t)(int)dxVal;
      dy[y * cols + x] = (int16_t)(int)dyVal;
    }
}

// projectPointsKernel, cudafuncs.cu:670-688 (cloud is float3 AoS)
void efo_project_to_point_cloud(const float* depth, int cols, int rows, float fx, float fy, float cx, float cy,
                                float* cloud) {
  const float invFx = 1.0f / fx, invFy = 1.0f / fy;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      float z = depth[y * cols + x];
      float* c = cloud + (size_t)(y * cols + x) * 3;
      c[0] = (float)((x - cx) * z * invFx);
      c[1] = (float)((y - cy) * z * invFy);
      c[2] = z;
    }
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Two-stage reductions of reduce.cu, with the reference's summation ORDER restated:
//   stage 1: <<<64,256>>> grid-stride (reduce.cu:313-317): virtual thread g sums pixels g, g+16384, ...
//            blockReduceSum (reduce.cu:97-125): warp32 shfl_down tree, shared[32], warp-0 tree
//   stage 2: reduceSum<<<1,1024>>> over the 64 partials (reduce.cu:127-140)
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kReduceBlocks = 64, kReduceThreads = 256, kWarp = 32, kMaxThreads = 1024;  // types.cuh:62-65

template <typename T, int K>
struct Acc {
  T v[K];
};

// lane-0 result of warpReduceSum (reduce.cu:57-95) over 32 lanes; out-of-range shfl_down returns
// the caller's own value, which never reaches lane 0.
template <typename T, int K>
Acc<T, K> warp_tree(const Acc<T, K>* lanes) {
  Acc<T, K> w[kWarp];
  for (int l = 0; l < kWarp; ++l) w[l] = lanes[l];
  for (int off = kWarp / 2; off > 0; off /= 2) {
    Acc<T, K> nw[kWarp];
    for (int l = 0; l < kWarp; ++l) {
      int src = (l + off < kWarp) ? l + off : l;
      for (int k = 0; k < K; ++k) nw[l].v[k] = w[l].v[k] + w[src].v[k];
    }
    for (int l = 0; l < kWarp; ++l) w[l] = nw[l];
  }
  return w[0];
}

template <typename T, int K>
Acc<T, K> block_reduce(const std::vector<Acc<T, K>>& threads) {  // threads.size() = blockDim
  const int nthreads = (int)threads.size();
  const int nwarps = nthreads / kWarp;
  Acc<T, K> shared[kWarp];
  for (auto& s : shared)
    for (int k = 0; k < K; ++k) s.v[k] = T(0);
  for (int w = 0; w < nwarps; ++w) shared[w] = warp_tree<T, K>(&threads[w * kWarp]);
  Acc<T, K> lanes[kWarp];
  for (int l = 0; l < kWarp; ++l) {
    if (l < nthreads / kWarp) lanes[l] = shared[l];
    else for (int k = 0; k < K; ++k) lanes[l].v[k] = T(0);
  }
  return warp_tree<T, K>(lanes);
}

// F: (int pixelIndex) -> Acc ; ADD: how a thread folds one value into its running sum
template <typename T, int K, typename F>
Acc<T, K> two_stage_reduce(int N, F&& products) {
  std::vector<Acc<T, K>> partial(kReduceBlocks);
  efo::parallel_for(kReduceBlocks, [&](int b0, int b1) {   // the 64 blocks are independent (reduce.cu:313-317)
    std::vector<Acc<T, K>> threads(kReduceThreads);
    for (int b = b0; b < b1; ++b) {
      for (int t = 0; t < kReduceThreads; ++t) {
        Acc<T, K> sum;
        for (int k = 0; k < K; ++k) sum.v[k] = T(0);
        for (int i = b * kReduceThreads + t; i < N; i += kReduceThreads * kReduceBlocks) products(i, sum);
        threads[t] = sum;
      }
      partial[b] = block_reduce<T, K>(threads);
    }
  });
  std::vector<Acc<T, K>> th2(kMaxThreads);
  for (int t = 0; t < kMaxThreads; ++t) {
    for (int k = 0; k < K; ++k) th2[t].v[k] = T(0);
    if (t < kReduceBlocks)
      for (int k = 0; k < K; ++k) th2[t].v[k] = th2[t].v[k] + partial[t].v[k];
  }
  return block_reduce<T, K>(th2);
}

#ifdef EFO_FAST_ORDER
// ---------------------------------------------------------------------------------------------
// THE FAST ORDER (round 4): the summation order of the SHIPPED build (libefusion_hip.so / libefo_oracle.so).  fp32 addition is not
// associative, so an order has to be specified; the reference's own (above: 16384 grid-stride threads, warp32 / 8-warp / 64-block
// trees) is kept by the reference-rounding pair (libefusion_hip_nofma.so / libefo_oracle_nofma.so, pinned against the compiled
// reduce.cu).  This one is laid out for wave64 hardware and depends on the pixel count N only:
//   row-group  = 64 consecutive pixels; RG = ceil(N / 64)
//   task       = U consecutive row-groups, U = max(1, ceil(RG / 1024)); T = ceil(RG / U) <= 1024 tasks
//   leaf (t,l) = lane l of task t: accumulates pixels 64 (U t + k) + l, k = 0 .. U-1, in that order (one FMA per product, from +0)
//   total      = the complete binary tree over the leaves in index order t * 64 + l, adjacent pairs first:
//                s[i] = s[2 i] + s[2 i + 1], level by level (missing leaves are +0: a running sum that starts at +0 is never -0, so
//                adding the padding is exact)
// On the GPU: a wavefront per task (per-lane register accumulation, DPP tree with offsets 1, 2, .. 32), four tasks per workgroup
// ((t0 + t1) + (t2 + t3) through LDS), a 256-leaf tree over the workgroup partials.  Integer sums (Acc<int, 2>) are order-free.
template <typename T, int K, typename F>
Acc<T, K> fast_reduce(int N, F&& products) {
  const int RG = (N + 63) / 64;
  const int U = std::max(1, (RG + 1023) / 1024);
  const int T_ = (RG + U - 1) / U;
  std::vector<Acc<T, K>> s((size_t)T_ * 64);
  efo::parallel_for(T_, [&](int t0, int t1) {
    for (int t = t0; t < t1; ++t)
      for (int l = 0; l < 64; ++l) {
        Acc<T, K> sum;
        for (int k = 0; k < K; ++k) sum.v[k] = T(0);
        for (int u = 0; u < U; ++u) {
          const long p = 64L * ((long)U * t + u) + l;
          if (p < N) products((int)p, sum);
        }
        s[(size_t)t * 64 + l] = sum;
      }
  });
  size_t n = s.size();
  while (n > 1) {
    const size_t h = (n + 1) / 2;
    for (size_t i = 0; i < h; ++i) {
      if (2 * i + 1 < n)
        for (int k = 0; k < K; ++k) s[i].v[k] = s[2 * i].v[k] + s[2 * i + 1].v[k];
      else
        s[i] = s[2 * i];
    }
    n = h;
  }
  return s[0];
}
#define EFO_REDUCE fast_reduce
#else
#define EFO_REDUCE two_stage_reduce
#endif

// sum.add(values) of the 29 JtJJtrSE3 members from a 7-vector row (types.cuh:104-143, reduce.cu:291-306).
// nvcc's default -fmad=true contracts "sum.x += a*b" into an FMA; restated explicitly (efo_common.h).
inline void add_products7(const float row[7], float found, Acc<float, 29>& sum) {
  int s = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) { sum.v[s] = EFO_FMA(row[i], row[j], sum.v[s]); ++s; }
  sum.v[27] = EFO_FMA(row[6], row[6], sum.v[27]);
  sum.v[28] += found;
}

// host unpack, reduce.cu:385-400
inline void unpack_se3(const Acc<float, 29>& h, float* A, float* b, float* residual) {
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      float value = h.v[shift++];
      if (j == 6) b[i] = value;
      else A[j * 6 + i] = A[i * 6 + j] = value;
    }
  if (residual) { residual[0] = h.v[27]; residual[1] = h.v[28]; }
}

}  // namespace

extern "C" {

// icpStep = icpKernel + reduceSum, reduce.cu:204-401
void efo_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr,
                  const float* Rprev_inv, const float* tprev, float fx, float fy, float cx, float cy,
                  const float* vmap_g_prev, const float* nmap_g_prev, float distThres, float angleThres, int cols,
                  int rows, float* A, float* b, float* residual) {
  const m33 Rc = m33_from(Rcurr), Rpi = m33_from(Rprev_inv);
  const f3 tc{tcurr[0], tcurr[1], tcurr[2]}, tp{tprev[0], tprev[1], tprev[2]};
  auto P = [&](const float* m, int plane, int y, int x) { return m[(y + plane * rows) * cols + x]; };
  auto products = [&](int i, Acc<float, 29>& sum) {
    int y = i / cols, x = i - y * cols;
    float row[7] = {0, 0, 0, 0, 0, 0, 0};
    bool found = false;
    // search(), reduce.cu:228-269
    f3 vcurr{P(vmap_curr, 0, y, x), P(vmap_curr, 1, y, x), P(vmap_curr, 2, y, x)};
    f3 vcurr_g = mul(Rc, vcurr) + tc;
    f3 vcurr_cp = mul(Rpi, vcurr_g - tp);
    int ux = f2i_rn(vcurr_cp.x * fx / vcurr_cp.z + cx);
    int uy = f2i_rn(vcurr_cp.y * fy / vcurr_cp.z + cy);
    if (!(ux < 0 || uy < 0 || ux >= cols || uy >= rows || vcurr_cp.z < 0)) {
      f3 vprev_g{P(vmap_g_prev, 0, uy, ux), P(vmap_g_prev, 1, uy, ux), P(vmap_g_prev, 2, uy, ux)};
      f3 ncurr{P(nmap_curr, 0, y, x), P(nmap_curr, 1, y, x), P(nmap_curr, 2, y, x)};
      f3 ncurr_g = mul(Rc, ncurr);
      f3 nprev_g{P(nmap_g_prev, 0, uy, ux), P(nmap_g_prev, 1, uy, ux), P(nmap_g_prev, 2, uy, ux)};
      float dist = norm(vprev_g - vcurr_g);
      float sine = norm(cross(ncurr_g, nprev_g));
      found = (sine < angleThres && dist <= distThres && !std::isnan(ncurr.x) && !std::isnan(nprev_g.x));
      if (found) {  // getProducts(), reduce.cu:271-309
        f3 s_cp = mul(Rpi, vcurr_g - tp);
        f3 d_cp = mul(Rpi, vprev_g - tp);
        f3 n_cp = mul(Rpi, nprev_g);
        f3 c = cross(s_cp, n_cp);
        row[0] = n_cp.x; row[1] = n_cp.y; row[2] = n_cp.z;
        row[3] = c.x; row[4] = c.y; row[5] = c.z;
        row[6] = dot(n_cp, s_cp - d_cp);
      }
    }
    add_products7(row, found ? 1.0f : 0.0f, sum);
  };
  Acc<float, 29> h = EFO_REDUCE<float, 29>(cols * rows, products);
  unpack_se3(h, A, b, residual);
}

// computeRgbResidual = residualKernel + reduceSum(int2), reduce.cu:603-787
void efo_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                      const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, void* corres_out,
                      float maxDepthDelta, const float* kt, const float* krkinv, int cols, int rows,
                      int* sigmaSum, int* count) {
  DataTerm* corresImg = (DataTerm*)corres_out;
  const m33 K = m33_from(krkinv);
  auto products = [&](int k, Acc<int, 2>& sum) {
    int i = k / cols, j0 = k - i * cols;
    DataTerm corres;
    std::memset(&corres, 0, sizeof(corres));  // reference leaves these uninitialised; only .valid is defined
    corres.valid = 0;
    int vx = 0, vy = 0;
    if (j0 < cols - 5 && i < rows - 1) {
      bool valid = true;
      for (int u = std::max(i - 2, 0); u < std::min(i + 2, rows); ++u)      // Q10: asymmetric [-2,+1]
        for (int v = std::max(j0 - 2, 0); v < std::min(j0 + 2, cols); ++v) valid = valid && (nextImage[u * cols + v] > 0);
      if (valid) {
        int16_t valx = dIdx[i * cols + j0], valy = dIdy[i * cols + j0];
        float mTwo = (float)((valx * valx) + (valy * valy));
        if (mTwo >= minScale) {
          int y = i, x = j0;
          float d1 = nextDepth[y * cols + x];
          if (!std::isnan(d1)) {
            float transformed_d1 = (float)(d1 * (K.r[2].x * x + K.r[2].y * y + K.r[2].z) + kt[2]);
            int u0 = f2i_rn((d1 * (K.r[0].x * x + K.r[0].y * y + K.r[0].z) + kt[0]) / transformed_d1);
            int v0 = f2i_rn((d1 * (K.r[1].x * x + K.r[1].y * y + K.r[1].z) + kt[1]) / transformed_d1);
            if (u0 >= 0 && v0 >= 0 && u0 < cols && v0 < rows) {
              float d0 = lastDepth[v0 * cols + u0];
              if (d0 > 0 && std::fabs(transformed_d1 - d0) <= maxDepthDelta && lastImage[v0 * cols + u0] != 0) {
                corres.zero_x = (int16_t)u0;
                corres.zero_y = (int16_t)v0;
                corres.one_x = (int16_t)x;
                corres.one_y = (int16_t)y;
                corres.diff = (float)nextImage[y * cols + x] - (float)lastImage[v0 * cols + u0];
                corres.valid = 1;
                vx = 1;
                vy = (int)(corres.diff * corres.diff);  // Q9: int accumulation
              }
            }
          }
        }
      }
    }
    corresImg[k] = corres;
    sum.v[0] += vx;
    sum.v[1] += vy;
  };
  Acc<int, 2> h = two_stage_reduce<int, 2>(cols * rows, products);
  *count = h.v[0];
  *sigmaSum = h.v[1];
}

#ifdef EFO_FAST_ORDER
// test hook (tests/test_oracle_fast_order.py): THE FAST ORDER applied to plain numbers — leaf (t, l) adds v[64 (U t + k) + l] for
// k = 0 .. U - 1 in order, then the adjacent-pair tree — so that the order itself is pinned by an independent restatement
void efo_fast_order_sum(const float* v, int N, float* out) {
  auto products = [&](int i, Acc<float, 1>& sum) { sum.v[0] = sum.v[0] + v[i]; };
  *out = fast_reduce<float, 1>(N, products).v[0];
}
#endif

// rgbStep = rgbKernel + reduceSum, reduce.cu:403-550
void efo_rgb_step(const void* corres_in, float sigma, const float* cloud, float fx, float fy, const int16_t* dIdx,
                  const int16_t* dIdy, float sobelScale, int cols, int rows, float* A, float* b) {
  const DataTerm* corresImg = (const DataTerm*)corres_in;
  auto products = [&](int i, Acc<float, 29>& sum) {
    const DataTerm& c = corresImg[i];
    float row[7] = {0, 0, 0, 0, 0, 0, 0};
    bool found = c.valid != 0;
    if (found) {
      float w = sigma + std::fabs(c.diff);
      w = w > 1.19209290E-07F ? 1.0f / w : 1.0f;
      if (sigma == -1) w = 1;
      row[6] = -w * c.diff;
      const float* cp = cloud + (size_t)(c.zero_y * cols + c.zero_x) * 3;
      f3 p{cp[0], cp[1], cp[2]};
      float invz = (float)(1.0 / p.z);
      float dI_dx_val = w * sobelScale * dIdx[c.one_y * cols + c.one_x];
      float dI_dy_val = w * sobelScale * dIdy[c.one_y * cols + c.one_x];
      float v0 = dI_dx_val * fx * invz;
      float v1 = dI_dy_val * fy * invz;
      float v2 = -(v0 * p.x + v1 * p.y) * invz;
      row[0] = v0; row[1] = v1; row[2] = v2;
      row[3] = -p.z * v1 + p.y * v2;
      row[4] = p.z * v0 - p.x * v2;
      row[5] = -p.y * v0 + p.x * v1;
    }
    add_products7(row, found ? 1.0f : 0.0f, sum);
  };
  Acc<float, 29> h = EFO_REDUCE<float, 29>(cols * rows, products);
  unpack_se3(h, A, b, nullptr);
}

// so3Step = so3Kernel + reduceSum, reduce.cu:789-973
void efo_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis, const float* kinv,
                  const float* krlr, int cols, int rows, float* A, float* b, float* residual) {
  const m33 IB = m33_from(imageBasis), KI = m33_from(kinv), KR = m33_from(krlr);
  auto grad = [&](const uint8_t* img, int x, int y, float& gx, float& gy) {  // getGradient, reduce.cu:804-818
    float actu = (float)img[y * cols + x];
    float back = (float)img[y * cols + x - 1], fore = (float)img[y * cols + x + 1];
    gx = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
    back = (float)img[(y - 1) * cols + x];
    fore = (float)img[(y + 1) * cols + x];
    gy = ((back + actu) / 2.0f) - ((fore + actu) / 2.0f);
  };
  auto products = [&](int k, Acc<float, 11>& sum) {
    int y = k / cols, x = k - y * cols;
    f3 unwarped{(float)x, (float)y, 1.0f};
    f3 warped = mul(IB, unwarped);
    int wx = f2i_rn(warped.x / warped.z), wy = f2i_rn(warped.y / warped.z);
    bool found = (wx >= 1 && wx < cols - 1 && wy >= 1 && wy < rows - 1 && x >= 1 && x < cols - 1 && y >= 1 && y < rows - 1);
    float row[4] = {0, 0, 0, 0};
    if (found) {
      float gnx, gny, glx, gly;
      grad(nextImage, wx, wy, gnx, gny);
      grad(lastImage, x, y, glx, gly);
      float gx = (gnx + glx) / 2.0f, gy = (gny + gly) / 2.0f;
      f3 point = mul(KI, unwarped);
      float z2 = point.z * point.z;
      float a = KR.r[0].x, bb = KR.r[0].y, c = KR.r[0].z;
      float d = KR.r[1].x, e = KR.r[1].y, f = KR.r[1].z;
      float g = KR.r[2].x, h = KR.r[2].y, ii = KR.r[2].z;
      f3 left{((point.z * (d * gy + a * gx)) - (gy * g * y) - (gx * g * x)) / z2,
              ((point.z * (e * gy + bb * gx)) - (gy * h * y) - (gx * h * x)) / z2,
              ((point.z * (f * gy + c * gx)) - (gy * ii * y) - (gx * ii * x)) / z2};
      f3 jr = cross(left, point);
      row[0] = jr.x; row[1] = jr.y; row[2] = jr.z;
      row[3] = -((float)nextImage[wy * cols + wx] - (float)lastImage[y * cols + x]);
    }
    int s = 0;
    for (int i = 0; i < 3; ++i)
      for (int j = i; j < 4; ++j) { sum.v[s] = EFO_FMA(row[i], row[j], sum.v[s]); ++s; }
    sum.v[9] = EFO_FMA(row[3], row[3], sum.v[9]);
    sum.v[10] += found ? 1.0f : 0.0f;
  };
  Acc<float, 11> h = EFO_REDUCE<float, 11>(cols * rows, products);
  int shift = 0;  // reduce.cu:958-969
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 4; ++j) {
      float value = h.v[shift++];
      if (j == 3) b[i] = value;
      else A[j * 3 + i] = A[i * 3 + j] = value;
    }
  residual[0] = h.v[9];
  residual[1] = h.v[10];
}

}  // extern "C"
