// HIP kernels (gfx950, wave64) for depth pre-processing and surfel-map maintenance.
// What the reference does with GLSL draws (point rasterisation + GL_LESS depth test + transform
// feedback) is done here with a 64-bit atomic z-buffer in HBM — key = (orderable depth bits << 32) | id,
// resolved by atomicMin, so "nearest wins, ties go to the lower surfel index" == GL draw order — and
// with order-preserving chunked stream compaction instead of transform feedback.
// GL-defined behaviour is specified as N1-N5 in SURVEY.md §8a (restated in DESIGN.md).
#include "ef_device.hpp"
#include <stdlib.h>
#include <mutex>
#include <hip/hip_ext.h>
#include "ef_map.hpp"

using namespace ef;

namespace efm {

namespace {

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
constexpr int SURFEL_GRID = 1024;   // grid-stride workgroups for per-surfel passes (count lives on the device)
constexpr int BLK = 256;
constexpr int CLEAN_GRID = 4096;     // grid-stride workgroups of the clean passes (one 256-element row each)
static_assert(CLEAN_ROW == BLK, "clean kernels run one element per thread");
static_assert(CLEAN_GRID % 8 == 0, "xcd_row");

__device__ __forceinline__ uint32_t depth_key(float z) {  // order-preserving float -> uint
  const uint32_t b = __float_as_uint(z);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long zkey(float z, uint32_t id) { return ((unsigned long long)depth_key(z) << 32) | id; }
// XCD-aware work order (round 6).  Workgroup b of a launch runs on XCD b % 8 and every XCD has its own L2.  The per-surfel and per-pixel passes
// that read a NEIGHBOURHOOD of the index maps (clean's and fuse's taps) walk surfels / pixels in column-major order, so consecutive workgroups read
// overlapping lines; dealt round-robin to the XCDs every such line is fetched into three L2s (k_clean_flags moved 2.4 x its unique bytes,
// profiles/pmc_traffic.json).  These helpers give the workgroups of one XCD CONTIGUOUS stretches of the work instead — which workgroup computes
// which element changes, no result does.
//   xcd_block(): a bijection of [0, gridDim.x): the blocks of XCD x, in launch order, take the x-th eighth
__device__ __forceinline__ unsigned xcd_block() {
  const unsigned g = gridDim.x, x = blockIdx.x & 7u, slot = blockIdx.x >> 3;
  return x * (g >> 3) + (x < (g & 7u) ? x : (g & 7u)) + slot;
}
//   xcd_row(nrows, it, r): row of this block's it-th trip when the rows are dealt to the XCDs in CHUNKS of XCD_CHUNK consecutive rows (chunk c to
//   XCD c % 8; gridDim.x % 8 == 0): neighbouring rows — which read overlapping lines — share an L2, and the heavy and the light stretches of
//   the map (old stable surfels at the front, candidate slots at the end) still spread over all XCDs (one contiguous eighth per XCD, measured:
//   k_clean_flags 18.5 -> 22.4 us, 61 -> 108 us at 1280 x 960 — the XCDs that own the stable surfels finish last)
constexpr unsigned XCD_CHUNK = 32;
__device__ __forceinline__ bool xcd_row(unsigned nrows, unsigned it, unsigned& r) {
  const unsigned x = blockIdx.x & 7u, slot = blockIdx.x >> 3, per = gridDim.x >> 3;
  const unsigned local = slot + it * per;
  r = (x + 8u * (local / XCD_CHUNK)) * XCD_CHUNK + (local % XCD_CHUNK);
  return r < nrows;
}

// uv attribute (FeedbackBuffer.cpp:44-52, GlobalModel.cpp:109-117) and x = texcoord.x * cols
__device__ __forceinline__ float pix_coord(int i, int n) {
  const float u = (float)((double)((float)i / (float)n) + 1.0 / (double)(2 * (float)n));
  return u * (float)n;
}
// color.glsl:19-34
__device__ __forceinline__ float encodeColor(f3 c) {
  int rgb = (int)roundf(c.x * 255.0f);
  rgb = (rgb << 8) + (int)roundf(c.y * 255.0f);
  rgb = (rgb << 8) + (int)roundf(c.z * 255.0f);
  return (float)rgb;
}
__device__ __forceinline__ f3 decodeColor(float c) {
  const int ic = (int)c;
  return {(float)((ic >> 16) & 0xFF) / 255.0f, (float)((ic >> 8) & 0xFF) / 255.0f, (float)(ic & 0xFF) / 255.0f};
}
// surfels.glsl:19-34
__device__ __forceinline__ float getRadius(float depth, float norm_z, float inv_fx, float inv_fy) {
  const float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
  const float sqrt2 = 1.41421356237f;
  const float radius = (depth / meanFocal) * sqrt2;
  const float radius_n = radius / fabsf(norm_z);
  return fminf(2.0f * radius, radius_n);
}
// surfels.glsl:36-46
__device__ __forceinline__ float confidence(float x, float y, float cx, float cy, float weighting) {
  const float maxRadDist = 400, twoSigmaSquared = 0.72f;
  const float px = x - cx, py = y - cy;
  const float radialDist = sqrtf(px * px + py * py) / maxRadDist;
  return ef_expf(-(radialDist * radialDist) / twoSigmaSquared) * weighting;
}
struct DepthF {  // float depth image, NEAREST + CLAMP_TO_EDGE (N4)
  const float* d; int cols, rows;
  __device__ __forceinline__ float at(int x, int y) const { return d[clampi(y, 0, rows - 1) * cols + clampi(x, 0, cols - 1)]; }
};
// geometry.glsl:21-40
__device__ __forceinline__ f3 getVertexF(const DepthF& D, int ix, int iy, float x, float y, float cx, float cy, float inv_fx, float inv_fy) {
  const float z = D.at(ix, iy);
  return {(x - cx) * z * inv_fx, (y - cy) * z * inv_fy, z};
}
__device__ __forceinline__ f3 half_sum(f3 a, f3 b) { return {(a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2}; }
__device__ __forceinline__ f3 getNormalF(const DepthF& D, f3 vPosition, int ix, int iy, float x, float y, float cx, float cy,
                                         float inv_fx, float inv_fy) {
  const f3 xf = getVertexF(D, ix + 1, iy, x + 1, y, cx, cy, inv_fx, inv_fy);
  const f3 xb = getVertexF(D, ix - 1, iy, x - 1, y, cx, cy, inv_fx, inv_fy);
  const f3 yf = getVertexF(D, ix, iy + 1, x, y + 1, cx, cy, inv_fx, inv_fy);
  const f3 yb = getVertexF(D, ix, iy - 1, x, y - 1, cx, cy, inv_fx, inv_fy);
  const f3 del_x = half_sum(xb, vPosition) - half_sum(xf, vPosition);
  const f3 del_y = half_sum(yb, vPosition) - half_sum(yf, vPosition);
  return normalized(cross(del_x, del_y));
}

// exclusive scan of one value per thread over a 256-thread workgroup (wave64 shuffles + 4-entry LDS)
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned* lds, unsigned& total) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  unsigned x = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned y = __shfl_up(x, off, 64);
    if (lane >= off) x += y;
  }
  __syncthreads();
  if (lane == 63) lds[w] = x;
  __syncthreads();
  unsigned base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < BLK / 64; ++i) {
    const unsigned s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  total = tot;
  return base + x - v;
}

// ------------------------------------------------------------------------------------------
// pre-processing: depth_bilateral.frag:30-76 + depth_metric.frag:28-40
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t bilateral_px(const uint16_t* __restrict__ raw, int cols, int rows, int x, int y, unsigned maxv) {
  const unsigned value = raw[y * cols + x];
  if (value > maxv || value < 300U) return 0;
  const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 0.000555556f;
  const int R = 6, D = R * 2 + 1;
  const int tx = min(x - D / 2 + D, cols), ty = min(y - D / 2 + D, rows);
  float sum1 = 0, sum2 = 0;
  for (int cy = max(y - D / 2, 0); cy < ty; ++cy)
    for (int cx = max(x - D / 2, 0); cx < tx; ++cx) {
      const unsigned tmp = raw[cy * cols + cx];
      const float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) + ((float)y - (float)cy) * ((float)y - (float)cy);
      const float color2 = ((float)value - (float)tmp) * ((float)value - (float)tmp);
      const float weight = ef_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
      sum1 += (float)tmp * weight;
      sum2 += weight;
    }
  return (uint16_t)(unsigned)roundf(sum1 / sum2);
}
#include "ef_preprocess.inc"
__global__ void k_bilateral_table(float* __restrict__ table) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BIL_ROWS * BIL_COLS) return;
  const int r = i / BIL_COLS, k = i - r * BIL_COLS;
  const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 0.000555556f;
  const float space2 = (float)BIL.distinct[r];
  const float d = (float)k;
  const float color2 = d * d;
  table[i] = k < BIL_ZERO ? ef_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half)) : 0.0f;
}
template <bool WITH_METRIC>
__global__ void __launch_bounds__(256) k_preprocess(const uint16_t* __restrict__ raw, int cols, int rows, unsigned maxv, const float* __restrict__ table,
                                                     uint16_t* __restrict__ filtered, float* __restrict__ metric,
                                                     float* __restrict__ metric_filtered, const uint8_t* __restrict__ rgb3,
                                                     uint8_t* __restrict__ next0, uint8_t* __restrict__ rgb_keep) {
  preprocess_tile<WITH_METRIC>((int)blockIdx.x, (int)blockIdx.y, raw, cols, rows, maxv, table, filtered, metric, metric_filtered, rgb3, next0, rgb_keep);
}
__global__ void k_metricise(const uint16_t* __restrict__ in, int n, unsigned maxv, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = metric_px(in[i], maxv);
}

// ------------------------------------------------------------------------------------------
// layout conversion
// ------------------------------------------------------------------------------------------
__global__ void k_aos_to_soa(const float4* __restrict__ aos, uint32_t count, SurfelSoA soa) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  soa.pos_conf[i] = aos[(size_t)i * 3];
  soa.col_time[i] = aos[(size_t)i * 3 + 1];
  soa.nrm_rad[i] = aos[(size_t)i * 3 + 2];
}
__global__ void k_soa_to_aos(SurfelSoA soa, uint32_t count, float4* __restrict__ aos) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  aos[(size_t)i * 3] = soa.pos_conf[i];
  aos[(size_t)i * 3 + 1] = soa.col_time[i];
  aos[(size_t)i * 3 + 2] = soa.nrm_rad[i];
}

// ------------------------------------------------------------------------------------------
// chunked order-preserving compaction: flags (1 byte / element) -> per-chunk counts -> scan -> scatter
// ------------------------------------------------------------------------------------------
// n = *count_dev + extra (count_dev may be null => n = extra)
__device__ __forceinline__ unsigned dyn_n(const unsigned* count_dev, unsigned extra) { return (count_dev ? *count_dev : 0u) + extra; }

// count_out / capacity / overflow_flag: optional clamped copy of the total (the new surfel count of clean())
__global__ void __launch_bounds__(1024) k_scan_chunks(const uint32_t* __restrict__ counts, const unsigned* count_dev, unsigned extra,
                                                       uint32_t* __restrict__ offsets, uint32_t* total_out, unsigned* count_out = nullptr,
                                                       uint32_t capacity = 0, int* overflow_flag = nullptr, unsigned chunk = CHUNK) {
  __shared__ unsigned wsum[16];
  __shared__ unsigned carry_s;
  const unsigned n = dyn_n(count_dev, extra);
  const unsigned nchunks = (n + chunk - 1) / chunk;
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  if (t == 0) carry_s = 0;
  __syncthreads();
  for (unsigned base = 0; base < nchunks; base += 1024) {
    const unsigned i = base + t;
    const unsigned v = i < nchunks ? counts[i] : 0u;
    unsigned x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const unsigned y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    unsigned wb = 0, tot = 0;
    for (int k = 0; k < 16; ++k) {
      const unsigned s = wsum[k];
      if (k < w) wb += s;
      tot += s;
    }
    const unsigned carry = carry_s;
    if (i < nchunks) offsets[i] = carry + wb + x - v;
    __syncthreads();
    if (t == 0) carry_s = carry + tot;
    __syncthreads();
  }
  if (t == 0) {
    *total_out = carry_s;
    if (count_out) {
      unsigned tot = carry_s;
      if (tot > capacity) { tot = capacity; if (overflow_flag) *overflow_flag = 1; }
      *count_out = tot;
    }
  }
}

// ------------------------------------------------------------------------------------------
// first-frame seeding (G3)
// ------------------------------------------------------------------------------------------
struct SeedArgs {
  Cam cam;
  const uint8_t* rgb3;
  const float* dm;
  const float* dmf;
  int time;
  float maxDepth;
};
// element e in column-major pixel order (the uv buffer's order): i = e / rows, j = e % rows
__global__ void __launch_bounds__(BLK) k_seed_flags(const SeedArgs A, uint8_t* __restrict__ flags_raw, uint8_t* __restrict__ flags_filt,
                                                     uint32_t* __restrict__ cnt_raw, uint32_t* __restrict__ cnt_filt) {
  __shared__ unsigned lds[BLK / 64];
  const int P = A.cam.cols * A.cam.rows;
  const int c = blockIdx.x;
  unsigned nr = 0, nf = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = c * CHUNK + threadIdx.x * 4 + k;
    if (e < P) {
      const int i = e / A.cam.rows, j = e - i * A.cam.rows;
      const float zr = A.dm[j * A.cam.cols + i], zf = A.dmf[j * A.cam.cols + i];
      const uint8_t fr = !(zr <= 0 || zr > A.maxDepth), ff = !(zf <= 0 || zf > A.maxDepth);
      flags_raw[e] = fr; flags_filt[e] = ff;
      nr += fr; nf += ff;
    }
  }
  unsigned tot;
  block_excl_scan(nr, lds, tot);
  if (threadIdx.x == 0) cnt_raw[c] = tot;
  block_excl_scan(nf, lds, tot);
  if (threadIdx.x == 0) cnt_filt[c] = tot;
}
__global__ void __launch_bounds__(BLK) k_seed_scatter(const SeedArgs A, const uint8_t* __restrict__ flags_raw,
                                                       const uint8_t* __restrict__ flags_filt, const uint32_t* __restrict__ off_raw,
                                                       const uint32_t* __restrict__ off_filt, const uint32_t* __restrict__ total_raw,
                                                       SurfelSoA out, unsigned* count_dev) {
  __shared__ unsigned lds[BLK / 64];
  const Cam cam = A.cam;
  const int P = cam.cols * cam.rows;
  const int c = blockIdx.x;
  const float inv_fx = 1.0f / cam.fx, inv_fy = 1.0f / cam.fy;  // FeedbackBuffer.cpp:91-95
  uint8_t fr[4], ff[4];
  unsigned nr = 0, nf = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = c * CHUNK + threadIdx.x * 4 + k;
    fr[k] = e < P ? flags_raw[e] : 0;
    ff[k] = e < P ? flags_filt[e] : 0;
    nr += fr[k]; nf += ff[k];
  }
  unsigned tot;
  unsigned pr = off_raw[c] + block_excl_scan(nr, lds, tot);
  unsigned pf = off_filt[c] + block_excl_scan(nf, lds, tot);
  const unsigned rawTotal = *total_raw;
  if (c == 0 && threadIdx.x == 0) *count_dev = rawTotal;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = c * CHUNK + threadIdx.x * 4 + k;
    if (e >= P) break;
    if (!fr[k] && !ff[k]) continue;
    const int i = e / cam.rows, j = e - i * cam.rows;
    const float x = pix_coord(i, cam.cols), y = pix_coord(j, cam.rows);
    if (fr[k]) {
      const DepthF D{A.dm, cam.cols, cam.rows};
      const f3 v = getVertexF(D, i, j, x, y, cam.cx, cam.cy, inv_fx, inv_fy);
      const uint8_t* cc = A.rgb3 + (size_t)(j * cam.cols + i) * 3;
      const f3 col{(float)cc[0] / 255.0f, (float)cc[1] / 255.0f, (float)cc[2] / 255.0f};
      out.pos_conf[pr] = make_float4(v.x, v.y, v.z, confidence(x, y, cam.cx, cam.cy, 1.0f));
      out.col_time[pr] = make_float4(encodeColor(col), 0.f, 1.f, (float)A.time);  // init_unstable.vert:33-34
      ++pr;
    }
    if (ff[k]) {
      const DepthF D{A.dmf, cam.cols, cam.rows};
      const f3 v = getVertexF(D, i, j, x, y, cam.cx, cam.cy, inv_fx, inv_fy);
      const f3 n = getNormalF(D, v, i, j, x, y, cam.cx, cam.cy, inv_fx, inv_fy);
      if (pf < rawTotal) out.nrm_rad[pf] = make_float4(n.x, n.y, n.z, getRadius(v.z, n.z, inv_fx, inv_fy));
      ++pf;
    }
  }
}

// ------------------------------------------------------------------------------------------
// IndexMap::predictIndices (G4): 1-pixel splat -> resolve
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void merge_surfel(const Candidates& cand, int r, uint32_t id, const SurfelSoA& map, int time, float4& s, float4& sc);
// MERGE (round 6): the frame's SECOND predictIndices (ElasticFusion.cpp:545) comes right behind the update pass (GlobalModel::fuse's second half,
// update.vert) — whose work is per SURFEL too: surfel id is rewritten by the one candidate that won its association, winner[id].  The lane that
// is about to splat surfel id merges that candidate into it first (merge_surfel: k_merge's arithmetic, in place) and splats what it wrote: one
// launch less per frame (k_merge: 7.5 us for 96 bytes x matched candidates), no second read of the map's matched surfels.
template <bool MERGE>
__global__ void __launch_bounds__(BLK) k_index_splat(const Cam cam, const float* __restrict__ T16, int time, SurfelSoA map,
                                                      const unsigned* __restrict__ count_dev, float maxDepth, int timeDelta,
                                                      unsigned long long* zbuf, int colmajor, Candidates cand, const uint32_t* __restrict__ winner) {
  const rt34 T = rt34_load16(T16);
  const unsigned count = *count_dev;
  const float ftime = (float)time, ftd = (float)timeDelta;
  for (unsigned id = blockIdx.x * blockDim.x + threadIdx.x; id < count; id += gridDim.x * blockDim.x) {
    float4 pc = map.pos_conf[id];
    float4 ct = map.col_time[id];
    if (MERGE) {
      const uint32_t r = winner[id];
      if (r != WINNER_EMPTY) merge_surfel(cand, (int)r, id, map, time, pc, ct);
    }
    const f3 p = xform(T, f3{pc.x, pc.y, pc.z});
    if (p.z > maxDepth || p.z < 0 || ftime - ct.w > ftd) continue;
    const float u = ((cam.fx * p.x) / p.z) + cam.cx;
    const float v = ((cam.fy * p.y) / p.z) + cam.cy;
    // index_map.vert:52-53 hands the point over in NDC (float) and the viewport transform brings it back, evaluated exactly on the float
    // NDC value (N1; the oracle's efo_predict_indices has the same lines): bit for bit the compiled shader's pixel, also on a pixel edge
    const float xn = (u - (float)cam.cols * 0.5f) / ((float)cam.cols * 0.5f), yn = (v - (float)cam.rows * 0.5f) / ((float)cam.rows * 0.5f);
    const double xw = ((double)xn + 1.0) * 0.5 * (double)cam.cols, yw = ((double)yn + 1.0) * 0.5 * (double)cam.rows;
    if (!(xw >= 0 && xw < cam.cols && yw >= 0 && yw < cam.rows)) continue;  // N1
    const int px = (int)floor(xw), py = (int)floor(yw);
    atomicMin(&zbuf[colmajor ? px * cam.rows + py : py * cam.cols + px], zkey(p.z, id));   // N2
  }
}
__global__ void __launch_bounds__(BLK) k_index_resolve(const Cam cam, const float* __restrict__ T16, SurfelSoA map,
                                                        unsigned long long* zbuf, IndexMaps out) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= cam.cols * cam.rows) return;
  const unsigned long long key = zbuf[pi];
  // (out.color_time null — the frame's FIRST predictIndices, whose only reader is the association: it taps index, vertex + confidence and
  // normal + radius — : the colour / time stream is neither gathered nor written, a quarter of this launch's bytes)
  if (key == ZBUF_EMPTY) {
    out.index[pi] = 0u;
    out.vert_conf[pi] = make_float4(0, 0, 0, 0);
    if (out.color_time) out.color_time[pi] = make_float4(0, 0, 0, 0);
    out.norm_rad[pi] = make_float4(0, 0, 0, 0);
    return;
  }
  zbuf[pi] = ZBUF_EMPTY;  // leave the buffer clean for the next splat: no separate clear pass
  const uint32_t id = (uint32_t)key;
  const rt34 T = rt34_load16(T16);
  const float4 pc = map.pos_conf[id];
  const float4 nr = map.nrm_rad[id];
  const f3 p = xform(T, f3{pc.x, pc.y, pc.z});
  const f3 n = normalized(mul(T.R, f3{nr.x, nr.y, nr.z}));
  out.index[pi] = id;
  out.vert_conf[pi] = make_float4(p.x, p.y, p.z, pc.w);
  if (out.color_time) out.color_time[pi] = map.col_time[id];
  out.norm_rad[pi] = make_float4(n.x, n.y, n.z, nr.w);
}

// ------------------------------------------------------------------------------------------
// IndexMap::combinedPredict (G5): oriented-disc splat -> resolve (+ fill-in G7, + denseEnough samples G8)
// ------------------------------------------------------------------------------------------
struct Sprite {
  f3 p, n;
  float rad, conf;
  float u, v, hs;
  bool ok;
};
// splat.vert:53-87 for one surfel
__device__ __forceinline__ Sprite make_sprite(const Cam& cam, const rt34& T, float4 pc, float4 ct, float4 nr, float maxDepth,
                                              float confThreshold, float ftime, float fmaxTime, float ftd) {
  Sprite S;
  S.ok = false;
  S.p = xform(T, f3{pc.x, pc.y, pc.z});
  if (S.p.z > maxDepth || S.p.z < 0 || pc.w < confThreshold || ftime - ct.w > ftd || ct.w > fmaxTime) return S;
  S.n = normalized(mul(T.R, f3{nr.x, nr.y, nr.z}));
  S.rad = nr.w;
  S.conf = pc.w;
  const f3 t1 = normalized(f3{S.n.y - S.n.z, -S.n.x, S.n.x});
  const f3 x1{(t1.x * S.rad) * 1.41421356f, (t1.y * S.rad) * 1.41421356f, (t1.z * S.rad) * 1.41421356f};
  const f3 y1 = cross(S.n, x1);
  const float fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
  const f3 a = S.p + x1, b = S.p + y1, c = S.p - y1, d = S.p - x1;
  const float q1x = ((fx * a.x) / a.z) + cx, q1y = ((fy * a.y) / a.z) + cy;
  const float q2x = ((fx * b.x) / b.z) + cx, q2y = ((fy * b.y) / b.z) + cy;
  const float q3x = ((fx * c.x) / c.z) + cx, q3y = ((fy * c.y) / c.z) + cy;
  const float q4x = ((fx * d.x) / d.z) + cx, q4y = ((fy * d.y) / d.z) + cy;
  const float xmin = fminf(q1x, fminf(q2x, fminf(q3x, q4x))), xmax = fmaxf(q1x, fmaxf(q2x, fmaxf(q3x, q4x)));
  const float ymin = fminf(q1y, fminf(q2y, fminf(q3y, q4y))), ymax = fmaxf(q1y, fmaxf(q2y, fmaxf(q3y, q4y)));
  // fminf/fmaxf drop NaNs, so test the operands themselves: any NaN corner => degenerate sprite, skipped (spec)
  if (q1x != q1x || q2x != q2x || q3x != q3x || q4x != q4x || q1y != q1y || q2y != q2y || q3y != q3y || q4y != q4y) return S;
  float size = fmaxf(0.f, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin)));
  if (size != size) return S;
  size = fminf(fmaxf(size, 1.0f), 2047.0f);  // N3
  S.u = ((fx * S.p.x) / S.p.z) + cx;
  S.v = ((fy * S.p.y) / S.p.z) + cy;
  if (!(S.u >= 0 && S.u < (float)cam.cols && S.v >= 0 && S.v < (float)cam.rows)) return S;
  S.hs = size * 0.5f;
  S.ok = true;
  return S;
}
// combo_splat.frag:35-61 for one fragment; returns false when discarded
__device__ __forceinline__ f3 pixel_ray(const Cam& cam, int px, int py) {
  const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
  return normalized(f3{(fcx - cam.cx) / cam.fx, (fcy - cam.cy) / cam.fy, 1.0f});
}
// ... with the ray of the pixel and dot(S.p, S.n) given (the same operations on the same values wherever they were evaluated)
__device__ __forceinline__ bool sprite_fragment_ray(const f3& Sp, const f3& Sn, float rad, float psn, const f3& l, float& z) {
  const float k = psn / dot(l, Sn);
  const f3 cp{k * l.x, k * l.y, k * l.z};
  const f3 diff = cp - Sp;
  if (!(dot(diff, diff) <= rad * rad)) return false;
  z = cp.z;
  return true;
}
__device__ __forceinline__ bool sprite_fragment(const Cam& cam, const Sprite& S, int px, int py, float& z) {
  return sprite_fragment_ray(S.p, S.n, S.rad, dot(S.p, S.n), pixel_ray(cam, px, py), z);
}
// The ray of every pixel (column-major like the z-buffer: texel (x, y) at x * rows + y), built once per context: a fragment of the surface splat
// then loads 16 bytes instead of evaluating two divisions, a square root and a reciprocal (52 of its ~90 instructions; the splat is VALU-bound,
// round 6) — pixel_ray's own value, so the intersection is bit-identical.
__global__ void k_ray_table(const Cam cam, float4* __restrict__ rays) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cam.cols * cam.rows) return;
  const int px = i / cam.rows, py = i - px * cam.rows;
  const f3 l = pixel_ray(cam, px, py);
  rays[i] = make_float4(l.x, l.y, l.z, 0.f);
}
// SPLAT_LANES consecutive lanes share one surfel and take every SPLAT_LANES-th fragment of its sprite: sprite areas vary
// from 1 to dozens of pixels, and with one surfel per lane a wave waits for its largest sprite
#ifndef EF_SPLAT_LANES
#define EF_SPLAT_LANES 4
#endif
constexpr int SPLAT_LANES = EF_SPLAT_LANES;
// The splat sends every fragment to a 64-bit atomicMin on the z-buffer in HBM.  north_star's "LDS-tiled binning" was built in round 3
// (-DEF_SPLAT_TILED, python -m elasticfusion_amd.build --variant splattiled -DEF_SPLAT_TILED): a workgroup takes a CONTIGUOUS range of
// SPLAT_CHUNK surfel ids (surfels are created in column-major pixel order and move little, so a short id range falls into a small
// window of the image), measures that window (bounding box of its sprites, LDS min / max), resolves every fragment inside it with a
// 64-bit atomicMin on an LDS tile and sends ONE global atomicMin per touched pixel afterwards; fragments outside the tile go to HBM
// directly.  min is associative and commutative: the z-buffer is bit-identical (the whole -m gpu suite passes on that build,
// profiles/r03e_gpu_tests_tiled_splat.log).  MEASURED on the mature 640x480 map: 40.3 us against 33.3 us for the all-global version, 1449
// against 1472 frames/s (profiles/r03e_ab_tiled_vs_global_splat.log, profiles/r03e_tiled_splat_bench_kernel_stats.csv): three barriers, the
// tile's clear and sweep and the box atomics per 128 surfels cost more than the overdraw they keep out of HBM, whose 64-bit atomics on
// neighbouring pixels already coalesce in L2.  Not the default.
#ifndef EF_SPLAT_TILED
constexpr int SPLAT_GRID = SURFEL_GRID * SPLAT_LANES;
#else
constexpr int SPLAT_ROUNDS = 2, SPLAT_CHUNK = SPLAT_ROUNDS * (BLK / SPLAT_LANES), SPLAT_TILE = 4096;
constexpr int SPLAT_GRID = 4096;   // workgroups; each strides over chunks of SPLAT_CHUNK ids (the count lives on the device)
#endif
// Round 6: a fragment's ray comes from the context's table (LUT; the operator tier, which has no context, evaluates it) and dot(S.p, S.n) is
// taken out of the fragment loop.  (Measured and dropped in the same round: an early-z load before the atomic — 28.0 against 24.5 us, plain or
// agent-scope: the atomics return nothing, the load makes every fragment wait — and one surfel per lane for the per-surfel part with the quad
// working through its four sprites — 34-37 us: a quarter of the wavefronts, each with a four times longer serial chain;
// profiles/r06i_*, r06k_*.)
template <bool LUT>
__global__ void __launch_bounds__(BLK) k_surface_splat(const Cam cam, const float* __restrict__ T16, SurfelSoA map,
                                                        const unsigned* __restrict__ count_dev, float maxDepth, float confThreshold,
                                                        int time, int maxTime, int timeDelta, unsigned long long* zbuf,
                                                        unsigned* consumed_mark, unsigned consumed_value, const float4* __restrict__ rays) {
  // (host-pointer frames: every kernel that reads the frame's landing buffers precedes this launch in the stream — their ring slot is free)
  if (consumed_mark && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(consumed_mark, consumed_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const rt34 T = rt34_load16(T16);
  const unsigned count = *count_dev;
  const unsigned sub = threadIdx.x % SPLAT_LANES;
#ifndef EF_SPLAT_TILED
  const unsigned stride = gridDim.x * blockDim.x / SPLAT_LANES;
  for (unsigned id = (blockIdx.x * blockDim.x + threadIdx.x) / SPLAT_LANES; id < count; id += stride) {
    const float4 pc = map.pos_conf[id];
#ifdef EF_SPLAT_EARLY_LOADS   // (A/B build "splat_early": all three streams of a surfel in one round trip, whatever its confidence)
    const float4 ct = map.col_time[id];
    const float4 nr = map.nrm_rad[id];
    if (pc.w < confThreshold) continue;
#else
    if (pc.w < confThreshold) continue;  // unstable surfels (the bulk of a young map) never reach the normal stream
    const float4 ct = map.col_time[id];
    const float4 nr = map.nrm_rad[id];
#endif
    const Sprite S = make_sprite(cam, T, pc, ct, nr, maxDepth, confThreshold, (float)time, (float)maxTime, (float)timeDelta);
    if (!S.ok) continue;
    const int px0 = max(0, (int)ceilf(S.u - S.hs - 0.5f)), px1 = min(cam.cols - 1, (int)ceilf(S.u + S.hs - 0.5f) - 1);
    const int py0 = max(0, (int)ceilf(S.v - S.hs - 0.5f)), py1 = min(cam.rows - 1, (int)ceilf(S.v + S.hs - 0.5f) - 1);
    const int hgt = py1 - py0 + 1, nfrag = (px1 - px0 + 1) * hgt;
    const float psn = dot(S.p, S.n);
    for (int f = (int)sub; f < nfrag; f += SPLAT_LANES) {   // (the sprite's fragments column by column: the lanes of a surfel stay in one cache line)
      const int fx = f / hgt, py = py0 + (f - fx * hgt), px = px0 + fx;
      const int zi = px * cam.rows + py;   // column-major z-buffer: see k_surface_resolve
      f3 l;
      if (LUT) { const float4 r = rays[zi]; l = f3{r.x, r.y, r.z}; }
      else l = pixel_ray(cam, px, py);
      float z;
      if (!sprite_fragment_ray(S.p, S.n, S.rad, psn, l, z)) continue;
      if (z != z) continue;
      unsigned long long* cell = &zbuf[zi];
      const unsigned long long key = zkey(z, id);
#if defined(EF_SPLAT_EARLYZ) && EF_SPLAT_EARLYZ == 1
      // early z (A/B): a fragment that is not nearer than what the cell already shows cannot change it (keys only decrease: a stale value is safe)
      if (*(volatile unsigned long long*)cell <= key) continue;
#elif defined(EF_SPLAT_EARLYZ) && EF_SPLAT_EARLYZ == 2
      if (__hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= key) continue;
#endif
      atomicMin(cell, key);
    }
  }
#else
  (void)rays;
  __shared__ unsigned long long tile[SPLAT_TILE];
  __shared__ int box[4];   // min x, min y, max x, max y of the chunk's sprites
  const unsigned group = threadIdx.x / SPLAT_LANES;
  for (unsigned c0 = blockIdx.x * SPLAT_CHUNK; c0 < count; c0 += gridDim.x * SPLAT_CHUNK) {
    if (threadIdx.x == 0) { box[0] = box[1] = 0x7fffffff; box[2] = box[3] = -1; }
    __syncthreads();
    Sprite S[SPLAT_ROUNDS];
    int px0[SPLAT_ROUNDS], px1[SPLAT_ROUNDS], py0[SPLAT_ROUNDS], py1[SPLAT_ROUNDS];
#pragma unroll
    for (int r = 0; r < SPLAT_ROUNDS; ++r) {
      const unsigned id = c0 + r * (BLK / SPLAT_LANES) + group;
      S[r].ok = false;
      px0[r] = py0[r] = 0;
      px1[r] = py1[r] = -1;
      if (id < count) {
        const float4 pc = map.pos_conf[id];
        if (!(pc.w < confThreshold)) {   // unstable surfels (the bulk of a young map) never reach the normal stream
          const float4 ct = map.col_time[id];
          const float4 nr = map.nrm_rad[id];
          S[r] = make_sprite(cam, T, pc, ct, nr, maxDepth, confThreshold, (float)time, (float)maxTime, (float)timeDelta);
        }
      }
      if (S[r].ok) {
        px0[r] = max(0, (int)ceilf(S[r].u - S[r].hs - 0.5f)); px1[r] = min(cam.cols - 1, (int)ceilf(S[r].u + S[r].hs - 0.5f) - 1);
        py0[r] = max(0, (int)ceilf(S[r].v - S[r].hs - 0.5f)); py1[r] = min(cam.rows - 1, (int)ceilf(S[r].v + S[r].hs - 0.5f) - 1);
        if (sub == 0 && px1[r] >= px0[r] && py1[r] >= py0[r]) {
          atomicMin(&box[0], px0[r]); atomicMin(&box[1], py0[r]);
          atomicMax(&box[2], px1[r]); atomicMax(&box[3], py1[r]);
        }
      }
    }
    __syncthreads();
    const int bx0 = box[0], by0 = box[1];
    const int bw = box[2] - bx0 + 1;                       // <= 0: no sprite in this chunk
    const int th = bw > 0 ? min(box[3] - by0 + 1, SPLAT_TILE / bw) : 0;   // rows of the window that fit the tile (0: window wider than the tile)
    const int tn = bw > 0 ? bw * th : 0;
    for (int i = threadIdx.x; i < tn; i += BLK) tile[i] = ~0ull;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < SPLAT_ROUNDS; ++r) {
      if (!S[r].ok) continue;
      const unsigned id = c0 + r * (BLK / SPLAT_LANES) + group;
      const int w = px1[r] - px0[r] + 1, nfrag = w * (py1[r] - py0[r] + 1);
      for (int f = (int)sub; f < nfrag; f += SPLAT_LANES) {
        const int fy = f / w, px = px0[r] + (f - fy * w), py = py0[r] + fy;
        float z;
        if (!sprite_fragment(cam, S[r], px, py, z)) continue;
        if (z != z) continue;
        const int ly = py - by0;
        if (ly < th) atomicMin(&tile[ly * bw + (px - bx0)], zkey(z, id));
        else atomicMin(&zbuf[px * cam.rows + py], zkey(z, id));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < tn; i += BLK) {
      const unsigned long long k = tile[i];
      if (k != ~0ull) {
        const int ly = i / bw, lx = i - ly * bw;
        atomicMin(&zbuf[(bx0 + lx) * cam.rows + (by0 + ly)], k);
      }
    }
    __syncthreads();   // the tile and the box are re-used by the next chunk
  }
#endif
}

// geometry.glsl:44-60 on the filtered u16 depth (integer pixel coords, forward differences; quirk Q4)
__device__ __forceinline__ f3 fill_vertex_at(const uint16_t* __restrict__ d, const Cam& cam, int sx, int sy, int x, int y, float inv_fx,
                                             float inv_fy) {
  const float z = (float)d[clampi(sy, 0, cam.rows - 1) * cam.cols + clampi(sx, 0, cam.cols - 1)] / 1000.0f;
  return {((float)x - cam.cx) * z * inv_fx, ((float)y - cam.cy) * z * inv_fy, z};
}
__device__ __forceinline__ void fill_in_px(const Cam& cam, int x, int y, uchar4 si, float4 sv, float4 sn,
                                           const uint16_t* __restrict__ depth_filtered, const uint8_t* __restrict__ rgb3, bool passthrough,
                                           bool passthroughImage, FillMaps out) {
  const int pi = y * cam.cols + x;
  const float inv_fx = 1.0f / cam.fx, inv_fy = 1.0f / cam.fy;  // FillIn.cpp:115-119
  if (sv.z == 0 || passthrough) {
    const f3 v = fill_vertex_at(depth_filtered, cam, x, y, x, y, inv_fx, inv_fy);
    out.vertex[pi] = make_float4(v.x, v.y, v.z, 1.f);
  } else {
    out.vertex[pi] = sv;
  }
  if (sn.z == 0 || passthrough) {
    const f3 v = fill_vertex_at(depth_filtered, cam, x, y, x, y, inv_fx, inv_fy);
    const f3 vx = fill_vertex_at(depth_filtered, cam, x + 1, y, x + 1, y, inv_fx, inv_fy);
    const f3 vy = fill_vertex_at(depth_filtered, cam, x, y + 1, x, y + 1, inv_fx, inv_fy);
    const f3 nn = normalized(cross(vx - v, vy - v));
    out.normal[pi] = make_float4(nn.x, nn.y, nn.z, 1.f);
  } else {
    out.normal[pi] = sn;
  }
  if ((si.x == 0 && si.y == 0 && si.z == 0) || passthroughImage) {
    out.image[pi] = make_uchar4(rgb3[(size_t)pi * 3], rgb3[(size_t)pi * 3 + 1], rgb3[(size_t)pi * 3 + 2], 255);
  } else {
    out.image[pi] = si;
  }
}
__device__ __forceinline__ void dense_sample(const Cam& cam, int x, int y, uchar4 si, unsigned* counter) {
  // Resize::image: dest (a,b) <- source texel (20a+10, 20b+10), consSample = 20 (ElasticFusion.cpp:62-70)
  if (counter && x % 20 == 10 && y % 20 == 10 && x / 20 < cam.cols / 20 && y / 20 < cam.rows / 20)
    if (si.x > 0 && si.y > 0 && si.z > 0) atomicAdd(counter, 1u);
}

template <bool FUSE_FILL>
__global__ void __launch_bounds__(BLK) k_surface_resolve(const Cam cam, const float* __restrict__ T16, SurfelSoA map, float maxDepth,
                                                          float confThreshold, int time, int maxTime, int timeDelta,
                                                          unsigned long long* zbuf, PredictMaps out, FillMaps fill,
                                                          const uint16_t* __restrict__ depth_filtered, const uint8_t* __restrict__ rgb3,
                                                          bool passthroughImage, unsigned* dense_counter, unsigned* nonempty_flag,
                                                          unsigned nonempty_value) {
  // Round 5: the surface z-buffer is COLUMN-major (texel (x, y) at x * rows + y), like the index maps of the frame tier: surfels are
  // created in column-major pixel order and move little, so consecutive ids splat onto vertically adjacent pixels and the 64-bit atomics of a
  // wavefront fall into a few cache lines instead of one line per sprite row (k_surface_splat: 33 us on the mature map, sigma 15).  The resolve
  // walks 16 x 16 pixel tiles with every wavefront on an 8 x 8 block, consecutive lanes going DOWN a column: its z-buffer reads are eight full
  // 64-byte lines, its surfel gathers follow consecutive ids, and its row-major outputs still leave in 8-pixel runs.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = blockIdx.x * 16 + (wave & 1) * 8 + (lane >> 3), py = blockIdx.y * 16 + (wave >> 1) * 8 + (lane & 7);
  if (px >= cam.cols || py >= cam.rows) return;
  const int pi = py * cam.cols + px, zi = px * cam.rows + py;
  const unsigned long long key = zbuf[zi];
  uchar4 im = make_uchar4(0, 0, 0, 0);
  float4 vt = make_float4(0, 0, 0, 0), nm = make_float4(0, 0, 0, 0);
  uint16_t tm = 0;
  if (key != ZBUF_EMPTY) {
    if (nonempty_flag) *nonempty_flag = nonempty_value;   // "this view shows at least one surfel", stamped with the caller's value (no reset pass)
    zbuf[zi] = ZBUF_EMPTY;
    const uint32_t id = (uint32_t)key;
    const rt34 T = rt34_load16(T16);
    const float4 pc = map.pos_conf[id], ct = map.col_time[id], nr = map.nrm_rad[id];
    const Sprite S = make_sprite(cam, T, pc, ct, nr, maxDepth, confThreshold, (float)time, (float)maxTime, (float)timeDelta);
    float z = 0.f;
    sprite_fragment(cam, S, px, py, z);  // same operations as the splat => same bits
    const f3 col = decodeColor(ct.x);
    im = make_uchar4((uint8_t)roundf(col.x * 255.0f), (uint8_t)roundf(col.y * 255.0f), (uint8_t)roundf(col.z * 255.0f), 255);
    const float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;
    vt = make_float4((fcx - cam.cx) * z * (1.f / cam.fx), (fcy - cam.cy) * z * (1.f / cam.fy), z, pc.w);
    nm = make_float4(S.n.x, S.n.y, S.n.z, nr.w);
    tm = (uint16_t)(unsigned)ct.z;
  }
  out.image[pi] = im;
  out.vertex[pi] = vt;
  out.normal[pi] = nm;
  out.time[pi] = tm;
  if (FUSE_FILL) {
    fill_in_px(cam, px, py, im, vt, nm, depth_filtered, rgb3, false, passthroughImage, fill);
    dense_sample(cam, px, py, im, dense_counter);
  }
}
// IndexMap::synthesizeDepth (G6): depth_splat.frag's only output is the intersection depth the z-buffer key already holds
__device__ __forceinline__ float depth_of_key(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__global__ void __launch_bounds__(BLK) k_depth_resolve(int cols, int rows, unsigned long long* zbuf, float* __restrict__ depth) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= cols * rows) return;
  const int py = pi / cols, px = pi - py * cols, zi = px * rows + py;   // (column-major z-buffer: k_surface_resolve)
  const unsigned long long key = zbuf[zi];
  float z = 0.f;   // glClearColor(0, 0, 0, 0)
  if (key != ZBUF_EMPTY) {
    zbuf[zi] = ZBUF_EMPTY;
    z = depth_of_key((uint32_t)(key >> 32));
  }
  depth[pi] = z;
}
__global__ void __launch_bounds__(BLK) k_fill_in(const Cam cam, PredictMaps pred, const uint16_t* __restrict__ depth_filtered,
                                                  const uint8_t* __restrict__ rgb3, bool passthrough, bool passthroughImage, FillMaps out) {
  const int pi = blockIdx.x * blockDim.x + threadIdx.x;
  if (pi >= cam.cols * cam.rows) return;
  const int py = pi / cam.cols, px = pi - py * cam.cols;
  fill_in_px(cam, px, py, pred.image[pi], pred.vertex[pi], pred.normal[pi], depth_filtered, rgb3, passthrough, passthroughImage, out);
}
__global__ void k_dense_count(const Cam cam, const uchar4* __restrict__ image, unsigned* counter) {
  const int dc = cam.cols / 20, dr = cam.rows / 20;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dc * dr) return;
  const int b = i / dc, a = i - b * dc;
  const uchar4 t = image[(20 * b + 10) * cam.cols + (20 * a + 10)];
  if (t.x > 0 && t.y > 0 && t.z > 0) atomicAdd(counter, 1u);
}

// ------------------------------------------------------------------------------------------
// GlobalModel::fuse (G9 data pass, G10 update pass)
// ------------------------------------------------------------------------------------------
struct FuseArgs {
  Cam cam;
  const float* pose16;
  int time;
  const uint8_t* rgb3;
  const float* dm;
  const float* dmf;
  IndexMaps im;
  float maxDepth;
  const float* weighting;
};
// data.vert:76-193.  One thread per fused pixel (W/2 x H/2, parity-selected: quirk Q12), threads walk rows
// (coalesced taps); the candidate lands in slot r = column-major rank == the reference's draw order.
__global__ void __launch_bounds__(BLK) k_associate(const FuseArgs A, Candidates cand, uint32_t* winner, int colwalk) {
  const Cam cam = A.cam;
  const int qc = cam.cols / 2, qr = cam.rows / 2;
  const int q = (int)(xcd_block() * blockDim.x + threadIdx.x);
  if (q >= qc * qr) return;
  // colwalk: consecutive lanes go down a column (the order of the candidate slots and of a column-major index map);
  // otherwise along a row (the order of the depth and colour images)
  const int qy = colwalk ? q % qr : q / qc, qx = colwalk ? q / qr : q - (q / qc) * qc;
  const int par = A.time % 2;
  const int i = 2 * qx + par, j = 2 * qy + par;
  const int r = qx * qr + qy;
  float4 c_col = make_float4(0, 0, 0, 0);  // tag 0: nothing emitted
  if (i < cam.cols && j < cam.rows) {
    const float cx = cam.cx, cy = cam.cy;
    const float inv_fx = (float)(1.0 / (double)cam.fx), inv_fy = (float)(1.0 / (double)cam.fy);  // GlobalModel.cpp:397-398
    const DepthF DR{A.dm, cam.cols, cam.rows}, DF{A.dmf, cam.cols, cam.rows};
    const float x = pix_coord(i, cam.cols), y = pix_coord(j, cam.rows);
    const float ftime = (float)A.time;
    const f3 vPosLocal = getVertexF(DR, i, j, x, y, cx, cy, inv_fx, inv_fy);
    const bool sel = ((int)x % 2 == (int)ftime % 2 && (int)y % 2 == (int)ftime % 2);
    const bool nb = !(DR.at(i - 1, j) == 0 || DR.at(i, j - 1) == 0 || DR.at(i + 1, j) == 0 || DR.at(i, j + 1) == 0);
#ifndef EF_ASSOC_LATE_LOADS
    // Round 6: everything the pixel reads — the filtered depth's cross, its colour, the 27 words of its 9 index-map texels — has an address that
    // depends on (i, j) only, so it is asked for HERE, beside the raw depth, not behind the test on the raw depth (a second dependent round
    // trip for every pixel that passes; clamped addresses: a pixel that fails the test reads valid memory it does not use)
    const float zf_c = DF.at(i, j), zf_xf = DF.at(i + 1, j), zf_xb = DF.at(i - 1, j), zf_yf = DF.at(i, j + 1), zf_yb = DF.at(i, j - 1);
    const uint8_t* c = A.rgb3 + (size_t)(j * cam.cols + i) * 3;
    const uint8_t c0 = c[0], c1 = c[1], c2 = c[2];
    const float weighting = *A.weighting;
    uint32_t idx9[3][3];
    float4 vc9[3][3], nr9[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int tx = clampi(i + a - 1, 0, cam.cols - 1);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ty = clampi(j + b - 1, 0, cam.rows - 1);
        const int ti = im_texel(A.im, cam, tx, ty);
        idx9[a][b] = A.im.index[ti];
        vc9[a][b] = A.im.vert_conf[ti];
        nr9[a][b] = A.im.norm_rad[ti];
      }
    }
    const rt34 pose = rt34_load16(A.pose16);
#endif
    if (sel && nb && vPosLocal.z > 0 && vPosLocal.z <= A.maxDepth) {
#ifdef EF_ASSOC_LATE_LOADS
      const rt34 pose = rt34_load16(A.pose16);
#endif
      const f3 vPos = xform(pose, vPosLocal);
#ifndef EF_ASSOC_LATE_LOADS
      // getVertexF / getNormalF (geometry.glsl:21-40) on the values loaded above: the same expressions
      auto vtx = [&](float z, float xx, float yy) { return f3{(xx - cx) * z * inv_fx, (yy - cy) * z * inv_fy, z}; };
      const f3 vPosition_f = vtx(zf_c, x, y);
      const f3 col{(float)c0 / 255.0f, (float)c1 / 255.0f, (float)c2 / 255.0f};
      const f3 xf = vtx(zf_xf, x + 1, y), xb = vtx(zf_xb, x - 1, y), yf = vtx(zf_yf, x, y + 1), yb = vtx(zf_yb, x, y - 1);
      const f3 del_x = half_sum(xb, vPosition_f) - half_sum(xf, vPosition_f);
      const f3 del_y = half_sum(yb, vPosition_f) - half_sum(yf, vPosition_f);
      const f3 vNormLocal = normalized(cross(del_x, del_y));
#else
      const f3 vPosition_f = getVertexF(DF, i, j, x, y, cx, cy, inv_fx, inv_fy);
      const uint8_t* c = A.rgb3 + (size_t)(j * cam.cols + i) * 3;
      const f3 col{(float)c[0] / 255.0f, (float)c[1] / 255.0f, (float)c[2] / 255.0f};
      const f3 vNormLocal = getNormalF(DF, vPosition_f, i, j, x, y, cx, cy, inv_fx, inv_fy);
      const float weighting = *A.weighting;
#endif
      const f3 nW = mul(pose.R, vNormLocal);
      int counter = 0;
      uint32_t best = 0;
      float bestDist = 1000;
      const float xl = (x - cx) * inv_fx, yl = (y - cy) * inv_fy;
      const float lambda = sqrtf(xl * xl + yl * yl + 1);
      const f3 ray{xl, yl, 1};
      const float lenRay = sqrtf(dot(ray, ray));
      const float lenN = sqrtf(dot(vNormLocal, vNormLocal));
      // N4: the 16 taps {-1, 0, 0, +1}^2 touch 9 distinct texels.  Round 5: all 27 loads (index, vertex + confidence, normal + radius of the 9
      // texels) are issued up front, unconditionally — inside the conditionals of the tap loop they formed chains of up to 48 DEPENDENT
      // round trips (index -> vertex -> normal, tap after tap: the compiler cannot speculate a load across a branch); the 16 taps are then
      // evaluated on registers in the reference's order (a duplicate tap never changes `best`: dist < bestDist is strict)
#ifdef EF_ASSOC_LATE_LOADS
      uint32_t idx9[3][3];
      float4 vc9[3][3], nr9[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const int tx = clampi(i + a - 1, 0, cam.cols - 1);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
          const int ty = clampi(j + b - 1, 0, cam.rows - 1);
          const int ti = im_texel(A.im, cam, tx, ty);
          idx9[a][b] = A.im.index[ti];
          vc9[a][b] = A.im.vert_conf[ti];
          nr9[a][b] = A.im.norm_rad[ti];
        }
      }
#endif
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int a3 = a == 0 ? 0 : (a == 3 ? 2 : 1);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const int b3 = b == 0 ? 0 : (b == 3 ? 2 : 1);
          const uint32_t current = idx9[a3][b3];
          if (current > 0U) {
            const float4 vc = vc9[a3][b3];
            if (fabsf((vc.z * lambda) - (vPosLocal.z * lambda)) < 0.05f) {
              const f3 cr = cross(ray, f3{vc.x, vc.y, vc.z});
              const float dist = sqrtf(dot(cr, cr)) / lenRay;
              const float4 nr = nr9[a3][b3];
              const f3 nn{nr.x, nr.y, nr.z};
              const float cang = dot(nn, vNormLocal) / (sqrtf(dot(nn, nn)) * lenN);
              const bool angOk = (cang > 0.87758255f && cang <= 1.0f);  // abs(acos(c)) < 0.5, NaN-false
              if (dist < bestDist && (fabsf(nr.z) < 0.75f || angOk)) {
                counter++;
                bestDist = dist;
                best = current;
              }
            }
          }
        }
      }
      const float tag = counter > 0 ? -1.0f : -2.0f;
      cand.pos_conf[r] = make_float4(vPos.x, vPos.y, vPos.z, confidence(x, y, cx, cy, weighting));
      cand.nrm_rad[r] = make_float4(nW.x, nW.y, nW.z, getRadius(vPosition_f.z, vNormLocal.z, inv_fx, inv_fy));
      c_col = make_float4(encodeColor(col), 0.f, ftime, tag);
      cand.best[r] = best;
      if (counter > 0) atomicMin(&winner[best], (uint32_t)r);  // N5: first pixel in draw order owns the update texel
    }
  }
  cand.col_time[r] = c_col;
}
// update.vert:37-92, in place, only for the surfels that won an association
// update.vert:37-92 for ONE surfel: candidate r (which won the association) merged into surfel id, in place; s / sc: the surfel's position +
// confidence and colour + times as they stand afterwards
__device__ __forceinline__ void merge_surfel(const Candidates& cand, int r, uint32_t id, const SurfelSoA& map, int time, float4& s, float4& sc) {
  const float4 ucol = cand.col_time[r];
  const float4 u = cand.pos_conf[r], un = cand.nrm_rad[r];
  float4 sn = map.nrm_rad[id];
  const float c_k = s.w, a = u.w, ftime = (float)time;
  if (un.w < (1.0f + 0.5f) * sn.w) {
    s.x = ((c_k * s.x) + (a * u.x)) / (c_k + a);
    s.y = ((c_k * s.y) + (a * u.y)) / (c_k + a);
    s.z = ((c_k * s.z) + (a * u.z)) / (c_k + a);
    s.w = c_k + a;
    const f3 oldCol = decodeColor(sc.x), newCol = decodeColor(ucol.x);
    const f3 avg{((c_k * oldCol.x) + (a * newCol.x)) / (c_k + a), ((c_k * oldCol.y) + (a * newCol.y)) / (c_k + a),
                 ((c_k * oldCol.z) + (a * newCol.z)) / (c_k + a)};
    sc.x = encodeColor(avg);
    sc.w = ftime;
    const float nx = ((c_k * sn.x) + (a * un.x)) / (c_k + a), ny = ((c_k * sn.y) + (a * un.y)) / (c_k + a),
                nz = ((c_k * sn.z) + (a * un.z)) / (c_k + a), nw = ((c_k * sn.w) + (a * un.w)) / (c_k + a);
    const f3 nn = normalized(f3{nx, ny, nz});
    sn = make_float4(nn.x, nn.y, nn.z, nw);
    map.pos_conf[id] = s;
    map.col_time[id] = sc;
    map.nrm_rad[id] = sn;
  } else {
    s.w = c_k + a;
    sc.w = ftime;
    map.pos_conf[id] = s;
    map.col_time[id] = sc;
  }
}
__global__ void __launch_bounds__(BLK) k_merge(Candidates cand, const uint32_t* __restrict__ winner, SurfelSoA map, int time) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= cand.n) return;
  if (cand.col_time[r].w != -1.0f) return;
  const uint32_t id = cand.best[r];
  if (winner[id] != (uint32_t)r) return;
  float4 s = map.pos_conf[id], sc = map.col_time[id];
  merge_surfel(cand, r, id, map, time, s, sc);
}

// ------------------------------------------------------------------------------------------
// GlobalModel::clean (G11): keep-test -> flags, then stable compaction of [old surfels | candidates]
// ------------------------------------------------------------------------------------------
struct CleanArgs {
  Cam cam;
  const float* T16;
  int time;
  IndexMaps im;
  float confThreshold;
  int timeDelta;
};
// The 4 taps of one axis (N4: pixel offsets {-1,-.5,0,+.5} -> texel floor(x+off), clamped) hit at most 3 distinct
// texels (the offsets span 1.5, so the floors span <= 2, and clamping is monotone).  The keep-test only counts taps,
// so each distinct texel is fetched once and counted with its multiplicity: 9 gathers instead of 16, all issued
// before the first use.
struct Taps3 { int u[3]; int m[3]; };
__device__ __forceinline__ Taps3 dedupe_taps(float x, int hi) {
  int t[4];
#pragma unroll
  for (int a = 0; a < 4; ++a) t[a] = clampi((int)floorf(x + (-1.0f + 0.5f * a)), 0, hi);
  Taps3 r;
  r.u[0] = t[0];
  r.u[2] = t[3];
  r.u[1] = (t[1] != t[0] && t[1] != t[3]) ? t[1] : t[2];
  r.m[0] = 1 + (t[1] == t[0]) + (t[2] == t[0]) + (t[3] == t[0]);
  r.m[2] = (t[3] != t[0]) ? (1 + (t[0] == t[3]) + (t[1] == t[3]) + (t[2] == t[3])) : 0;
  r.m[1] = (r.u[1] != t[0] && r.u[1] != t[3]) ? ((t[1] == r.u[1]) + (t[2] == r.u[1])) : 0;
  return r;
}
// copy_unstable.vert:49-130 (nodes == 0); returns keep flag; ct.w tag -2 is rewritten to time by the caller
__device__ __forceinline__ bool clean_test(const CleanArgs& A, const rt34& T, float4 pc, float4 ct, float4 nr) {
  const Cam& cam = A.cam;
  const float ftime = (float)A.time, ftd = (float)A.timeDelta;
  int test = 1;
  const f3 localPos = xform(T, f3{pc.x, pc.y, pc.z});
  const float x = ((cam.fx * localPos.x) / localPos.z) + cam.cx;
  const float y = ((cam.fy * localPos.y) / localPos.z) + cam.cy;
  const f3 localNorm = normalized(mul(T.R, f3{nr.x, nr.y, nr.z}));
  int cnt = 0, zCount = 0;
  // (round 6) the taps only ever turn a 1 into a 0, and the two time rules below override them: an element they send away whatever the taps say
  // — a matched candidate (tag -1: most candidate slots of a mature map), an old unstable surfel — does not ask for its 27 texels
  const float tagTime = ct.w == -2 ? ftime : ct.w;
#ifdef EF_CLEAN_ALL_TAPS   // (A/B build "alltaps": rounds 1-5)
  const bool decided = false;
  (void)tagTime;
#else
  const bool decided = tagTime == -1 || ((ftime - tagTime) > 20 && pc.w < A.confThreshold);
#endif
  if (!decided && ftime - ct.w < ftd && localPos.z > 0 && x > 0 && y > 0 && x < (float)cam.cols && y < (float)cam.rows) {
    const Taps3 tx = dedupe_taps(x, cam.cols - 1), ty = dedupe_taps(y, cam.rows - 1);
    uint32_t idx[9];
    float4 vcs[9], c2s[9];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int ti = im_texel(A.im, cam, tx.u[a], ty.u[b]);
        idx[a * 3 + b] = A.im.index[ti];
        vcs[a * 3 + b] = A.im.vert_conf[ti];
        c2s[a * 3 + b] = A.im.color_time[ti];
      }
    const bool steep = fabsf(localNorm.z) > 0.85f;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const int w = tx.m[a] * ty.m[b];   // how many of the 16 taps land on this texel
        const float4 vc = vcs[a * 3 + b], c2 = c2s[a * 3 + b];
        if (w > 0 && idx[a * 3 + b] > 0U) {
          const float dx = vc.x - localPos.x, dy = vc.y - localPos.y;
          if (c2.z < ct.z && vc.w > A.confThreshold && vc.z > localPos.z && vc.z - localPos.z < 0.01f &&
              sqrtf(dx * dx + dy * dy) < nr.w * 1.4f)
            cnt += w;
          if (c2.w == ftime && vc.w > A.confThreshold && vc.z > localPos.z && vc.z - localPos.z > 0.01f && steep)
            zCount += w;
        }
      }
  }
  if (cnt > 8 || zCount > 4) test = 0;
  float lastTime = ct.w;
  if (lastTime == -2) lastTime = ftime;
  if (lastTime == -1 || ((ftime - lastTime) > 20 && pc.w < A.confThreshold)) test = 0;
  if (lastTime > 0 && ftime - lastTime > ftd) test = 1;
  return test != 0;
}
__device__ __forceinline__ bool load_element(const SurfelSoA& map, const Candidates& cand, unsigned count, unsigned e, float4& pc,
                                             float4& ct, float4& nr) {
  if (e < count) {
    pc = map.pos_conf[e]; ct = map.col_time[e]; nr = map.nrm_rad[e];
    return true;
  }
  const unsigned r = e - count;
  ct = cand.col_time[r];
  if (ct.w == 0.0f) return false;  // slot never emitted by the data pass
  pc = cand.pos_conf[r]; nr = cand.nrm_rad[r];
  return true;
}
// ---- deformation-graph application, copy_unstable.vert:128-322 ----
// graph: nodes x 16 floats sorted by time {position 3, rotation 9 column-major, translation 3, time}, the content of the
// reference's 1 x 16384 node texture (GlobalModel.cpp:540-546); texel index = float index, CLAMP_TO_EDGE, zeros past the end.
struct DeformArgs {
  const float* graph;
  int nodes;
  const float* depth;     // IndexMap::synthesizeDepth image (row-major), read when !isFern
  int isFern;
  float maxDepth;
};
__device__ __forceinline__ float node_tex(const DeformArgs& D, int idx) {
  idx = clampi(idx, 0, 16384 - 1);
  return idx < D.nodes * 16 ? D.graph[idx] : 0.0f;
}
__device__ __forceinline__ f3 node_pos(const DeformArgs& D, int j) { return f3{node_tex(D, j * 16), node_tex(D, j * 16 + 1), node_tex(D, j * 16 + 2)}; }
__device__ __forceinline__ float node_dist(const DeformArgs& D, f3 pos, int j) {
  const f3 d = pos - node_pos(D, j);
  return sqrtf(dot(d, d));
}
struct M3c { float m[3][3]; };   // column-major mat3, m[col][row]
__device__ __forceinline__ f3 mulc(const M3c& a, f3 v) {
  return {dot(f3{a.m[0][0], a.m[1][0], a.m[2][0]}, v), dot(f3{a.m[0][1], a.m[1][1], a.m[2][1]}, v), dot(f3{a.m[0][2], a.m[1][2], a.m[2][2]}, v)};
}
__device__ __forceinline__ M3c inverse3(const M3c& a) {   // adjugate x 1/det: GLSL leaves inverse() to the implementation, this formula is the specification
  const float (*m)[3] = a.m;
  const float c00 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c01 = m[2][1] * m[0][2] - m[0][1] * m[2][2], c02 = m[0][1] * m[1][2] - m[1][1] * m[0][2];
  const float det = (m[0][0] * c00 + m[1][0] * c01) + m[2][0] * c02;
  const float id = 1.0f / det;
  M3c r;
  r.m[0][0] = c00 * id; r.m[0][1] = c01 * id; r.m[0][2] = c02 * id;
  r.m[1][0] = (m[2][0] * m[1][2] - m[1][0] * m[2][2]) * id; r.m[1][1] = (m[0][0] * m[2][2] - m[2][0] * m[0][2]) * id; r.m[1][2] = (m[1][0] * m[0][2] - m[0][0] * m[1][2]) * id;
  r.m[2][0] = (m[1][0] * m[2][1] - m[2][0] * m[1][1]) * id; r.m[2][1] = (m[2][0] * m[0][1] - m[0][0] * m[2][1]) * id; r.m[2][2] = (m[0][0] * m[1][1] - m[1][0] * m[0][1]) * id;
  return r;
}
__device__ void deform_vertex(const CleanArgs& A, const DeformArgs& D, const rt34& T, float4& pc, float4& ct, float4& nr) {
  constexpr int k = 4, lookBack = 20;
  const int nodes = D.nodes;
  int nearNodes[lookBack];
  float nearDists[lookBack];
  for (int i = 0; i < lookBack; ++i) { nearNodes[i] = -1; nearDists[i] = 16777216.0f; }
  const int poseTime = (int)ct.z;
  int foundIndex = 0, imin = 0, imax = nodes - 1, imid = (imin + imax) / 2;
  while (imax >= imin) {
    imid = (imin + imax) / 2;
    const int nodeTime = (int)node_tex(D, imid * 16 + 15);
    if (nodeTime < poseTime) imin = imid + 1;
    else if (nodeTime > poseTime) imax = imid - 1;
    else break;
  }
  imin = min(imin, nodes - 1);
  const int nodeMin = (int)node_tex(D, imin * 16 + 15), nodeMid = (int)node_tex(D, imid * 16 + 15), nodeMax = (int)node_tex(D, imax * 16 + 15);
  if (abs(nodeMin - poseTime) <= abs(nodeMid - poseTime) && abs(nodeMin - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imin;
  else if (abs(nodeMid - poseTime) <= abs(nodeMin - poseTime) && abs(nodeMid - poseTime) <= abs(nodeMax - poseTime)) foundIndex = imid;
  else foundIndex = imax;
  if (foundIndex == nodes) foundIndex = nodes - 1;
  const f3 pos{pc.x, pc.y, pc.z};
  int nearNodeIndex = 0, distanceBack = 0;
  for (int j = foundIndex; j >= 0; --j) {
    nearNodes[nearNodeIndex] = j;
    nearDists[nearNodeIndex] = node_dist(D, pos, j);
    nearNodeIndex++;
    if (++distanceBack == lookBack / 2) break;
  }
  for (int j = foundIndex + 1; j < nodes; ++j) {
    nearNodes[nearNodeIndex] = j;
    nearDists[nearNodeIndex] = node_dist(D, pos, j);
    nearNodeIndex++;
    if (++distanceBack == lookBack) break;
  }
  for (int i = 0; i < lookBack - 1; ++i)     // the shader's exchange sort, tie behaviour included
    for (int j = i + 1; j < lookBack; ++j)
      if (nearDists[j] < nearDists[i]) {
        const float tf = nearDists[i]; nearDists[i] = nearDists[j]; nearDists[j] = tf;
        const int ti = nearNodes[i]; nearNodes[i] = nearNodes[j]; nearNodes[j] = ti;
      }
  const float dMax = nearDists[k];
  float nodeWeights[k];
  float weightSum = 0;
  for (int j = 0; j < k; ++j) {
    const float q = 1.0f - (node_dist(D, pos, nearNodes[j]) / dMax);
    nodeWeights[j] = q * q;
    weightSum += nodeWeights[j];
  }
  for (int j = 0; j < k; ++j) nodeWeights[j] /= weightSum;
  f3 newPos{0.f, 0.f, 0.f}, newNorm{0.f, 0.f, 0.f};
  const f3 nrm{nr.x, nr.y, nr.z};
  for (int i = 0; i < k; ++i) {
    const int n = nearNodes[i];
    const f3 g = node_pos(D, n);
    M3c R;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) R.m[c][r] = node_tex(D, n * 16 + 3 + c * 3 + r);
    const f3 t{node_tex(D, n * 16 + 12), node_tex(D, n * 16 + 13), node_tex(D, n * 16 + 14)};
    const f3 moved = (mulc(R, pos - g) + g) + t;
    newPos = newPos + f3{nodeWeights[i] * moved.x, nodeWeights[i] * moved.y, nodeWeights[i] * moved.z};
    const M3c Ri = inverse3(R);
    M3c Rit;
    for (int c = 0; c < 3; ++c)
      for (int r = 0; r < 3; ++r) Rit.m[c][r] = Ri.m[r][c];
    const f3 rn = mulc(Rit, nrm);
    newNorm = newNorm + f3{nodeWeights[i] * rn.x, nodeWeights[i] * rn.y, nodeWeights[i] * rn.z};
  }
  pc.x = newPos.x; pc.y = newPos.y; pc.z = newPos.z;
  const f3 nn = normalized(newNorm);
  nr.x = nn.x; nr.y = nn.y; nr.z = nn.z;
  if (pc.w > A.confThreshold && D.isFern == 0) {
    const Cam& cam = A.cam;
    const f3 lp = xform(T, f3{pc.x, pc.y, pc.z});
    const float x = ((cam.fx * lp.x) / lp.z) + cam.cx, y = ((cam.fy * lp.y) / lp.z) + cam.cy;
    if (lp.z > 0 && lp.z < D.maxDepth && x > 0 && y > 0 && x < (float)cam.cols && y < (float)cam.rows) {
      const float currentDepth = D.depth[clampi((int)floorf(y), 0, cam.rows - 1) * cam.cols + clampi((int)floorf(x), 0, cam.cols - 1)];   // N4
      if (currentDepth > 0.0f && lp.z < currentDepth + 0.1f) ct.w = (float)A.time;
    }
  }
}
// runs between the keep-test and the scatter: kept elements (flag 1) not initialised this frame are deformed IN PLACE in
// the source buffers (the old map / the candidate slots are dead after this clean()); a new point's -2 tag is resolved
// first, as the shader does before its deformation block
__global__ void __launch_bounds__(BLK) k_clean_deform(const CleanArgs A, const DeformArgs D, SurfelSoA map, const unsigned* __restrict__ count_dev,
                                                       Candidates cand, const uint8_t* __restrict__ flags) {
  const unsigned count = *count_dev;
  const unsigned n = count + (unsigned)cand.n;
  const rt34 T = rt34_load16(A.T16);
  for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    if (!flags[e]) continue;
    float4* ppc = e < count ? &map.pos_conf[e] : &cand.pos_conf[e - count];
    float4* pct = e < count ? &map.col_time[e] : &cand.col_time[e - count];
    float4* pnr = e < count ? &map.nrm_rad[e] : &cand.nrm_rad[e - count];
    float4 pc = *ppc, ct = *pct, nr = *pnr;
    if (ct.w == -2.0f) ct.w = (float)A.time;
    if (ct.z == (float)A.time) { *pct = ct; continue; }
    deform_vertex(A, D, T, pc, ct, nr);
    *ppc = pc; *pct = ct; *pnr = nr;
  }
}

// one element per thread, one CLEAN_ROW-element compaction chunk per workgroup iteration: everything a row needs
// is in flight at once (the element count is device-resident, hence the grid-stride over rows)
// gsum_now / gsum_zero (CompactScratch::group_sum, or null): the row's count is also added to its group's sum, and the other half is cleared
__global__ void __launch_bounds__(BLK) k_clean_flags(const CleanArgs A, SurfelSoA map, const unsigned* __restrict__ count_dev,
                                                      Candidates cand, uint32_t* winner, uint8_t* __restrict__ flags,
                                                      uint32_t* __restrict__ chunk_count, uint32_t* gsum_now, uint32_t* __restrict__ gsum_zero,
                                                      int max_groups) {
  __shared__ unsigned lds[BLK / 64];
  const unsigned count = *count_dev;
  const unsigned n = count + (unsigned)cand.n;
  const unsigned nrows = (n + CLEAN_ROW - 1) / CLEAN_ROW;
  const rt34 T = rt34_load16(A.T16);
  if (gsum_zero)
    for (int i = blockIdx.x * BLK + threadIdx.x; i < max_groups; i += gridDim.x * BLK) gsum_zero[(size_t)i * CLEAN_GSTRIDE] = 0u;
  unsigned r;
  for (unsigned it = 0; xcd_row(nrows, it, r); ++it) {   // (uniform per workgroup)
    const unsigned e = r * CLEAN_ROW + threadIdx.x;
    bool f = false;
    if (e < n) {
      float4 pc, ct, nr;
      if (load_element(map, cand, count, e, pc, ct, nr)) f = clean_test(A, T, pc, ct, nr);
      if (e < count) winner[e] = WINNER_EMPTY;  // re-arm the association winners for the next frame
      flags[e] = f ? 1 : 0;
    }
    const unsigned wave_keep = (unsigned)__popcll(__ballot(f));
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = wave_keep;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned tot = 0;
#pragma unroll
      for (int i = 0; i < BLK / 64; ++i) tot += lds[i];
      chunk_count[r] = tot;
      if (gsum_now && tot) atomicAdd(&gsum_now[(size_t)(r / CLEAN_GROUP) * CLEAN_GSTRIDE], tot);
    }
    __syncthreads();
  }
}
// block_excl_scan of v together with the workgroup's sum of a second value (same two barriers)
__device__ __forceinline__ unsigned block_excl_scan_and_sum(unsigned v, unsigned a, unsigned* lds, unsigned& total, unsigned& asum) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  unsigned x = v, y = a;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned u = __shfl_up(x, off, 64);
    if (lane >= off) x += u;
    y += __shfl_xor(y, off, 64);
  }
  __syncthreads();
  if (lane == 63) { lds[w] = x; lds[BLK / 64 + w] = y; }
  __syncthreads();
  unsigned base = 0, tot = 0, as = 0;
#pragma unroll
  for (int i = 0; i < BLK / 64; ++i) {
    const unsigned s = lds[i];
    if (i < w) base += s;
    tot += s;
    as += lds[BLK / 64 + i];
  }
  total = tot;
  asum = as;
  return base + x - v;
}
// gsum given (CompactScratch::group_sum): no scan launch in front of this one — the workgroup adds up the groups and the rows in front of its row
// itself (<= max_groups + CLEAN_GROUP - 1 words, one or two loads per thread, in flight with the row's elements), and the workgroup of the
// LAST row leaves the totals k_scan_chunks left (total_out, the clamped count_out, the overflow flag)
__global__ void __launch_bounds__(BLK) k_clean_scatter(SurfelSoA map, const unsigned* __restrict__ count_dev, Candidates cand,
                                                        const uint8_t* __restrict__ flags, const uint32_t* __restrict__ chunk_offset,
                                                        int time, SurfelSoA out, uint32_t capacity, const uint32_t* __restrict__ gsum,
                                                        const uint32_t* __restrict__ chunk_count, uint32_t* total_out, unsigned* count_out,
                                                        int* overflow_flag) {
  __shared__ unsigned lds[2 * BLK / 64];
  const unsigned count = *count_dev;
  const unsigned n = count + (unsigned)cand.n;
  const unsigned nrows = (n + CLEAN_ROW - 1) / CLEAN_ROW;
  if (gsum && nrows == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    *total_out = 0u;
    if (count_out) *count_out = 0u;
  }
  unsigned r;
  for (unsigned it = 0; xcd_row(nrows, it, r); ++it) {   // (uniform per workgroup)
    const unsigned e = r * CLEAN_ROW + threadIdx.x;
    const unsigned f = (e < n) ? flags[e] : 0u;
    unsigned before = 0;
    if (gsum) {
      const unsigned g = r / CLEAN_GROUP;
      for (unsigned i = threadIdx.x; i < g; i += BLK) before += gsum[(size_t)i * CLEAN_GSTRIDE];
      for (unsigned i = g * CLEAN_GROUP + threadIdx.x; i < r; i += BLK) before += chunk_count[i];
    }
    float4 pc = make_float4(0, 0, 0, 0), ct = pc, nr = pc;
    if (f) load_element(map, cand, count, e, pc, ct, nr);
    unsigned tot, row0;
    unsigned pos = block_excl_scan_and_sum(f, before, lds, tot, row0);
    if (!gsum) row0 = chunk_offset[r];
    pos += row0;
    if (f && pos < capacity) {
      if (ct.w == -2.0f) ct.w = (float)time;  // copy_unstable.vert:114-117
      out.pos_conf[pos] = pc;
      out.col_time[pos] = ct;
      out.nrm_rad[pos] = nr;
    }
    if (gsum && r == nrows - 1 && threadIdx.x == 0) {
      unsigned all = row0 + tot;
      *total_out = all;
      if (count_out) {
        if (all > capacity) { all = capacity; if (overflow_flag) *overflow_flag = 1; }
        *count_out = all;
      }
    }
    __syncthreads();
  }
}

// candidates (tag != 0) -> AoS list in draw order
__global__ void __launch_bounds__(BLK) k_cand_flags(Candidates cand, uint8_t* __restrict__ flags, uint32_t* __restrict__ chunk_count) {
  __shared__ unsigned lds[BLK / 64];
  const unsigned n = cand.n, c = blockIdx.x;
  unsigned keep = 0;
  for (int k = 0; k < 4; ++k) {
    const unsigned e = c * CHUNK + k * BLK + threadIdx.x;
    if (e < n) {
      const uint8_t f = cand.col_time[e].w != 0.0f;
      flags[e] = f;
      keep += f;
    }
  }
  unsigned tot;
  block_excl_scan(keep, lds, tot);
  if (threadIdx.x == 0) chunk_count[c] = tot;
}
__global__ void __launch_bounds__(BLK) k_cand_scatter(Candidates cand, const uint8_t* __restrict__ flags,
                                                       const uint32_t* __restrict__ chunk_offset, float4* __restrict__ aos) {
  __shared__ unsigned lds[BLK / 64];
  const unsigned n = cand.n, c = blockIdx.x;
  unsigned base = chunk_offset[c];
  for (int k = 0; k < 4; ++k) {
    const unsigned e = c * CHUNK + k * BLK + threadIdx.x;
    const unsigned f = (e < n) ? flags[e] : 0u;
    unsigned tot;
    const unsigned pos = base + block_excl_scan(f, lds, tot);
    if (f) {
      aos[(size_t)pos * 3] = cand.pos_conf[e];
      aos[(size_t)pos * 3 + 1] = cand.col_time[e];
      aos[(size_t)pos * 3 + 2] = cand.nrm_rad[e];
    }
    base += tot;
    __syncthreads();
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
static inline dim3 tgrid(int cols, int rows) { return dim3(ceil_div(cols, 64), ceil_div(rows, 4)); }

// The bilateral filter's weight table (k_bilateral_table): one 43 KB buffer per device for the life of the process, built on first use
// (ef_create asks for it, so the first use is never inside a stream capture) and complete before the call returns.
const float* bilateral_table() {
  static std::mutex mu;
  static float* tables[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!tables[dev]) {
    float* p = nullptr;
    if (hipMalloc(&p, sizeof(float) * BIL_ROWS * BIL_COLS) != hipSuccess) return nullptr;
    hipLaunchKernelGGL(k_bilateral_table, dim3(ceil_div(BIL_ROWS * BIL_COLS, 256)), dim3(256), 0, (hipStream_t)0, p);
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipFree(p); return nullptr; }
    tables[dev] = p;
  }
  return tables[dev];
}
bool filter_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered, hipStream_t s) {
  const float* table = bilateral_table();
  if (!table) return false;   // (device ordinal >= 64, allocation or launch failure: an error to the caller, not a null dereference on the device)
  hipLaunchKernelGGL(k_preprocess<false>, dim3(ceil_div(cols, PRE_TW), ceil_div(rows, PRE_TH)), dim3(64, 4), 0, s, raw, cols, rows, (unsigned)(maxD * 1000.0f), table,
                     filtered, (float*)nullptr, (float*)nullptr, (const uint8_t*)nullptr, (uint8_t*)nullptr, (uint8_t*)nullptr);
  return true;
}
void metricise_depth(const uint16_t* in, int cols, int rows, float maxD, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_metricise, dim3(ceil_div(cols * rows, 256)), dim3(256), 0, s, in, cols * rows, (unsigned)(maxD * 1000.0f), out);
}
bool preprocess_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered, float* metric, float* metric_filtered,
                      hipStream_t s, unsigned extra_lds, const uint8_t* rgb3, uint8_t* next0, uint8_t* rgb_keep, const float* table) {
  if (!table) table = bilateral_table();   // (a context hands in the pointer it asked for at ef_create: no process-wide lock per frame)
  if (!table) return false;
  // static + dynamic LDS of a launch that never asked for more than the default 64 KB: beyond it the launch would fail silently
  constexpr unsigned STATIC_LDS = sizeof(float) * (BIL_ROWS * BIL_COLS + PRE_LH * PRE_LW);
  static_assert(STATIC_LDS < 65536u, "k_preprocess fits the default LDS limit");
  if (extra_lds > 65536u - STATIC_LDS) extra_lds = 65536u - STATIC_LDS;
  hipLaunchKernelGGL(k_preprocess<true>, dim3(ceil_div(cols, PRE_TW), ceil_div(rows, PRE_TH)), dim3(64, 4), extra_lds, s, raw, cols, rows, (unsigned)(maxD * 1000.0f), table,
                     filtered, metric, metric_filtered, rgb3, next0, rgb_keep);
  return true;
}
namespace {
__global__ void k_copy_map(SurfelSoA src, const unsigned* __restrict__ count_dev, SurfelSoA dst) {
  const unsigned n = *count_dev;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    dst.pos_conf[i] = src.pos_conf[i];
    dst.col_time[i] = src.col_time[i];
    dst.nrm_rad[i] = src.nrm_rad[i];
  }
}
}  // namespace
void copy_map(SurfelSoA src, const unsigned* count_dev, SurfelSoA dst, hipStream_t s) {
  hipLaunchKernelGGL(k_copy_map, dim3(SURFEL_GRID), dim3(BLK), 0, s, src, count_dev, dst);
}
void aos_to_soa(const float* aos, uint32_t count, SurfelSoA soa, hipStream_t s) {
  if (count) hipLaunchKernelGGL(k_aos_to_soa, dim3(ceil_div((int)count, 256)), dim3(256), 0, s, (const float4*)aos, count, soa);
}
void soa_to_aos(SurfelSoA soa, uint32_t count, float* aos, hipStream_t s) {
  if (count) hipLaunchKernelGGL(k_soa_to_aos, dim3(ceil_div((int)count, 256)), dim3(256), 0, s, soa, count, (float4*)aos);
}

void seed_map(const Cam& cam, const uint8_t* rgb3, const float* dm, const float* dmf, int time, float maxDepth, SurfelSoA out,
              unsigned* count_dev, const CompactScratch& cs, hipStream_t s) {
  const int P = cam.cols * cam.rows;
  const int nch = ceil_div(P, CHUNK);
  SeedArgs A{cam, rgb3, dm, dmf, time, maxDepth};
  uint8_t* fr = cs.flags;
  uint8_t* ff = cs.flags + P;
  uint32_t* cr = cs.chunk_count;
  uint32_t* cf = cs.chunk_count + nch;
  uint32_t* orw = cs.chunk_offset;
  uint32_t* ofl = cs.chunk_offset + nch;
  (void)hipMemsetAsync(out.nrm_rad, 0, (size_t)P * sizeof(float4), s);  // "stale zeros" past the end of the filtered stream
  hipLaunchKernelGGL(k_seed_flags, dim3(nch), dim3(BLK), 0, s, A, fr, ff, cr, cf);
  hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, (const uint32_t*)cr, (const unsigned*)nullptr, (unsigned)P, orw, cs.totals);
  hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, (const uint32_t*)cf, (const unsigned*)nullptr, (unsigned)P, ofl, cs.totals + 1);
  hipLaunchKernelGGL(k_seed_scatter, dim3(nch), dim3(BLK), 0, s, A, (const uint8_t*)fr, (const uint8_t*)ff, (const uint32_t*)orw,
                     (const uint32_t*)ofl, (const uint32_t*)cs.totals, out, count_dev);
}

void predict_indices(const Cam& cam, const float* T_cw16_dev, int time, SurfelSoA map, const unsigned* count_dev, float maxDepth,
                     int timeDelta, unsigned long long* zbuf, IndexMaps out, hipStream_t s, eft::KernelProbe* probe, const Candidates* merge_cand,
                     const uint32_t* merge_winner) {
  if (merge_cand) {   // fuse(..., defer_merge = true) in front of this call: the update pass rides on the splat
    hipLaunchKernelGGL(k_index_splat<true>, dim3(SURFEL_GRID), dim3(BLK), 0, s, cam, T_cw16_dev, time, map, count_dev, maxDepth, timeDelta, zbuf,
                       out.colmajor, *merge_cand, merge_winner);
    hipLaunchKernelGGL(k_index_resolve, dim3(ceil_div(cam.cols * cam.rows, BLK)), dim3(BLK), 0, s, cam, T_cw16_dev, map, zbuf, out);
    return;
  }
  // the probe's events receive the kernel's own begin / end timestamps (what rocprofv3 --kernel-trace reports as its duration)
  const bool sample = probe && probe->used < probe->capacity;
  hipEvent_t e0 = sample ? probe->start[probe->used] : nullptr, e1 = sample ? probe->stop[probe->used] : nullptr;
  if (sample) probe->used++;
  hipExtLaunchKernelGGL(k_index_splat<false>, dim3(SURFEL_GRID), dim3(BLK), 0, s, e0, e1, 0, cam, T_cw16_dev, time, map, count_dev, maxDepth, timeDelta,
                        zbuf, out.colmajor, Candidates{}, (const uint32_t*)nullptr);
  hipLaunchKernelGGL(k_index_resolve, dim3(ceil_div(cam.cols * cam.rows, BLK)), dim3(BLK), 0, s, cam, T_cw16_dev, map, zbuf, out);
}

void combined_predict(const Cam& cam, const float* T_cw16_dev, SurfelSoA map, const unsigned* count_dev, float maxDepth,
                      float confThreshold, int time, int maxTime, int timeDelta, unsigned long long* zbuf, PredictMaps out, FillMaps fill,
                      const uint16_t* depth_filtered, const uint8_t* rgb3, bool passthroughImage, unsigned* dense_counter, hipStream_t s,
                      unsigned* nonempty_flag, unsigned nonempty_value, unsigned* consumed_mark, unsigned consumed_value, const float* rays4) {
#ifdef EF_SPLAT_NO_LUT
  rays4 = nullptr;   // (A/B: every fragment evaluates its ray)
#endif
  if (rays4)
    hipLaunchKernelGGL(k_surface_splat<true>, dim3(SPLAT_GRID), dim3(BLK), 0, s, cam, T_cw16_dev, map, count_dev, maxDepth, confThreshold,
                       time, maxTime, timeDelta, zbuf, consumed_mark, consumed_value, (const float4*)rays4);
  else
    hipLaunchKernelGGL(k_surface_splat<false>, dim3(SPLAT_GRID), dim3(BLK), 0, s, cam, T_cw16_dev, map, count_dev, maxDepth, confThreshold,
                       time, maxTime, timeDelta, zbuf, consumed_mark, consumed_value, (const float4*)nullptr);
  const dim3 g(ceil_div(cam.cols, 16), ceil_div(cam.rows, 16));   // 16 x 16 pixel tiles, 8 x 8 per wavefront
  if (fill.image)
    hipLaunchKernelGGL(k_surface_resolve<true>, g, dim3(BLK), 0, s, cam, T_cw16_dev, map, maxDepth, confThreshold, time, maxTime, timeDelta,
                       zbuf, out, fill, depth_filtered, rgb3, passthroughImage, dense_counter, nonempty_flag, nonempty_value);
  else
    hipLaunchKernelGGL(k_surface_resolve<false>, g, dim3(BLK), 0, s, cam, T_cw16_dev, map, maxDepth, confThreshold, time, maxTime, timeDelta,
                       zbuf, out, fill, depth_filtered, rgb3, passthroughImage, dense_counter, nonempty_flag, nonempty_value);
}
void synthesize_depth(const Cam& cam, const float* T_cw16_dev, SurfelSoA map, const unsigned* count_dev, float maxDepth, float confThreshold,
                      int time, int maxTime, int timeDelta, unsigned long long* zbuf, float* depth, hipStream_t s, const float* rays4) {
  if (rays4)
    hipLaunchKernelGGL(k_surface_splat<true>, dim3(SPLAT_GRID), dim3(BLK), 0, s, cam, T_cw16_dev, map, count_dev, maxDepth, confThreshold,
                       time, maxTime, timeDelta, zbuf, (unsigned*)nullptr, 0u, (const float4*)rays4);
  else
    hipLaunchKernelGGL(k_surface_splat<false>, dim3(SPLAT_GRID), dim3(BLK), 0, s, cam, T_cw16_dev, map, count_dev, maxDepth, confThreshold,
                       time, maxTime, timeDelta, zbuf, (unsigned*)nullptr, 0u, (const float4*)nullptr);
  const int n = cam.cols * cam.rows;
  hipLaunchKernelGGL(k_depth_resolve, dim3(ceil_div(n, BLK)), dim3(BLK), 0, s, cam.cols, cam.rows, zbuf, depth);
}
void build_ray_table(const Cam& cam, float* rays4, hipStream_t s) {
  hipLaunchKernelGGL(k_ray_table, dim3(ceil_div(cam.cols * cam.rows, BLK)), dim3(BLK), 0, s, cam, (float4*)rays4);
}
void fill_in(const Cam& cam, PredictMaps pred, const uint16_t* depth_filtered, const uint8_t* rgb3, bool passthrough, bool passthroughImage,
             FillMaps out, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_in, dim3(ceil_div(cam.cols * cam.rows, BLK)), dim3(BLK), 0, s, cam, pred, depth_filtered, rgb3, passthrough,
                     passthroughImage, out);
}
void dense_count(const Cam& cam, const uchar4* image, unsigned* counter, hipStream_t s) {
  const int n = (cam.cols / 20) * (cam.rows / 20);
  hipLaunchKernelGGL(k_dense_count, dim3(ceil_div(n, 256)), dim3(256), 0, s, cam, image, counter);
}

void fuse(const Cam& cam, const float* pose_f16_dev, int time, const uint8_t* rgb3, const float* dm, const float* dmf, IndexMaps im,
          float maxDepth, const float* weighting_dev, SurfelSoA map, const unsigned* count_dev, Candidates cand, uint32_t* winner,
          hipStream_t s, bool defer_merge) {
  (void)count_dev;
  FuseArgs A{cam, pose_f16_dev, time, rgb3, dm, dmf, im, maxDepth, weighting_dev};
  // column-major index maps: walking columns measured 24.7 us vs 34.8 us for walking rows (profiles/, round 1)
  hipLaunchKernelGGL(k_associate, dim3(ceil_div(cand.n, BLK)), dim3(BLK), 0, s, A, cand, winner, im.colmajor ? 1 : 0);
  // (defer_merge: the caller's next launch is predict_indices(..., &cand, winner), whose splat merges every surfel before it projects it)
  if (!defer_merge) hipLaunchKernelGGL(k_merge, dim3(ceil_div(cand.n, BLK)), dim3(BLK), 0, s, cand, (const uint32_t*)winner, map, time);
}

void clean(const Cam& cam, const float* T_cw16_dev, int time, IndexMaps im, float confThreshold, int timeDelta, SurfelSoA map,
           const unsigned* count_dev, Candidates cand, uint32_t* winner, SurfelSoA out, unsigned* count_out_dev, uint32_t capacity,
           const CompactScratch& cs, int* overflow_flag, hipStream_t s, const Deformation* deform) {
  CleanArgs A{cam, T_cw16_dev, time, im, confThreshold, timeDelta};
#ifdef EF_SEPARATE_SCAN   // (A/B build "sepscan": rounds 1-5's three launches)
  const bool fold = false;
#else
  const bool fold = cs.group_sum != nullptr;
#endif
  uint32_t* const gnow = fold ? cs.group_sum + (size_t)(cs.flip & 1) * cs.max_groups * CLEAN_GSTRIDE : nullptr;
  uint32_t* const gzero = fold ? cs.group_sum + (size_t)((cs.flip & 1) ^ 1) * cs.max_groups * CLEAN_GSTRIDE : nullptr;
  hipLaunchKernelGGL(k_clean_flags, dim3(CLEAN_GRID), dim3(BLK), 0, s, A, map, count_dev, cand, winner, cs.flags,
                     cs.chunk_count, gnow, gzero, cs.max_groups);
  if (deform && deform->nodes > 0) {
    const DeformArgs D{deform->graph_dev, deform->nodes, deform->depth_dev, deform->is_fern, deform->max_depth};
    hipLaunchKernelGGL(k_clean_deform, dim3(CLEAN_GRID), dim3(BLK), 0, s, A, D, map, count_dev, cand, (const uint8_t*)cs.flags);
  }
  // (the scan of the rows' counts: with CompactScratch::group_sum every workgroup of the scatter finds its own row's offset — one launch of one
  // workgroup less on the frame's chain)
  if (!fold)
    hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, (const uint32_t*)cs.chunk_count, (const unsigned*)count_dev, (unsigned)cand.n,
                       cs.chunk_offset, cs.totals, count_out_dev, capacity, overflow_flag, (unsigned)CLEAN_ROW);
  hipLaunchKernelGGL(k_clean_scatter, dim3(CLEAN_GRID), dim3(BLK), 0, s, map, count_dev, cand, (const uint8_t*)cs.flags,
                     (const uint32_t*)cs.chunk_offset, time, out, capacity, (const uint32_t*)gnow, (const uint32_t*)cs.chunk_count, cs.totals,
                     count_out_dev, overflow_flag);
}

void candidates_to_aos(Candidates cand, float* aos, unsigned* count_dev, const CompactScratch& cs, hipStream_t s) {
  const int nch = ceil_div(cand.n, CHUNK);
  hipLaunchKernelGGL(k_cand_flags, dim3(nch), dim3(BLK), 0, s, cand, cs.flags, cs.chunk_count);
  hipLaunchKernelGGL(k_scan_chunks, dim3(1), dim3(1024), 0, s, (const uint32_t*)cs.chunk_count, (const unsigned*)nullptr, (unsigned)cand.n,
                     cs.chunk_offset, count_dev);
  hipLaunchKernelGGL(k_cand_scatter, dim3(nch), dim3(BLK), 0, s, cand, (const uint8_t*)cs.flags, (const uint32_t*)cs.chunk_offset, (float4*)aos);
}

// Deformation::sampleGraphModel (Deformation.cpp:232-306; sample.vert + sample.geom): every 5000th surfel of the model, in
// map order, as {position, initTime}
__global__ void k_sample_graph(SurfelSoA map, const unsigned* __restrict__ count_dev, int stride, int max_nodes, float4* __restrict__ out,
                               unsigned* __restrict__ n_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned count = *count_dev;
  const unsigned n = count == 0 ? 0u : (count - 1u) / (unsigned)stride + 1u;
  if (i == 0) *n_out = n < (unsigned)max_nodes ? n : (unsigned)max_nodes;
  if (i >= max_nodes || (unsigned)i >= n) return;
  const float4 p = map.pos_conf[(size_t)i * stride];
  const float4 c = map.col_time[(size_t)i * stride];
  out[i] = make_float4(p.x, p.y, p.z, c.z);
}
void sample_graph(SurfelSoA map, const unsigned* count_dev, int stride, int max_nodes, float* out4, unsigned* n_out, hipStream_t s) {
  hipLaunchKernelGGL(k_sample_graph, dim3(ceil_div(max_nodes, 256)), dim3(256), 0, s, map, count_dev, stride, max_nodes, (float4*)out4, n_out);
}

}  // namespace efm
