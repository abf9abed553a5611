// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// CPU restatement of the reference's GLSL "compute" passes and the surfel-map maintenance:
//   Core/Shaders/depth_bilateral.frag, depth_metric.frag            (G1, G2)
//   vertex_feedback.{vert,geom}, init_unstable.vert, surfels.glsl, geometry.glsl, color.glsl (G3)
//   index_map.{vert,frag}                      via IndexMap::predictIndices   (G4)
//   splat.vert + combo_splat.frag              via IndexMap::combinedPredict  (G5)
//   fill_{vertex,normal,rgb}.frag              via FillIn                     (G7)
//   resize.frag + ElasticFusion::denseEnough                                  (G8)
//   data.{vert,geom,frag}, update.vert         via GlobalModel::fuse          (G9, G10)
//   copy_unstable.{vert,geom}                  via GlobalModel::clean         (G11)
// GL-defined behaviour that cannot be observed here is *specified* (SURVEY.md §8a N1-N5):
//   N1 size-1 point -> pixel (floor u, floor v), culled when the centre is outside the viewport
//   N2 depth test on the float camera-space z, ties -> lower surfel index (draw order, GL_LESS)
//   N3 sprite = pixel centres in [u-s/2, u+s/2) x [v-s/2, v+s/2), s clamped to [1, 2047]
//   N4 NEAREST sampling, CLAMP_TO_EDGE; the 4 taps/axis of data.vert / copy_unstable.vert sit at
//      pixel offsets {-1,-1/2,0,+1/2} -> texel floor(x+off)
//   N5 update-map collisions: first pixel in draw (column-major) order wins
// parity PINNED: built with -DEFO_NO_FMA this file reproduces, bit for bit, the reference's own shaders compiled for the
// CPU (oracle/_ref/libefr_glsl.so, oracle/glsl_on_cpu/) and the golden vectors made from them (tests/golden/); N1-N5 stay specified.
#include "efo_common.h"
#include "efo_linalg.h"
#include "efo_pose.h"
#include "efo_api.h"
#include <algorithm>
#include <vector>

using namespace efo;

namespace {

constexpr int kTexDim = 3072;  // GlobalModel::TEXTURE_DIMENSION, GlobalModel.cpp:22

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// uv attribute of FeedbackBuffer.cpp:44-52 / GlobalModel.cpp:109-117 and x = texcoord.x * cols
inline float pix_coord(int i, int n) {
  float u = (float)((double)((float)i / (float)n) + 1.0 / (double)(2 * (float)n));
  return u * (float)n;
}

inline m33 rot_of(const Mat4f& M) {
  return m33{{{M.m[0], M.m[1], M.m[2]}, {M.m[4], M.m[5], M.m[6]}, {M.m[8], M.m[9], M.m[10]}}};
}
inline f3 trans_of(const Mat4f& M) { return {M.m[3], M.m[7], M.m[11]}; }
// mat4 * vec4(p,1): fma chain per row, then + translation column
inline f3 xform(const Mat4f& M, f3 p) { return mul(rot_of(M), p) + trans_of(M); }

// color.glsl:19-34
inline float encodeColor(f3 c) {
  int rgb = (int)roundf(c.x * 255.0f);
  rgb = (rgb << 8) + (int)roundf(c.y * 255.0f);
  rgb = (rgb << 8) + (int)roundf(c.z * 255.0f);
  return (float)rgb;
}
inline f3 decodeColor(float c) {
  int ic = (int)c;
  return {(float)((ic >> 16) & 0xFF) / 255.0f, (float)((ic >> 8) & 0xFF) / 255.0f, (float)(ic & 0xFF) / 255.0f};
}
// surfels.glsl:19-34 ; cam.z = 1/fx, cam.w = 1/fy
inline float getRadius(float depth, float norm_z, float inv_fx, float inv_fy) {
  float meanFocal = ((1.0f / fabsf(inv_fx)) + (1.0f / fabsf(inv_fy))) / 2.0f;
  const float sqrt2 = 1.41421356237f;
  float radius = (depth / meanFocal) * sqrt2;
  float radius_n = radius / fabsf(norm_z);
  return fminf(2.0f * radius, radius_n);
}
// surfels.glsl:36-46
inline float confidence(float x, float y, float cx, float cy, float weighting) {
  const float maxRadDist = 400, twoSigmaSquared = 0.72f;
  float px = x - cx, py = y - cy;
  float radialDist = sqrtf(px * px + py * py) / maxRadDist;
  return efo_expf(-(radialDist * radialDist) / twoSigmaSquared) * weighting;
}

struct DepthF {  // float depth texture, NEAREST + CLAMP_TO_EDGE
  const float* d; int cols, rows;
  float at(int x, int y) const { return d[clampi(y, 0, rows - 1) * cols + clampi(x, 0, cols - 1)]; }
};
// geometry.glsl:21-25 (float depth): vertex at float pixel coords (x,y) from texel (ix,iy)
inline f3 getVertexF(const DepthF& D, int ix, int iy, float x, float y, float cx, float cy, float inv_fx, float inv_fy) {
  float z = D.at(ix, iy);
  return {(x - cx) * z * inv_fx, (y - cy) * z * inv_fy, z};
}
// geometry.glsl:28-40 central difference
inline f3 getNormalF(const DepthF& D, f3 vPosition, int ix, int iy, float x, float y, float cx, float cy, float inv_fx, float inv_fy) {
  f3 xf = getVertexF(D, ix + 1, iy, x + 1, y, cx, cy, inv_fx, inv_fy);
  f3 xb = getVertexF(D, ix - 1, iy, x - 1, y, cx, cy, inv_fx, inv_fy);
  f3 yf = getVertexF(D, ix, iy + 1, x, y + 1, cx, cy, inv_fx, inv_fy);
  f3 yb = getVertexF(D, ix, iy - 1, x, y - 1, cx, cy, inv_fx, inv_fy);
  auto half = [](f3 a, f3 b) { return f3{(a.x + b.x) / 2, (a.y + b.y) / 2, (a.z + b.z) / 2}; };
  f3 del_x = half(xb, vPosition) - half(xf, vPosition);
  f3 del_y = half(yb, vPosition) - half(yf, vPosition);
  return normalized(cross(del_x, del_y));
}

}  // namespace

extern "C" {

// depth_bilateral.frag:30-76 (G1)
void efo_filter_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered) {
  const float sigma_space2_inv_half = 0.024691358f, sigma_color2_inv_half = 0.000555556f;
  const int R = 6, D = R * 2 + 1;
  const unsigned maxv = (unsigned)(maxD * 1000.0f);
  efo::parallel_for(rows, [&](int y0, int y1) {
  for (int y = y0; y < y1; ++y)
    for (int x = 0; x < cols; ++x) {
      unsigned value = raw[y * cols + x];
      if (value > maxv || value < 300U) { filtered[y * cols + x] = 0; continue; }
      int tx = std::min(x - D / 2 + D, cols), ty = std::min(y - D / 2 + D, rows);
      float sum1 = 0, sum2 = 0;
      for (int cy = std::max(y - D / 2, 0); cy < ty; ++cy)
        for (int cx = std::max(x - D / 2, 0); cx < tx; ++cx) {
          unsigned tmp = raw[cy * cols + cx];
          float space2 = ((float)x - (float)cx) * ((float)x - (float)cx) + ((float)y - (float)cy) * ((float)y - (float)cy);
          float color2 = ((float)value - (float)tmp) * ((float)value - (float)tmp);
          float weight = efo_expf(-(space2 * sigma_space2_inv_half + color2 * sigma_color2_inv_half));
          sum1 += (float)tmp * weight;
          sum2 += weight;
        }
      filtered[y * cols + x] = (uint16_t)(unsigned)roundf(sum1 / sum2);
    }
  });
}

// depth_metric.frag:28-40 (G2)
void efo_metricise_depth(const uint16_t* in, int cols, int rows, float maxD, float* out) {
  const unsigned maxv = (unsigned)(maxD * 1000.0f);
  for (int i = 0; i < cols * rows; ++i) {
    unsigned value = in[i];
    out[i] = (value > maxv || value < 300U) ? 0.0f : (float)value / 1000.0f;
  }
}

// vertex_feedback.{vert,geom} run twice (raw, filtered) + init_unstable.vert (G3).
// ElasticFusion.cpp:240-254, FeedbackBuffer.cpp:81-138, GlobalModel.cpp:229-284.
// The two transform-feedback streams are compacted independently and zipped by output index, exactly
// as the reference binds attribute 0/1 from RAW and attribute 2 from FILTERED.
int efo_seed_map(const efo_cam* cam, const uint8_t* rgb, const float* depthMetric, const float* depthMetricFiltered,
                 int time, float maxDepth, float* out) {
  const int cols = cam->cols, rows = cam->rows;
  const float inv_fx = 1.0f / cam->fx, inv_fy = 1.0f / cam->fy;  // FeedbackBuffer.cpp:91-95
  std::vector<float> rawStream, filtStream;  // 12 floats per emitted vertex
  for (int pass = 0; pass < 2; ++pass) {
    DepthF D{pass == 0 ? depthMetric : depthMetricFiltered, cols, rows};
    std::vector<float>& S = pass == 0 ? rawStream : filtStream;
    for (int i = 0; i < cols; ++i)
      for (int j = 0; j < rows; ++j) {
        float x = pix_coord(i, cols), y = pix_coord(j, rows);
        f3 v = getVertexF(D, i, j, x, y, cam->cx, cam->cy, inv_fx, inv_fy);
        f3 n = getNormalF(D, v, i, j, x, y, cam->cx, cam->cy, inv_fx, inv_fy);
        float zVal = (v.z <= 0 || v.z > maxDepth) ? 0.f : v.z;
        if (!(zVal > 0)) continue;
        const uint8_t* c = rgb + (size_t)(j * cols + i) * 3;
        f3 col{(float)c[0] / 255.0f, (float)c[1] / 255.0f, (float)c[2] / 255.0f};
        float rec[12] = {v.x, v.y, v.z, confidence(x, y, cam->cx, cam->cy, 1.0f),
                         encodeColor(col), 0.f, col.z, (float)time,
                         n.x, n.y, n.z, getRadius(v.z, n.z, inv_fx, inv_fy)};
        S.insert(S.end(), rec, rec + 12);
      }
  }
  const int count = (int)(rawStream.size() / 12);
  filtStream.resize((size_t)std::max<size_t>(filtStream.size(), rawStream.size()), 0.f);  // stale zeros beyond its end
  for (int k = 0; k < count; ++k) {
    float* o = out + (size_t)k * 12;
    for (int c = 0; c < 8; ++c) o[c] = rawStream[(size_t)k * 12 + c];
    o[5] = 0;  // init_unstable.vert:33-34
    o[6] = 1;
    for (int c = 8; c < 12; ++c) o[c] = filtStream[(size_t)k * 12 + c];
  }
  return count;
}

// IndexMap::predictIndices + index_map.{vert,frag} (G4)
void efo_predict_indices(const efo_cam* cam, const double* T_wc16, int time, const float* surfels, int count,
                         float maxDepth, int timeDelta, uint32_t* indexMap, float* vertConf, float* colorTime,
                         float* normRad) {
  const int cols = cam->cols, rows = cam->rows, P = cols * rows;
  const Mat4f T = T_cw_float(T_wc16);
  const m33 R = rot_of(T);
  std::vector<float> zbuf(P, std::numeric_limits<float>::infinity());
  std::fill(indexMap, indexMap + P, 0u);
  std::fill(vertConf, vertConf + 4 * (size_t)P, 0.f);
  std::fill(colorTime, colorTime + 4 * (size_t)P, 0.f);
  std::fill(normRad, normRad + 4 * (size_t)P, 0.f);
  for (int id = 0; id < count; ++id) {
    const float* s = surfels + (size_t)id * 12;
    f3 p = xform(T, f3{s[0], s[1], s[2]});
    if (p.z > maxDepth || p.z < 0 || (float)time - s[7] > (float)timeDelta) continue;
    float u = ((cam->fx * p.x) / p.z) + cam->cx;
    float v = ((cam->fy * p.y) / p.z) + cam->cy;
    // index_map.vert:52-53 hands the point over in NDC — x = (u - cols * 0.5) / (cols * 0.5), in float — and the viewport transform brings it
    // back to a window position, (x + 1) * cols / 2, evaluated exactly on the float NDC value (N1): a point within ~1e-5 px of a pixel edge
    // lands in the neighbouring pixel, which floor(u) would not see (rounds 1-4 tolerated 2e-4 of the pixels for it)
    const float xn = (u - (float)cols * 0.5f) / ((float)cols * 0.5f), yn = (v - (float)rows * 0.5f) / ((float)rows * 0.5f);
    const double xw = ((double)xn + 1.0) * 0.5 * (double)cols, yw = ((double)yn + 1.0) * 0.5 * (double)rows;
    if (!(xw >= 0 && xw < cols && yw >= 0 && yw < rows)) continue;  // N1: culled by its centre
    int px = (int)std::floor(xw), py = (int)std::floor(yw);
    int pi = py * cols + px;
    if (!(p.z < zbuf[pi])) continue;  // N2 (strict: earlier id keeps ties)
    zbuf[pi] = p.z;
    indexMap[pi] = (uint32_t)id;
    float* vc = vertConf + (size_t)pi * 4;
    vc[0] = p.x; vc[1] = p.y; vc[2] = p.z; vc[3] = s[3];
    std::memcpy(colorTime + (size_t)pi * 4, s + 4, 16);
    f3 n = normalized(mul(R, f3{s[8], s[9], s[10]}));
    float* nr = normRad + (size_t)pi * 4;
    nr[0] = n.x; nr[1] = n.y; nr[2] = n.z; nr[3] = s[11];
  }
}

// IndexMap::combinedPredict + splat.vert + combo_splat.frag (G5)
void efo_combined_predict(const efo_cam* cam, const double* T_wc16, const float* surfels, int count, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, uint8_t* image, float* vertex,
                          float* normal, uint16_t* timeMap) {
  const int cols = cam->cols, rows = cam->rows, P = cols * rows;
  const Mat4f T = T_cw_float(T_wc16);
  const m33 R = rot_of(T);
  const float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
  std::vector<float> zbuf(P, std::numeric_limits<float>::infinity());
  std::fill(image, image + 4 * (size_t)P, (uint8_t)0);
  std::fill(vertex, vertex + 4 * (size_t)P, 0.f);
  std::fill(normal, normal + 4 * (size_t)P, 0.f);
  std::fill(timeMap, timeMap + P, (uint16_t)0);
  auto projImage = [&](f3 p) { return f3{((fx * p.x) / p.z) + cx, ((fy * p.y) / p.z) + cy, p.z}; };
  for (int id = 0; id < count; ++id) {
    const float* s = surfels + (size_t)id * 12;
    f3 p = xform(T, f3{s[0], s[1], s[2]});
    if (p.z > maxDepth || p.z < 0 || s[3] < confThreshold || (float)time - s[7] > (float)timeDelta || s[7] > (float)maxTime)
      continue;
    f3 n = normalized(mul(R, f3{s[8], s[9], s[10]}));
    float rad = s[11];
    // splat.vert:70  normalize(..) * normRad.w * 1.41421356  ==  (t*r)*c per component
    f3 t1 = normalized(f3{n.y - n.z, -n.x, n.x});
    f3 x1{(t1.x * rad) * 1.41421356f, (t1.y * rad) * 1.41421356f, (t1.z * rad) * 1.41421356f};
    f3 y1 = cross(n, x1);
    f3 q1 = projImage(p + x1), q2 = projImage(p + y1), q3 = projImage(p - y1), q4 = projImage(p - x1);
    float xmin = fminf(q1.x, fminf(q2.x, fminf(q3.x, q4.x))), xmax = fmaxf(q1.x, fmaxf(q2.x, fmaxf(q3.x, q4.x)));
    float ymin = fminf(q1.y, fminf(q2.y, fminf(q3.y, q4.y))), ymax = fmaxf(q1.y, fmaxf(q2.y, fmaxf(q3.y, q4.y)));
    float size = fmaxf(0.f, fmaxf(fabsf(xmax - xmin), fabsf(ymax - ymin)));
    if (std::isnan(size) || std::isnan(xmin) || std::isnan(ymin)) continue;  // degenerate sprite axis: specified skip
    size = fminf(fmaxf(size, 1.0f), 2047.0f);                                // N3
    float u = ((fx * p.x) / p.z) + cx, v = ((fy * p.y) / p.z) + cy;
    if (!(u >= 0 && u < (float)cols && v >= 0 && v < (float)rows)) continue;  // point clipped by its centre
    float hs = size * 0.5f;
    int px0 = std::max(0, (int)ceilf(u - hs - 0.5f)), px1 = std::min(cols - 1, (int)ceilf(u + hs - 0.5f) - 1);
    int py0 = std::max(0, (int)ceilf(v - hs - 0.5f)), py1 = std::min(rows - 1, (int)ceilf(v + hs - 0.5f) - 1);
    const float sqrRad = rad * rad;
    const float pn = dot(p, n);
    f3 colr = decodeColor(s[4]);
    for (int py = py0; py <= py1; ++py)
      for (int px = px0; px <= px1; ++px) {
        float fcx = (float)px + 0.5f, fcy = (float)py + 0.5f;  // gl_FragCoord
        f3 l = normalized(f3{(fcx - cx) / fx, (fcy - cy) / fy, 1.0f});
        float k = pn / dot(l, n);
        f3 cp{k * l.x, k * l.y, k * l.z};
        f3 diff = cp - p;
        if (!(dot(diff, diff) <= sqrRad)) continue;  // discard (also on NaN)
        float z = cp.z;
        int pi = py * cols + px;
        if (!(z < zbuf[pi])) continue;  // N2
        zbuf[pi] = z;
        uint8_t* im = image + (size_t)pi * 4;
        im[0] = (uint8_t)roundf(colr.x * 255.0f); im[1] = (uint8_t)roundf(colr.y * 255.0f);
        im[2] = (uint8_t)roundf(colr.z * 255.0f); im[3] = 255;
        float* vt = vertex + (size_t)pi * 4;
        vt[0] = (fcx - cx) * z * (1.f / fx); vt[1] = (fcy - cy) * z * (1.f / fy); vt[2] = z; vt[3] = s[3];
        float* nm = normal + (size_t)pi * 4;
        nm[0] = n.x; nm[1] = n.y; nm[2] = n.z; nm[3] = rad;
        timeMap[pi] = (uint16_t)(unsigned)s[6];
      }
  }
}

// IndexMap::synthesizeDepth + splat.vert + depth_splat.frag (G6), IndexMap.cpp:395-476.  depth_splat.frag:35-46 is
// combo_splat.frag:37-48 with FragColor = corrected_pos.z as the only output, same sprite, same depth test: the synthesized
// depth IS the z channel of combinedPredict's vertex map (0 where nothing was drawn: glClearColor(0,0,0,0)).
void efo_synthesize_depth(const efo_cam* cam, const double* T_wc16, const float* surfels, int count, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, float* depth) {
  const size_t P = (size_t)cam->cols * cam->rows;
  std::vector<uint8_t> image(4 * P);
  std::vector<float> vertex(4 * P), normal(4 * P);
  std::vector<uint16_t> timeMap(P);
  efo_combined_predict(cam, T_wc16, surfels, count, maxDepth, confThreshold, time, maxTime, timeDelta, image.data(), vertex.data(),
                       normal.data(), timeMap.data());
  for (size_t i = 0; i < P; ++i) depth[i] = vertex[4 * i + 2];
}

// FillIn::{vertex,normal,image} + fill_*.frag (G7).  cam = (cx, cy, 1/fx, 1/fy), FillIn.cpp:115-119
void efo_fill_in(const efo_cam* cam, const uint8_t* image, const float* vertex, const float* normal,
                 const uint16_t* depthFiltered, const uint8_t* rgb, int passthrough, int passthroughImage,
                 uint8_t* fimage, float* fvertex, float* fnormal) {
  const int cols = cam->cols, rows = cam->rows;
  const float cx = cam->cx, cy = cam->cy, inv_fx = 1.0f / cam->fx, inv_fy = 1.0f / cam->fy;
  auto depthAt = [&](int x, int y) { return (float)depthFiltered[clampi(y, 0, rows - 1) * cols + clampi(x, 0, cols - 1)] / 1000.0f; };
  auto vtx = [&](int sx, int sy, int x, int y) {  // geometry.glsl:44-48 (int pixel coords, Q4)
    float z = depthAt(sx, sy);
    return f3{((float)x - cx) * z * inv_fx, ((float)y - cy) * z * inv_fy, z};
  };
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      size_t pi = (size_t)y * cols + x;
      const float* sv = vertex + pi * 4;
      float* ov = fvertex + pi * 4;
      if (sv[2] == 0 || passthrough) {
        f3 v = vtx(x, y, x, y);
        ov[0] = v.x; ov[1] = v.y; ov[2] = v.z; ov[3] = 1;
      } else std::memcpy(ov, sv, 16);
      const float* sn = normal + pi * 4;
      float* on = fnormal + pi * 4;
      if (sn[2] == 0 || passthrough) {  // fill_normal.frag tests the normal image's own z
        f3 v = vtx(x, y, x, y);
        f3 vx = vtx(x + 1, y, x + 1, y), vy = vtx(x, y + 1, x, y + 1);  // geometry.glsl:52-60 forward difference
        f3 nn = normalized(cross(vx - v, vy - v));
        on[0] = nn.x; on[1] = nn.y; on[2] = nn.z; on[3] = 1;
      } else std::memcpy(on, sn, 16);
      const uint8_t* si = image + pi * 4;
      uint8_t* oi = fimage + pi * 4;
      float sum = (float)si[0] / 255.0f + (float)si[1] / 255.0f + (float)si[2] / 255.0f;
      if (sum == 0 || passthroughImage) {
        oi[0] = rgb[pi * 3]; oi[1] = rgb[pi * 3 + 1]; oi[2] = rgb[pi * 3 + 2]; oi[3] = 255;
      } else std::memcpy(oi, si, 4);
    }
}

// Resize::image + ElasticFusion::denseEnough (G8), consSample = 20 (ElasticFusion.cpp:62-70,256-268)
// the two float pose matrices every map pass derives from T_wc (efo_pose.h), row-major: T_cw = T_wc.inverse().matrix().cast<float>()
// (IndexMap.cpp:208, GlobalModel.cpp:567) and pose = T_wc.cast<float>().matrix() (GlobalModel.cpp:403)
void efo_pose_matrices(const double* T_wc16, float* T_cw16, float* pose16) {
  const Mat4f a = T_cw_float(T_wc16), b = pose_castf(T_wc16);
  std::memcpy(T_cw16, a.m, sizeof(a.m));
  std::memcpy(pose16, b.m, sizeof(b.m));
}

// Resize::{image,vertex,time} (Resize.cpp:50-159; empty.vert + quad.geom + resize.frag): NEAREST downsample, destination pixel
// (a, b) <- source texel (f*a + f/2, f*b + f/2) (the sample point (a + 0.5) * f falls exactly on a texel boundary and belongs to
// the upper texel, N4 / G8).  Used with f = 20 for the constraint grid and denseEnough, f = 8 by the fern database (Ferns.cpp:31-36).
void efo_resize_nearest(const void* src, int cols, int rows, int elemBytes, int factor, void* dst) {
  const int dw = cols / factor, dh = rows / factor;
  for (int b = 0; b < dh; ++b)
    for (int a = 0; a < dw; ++a)
      std::memcpy((char*)dst + ((size_t)b * dw + a) * elemBytes,
                  (const char*)src + ((size_t)(b * factor + factor / 2) * cols + (a * factor + factor / 2)) * elemBytes, elemBytes);
}

int efo_dense_enough(const efo_cam* cam, const uint8_t* image) {
  const int dc = cam->cols / 20, dr = cam->rows / 20;
  int sum = 0;
  for (int b = 0; b < dr; ++b)
    for (int a = 0; a < dc; ++a) {
      const uint8_t* t = image + ((size_t)(20 * b + 10) * cam->cols + (20 * a + 10)) * 4;
      sum += (t[0] > 0 && t[1] > 0 && t[2] > 0);
    }
  return (float)sum / (float)(dr * dc) > 0.75f;
}

// GlobalModel::fuse = data pass (data.vert/geom/frag) + update pass (update.vert)  (G9, G10)
int efo_fuse(const efo_cam* cam, const double* T_wc16, int time, const uint8_t* rgb, const float* depthMetric,
             const float* depthMetricFiltered, const uint32_t* indexMap, const float* vertConf,
             const float* colorTime, const float* normRad, float maxDepth, float weighting, float* surfels, int count,
             float* newUnstable) {
  (void)colorTime;
  const int cols = cam->cols, rows = cam->rows;
  const float cx = cam->cx, cy = cam->cy;
  const float inv_fx = (float)(1.0 / (double)cam->fx), inv_fy = (float)(1.0 / (double)cam->fy);  // GlobalModel.cpp:397-398
  const Mat4f pose = pose_castf(T_wc16);
  const m33 Rp = rot_of(pose);
  const DepthF DR{depthMetric, cols, rows}, DF{depthMetricFiltered, cols, rows};
  const float ftime = (float)time;
  // update "textures": one winner per surfel id (N5)
  std::vector<int> winner(count > 0 ? count : 1, -1);
  std::vector<float> upd;  // 12 floats per winner, indexed through winner[]
  int nNew = 0;
  for (int i = 0; i < cols; ++i)
    for (int j = 0; j < rows; ++j) {
      float x = pix_coord(i, cols), y = pix_coord(j, rows);
      f3 vPosLocal = getVertexF(DR, i, j, x, y, cx, cy, inv_fx, inv_fy);
      if (!((int)x % 2 == (int)ftime % 2 && (int)y % 2 == (int)ftime % 2)) continue;
      // checkNeighbours, data.vert:50-69
      if (DR.at(i - 1, j) == 0 || DR.at(i, j - 1) == 0 || DR.at(i + 1, j) == 0 || DR.at(i, j + 1) == 0) continue;
      if (!(vPosLocal.z > 0 && vPosLocal.z <= maxDepth)) continue;
      f3 vPos = xform(pose, vPosLocal);
      f3 vPosition_f = getVertexF(DF, i, j, x, y, cx, cy, inv_fx, inv_fy);
      const uint8_t* c = rgb + (size_t)(j * cols + i) * 3;
      f3 col{(float)c[0] / 255.0f, (float)c[1] / 255.0f, (float)c[2] / 255.0f};
      f3 vNormLocal = getNormalF(DF, vPosition_f, i, j, x, y, cx, cy, inv_fx, inv_fy);
      f3 nW = mul(Rp, vNormLocal);
      float rec[12] = {vPos.x, vPos.y, vPos.z, confidence(x, y, cx, cy, weighting),
                       encodeColor(col), 0.f, ftime, 0.f,
                       nW.x, nW.y, nW.z, getRadius(vPosition_f.z, vNormLocal.z, inv_fx, inv_fy)};
      // association, data.vert:114-158
      int counter = 0;
      uint32_t best = 0;
      float bestDist = 1000;
      float xl = (x - cx) * inv_fx, yl = (y - cy) * inv_fy;
      float lambda = sqrtf(xl * xl + yl * yl + 1);
      f3 ray{xl, yl, 1};
      const float lenRay = sqrtf(dot(ray, ray));
      static const int tapOff[4] = {-1, 0, 0, 1};  // N4: pixel offsets {-1,-1/2,0,+1/2} from the centre i+0.5
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
          int tx = clampi(i + tapOff[a], 0, cols - 1), ty = clampi(j + tapOff[b], 0, rows - 1);
          size_t ti = (size_t)ty * cols + tx;
          uint32_t current = indexMap[ti];
          if (current > 0U) {
            const float* vc = vertConf + ti * 4;
            if (fabsf((vc[2] * lambda) - (vPosLocal.z * lambda)) < 0.05f) {
              f3 cr = cross(ray, f3{vc[0], vc[1], vc[2]});
              float dist = sqrtf(dot(cr, cr)) / lenRay;
              const float* nr = normRad + ti * 4;
              f3 nn{nr[0], nr[1], nr[2]};
              // abs(acos(c)) < 0.5  <=>  cos(0.5) < c <= 1   (acos(c>1) is NaN -> false)
              float cang = dot(nn, vNormLocal) / (sqrtf(dot(nn, nn)) * sqrtf(dot(vNormLocal, vNormLocal)));
              bool angOk = (cang > 0.87758255f && cang <= 1.0f);
              if (dist < bestDist && (fabsf(nr[2]) < 0.75f || angOk)) {
                counter++;
                bestDist = dist;
                best = current;
              }
            }
          }
        }
      if (counter > 0) {
        rec[7] = -1;
        if ((int)best < count && winner[best] < 0) {  // first in draw order wins the update texel (N5)
          winner[best] = (int)(upd.size() / 12);
          upd.insert(upd.end(), rec, rec + 12);
        }
      } else {
        rec[7] = -2;
      }
      std::memcpy(newUnstable + (size_t)nNew * 12, rec, sizeof(rec));  // data.geom:37-48 emits both kinds
      ++nNew;
    }
  // update.vert:37-92
  for (int id = 0; id < count; ++id) {
    if (winner[id] < 0) continue;
    const float* u = &upd[(size_t)winner[id] * 12];
    float* s = surfels + (size_t)id * 12;
    float c_k = s[3], a = u[3];
    if (u[11] < (1.0f + 0.5f) * s[11]) {
      for (int k = 0; k < 3; ++k) s[k] = ((c_k * s[k]) + (a * u[k])) / (c_k + a);
      s[3] = c_k + a;
      f3 oldCol = decodeColor(s[4]), newCol = decodeColor(u[4]);
      f3 avg{((c_k * oldCol.x) + (a * newCol.x)) / (c_k + a), ((c_k * oldCol.y) + (a * newCol.y)) / (c_k + a),
             ((c_k * oldCol.z) + (a * newCol.z)) / (c_k + a)};
      s[4] = encodeColor(avg);
      s[7] = ftime;
      float nr[4];
      for (int k = 0; k < 4; ++k) nr[k] = ((c_k * s[8 + k]) + (a * u[8 + k])) / (c_k + a);
      f3 nn = normalized(f3{nr[0], nr[1], nr[2]});
      s[8] = nn.x; s[9] = nn.y; s[10] = nn.z; s[11] = nr[3];
    } else {
      s[3] = c_k + a;
      s[7] = ftime;
    }
  }
  return nNew;
}

// GlobalModel::clean + copy_unstable.{vert,geom} (G11), with the deformation-graph application of
// copy_unstable.vert:128-322 when nodes > 0: graph = nodes x 16 floats as GlobalModel.cpp:540-546
// uploads them into the 1-row node texture {position 3, rotation 9 (column major), translation 3, time 1}, sorted by time.
// depth = IndexMap::synthesizeDepth's image (G6) for the "seen again" test, may be null when nodes == 0.
}  // extern "C"
namespace {
// 3x3 inverse as glsl_on_cpu specifies it: adjugate (cofactors from 2x2 products) times 1/det; m[col][row]
struct M3c { float m[3][3]; };
inline M3c inverse3(const M3c& a) {
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
// mat3 * vec3, column-major storage: row i = dot((m[0][i], m[1][i], m[2][i]), v), like every other mat * vec of this file
inline f3 mulc(const M3c& a, f3 v) {
  return {dot(f3{a.m[0][0], a.m[1][0], a.m[2][0]}, v), dot(f3{a.m[0][1], a.m[1][1], a.m[2][1]}, v), dot(f3{a.m[0][2], a.m[1][2], a.m[2][2]}, v)};
}

// copy_unstable.vert:128-322 for one kept vertex; v = {pos conf | colour 0 initTime lastTime | normal radius}
void deform_vertex(float* v, const float* graph, int nodes, const Mat4f& T, const efo_cam* cam, const float* depth, float confThreshold,
                   float maxDepth, float ftime, int isFern) {
  const int k = 4, lookBack = 20;
  auto node = [&](int idx) {   // 1-row node texture, NEAREST + CLAMP_TO_EDGE (texel = idx: the shader's coordinates are exact)
    const int width = 16384;   // GlobalModel::NODE_TEXTURE_DIMENSION
    idx = clampi(idx, 0, width - 1);
    return idx < nodes * 16 ? graph[idx] : 0.0f;
  };
  int nearNodes[lookBack];
  float nearDists[lookBack];
  for (int i = 0; i < lookBack; ++i) { nearNodes[i] = -1; nearDists[i] = 16777216.0f; }
  const int poseTime = (int)v[6];
  int foundIndex = 0, imin = 0, imax = nodes - 1, imid = (imin + imax) / 2;
  while (imax >= imin) {
    imid = (imin + imax) / 2;
    const int nodeTime = (int)node(imid * 16 + 15);
    if (nodeTime < poseTime) imin = imid + 1;
    else if (nodeTime > poseTime) imax = imid - 1;
    else break;
  }
  imin = std::min(imin, nodes - 1);
  const int nodeMin = (int)node(imin * 16 + 15), nodeMid = (int)node(imid * 16 + 15), nodeMax = (int)node(imax * 16 + 15);
  if (std::abs(nodeMin - poseTime) <= std::abs(nodeMid - poseTime) && std::abs(nodeMin - poseTime) <= std::abs(nodeMax - poseTime)) foundIndex = imin;
  else if (std::abs(nodeMid - poseTime) <= std::abs(nodeMin - poseTime) && std::abs(nodeMid - poseTime) <= std::abs(nodeMax - poseTime)) foundIndex = imid;
  else foundIndex = imax;
  if (foundIndex == nodes) foundIndex = nodes - 1;
  const f3 pos{v[0], v[1], v[2]};
  auto node_pos = [&](int j) { return f3{node(j * 16), node(j * 16 + 1), node(j * 16 + 2)}; };
  auto dist_to = [&](int j) { const f3 d = pos - node_pos(j); return sqrtf(dot(d, d)); };
  int nearNodeIndex = 0, distanceBack = 0;
  for (int j = foundIndex; j >= 0; --j) {
    nearNodes[nearNodeIndex] = j;
    nearDists[nearNodeIndex] = dist_to(j);
    nearNodeIndex++;
    if (++distanceBack == lookBack / 2) break;
  }
  for (int j = foundIndex + 1; j < nodes; ++j) {
    nearNodes[nearNodeIndex] = j;
    nearDists[nearNodeIndex] = dist_to(j);
    nearNodeIndex++;
    if (++distanceBack == lookBack) break;
  }
  for (int i = 0; i < lookBack - 1; ++i)
    for (int j = i + 1; j < lookBack; ++j)
      if (nearDists[j] < nearDists[i]) { std::swap(nearDists[i], nearDists[j]); std::swap(nearNodes[i], nearNodes[j]); }
  const float dMax = nearDists[k];
  float nodeWeights[k];
  float weightSum = 0;
  for (int j = 0; j < k; ++j) {
    const float q = 1.0f - (dist_to(nearNodes[j]) / dMax);
    nodeWeights[j] = q * q;   // pow(x, 2)
    weightSum += nodeWeights[j];
  }
  for (int j = 0; j < k; ++j) nodeWeights[j] /= weightSum;
  f3 newPos{0, 0, 0}, newNorm{0, 0, 0};
  const f3 nrm{v[8], v[9], v[10]};
  for (int i = 0; i < k; ++i) {
    const int n = nearNodes[i];
    const f3 g = node_pos(n);
    M3c R;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R.m[c][r] = node(n * 16 + 3 + c * 3 + r);
    const f3 t{node(n * 16 + 12), node(n * 16 + 13), node(n * 16 + 14)};
    const f3 moved = (mulc(R, pos - g) + g) + t;
    newPos = newPos + f3{nodeWeights[i] * moved.x, nodeWeights[i] * moved.y, nodeWeights[i] * moved.z};
    const M3c Ri = inverse3(R);
    M3c Rit;   // transpose(inverse(rotation))
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) Rit.m[c][r] = Ri.m[r][c];
    const f3 rn = mulc(Rit, nrm);
    newNorm = newNorm + f3{nodeWeights[i] * rn.x, nodeWeights[i] * rn.y, nodeWeights[i] * rn.z};
  }
  v[0] = newPos.x; v[1] = newPos.y; v[2] = newPos.z;
  const f3 nn = normalized(newNorm);
  v[8] = nn.x; v[9] = nn.y; v[10] = nn.z;
  if (v[3] > confThreshold && isFern == 0) {
    const f3 lp = xform(T, f3{v[0], v[1], v[2]});
    const float x = ((cam->fx * lp.x) / lp.z) + cam->cx, y = ((cam->fy * lp.y) / lp.z) + cam->cy;
    if (lp.z > 0 && lp.z < maxDepth && x > 0 && y > 0 && x < (float)cam->cols && y < (float)cam->rows) {
      // textureLod(depthSampler, vec2(x / cols, y / rows)): NEAREST -> texel floor(x), floor(y) (N4)
      const float currentDepth = depth[clampi((int)floorf(y), 0, cam->rows - 1) * cam->cols + clampi((int)floorf(x), 0, cam->cols - 1)];
      if (currentDepth > 0.0f && lp.z < currentDepth + 0.1f) v[7] = ftime;
    }
  }
}
}  // namespace
extern "C" {

int efo_clean_deform(const efo_cam* cam, const double* T_wc16, int time, const uint32_t* indexMap, const float* vertConf,
                     const float* colorTime, const float* normRad, float confThreshold, int timeDelta, float maxDepth,
                     const float* surfels, int count, const float* newUnstable, int newCount, const float* graph, int nodes,
                     const float* depth, int isFern, float* out) {
  (void)normRad;
  const int cols = cam->cols, rows = cam->rows;
  const float fx = cam->fx, fy = cam->fy, cx = cam->cx, cy = cam->cy;
  const Mat4f T = T_cw_float(T_wc16);
  const m33 R = rot_of(T);
  const float ftime = (float)time, ftd = (float)timeDelta;
  int outCount = 0;
  static const float tapOff[4] = {-1.0f, -0.5f, 0.0f, 0.5f};  // N4
  for (int k = 0; k < count + newCount; ++k) {
    const float* s = k < count ? surfels + (size_t)k * 12 : newUnstable + (size_t)(k - count) * 12;
    float v[12];
    std::memcpy(v, s, sizeof(v));
    int test = 1;
    f3 localPos = xform(T, f3{v[0], v[1], v[2]});
    float x = ((fx * localPos.x) / localPos.z) + cx;
    float y = ((fy * localPos.y) / localPos.z) + cy;
    f3 localNorm = normalized(mul(R, f3{v[8], v[9], v[10]}));
    int cnt = 0, zCount = 0;
    if (ftime - v[7] < ftd && localPos.z > 0 && x > 0 && y > 0 && x < (float)cols && y < (float)rows) {
      for (int a = 0; a < 4; ++a)
        for (int b = 0; b < 4; ++b) {
          int tx = clampi((int)floorf(x + tapOff[a]), 0, cols - 1), ty = clampi((int)floorf(y + tapOff[b]), 0, rows - 1);
          size_t ti = (size_t)ty * cols + tx;
          if (indexMap[ti] > 0U) {
            const float* vc = vertConf + ti * 4;
            const float* ct = colorTime + ti * 4;
            float dx = vc[0] - localPos.x, dy = vc[1] - localPos.y;
            if (ct[2] < v[6] && vc[3] > confThreshold && vc[2] > localPos.z && vc[2] - localPos.z < 0.01f &&
                sqrtf(dx * dx + dy * dy) < v[11] * 1.4f)
              cnt++;
            if (ct[3] == ftime && vc[3] > confThreshold && vc[2] > localPos.z && vc[2] - localPos.z > 0.01f &&
                fabsf(localNorm.z) > 0.85f)
              zCount++;
          }
        }
    }
    if (cnt > 8 || zCount > 4) test = 0;
    if (v[7] == -2) v[7] = ftime;                                        // new unstable point
    if (v[7] == -1 || ((ftime - v[7]) > 20 && v[3] < confThreshold)) test = 0;
    if (v[7] > 0 && ftime - v[7] > ftd) test = 1;
    // vColor.z != time: points initialised this frame were fused with the updated pose already (copy_unstable.vert:130-131)
    if (test == 1 && nodes > 0 && v[6] != ftime) deform_vertex(v, graph, nodes, T, cam, depth, confThreshold, maxDepth, ftime, isFern);
    if (test) {
      std::memcpy(out + (size_t)outCount * 12, v, sizeof(v));
      ++outCount;
    }
  }
  return outCount;
}

int efo_clean(const efo_cam* cam, const double* T_wc16, int time, const uint32_t* indexMap, const float* vertConf,
              const float* colorTime, const float* normRad, float confThreshold, int timeDelta, float maxDepth,
              const float* surfels, int count, const float* newUnstable, int newCount, float* out) {
  return efo_clean_deform(cam, T_wc16, time, indexMap, vertConf, colorTime, normRad, confThreshold, timeDelta, maxDepth, surfels, count,
                          newUnstable, newCount, nullptr, 0, nullptr, 0, out);
}

}  // extern "C"
