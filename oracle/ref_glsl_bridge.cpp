// TEST INFRASTRUCTURE — runs the REFERENCE's own shaders (Core/Shaders/*, compiled where they lie through
// oracle/glsl_on_cpu/) pass by pass, with the host-side set-up of the reference's draw calls restated here:
// uniforms as Core/{IndexMap,GlobalModel}.cpp, Core/Shaders/{FeedbackBuffer,FillIn,ComputePack}.cpp set them, vertex
// order as the reference's buffers have it, and the fixed-function stages specified as N1-N5: point
// rasterisation, depth test, transform feedback.  efg_<pass> has exactly the signature of the oracle's efo_<pass>
// (oracle/efo_api.h) so tests/test_oracle_vs_reference_glsl.py can push the same inputs through both.
// This file is ours; the shader bodies it calls are the reference's (namespaces glsl::sh_* emitted by glsl2cpp.pl
// ahead of this file in the same translation unit — see oracle/Makefile target refglsl).
#include <algorithm>
#include <limits>
#include <vector>

#include "efo_api.h"
#include "efo_pose.h"

namespace glsl {
float (*exp_hook)(float) = ::expf;
double texel_snap = 1.0 / 1024.0;
vec4 gl_Position, gl_FragCoord;
float gl_PointSize = 1, gl_FragDepth = 0;
int gl_VertexID = 0;
bool discard_flag = false;
gl_PerVertex gl_in[1];
std::function<void()> emit_hook;
}  // namespace glsl

using namespace glsl;

namespace {

constexpr int kTexDim = 3072;   // GlobalModel::TEXTURE_DIMENSION, GlobalModel.cpp:22

Texture tex(const void* data, int w, int h, int ch, Texture::Format f) {
  Texture t;
  t.data = data; t.width = w; t.height = h; t.channels = ch; t.format = f;
  return t;
}
mat4 to_mat4(const efo::Mat4f& M) {   // row-major -> column-major
  mat4 r;
  for (int row = 0; row < 4; ++row) for (int col = 0; col < 4; ++col) r.m[col][row] = M.m[row * 4 + col];
  return r;
}
// uv attribute buffer, FeedbackBuffer.cpp:44-52 == GlobalModel.cpp:109-117 (column-major pixel order; float + double mix as written)
inline vec2 uv_of(int i, int j, int cols, int rows) {
  return vec2((float)(((float)i / (float)cols) + 1.0 / (2 * (float)cols)), (float)(((float)j / (float)rows) + 1.0 / (2 * (float)rows)));
}
// texcoord of the full-screen quad (empty.vert + quad.geom) at pixel (x, y): linear interpolation of 0..1 at the pixel centre
inline vec2 quad_texcoord(int x, int y, int cols, int rows) { return vec2(((float)x + 0.5f) / (float)cols, ((float)y + 0.5f) / (float)rows); }
void put4(float* dst, const vec4& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
vec4 get4(const float* s) { return vec4(s[0], s[1], s[2], s[3]); }
// NDC -> window coordinate of the viewport transform (evaluated exactly on the float NDC value)
inline double window_coord(float ndc, int size) { return ((double)ndc + 1.0) * 0.5 * (double)size; }

}  // namespace

extern "C" {

void efg_use_specified_exp(int on) { glsl::exp_hook = on ? efo::efo_expf : ::expf; }
void efg_set_texel_snap(double s) { glsl::texel_snap = s; }
// N2, the depth test.  0: compare what the shaders write (gl_Position.z = z / maxDepth, gl_FragDepth = z / (2 maxDepth) + 0.5,
// both rounded to float32: two fragments closer than ~2.4 um tie and the earlier draw wins).  1: compare the
// camera-space z itself, which is what the oracle and the HIP kernels specify.
static int g_depth_mode = 1;
void efg_set_depth_compare(int mode) { g_depth_mode = mode; }

// ComputePack FILTER: empty.vert + quad.geom + depth_bilateral.frag (ElasticFusion.cpp:655-673, ComputePack.cpp:44-66)
void efg_filter_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered) {
  namespace S = glsl::sh_depth_bilateral_frag;
  const Texture t = tex(raw, cols, rows, 1, Texture::U16);
  S::gSampler.t = &t; S::cols = (float)cols; S::rows = (float)rows; S::maxD = maxD;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      S::texcoord = quad_texcoord(x, y, cols, rows);
      S::shader_main();
      filtered[y * cols + x] = (uint16_t)S::FragColor;
    }
}
// ComputePack METRIC / METRIC_FILTERED: depth_metric.frag
void efg_metricise_depth(const uint16_t* in, int cols, int rows, float maxD, float* out) {
  namespace S = glsl::sh_depth_metric_frag;
  const Texture t = tex(in, cols, rows, 1, Texture::U16);
  S::gSampler.t = &t; S::maxD = maxD;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      S::texcoord = quad_texcoord(x, y, cols, rows);
      S::shader_main();
      out[y * cols + x] = S::FragColor;
    }
}

// FeedbackBuffer::compute x2 (vertex_feedback.vert + .geom, transform feedback) then GlobalModel::initialise
// (init_unstable.vert; attributes 0,1 from the RAW stream, attribute 2 from the FILTERED stream, GlobalModel.cpp:229-284)
int efg_seed_map(const efo_cam* cam, const uint8_t* rgb, const float* depthMetric, const float* depthMetricFiltered, int time, float maxDepth,
                 float* out) {
  namespace V = glsl::sh_vertex_feedback_vert;
  namespace G = glsl::sh_vertex_feedback_geom;
  namespace I = glsl::sh_init_unstable_vert;
  const int cols = cam->cols, rows = cam->rows;
  const Texture tc = tex(rgb, cols, rows, 3, Texture::U8_NORM);
  std::vector<float> stream[2];
  for (int pass = 0; pass < 2; ++pass) {
    const Texture td = tex(pass == 0 ? depthMetric : depthMetricFiltered, cols, rows, 1, Texture::F32);
    V::gSampler.t = &td; V::cSampler.t = &tc;
    V::cam = vec4(cam->cx, cam->cy, 1.0f / cam->fx, 1.0f / cam->fy);   // FeedbackBuffer.cpp:88-92
    V::cols = (float)cols; V::rows = (float)rows; V::time = time; V::maxDepth = maxDepth;
    std::vector<float>& S = stream[pass];
    glsl::emit_hook = [&S]() {
      float rec[12];
      put4(rec, G::vPosition0); put4(rec + 4, G::vColor0); put4(rec + 8, G::vNormRad0);
      S.insert(S.end(), rec, rec + 12);
    };
    for (int i = 0; i < cols; ++i)
      for (int j = 0; j < rows; ++j) {
        V::texcoord = uv_of(i, j, cols, rows);
        V::shader_main();
        G::vPosition[0] = V::vPosition; G::vColor[0] = V::vColor; G::vNormRad[0] = V::vNormRad; G::zVal[0] = V::zVal;
        G::shader_main();
      }
    glsl::emit_hook = nullptr;
  }
  const int count = (int)(stream[0].size() / 12);
  stream[1].resize(std::max(stream[1].size(), stream[0].size()), 0.f);   // the filtered buffer keeps stale zeros past its end
  for (int k = 0; k < count; ++k) {
    I::vPosition = get4(&stream[0][(size_t)k * 12]);
    I::vColor = get4(&stream[0][(size_t)k * 12 + 4]);
    I::vNormRad = get4(&stream[1][(size_t)k * 12 + 8]);
    I::shader_main();
    put4(out + (size_t)k * 12, I::vPosition0); put4(out + (size_t)k * 12 + 4, I::vColor0); put4(out + (size_t)k * 12 + 8, I::vNormRad0);
  }
  return count;
}

// IndexMap::predictIndices: index_map.vert + index_map.frag, GL_POINTS of size 1, GL_LESS depth test (IndexMap.cpp:190-258)
void efg_predict_indices(const efo_cam* cam, const double* T_wc16, int time, const float* surfels, int count, float maxDepth, int timeDelta,
                         uint32_t* indexMap, float* vertConf, float* colorTime, float* normRad) {
  namespace V = glsl::sh_index_map_vert;
  namespace F = glsl::sh_index_map_frag;
  const int cols = cam->cols, rows = cam->rows, P = cols * rows;
  V::t_inv = to_mat4(efo::T_cw_float(T_wc16));
  V::cam = vec4(cam->cx, cam->cy, cam->fx, cam->fy);   // IndexMap.cpp:210-214 (FACTOR == 1)
  V::cols = (float)cols; V::rows = (float)rows; V::maxDepth = maxDepth; V::time = time; V::timeDelta = timeDelta;
  std::vector<float> zbuf(P, std::numeric_limits<float>::infinity());
  std::fill(indexMap, indexMap + P, 0u);
  std::fill(vertConf, vertConf + 4 * (size_t)P, 0.f);
  std::fill(colorTime, colorTime + 4 * (size_t)P, 0.f);
  std::fill(normRad, normRad + 4 * (size_t)P, 0.f);
  for (int id = 0; id < count; ++id) {
    const float* s = surfels + (size_t)id * 12;
    V::vPosition = get4(s); V::vColorTime = get4(s + 4); V::vNormRad = get4(s + 8);
    glsl::gl_VertexID = id;
    V::shader_main();
    const double xw = window_coord(glsl::gl_Position.x, cols), yw = window_coord(glsl::gl_Position.y, rows);
    if (!(xw >= 0 && xw < cols && yw >= 0 && yw < rows)) continue;              // N1: culled by its centre
    if (!(glsl::gl_Position.z >= -1.0f && glsl::gl_Position.z <= 1.0f)) continue;  // clip volume
    const int pi = (int)std::floor(yw) * cols + (int)std::floor(xw);
    const float z = g_depth_mode ? V::vPosition0.z : glsl::gl_Position.z;       // N2: GL_LESS, ties keep the earlier point
    if (!(z < zbuf[pi])) continue;
    F::vPosition0 = V::vPosition0; F::vColorTime0 = V::vColorTime0; F::vNormRad0 = V::vNormRad0; F::vertexId = V::vertexId;
    F::shader_main();
    zbuf[pi] = z;
    indexMap[pi] = (uint32_t)F::FragColor;
    put4(vertConf + (size_t)pi * 4, F::vPosition1); put4(colorTime + (size_t)pi * 4, F::vColorTime1); put4(normRad + (size_t)pi * 4, F::vNormRad1);
  }
}

// IndexMap::combinedPredict: splat.vert + combo_splat.frag, point sprites, GL_LESS on gl_FragDepth (IndexMap.cpp:293-393)
void efg_combined_predict(const efo_cam* cam, const double* T_wc16, const float* surfels, int count, float maxDepth, float confThreshold,
                          int time, int maxTime, int timeDelta, uint8_t* image, float* vertex, float* normal, uint16_t* timeMap) {
  namespace V = glsl::sh_splat_vert;
  namespace F = glsl::sh_combo_splat_frag;
  const int cols = cam->cols, rows = cam->rows, P = cols * rows;
  V::t_inv = to_mat4(efo::T_cw_float(T_wc16));
  V::cam = vec4(cam->cx, cam->cy, cam->fx, cam->fy);
  V::cols = (float)cols; V::rows = (float)rows; V::maxDepth = maxDepth; V::confThreshold = confThreshold;
  V::time = time; V::maxTime = maxTime; V::timeDelta = timeDelta;
  F::cam = V::cam; F::maxDepth = maxDepth;
  std::vector<float> zbuf(P, std::numeric_limits<float>::infinity());
  std::fill(image, image + 4 * (size_t)P, (uint8_t)0);
  std::fill(vertex, vertex + 4 * (size_t)P, 0.f);
  std::fill(normal, normal + 4 * (size_t)P, 0.f);
  std::fill(timeMap, timeMap + P, (uint16_t)0);
  for (int id = 0; id < count; ++id) {
    const float* s = surfels + (size_t)id * 12;
    V::vPosition = get4(s); V::vColor = get4(s + 4); V::vNormRad = get4(s + 8);
    V::shader_main();
    if (glsl::gl_Position.w != 1.0f) continue;   // the rejected-vertex marker of splat.vert:58-62 (w = 1000): far outside the clip volume
    float size = glsl::gl_PointSize;
    if (std::isnan(size) || std::isnan(glsl::gl_Position.x) || std::isnan(glsl::gl_Position.y)) continue;   // degenerate sprite axis: specified skip
    size = std::min(std::max(size, 1.0f), 2047.0f);   // N3
    const double u = window_coord(glsl::gl_Position.x, cols), v = window_coord(glsl::gl_Position.y, rows);
    if (!(u >= 0 && u < cols && v >= 0 && v < rows)) continue;   // N1: a point is clipped by its centre
    const double hs = (double)size * 0.5;
    const int px0 = std::max(0, (int)std::ceil(u - hs - 0.5)), px1 = std::min(cols - 1, (int)std::ceil(u + hs - 0.5) - 1);
    const int py0 = std::max(0, (int)std::ceil(v - hs - 0.5)), py1 = std::min(rows - 1, (int)std::ceil(v + hs - 0.5) - 1);
    F::position = V::position; F::normRad = V::normRad; F::colTime = V::colTime;
    for (int py = py0; py <= py1; ++py)
      for (int px = px0; px <= px1; ++px) {
        glsl::gl_FragCoord = vec4((float)px + 0.5f, (float)py + 0.5f, glsl::gl_Position.z, 1.0f);
        glsl::discard_flag = false;
        F::shader_main();
        if (glsl::discard_flag) continue;
        const int pi = py * cols + px;
        const float zkey = g_depth_mode ? F::vertex.z : glsl::gl_FragDepth;
        if (!(zkey < zbuf[pi])) continue;   // N2
        zbuf[pi] = zkey;
        uint8_t* im = image + (size_t)pi * 4;   // RGBA8 attachment: round(c * 255)
        im[0] = (uint8_t)roundf(F::image.x * 255.0f); im[1] = (uint8_t)roundf(F::image.y * 255.0f);
        im[2] = (uint8_t)roundf(F::image.z * 255.0f); im[3] = (uint8_t)roundf(F::image.w * 255.0f);
        put4(vertex + (size_t)pi * 4, F::vertex); put4(normal + (size_t)pi * 4, F::normal);
        timeMap[pi] = (uint16_t)F::time;
      }
  }
}

// IndexMap::synthesizeDepth: splat.vert + depth_splat.frag into an R32F attachment cleared to 0 (IndexMap.cpp:395-476)
void efg_synthesize_depth(const efo_cam* cam, const double* T_wc16, const float* surfels, int count, float maxDepth, float confThreshold,
                          int time, int maxTime, int timeDelta, float* depth) {
  namespace V = glsl::sh_splat_vert;
  namespace F = glsl::sh_depth_splat_frag;
  const int cols = cam->cols, rows = cam->rows, P = cols * rows;
  V::t_inv = to_mat4(efo::T_cw_float(T_wc16));
  V::cam = vec4(cam->cx, cam->cy, cam->fx, cam->fy);
  V::cols = (float)cols; V::rows = (float)rows; V::maxDepth = maxDepth; V::confThreshold = confThreshold;
  V::time = time; V::maxTime = maxTime; V::timeDelta = timeDelta;
  F::cam = V::cam; F::maxDepth = maxDepth;
  std::vector<float> zbuf(P, std::numeric_limits<float>::infinity());
  std::fill(depth, depth + P, 0.f);
  for (int id = 0; id < count; ++id) {
    const float* s = surfels + (size_t)id * 12;
    V::vPosition = get4(s); V::vColor = get4(s + 4); V::vNormRad = get4(s + 8);
    V::shader_main();
    if (glsl::gl_Position.w != 1.0f) continue;
    float size = glsl::gl_PointSize;
    if (std::isnan(size) || std::isnan(glsl::gl_Position.x) || std::isnan(glsl::gl_Position.y)) continue;
    size = std::min(std::max(size, 1.0f), 2047.0f);
    const double u = window_coord(glsl::gl_Position.x, cols), v = window_coord(glsl::gl_Position.y, rows);
    if (!(u >= 0 && u < cols && v >= 0 && v < rows)) continue;
    const double hs = (double)size * 0.5;
    const int px0 = std::max(0, (int)std::ceil(u - hs - 0.5)), px1 = std::min(cols - 1, (int)std::ceil(u + hs - 0.5) - 1);
    const int py0 = std::max(0, (int)std::ceil(v - hs - 0.5)), py1 = std::min(rows - 1, (int)std::ceil(v + hs - 0.5) - 1);
    F::position = V::position; F::normRad = V::normRad;
    for (int py = py0; py <= py1; ++py)
      for (int px = px0; px <= px1; ++px) {
        glsl::gl_FragCoord = vec4((float)px + 0.5f, (float)py + 0.5f, glsl::gl_Position.z, 1.0f);
        glsl::discard_flag = false;
        F::shader_main();
        if (glsl::discard_flag) continue;
        const int pi = py * cols + px;
        const float zkey = g_depth_mode ? F::FragColor : glsl::gl_FragDepth;
        if (!(zkey < zbuf[pi])) continue;
        zbuf[pi] = zkey;
        depth[pi] = F::FragColor;
      }
  }
}

// FillIn::{vertex,normal,image}: quad + fill_vertex.frag / fill_normal.frag / fill_rgb.frag (FillIn.cpp:62-191)
void efg_fill_in(const efo_cam* cam, const uint8_t* image, const float* vertex, const float* normal, const uint16_t* depthFiltered,
                 const uint8_t* rgb, int passthrough, int passthroughImage, uint8_t* fimage, float* fvertex, float* fnormal) {
  namespace FV = glsl::sh_fill_vertex_frag;
  namespace FN = glsl::sh_fill_normal_frag;
  namespace FC = glsl::sh_fill_rgb_frag;
  const int cols = cam->cols, rows = cam->rows;
  const Texture tv = tex(vertex, cols, rows, 4, Texture::F32), tn = tex(normal, cols, rows, 4, Texture::F32);
  const Texture td = tex(depthFiltered, cols, rows, 1, Texture::U16);
  const Texture ti = tex(image, cols, rows, 4, Texture::U8_NORM), tc = tex(rgb, cols, rows, 3, Texture::U8_NORM);
  const vec4 camv(cam->cx, cam->cy, 1.0f / cam->fx, 1.0f / cam->fy);   // FillIn.cpp:115-119
  FV::eSampler.t = &tv; FV::rSampler.t = &td; FV::cam = camv; FV::cols = (float)cols; FV::rows = (float)rows; FV::passthrough = passthrough;
  FN::eSampler.t = &tn; FN::rSampler.t = &td; FN::cam = camv; FN::cols = (float)cols; FN::rows = (float)rows; FN::passthrough = passthrough;
  FC::eSampler.t = &ti; FC::rSampler.t = &tc; FC::passthrough = passthroughImage;
  for (int y = 0; y < rows; ++y)
    for (int x = 0; x < cols; ++x) {
      const size_t pi = (size_t)y * cols + x;
      const vec2 tcd = quad_texcoord(x, y, cols, rows);
      FV::texcoord = tcd; FV::shader_main(); put4(fvertex + pi * 4, FV::FragColor);
      FN::texcoord = tcd; FN::shader_main(); put4(fnormal + pi * 4, FN::FragColor);
      FC::texcoord = tcd; FC::shader_main();
      const vec4 c = FC::FragColor;
      fimage[pi * 4] = (uint8_t)roundf(c.x * 255.0f); fimage[pi * 4 + 1] = (uint8_t)roundf(c.y * 255.0f);
      fimage[pi * 4 + 2] = (uint8_t)roundf(c.z * 255.0f); fimage[pi * 4 + 3] = (uint8_t)roundf(c.w * 255.0f);
    }
}

// GlobalModel::fuse: data pass (data.vert + data.geom + data.frag into the 3072^2 update maps, transform feedback of
// the new-unstable stream) then update pass (update.vert over the model).  GlobalModel.cpp:368-524
int efg_fuse(const efo_cam* cam, const double* T_wc16, int time, const uint8_t* rgb, const float* depthMetric, const float* depthMetricFiltered,
             const uint32_t* indexMap, const float* vertConf, const float* colorTime, const float* normRad, float maxDepth, float weighting,
             float* surfels, int count, float* newUnstable) {
  namespace V = glsl::sh_data_vert;
  namespace G = glsl::sh_data_geom;
  namespace F = glsl::sh_data_frag;
  namespace U = glsl::sh_update_vert;
  const int cols = cam->cols, rows = cam->rows;
  const Texture tc = tex(rgb, cols, rows, 3, Texture::U8_NORM), tdr = tex(depthMetric, cols, rows, 1, Texture::F32);
  const Texture tdf = tex(depthMetricFiltered, cols, rows, 1, Texture::F32), tix = tex(indexMap, cols, rows, 1, Texture::U32);
  const Texture tvc = tex(vertConf, cols, rows, 4, Texture::F32), tct = tex(colorTime, cols, rows, 4, Texture::F32);
  const Texture tnr = tex(normRad, cols, rows, 4, Texture::F32);
  V::cSampler.t = &tc; V::drSampler.t = &tdr; V::drfSampler.t = &tdf; V::indexSampler.t = &tix; V::vertConfSampler.t = &tvc;
  V::colorTimeSampler.t = &tct; V::normRadSampler.t = &tnr;
  V::cam = vec4(cam->cx, cam->cy, (float)(1.0 / (double)cam->fx), (float)(1.0 / (double)cam->fy));   // GlobalModel.cpp:392-398
  V::cols = (float)cols; V::rows = (float)rows; V::scale = 1.0f; V::texDim = (float)kTexDim; V::pose = to_mat4(efo::pose_castf(T_wc16));
  V::maxDepth = maxDepth; V::time = (float)time; V::weighting = weighting;
  // update maps: 3 x RGBA32F 3072^2, cleared every frame (lazily committed zero pages)
  const size_t texels = (size_t)kTexDim * kTexDim;
  float* um[3];
  for (auto& m : um) m = (float*)calloc(texels * 4, sizeof(float));
  std::vector<uint8_t> written(texels, 0);
  int nNew = 0;
  glsl::emit_hook = [&]() {
    float* rec = newUnstable + (size_t)nNew * 12;   // transform feedback: every emitted vertex, in draw order
    put4(rec, G::vPosition0); put4(rec + 4, G::vColor0); put4(rec + 8, G::vNormRad0);
    ++nNew;
    // rasterise the point into the update maps
    const double xw = window_coord(glsl::gl_Position.x, kTexDim), yw = window_coord(glsl::gl_Position.y, kTexDim);
    if (!(xw >= 0 && xw < kTexDim && yw >= 0 && yw < kTexDim)) return;   // (-10,-10): the new-unstable vertices are off screen
    const size_t ti = (size_t)std::floor(yw) * kTexDim + (size_t)std::floor(xw);
    if (written[ti]) return;   // N5: the first point in draw order owns the texel
    F::vPosition0 = G::vPosition0; F::vColor0 = G::vColor0; F::vNormRad0 = G::vNormRad0; F::updateId0 = G::updateId0;
    F::vPosition1 = vec4(); F::vColor1 = vec4(); F::vNormRad1 = vec4();
    F::shader_main();
    written[ti] = 1;
    put4(um[0] + ti * 4, F::vPosition1); put4(um[1] + ti * 4, F::vColor1); put4(um[2] + ti * 4, F::vNormRad1);
  };
  for (int i = 0; i < cols; ++i)
    for (int j = 0; j < rows; ++j) {
      V::texcoord = uv_of(i, j, cols, rows);
      V::shader_main();
      G::vPosition[0] = V::vPosition; G::vColor[0] = V::vColor; G::vNormRad[0] = V::vNormRad; G::updateId[0] = V::updateId;
      glsl::gl_in[0].gl_Position = glsl::gl_Position;
      G::shader_main();
    }
  glsl::emit_hook = nullptr;
  const Texture t0 = tex(um[0], kTexDim, kTexDim, 4, Texture::F32), t1 = tex(um[1], kTexDim, kTexDim, 4, Texture::F32);
  const Texture t2 = tex(um[2], kTexDim, kTexDim, 4, Texture::F32);
  U::vertSamp.t = &t0; U::colorSamp.t = &t1; U::normSamp.t = &t2; U::texDim = (float)kTexDim; U::time = time;
  for (int id = 0; id < count; ++id) {
    float* s = surfels + (size_t)id * 12;
    U::vPosition = get4(s); U::vColor = get4(s + 4); U::vNormRad = get4(s + 8);
    glsl::gl_VertexID = id;
    U::shader_main();
    put4(s, U::vPosition0); put4(s + 4, U::vColor0); put4(s + 8, U::vNormRad0);
  }
  for (auto& m : um) free(m);
  return nNew;
}

// GlobalModel::clean: copy_unstable.vert + copy_unstable.geom over the model then over the new-unstable stream,
// transform feedback (GlobalModel.cpp:527-671).  graph / nodes: the deformation graph as GlobalModel.cpp:540-546 uploads it
// into the 1 x 16384 LUMINANCE32F node texture (16 floats per node); depth: IndexMap::synthesizeDepth's image.
int efg_clean_deform(const efo_cam* cam, const double* T_wc16, int time, const uint32_t* indexMap, const float* vertConf,
                     const float* colorTime, const float* normRad, float confThreshold, int timeDelta, float maxDepth, const float* surfels,
                     int count, const float* newUnstable, int newCount, const float* graph, int nodes, const float* depth, int isFern,
                     float* out) {
  namespace V = glsl::sh_copy_unstable_vert;
  namespace G = glsl::sh_copy_unstable_geom;
  const int cols = cam->cols, rows = cam->rows;
  const int kNodeDim = 16384;   // GlobalModel::NODE_TEXTURE_DIMENSION, GlobalModel.cpp:23
  const Texture tix = tex(indexMap, cols, rows, 1, Texture::U32), tvc = tex(vertConf, cols, rows, 4, Texture::F32);
  const Texture tct = tex(colorTime, cols, rows, 4, Texture::F32), tnr = tex(normRad, cols, rows, 4, Texture::F32);
  std::vector<float> nodeRow(kNodeDim, 0.f);
  if (nodes > 0) std::copy(graph, graph + (size_t)nodes * 16, nodeRow.begin());
  const Texture tnode = tex(nodeRow.data(), kNodeDim, 1, 1, Texture::F32);
  const Texture tdepth = tex(depth, cols, rows, 1, Texture::F32);
  V::indexSampler.t = &tix; V::vertConfSampler.t = &tvc; V::colorTimeSampler.t = &tct; V::normRadSampler.t = &tnr;
  V::nodeSampler.t = &tnode; V::depthSampler.t = depth ? &tdepth : nullptr;
  V::time = time; V::confThreshold = confThreshold; V::scale = 1.0f; V::nodes = (float)nodes; V::nodeCols = (float)kNodeDim;
  V::timeDelta = timeDelta; V::maxDepth = maxDepth; V::isFern = isFern;
  V::t_inv = to_mat4(efo::T_cw_float(T_wc16));
  V::cam = vec4(cam->cx, cam->cy, cam->fx, cam->fy);   // GlobalModel.cpp:570-575
  V::cols = (float)cols; V::rows = (float)rows;
  int outCount = 0;
  glsl::emit_hook = [&]() {
    float* rec = out + (size_t)outCount * 12;
    put4(rec, G::vPosition0); put4(rec + 4, G::vColor0); put4(rec + 8, G::vNormRad0);
    ++outCount;
  };
  for (int k = 0; k < count + newCount; ++k) {
    const float* s = k < count ? surfels + (size_t)k * 12 : newUnstable + (size_t)(k - count) * 12;
    V::vPos = get4(s); V::vCol = get4(s + 4); V::vNormR = get4(s + 8);
    V::shader_main();
    G::vPosition[0] = V::vPosition; G::vColor[0] = V::vColor; G::vNormRad[0] = V::vNormRad; G::test[0] = V::test;
    G::shader_main();
  }
  glsl::emit_hook = nullptr;
  return outCount;
}
int efg_clean(const efo_cam* cam, const double* T_wc16, int time, const uint32_t* indexMap, const float* vertConf, const float* colorTime,
              const float* normRad, float confThreshold, int timeDelta, float maxDepth, const float* surfels, int count,
              const float* newUnstable, int newCount, float* out) {
  return efg_clean_deform(cam, T_wc16, time, indexMap, vertConf, colorTime, normRad, confThreshold, timeDelta, maxDepth, surfels, count,
                          newUnstable, newCount, nullptr, 0, nullptr, 0, out);
}

// Deformation::sampleGraphModel (Deformation.cpp:232-306): sample.vert + sample.geom under transform feedback, every 5000th
// surfel of the model -> {position, initTime}
int efg_sample_graph(const float* surfels, int count, float* out4) {
  namespace V = glsl::sh_sample_vert;
  namespace G = glsl::sh_sample_geom;
  int n = 0;
  glsl::emit_hook = [&]() { put4(out4 + (size_t)n * 4, G::vData); ++n; };
  for (int k = 0; k < count; ++k) {
    const float* s = surfels + (size_t)k * 12;
    V::vPosition = get4(s); V::vColorTime = get4(s + 4); V::vNormRad = get4(s + 8);
    glsl::gl_VertexID = k;
    V::shader_main();
    G::vPosition0[0] = V::vPosition0; G::vColorTime0[0] = V::vColorTime0; G::vNormRad0[0] = V::vNormRad0; G::id[0] = V::id;
    G::shader_main();
  }
  glsl::emit_hook = nullptr;
  return n;
}

// Resize::vertex / Resize::image (Resize.cpp:50-121): one point expanded by quad.geom to a full-viewport quad whose texcoord runs
// 0..1, so fragment (a, b) of the (cols/f x rows/f) target samples the source at its own centre, ((a + 0.5) / dw, (b + 0.5) / dh).
// elemBytes 16 = RGBA32F texture, 4 = RGBA8 (read back through the normalised float path).
void efg_resize_nearest(const void* src, int cols, int rows, int elemBytes, int factor, void* dst) {
  namespace F = glsl::sh_resize_frag;
  const int dw = cols / factor, dh = rows / factor;
  const Texture t = elemBytes == 16 ? tex((const float*)src, cols, rows, 4, Texture::F32) : tex((const uint8_t*)src, cols, rows, 4, Texture::U8_NORM);
  F::eSampler.t = &t;
  for (int b = 0; b < dh; ++b)
    for (int a = 0; a < dw; ++a) {
      F::texcoord = vec2(((float)a + 0.5f) / (float)dw, ((float)b + 0.5f) / (float)dh);
      F::shader_main();
      if (elemBytes == 16) {
        put4((float*)dst + ((size_t)b * dw + a) * 4, F::FragColor);
      } else {
        float c[4];
        put4(c, F::FragColor);
        for (int k = 0; k < 4; ++k) ((uint8_t*)dst)[((size_t)b * dw + a) * 4 + k] = (uint8_t)std::lround(c[k] * 255.0f);
      }
    }
}

const char* efg_about() {
  return "reference Core/Shaders/*.{vert,geom,frag,glsl} compiled by g++ through oracle/glsl_on_cpu (-ffp-contract=off); "
         "fixed-function stages as specified in SURVEY.md 8a N1-N5";
}

}  // extern "C"
