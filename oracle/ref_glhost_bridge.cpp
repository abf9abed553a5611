// TEST INFRASTRUCTURE — runs the REFERENCE's own map-side HOST code (Core/IndexMap.cpp, GlobalModel.cpp,
// Shaders/{FillIn,ComputePack,FeedbackBuffer,Resize}.cpp + Shaders.h, Uniform.h), compiled from /root/reference where it lies against
// host_on_cpu/ — OpenGL and Pangolin as a tape recorder (host_on_cpu/gl_record.h) — and hands back the transcript of what each
// call asked the GL to do (oracle/Makefile, target `refglhost` -> _ref/libefr_glhost.so).  This file is ours.  The ComputePack /
// FeedbackBuffer objects are constructed here the way ElasticFusion::createCompute / createFeedbackBuffers do
// (ElasticFusion.cpp:165-206); what runs inside them is the reference's.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "GlobalModel.h"
#include "IndexMap.h"
#include "Shaders/ComputePack.h"
#include "Shaders/FeedbackBuffer.h"
#include "Shaders/FillIn.h"
#include "Shaders/Resize.h"
#include "Utils/Img.h"
#include "efo_linalg.h"

const std::string GPUTexture::RGB = "RGB";
const std::string GPUTexture::DEPTH_RAW = "DEPTH";
const std::string GPUTexture::DEPTH_FILTERED = "DEPTH_FILTERED";
const std::string GPUTexture::DEPTH_METRIC = "DEPTH_METRIC";
const std::string GPUTexture::DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
const std::string GPUTexture::DEPTH_NORM = "DEPTH_NORM";
// Core/GPUTexture.cpp:28-46 without the CUDA registration
GPUTexture::GPUTexture(const int w, const int h, const GLenum internalFormat_, const GLenum format_, const GLenum dataType_, const bool draw_)
    : texture(new pangolin::GlTexture(w, h, internalFormat_, draw_, 0, format_, dataType_)), cudaRes(nullptr), draw(draw_), width(w), height(h),
      internalFormat(internalFormat_), format(format_), dataType(dataType_) {}
GPUTexture::~GPUTexture() { delete texture; }

namespace {
struct Host {
  int w, h;
  // the frame textures of ElasticFusion::createTextures (ElasticFusion.cpp:165-206)
  GPUTexture rgb, depthRaw, depthFiltered, depthMetric, depthMetricFiltered;
  IndexMap indexMap;
  GlobalModel globalModel;
  FillIn fillIn;
  Resize resize;
  FeedbackBuffer rawFeedback, filteredFeedback;
  ComputePack filter, metric, metricFiltered;
  std::string out;
  Host(int w_, int h_)
      : w(w_), h(h_),
        rgb(w_, h_, GL_RGBA, GL_RGB, GL_UNSIGNED_BYTE, true),
        depthRaw(w_, h_, GL_LUMINANCE16UI_EXT, GL_LUMINANCE_INTEGER_EXT, GL_UNSIGNED_SHORT, false),
        depthFiltered(w_, h_, GL_LUMINANCE16UI_EXT, GL_LUMINANCE_INTEGER_EXT, GL_UNSIGNED_SHORT, false),
        depthMetric(w_, h_, GL_LUMINANCE32F_ARB, GL_LUMINANCE, GL_FLOAT, false),
        depthMetricFiltered(w_, h_, GL_LUMINANCE32F_ARB, GL_LUMINANCE, GL_FLOAT, false),
        resize(w_, h_, w_ / 20, h_ / 20),
        rawFeedback(loadProgramGeomFromFile("vertex_feedback.vert", "vertex_feedback.geom")),
        filteredFeedback(loadProgramGeomFromFile("vertex_feedback.vert", "vertex_feedback.geom")),
        filter(loadProgramFromFile("empty.vert", "depth_bilateral.frag", "quad.geom"), depthFiltered.texture),
        metric(loadProgramFromFile("empty.vert", "depth_metric.frag", "quad.geom"), depthMetric.texture),
        metricFiltered(loadProgramFromFile("empty.vert", "depth_metric.frag", "quad.geom"), depthMetricFiltered.texture) {}
};
Sophus::SE3d pose_of(const double* T16) { return Sophus::SE3d(efo::se3_from_matrix(T16)); }
const char* take(Host* h) {
  h->out.swap(glrec::S().log);
  glrec::S().log.clear();
  return h->out.c_str();
}
}  // namespace

extern "C" {

void* efh_create(int w, int h, float fx, float fy, float cx, float cy) {
  Resolution::getInstance(w, h);
  Intrinsics::getInstance(fx, fy, cx, cy);
  return new Host(w, h);
}
void efh_destroy(void* p) { delete (Host*)p; }
const char* efh_take_log(void* p) { return take((Host*)p); }             // everything recorded since the last call (construction first)
unsigned efh_constant(const char* name) { return glrec::S().constants.count(name) ? glrec::S().constants[name] : 0xFFFFFFFFu; }
void efh_set_query_result(unsigned n) { glrec::S().query_result = n; }    // what "primitives written" queries return
// texture / object ids by the reference's own accessor names
unsigned efh_tid(void* p, const char* name) {
  Host* h = (Host*)p;
  const std::string n = name;
  IndexMap& im = h->indexMap;
  GPUTexture* t = n == "index" ? im.indexTex() : n == "vertConf" ? im.vertConfTex() : n == "colorTime" ? im.colorTimeTex() : n == "normalRad" ? im.normalRadTex()
                : n == "image" ? im.imageTex() : n == "vertex" ? im.vertexTex() : n == "normal" ? im.normalTex() : n == "time" ? im.timeTex()
                : n == "oldImage" ? im.oldImageTex() : n == "oldVertex" ? im.oldVertexTex() : n == "oldNormal" ? im.oldNormalTex() : n == "oldTime" ? im.oldTimeTex()
                : n == "depth" ? im.depthTex() : n == "rgb" ? &h->rgb : n == "depthRaw" ? &h->depthRaw : n == "depthFiltered" ? &h->depthFiltered
                : n == "depthMetric" ? &h->depthMetric : n == "depthMetricFiltered" ? &h->depthMetricFiltered : n == "fillImage" ? &h->fillIn.imageTexture
                : n == "fillVertex" ? &h->fillIn.vertexTexture : n == "fillNormal" ? &h->fillIn.normalTexture : nullptr;
  return t ? t->texture->tid : 0;
}
void efh_model(void* p, unsigned* vbo, unsigned* fid) {
  const std::pair<GLuint, GLuint>& m = ((Host*)p)->globalModel.model();
  *vbo = m.first; *fid = m.second;
}
const char* efh_predict_indices(void* p, const double* T16, int time, float depthCutoff, int timeDelta) {
  Host* h = (Host*)p;
  h->indexMap.predictIndices(pose_of(T16), time, h->globalModel.model(), depthCutoff, timeDelta);
  return take(h);
}
const char* efh_combined_predict(void* p, const double* T16, float depthCutoff, float confThreshold, int time, int maxTime, int timeDelta, int inactive) {
  Host* h = (Host*)p;
  h->indexMap.combinedPredict(pose_of(T16), h->globalModel.model(), depthCutoff, confThreshold, time, maxTime, timeDelta,
                              inactive ? IndexMap::INACTIVE : IndexMap::ACTIVE);
  return take(h);
}
const char* efh_synthesize_depth(void* p, const double* T16, float depthCutoff, float confThreshold, int time, int maxTime, int timeDelta) {
  Host* h = (Host*)p;
  h->indexMap.synthesizeDepth(pose_of(T16), h->globalModel.model(), depthCutoff, confThreshold, time, maxTime, timeDelta);
  return take(h);
}
const char* efh_fuse(void* p, const double* T16, int time, float depthCutoff, float weighting) {
  Host* h = (Host*)p;
  h->globalModel.fuse(pose_of(T16), time, &h->rgb, &h->depthMetric, &h->depthMetricFiltered, h->indexMap.indexTex(), h->indexMap.vertConfTex(),
                      h->indexMap.colorTimeTex(), h->indexMap.normalRadTex(), depthCutoff, weighting);
  return take(h);
}
const char* efh_clean(void* p, const double* T16, int time, float confThreshold, const float* graph, int nodes, int timeDelta, float maxDepth, int isFern) {
  Host* h = (Host*)p;
  std::vector<float> g(graph, graph + (size_t)nodes * 16);
  h->globalModel.clean(pose_of(T16), time, h->indexMap.indexTex(), h->indexMap.vertConfTex(), h->indexMap.colorTimeTex(), h->indexMap.normalRadTex(),
                       h->indexMap.depthTex(), confThreshold, g, timeDelta, maxDepth, isFern != 0);
  return take(h);
}
const char* efh_feedback_and_initialise(void* p, int time, float depthCutoff) {
  Host* h = (Host*)p;
  h->rawFeedback.compute(h->rgb.texture, h->depthMetric.texture, time, depthCutoff);                   // ElasticFusion.cpp:208-221
  h->filteredFeedback.compute(h->rgb.texture, h->depthMetricFiltered.texture, time, depthCutoff);
  h->globalModel.initialise(h->rawFeedback, h->filteredFeedback);
  return take(h);
}
const char* efh_fill_in(void* p, int lost) {
  Host* h = (Host*)p;
  h->fillIn.vertex(h->indexMap.vertexTex(), &h->depthFiltered, lost != 0);                             // ElasticFusion.cpp:636-650
  h->fillIn.normal(h->indexMap.normalTex(), &h->depthFiltered, lost != 0);
  h->fillIn.image(h->indexMap.imageTex(), &h->rgb, lost != 0);
  return take(h);
}
const char* efh_resize(void* p) {
  Host* h = (Host*)p;
  Img<Eigen::Matrix<uint8_t, 3, 1>> imageBuff(h->h / 20, h->w / 20);
  Img<Eigen::Vector4f> consBuff(h->h / 20, h->w / 20);
  Img<uint16_t> timesBuff(h->h / 20, h->w / 20);
  h->resize.image(h->indexMap.imageTex(), imageBuff);
  h->resize.vertex(h->indexMap.vertexTex(), consBuff);
  h->resize.time(h->indexMap.oldTimeTex(), timesBuff);
  return take(h);
}
// ElasticFusion::filterDepth / metriciseDepth (ElasticFusion.cpp:655-673): the uniform lists are built there (not compiled); what
// ComputePack::compute does with them is
const char* efh_compute_packs(void* p, float depthCutoff) {
  Host* h = (Host*)p;
  std::vector<Uniform> uf;
  uf.push_back(Uniform("cols", (float)Resolution::getInstance().cols()));
  uf.push_back(Uniform("rows", (float)Resolution::getInstance().rows()));
  uf.push_back(Uniform("maxD", depthCutoff));
  h->filter.compute(h->depthRaw.texture, &uf);
  std::vector<Uniform> um;
  um.push_back(Uniform("maxD", depthCutoff));
  h->metric.compute(h->depthRaw.texture, &um);
  h->metricFiltered.compute(h->depthFiltered.texture, &um);
  return take(h);
}

}  // extern "C"
