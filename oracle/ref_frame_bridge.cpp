// TEST INFRASTRUCTURE — runs the REFERENCE's own per-frame orchestration, Core/ElasticFusion.cpp (constructor, processFrame, predict,
// savePly, the destructor's trajectory dump), compiled from /root/reference where it lies and linked with its own IndexMap /
// GlobalModel / FillIn / ComputePack / FeedbackBuffer / Resize / Ferns / Deformation sources, over
//   * OpenGL + Pangolin as a tape recorder (host_on_cpu/gl_record.h), and
//   * recording doubles (below, ours) for the two things that would need real pixels: the tracker (RGBDOdometry) and the graph
//     optimiser (DeformationGraph).
// processFrame then leaves a transcript of the whole frame: which passes run in which order with which parameters, what the tracker
// is initialised from and asked to do, how the pose flows.  The transcript is what tests compare with the oracle's frame loop
// (oracle/efo_frame.cpp); savePly and the .freiburg dump are compared byte for byte with the product's writers.
// oracle/Makefile, target `refframe` -> _ref/libefr_frame.so.
#include <algorithm>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <random>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

// the fern database's generator and its two measures are private members (Ferns.h:157-168); the tests call them directly
#define private public
#include "Ferns.h"
#undef private
#include "ElasticFusion.h"
#include "efo_linalg.h"

const std::string GPUTexture::RGB = "RGB";
const std::string GPUTexture::DEPTH_RAW = "DEPTH";
const std::string GPUTexture::DEPTH_FILTERED = "DEPTH_FILTERED";
const std::string GPUTexture::DEPTH_METRIC = "DEPTH_METRIC";
const std::string GPUTexture::DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
const std::string GPUTexture::DEPTH_NORM = "DEPTH_NORM";
GPUTexture::GPUTexture(const int w, const int h, const GLenum internalFormat_, const GLenum format_, const GLenum dataType_, const bool draw_)
    : texture(new pangolin::GlTexture(w, h, internalFormat_, draw_, 0, format_, dataType_)), cudaRes(nullptr), draw(draw_), width(w), height(h),
      internalFormat(internalFormat_), format(format_), dataType(dataType_) {}
GPUTexture::~GPUTexture() { delete texture; }

namespace {
struct TrackerScript {   // what the recording tracker answers
  double delta[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};   // T_wc := T_wc * delta
  float icpError = 1e-6f, icpCount = 100000.f;
  double covDiag = 1e-7;
  int constrainResult = 0;   // what DeformationGraph::optimiseGraphSparse / Deformation::constrain see
} g_script;
int tex_id(GPUTexture* t) { return t && t->texture ? (int)t->texture->tid : 0; }
void rec_pose(const char* what, const Sophus::SE3d& T) {
  const Eigen::Matrix4d M = T.matrix();
  std::string s;
  char b[40];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) { snprintf(b, sizeof(b), " %.17g", M(i, j)); s += b; }
  rec("%s T_wc:%s", what, s.c_str());
}
}  // namespace

// ---- recording double of Core/Utils/RGBDOdometry.cpp ----
RGBDOdometry::RGBDOdometry(int width_, int height_, float cx_, float cy_, float fx_, float fy_, float distThresh, float angleThresh)
    : lastICPError(0), lastICPCount(width_ * height_), lastRGBError(0), lastRGBCount(width_ * height_), lastSO3Error(0), lastSO3Count(width_ * height_),
      lastA(Eigen::Matrix<double, 6, 6, Eigen::RowMajor>::Zero()), lastb(Eigen::Matrix<double, 6, 1>::Zero()), sobelSize(3), sobelScale(1.0f / 8.0f),
      maxDepthDeltaRGB(0.07f), maxDepthRGB(6.0f), distThres_(distThresh), angleThres_(angleThresh), width(width_), height(height_), cx(cx_), cy(cy_),
      fx(fx_), fy(fy_) {
  rec("RGBDOdometry[%dx%d] created at %p", width, height, (void*)this);
}
RGBDOdometry::~RGBDOdometry() {}
void RGBDOdometry::initICP(GPUTexture* filteredDepth, const float depthCutoff) { rec("RGBDOdometry@%p[%dx%d]::initICP depth=tex%d cutoff=%.9g", (void*)this, width, height, tex_id(filteredDepth), depthCutoff); }
void RGBDOdometry::initICP(GPUTexture* v, GPUTexture* n) { rec("RGBDOdometry@%p[%dx%d]::initICP vertices=tex%d normals=tex%d", (void*)this, width, height, tex_id(v), tex_id(n)); }
void RGBDOdometry::initICPModel(GPUTexture* v, GPUTexture* n, const Sophus::SE3d& T_wc) {
  rec("RGBDOdometry@%p[%dx%d]::initICPModel vertices=tex%d normals=tex%d", (void*)this, width, height, tex_id(v), tex_id(n));
  rec_pose("  initICPModel", T_wc);
}
void RGBDOdometry::initRGB(GPUTexture* rgb) { rec("RGBDOdometry@%p[%dx%d]::initRGB image=tex%d", (void*)this, width, height, tex_id(rgb)); }
void RGBDOdometry::initRGBModel(GPUTexture* rgb) { rec("RGBDOdometry@%p[%dx%d]::initRGBModel image=tex%d", (void*)this, width, height, tex_id(rgb)); }
void RGBDOdometry::initFirstRGB(GPUTexture* rgb) { rec("RGBDOdometry@%p[%dx%d]::initFirstRGB image=tex%d", (void*)this, width, height, tex_id(rgb)); }
void RGBDOdometry::getIncrementalTransformation(Sophus::SE3d& T_wc, const bool& rgbOnly, const float& icpWeight, const bool& pyramid,
                                                const bool& fastOdom, const bool& so3) {
  rec("RGBDOdometry@%p[%dx%d]::getIncrementalTransformation rgbOnly=%d icpWeight=%.9g pyramid=%d fastOdom=%d so3=%d", (void*)this, width, height, (int)rgbOnly,
      icpWeight, (int)pyramid, (int)fastOdom, (int)so3);
  rec_pose("  track in ", T_wc);
  T_wc = T_wc * Sophus::SE3d(efo::se3_from_matrix(g_script.delta));
  rec_pose("  track out", T_wc);
  lastICPError = g_script.icpError;
  lastICPCount = g_script.icpCount;
}
Eigen::MatrixXd RGBDOdometry::getCovariance() {
  rec("RGBDOdometry@%p[%dx%d]::getCovariance", (void*)this, width, height);
  Eigen::Matrix<double, 6, 6, Eigen::RowMajor> c = Eigen::Matrix<double, 6, 6, Eigen::RowMajor>::Zero();
  for (int i = 0; i < 6; ++i) c(i, i) = g_script.covDiag;
  return Eigen::MatrixXd(c);
}

// ---- recording double of Core/Utils/DeformationGraph.cpp (the CHOLMOD-based optimiser) ----
DeformationGraph::DeformationGraph(int k_, std::vector<Eigen::Vector3d>* sv) : k(k_), initialised(false), wRot(1), wReg(10), wCon(100), sourceVertices(sv) {}
DeformationGraph::~DeformationGraph() {}
void DeformationGraph::initialiseGraph(std::vector<Eigen::Vector3d>* g, std::vector<uint64_t>* t) {
  rec("DeformationGraph::initialiseGraph nodes=%d", (int)g->size());
  graphNodes.clear();   // the double keeps the nodes (identity transforms), so that an accepted deformation hands a graph on
  graph.clear();
  graphNodes.resize(g->size());
  sampledGraphTimes = *t;
  for (size_t i = 0; i < g->size(); ++i) {
    graphNodes[i].id = (int)i;
    graphNodes[i].enabled = true;
    graphNodes[i].position = g->at(i);
    graphNodes[i].translation = Eigen::Vector3d::Zero();
    graphNodes[i].rotation.setIdentity();
    graph.push_back(&graphNodes[i]);
  }
  initialised = true;
}
void DeformationGraph::appendVertices(std::vector<uint64_t>* t, uint32_t originalPointEnd) {
  rec("DeformationGraph::appendVertices %d from %u", (int)t->size(), originalPointEnd);
  for (size_t i = originalPointEnd; i < sourceVertices->size(); ++i)      // the constraint sources (Deformation::constrain, Deformation.cpp:120-131)
    rec("  vertex %d time=%lu %.17g %.17g %.17g", (int)i, (unsigned long)t->at(i), (*sourceVertices)[i](0), (*sourceVertices)[i](1), (*sourceVertices)[i](2));
}
void DeformationGraph::setPosesSeq(std::vector<uint64_t>* t, const std::vector<Sophus::SE3d>& T) { rec("DeformationGraph::setPosesSeq %d", (int)T.size()); }
std::vector<GraphNode*>& DeformationGraph::getGraph() { return graph; }
std::vector<uint64_t>& DeformationGraph::getGraphTimes() { return sampledGraphTimes; }
void DeformationGraph::addConstraint(int vertexId, Eigen::Vector3d& target) { rec("DeformationGraph::addConstraint vertex=%d target=%.17g %.17g %.17g", vertexId, target(0), target(1), target(2)); }
void DeformationGraph::addRelativeConstraint(int a, int b) { rec("DeformationGraph::addRelativeConstraint %d %d", a, b); }
void DeformationGraph::clearConstraints() { rec("DeformationGraph::clearConstraints"); }
void DeformationGraph::applyGraphToVertices() { rec("DeformationGraph::applyGraphToVertices"); }
void DeformationGraph::applyGraphToPoses(std::vector<Sophus::SE3d*> p) { rec("DeformationGraph::applyGraphToPoses %d", (int)p.size()); }
bool DeformationGraph::optimiseGraphSparse(float& error, float& meanConsErr, const bool fernMatch, const uint64_t lastDeformTime) {
  rec("DeformationGraph::optimiseGraphSparse fernMatch=%d lastDeformTime=%lu -> %d", (int)fernMatch, (unsigned long)lastDeformTime, g_script.constrainResult);
  error = 0; meanConsErr = 0;
  return g_script.constrainResult != 0;
}

namespace {
struct Frame {
  ElasticFusion* ef;
  std::string out;
};
const char* take(Frame* f) { f->out.swap(glrec::S().log); glrec::S().log.clear(); return f->out.c_str(); }
}  // namespace

extern "C" {
// flags: bit 0 = closeLoops, bit 1 = reloc (the constructor's relocalisation mode, ElasticFusion.cpp:29,79)
void* efe_create(int w, int h, float fx, float fy, float cx, float cy, int timeDelta, int countThresh, float errThresh, float covThresh, int flags,
                 float confidence, float depthCut, float icpThresh, int fastOdom, int so3, int frameToFrameRGB, const char* fileName) {
  Resolution::getInstance(w, h);
  Intrinsics::getInstance(fx, fy, cx, cy);
  Frame* f = new Frame();
  f->ef = new ElasticFusion(timeDelta, countThresh, errThresh, covThresh, (flags & 1) != 0, false, (flags & 2) != 0, 115, confidence, depthCut, icpThresh,
                            fastOdom != 0, 0.3095f, so3 != 0, frameToFrameRGB != 0, fileName);
  return f;
}
int efe_lost(void* p) { return ((Frame*)p)->ef->getLost() ? 1 : 0; }
const char* efe_take_log(void* p) { return take((Frame*)p); }
void efe_destroy(void* p) { Frame* f = (Frame*)p; delete f->ef; delete f; }   // the destructor writes <fileName>.freiburg
const char* efe_process_frame(void* p, const unsigned char* rgb, const unsigned short* depth, long long timestamp, float weightMultiplier, const double* T16) {
  Frame* f = (Frame*)p;
  if (T16) {
    const Sophus::SE3d T(efo::se3_from_matrix(T16));
    f->ef->processFrame(rgb, depth, timestamp, weightMultiplier, &T);
  } else {
    f->ef->processFrame(rgb, depth, timestamp, weightMultiplier, nullptr);
  }
  return take(f);
}
void efe_save_ply(void* p) { ((Frame*)p)->ef->savePly(); }
void efe_get_pose(void* p, double* T16) {
  const Eigen::Matrix4d M = ((Frame*)p)->ef->get_T_wc().matrix();
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T16[i * 4 + j] = M(i, j);
}
int efe_tick(void* p) { return ((Frame*)p)->ef->getTick(); }
// scripting the doubles and the recorder's readbacks
void efe_script_tracker(const double* delta16, float icpError, float icpCount, double covDiag, int constrainResult) {
  std::memcpy(g_script.delta, delta16, sizeof(g_script.delta));
  g_script.icpError = icpError; g_script.icpCount = icpCount; g_script.covDiag = covDiag; g_script.constrainResult = constrainResult;
}
void efe_script_readbacks(int readpixels_fill, unsigned query_result, const unsigned char* buffer, long buffer_bytes) {
  glrec::S().readpixels_fill = readpixels_fill;
  glrec::S().query_result = query_result;
  glrec::S().buffer_data.assign(buffer, buffer + (buffer ? buffer_bytes : 0));
}
// the reference's own setters (ElasticFusion.h:135-183)
void efe_set(void* p, const char* what, float v) {
  ElasticFusion* e = ((Frame*)p)->ef;
  const std::string n = what;
  if (n == "rgbOnly") e->setRgbOnly(v != 0);
  else if (n == "icpWeight") e->setIcpWeight(v);
  else if (n == "pyramid") e->setPyramid(v != 0);
  else if (n == "fastOdom") e->setFastOdom(v != 0);
  else if (n == "so3") e->setSo3(v != 0);
  else if (n == "frameToFrameRGB") e->setFrameToFrameRGB(v != 0);
  else if (n == "confidence") e->setConfidenceThreshold(v);
  else if (n == "depthCutoff") e->setDepthCutoff(v);
}
void efe_queue_readpixels(const unsigned char* data, long bytes) { glrec::S().readpixels_queue.emplace_back(data, data + bytes); }
void efe_queue_query(int n) { glrec::S().query_queue.push_back(n); }
void efe_clear_queues() { glrec::S().readpixels_queue.clear(); glrec::S().query_queue.clear(); }
void efe_script_next_query(int n) { glrec::S().query_once = n; }   // the next "primitives written" query only (e.g. the clean pass's count)
unsigned efe_tid(void* p, const char* name) {
  ElasticFusion* e = ((Frame*)p)->ef;
  const std::string n = name;
  IndexMap& im = e->getIndexMap();
  GPUTexture* t = n == "index" ? im.indexTex() : n == "vertConf" ? im.vertConfTex() : n == "colorTime" ? im.colorTimeTex() : n == "normalRad" ? im.normalRadTex()
                : n == "image" ? im.imageTex() : n == "vertex" ? im.vertexTex() : n == "normal" ? im.normalTex() : n == "time" ? im.timeTex()
                : n == "oldImage" ? im.oldImageTex() : n == "oldVertex" ? im.oldVertexTex() : n == "oldNormal" ? im.oldNormalTex() : n == "oldTime" ? im.oldTimeTex()
                : n == "depth" ? im.depthTex() : nullptr;
  if (!t && e->getTextures().count(n)) t = e->getTextures()[n];
  return t ? t->texture->tid : 0;
}

// ---- the instance's fern database (the compiled Core/Ferns.cpp) driven directly: the three Resize read-backs of every call are queued
// from the caller's arrays, the 80x60 tracker inside findFrame is the scripted double above
static void queue_view(Ferns& f, const unsigned char* rgb3, const float* verts4, const float* norms4) {
  const size_t px = (size_t)f.width * f.height;
  glrec::S().readpixels_queue.clear();
  glrec::S().readpixels_queue.emplace_back(rgb3, rgb3 + px * 3);
  glrec::S().readpixels_queue.emplace_back((const unsigned char*)verts4, (const unsigned char*)verts4 + px * 16);
  glrec::S().readpixels_queue.emplace_back((const unsigned char*)norms4, (const unsigned char*)norms4 + px * 16);
}
static void table_out(Ferns& f, int* t) {
  for (int i = 0; i < f.num; ++i) {
    t[i * 6] = f.conservatory[i].pos(0); t[i * 6 + 1] = f.conservatory[i].pos(1);
    for (int k = 0; k < 4; ++k) t[i * 6 + 2 + k] = f.conservatory[i].rgbd(k);
  }
}
int efe_ferns_num(void* p) { return ((Frame*)p)->ef->getFerns().num; }
void efe_ferns_reseed(void* p, unsigned seed, int* table6) {   // Ferns::generateFerns itself, from a known seed instead of time(0)
  Ferns& f = ((Frame*)p)->ef->getFerns();
  f.conservatory.clear();
  f.random.seed(seed);
  f.generateFerns();
  table_out(f, table6);
}
int efe_ferns_add_frame(void* p, const unsigned char* rgb3, const float* verts4, const float* norms4, const double* T16, int srcTime, float threshold) {
  ElasticFusion* e = ((Frame*)p)->ef;
  Ferns& f = e->getFerns();
  queue_view(f, rgb3, verts4, norms4);
  GPUTexture* t = e->getIndexMap().imageTex();
  const bool r = f.addFrame(t, t, t, Sophus::SE3d(efo::se3_from_matrix(T16)), srcTime, threshold);
  glrec::S().readpixels_queue.clear(); glrec::S().log.clear();
  return r ? 1 : 0;
}
int efe_ferns_find_frame(void* p, const unsigned char* rgb3, const float* verts4, const float* norms4, const double* T16, int time, int lost,
                         double* T_est16, double* cons6, int max_cons, int* n_out) {
  ElasticFusion* e = ((Frame*)p)->ef;
  Ferns& f = e->getFerns();
  queue_view(f, rgb3, verts4, norms4);
  GPUTexture* t = e->getIndexMap().imageTex();
  std::vector<Ferns::SurfaceConstraint> cons;
  const Sophus::SE3d T_est = f.findFrame(cons, Sophus::SE3d(efo::se3_from_matrix(T16)), t, t, t, time, lost != 0);
  glrec::S().readpixels_queue.clear(); glrec::S().log.clear();
  const Eigen::Matrix4d M = T_est.matrix();
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) T_est16[i * 4 + j] = M(i, j);
  *n_out = (int)cons.size();
  for (int i = 0; i < (int)cons.size() && i < max_cons; ++i)
    for (int k = 0; k < 3; ++k) { cons6[i * 6 + k] = cons[i].sourcePoint(k); cons6[i * 6 + 3 + k] = cons[i].targetPoint(k); }
  return f.lastClosest;
}
void efe_set_tick(void* p, int tick) { ((Frame*)p)->ef->setTick(tick); }
int efe_ferns_last_closest(void* p) { return ((Frame*)p)->ef->getFerns().lastClosest; }
int efe_fern_deforms(void* p) { return ((Frame*)p)->ef->getFernDeforms(); }
int efe_ferns_count(void* p) { return (int)((Frame*)p)->ef->getFerns().frames.size(); }
void efe_ferns_frame(void* p, int i, unsigned char* codes, int* good, int* srcTime, double* T16) {
  Ferns& f = ((Frame*)p)->ef->getFerns();
  const Ferns::Frame* s = f.frames.at(i);
  std::memcpy(codes, s->codes, f.num);
  *good = s->goodCodes; *srcTime = s->srcTime;
  const Eigen::Matrix4d M = s->T_wc.matrix();
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) T16[r * 4 + c] = M(r, c);
}
float efe_ferns_block_hd_aware(void* p, int a, int b) {
  Ferns& f = ((Frame*)p)->ef->getFerns();
  return f.blockHDAware(f.frames.at(a), f.frames.at(b));
}
float efe_ferns_photometric_check(void* p, const unsigned char* rgb3, const float* verts4, const double* T_est16, int id) {
  Ferns& f = ((Frame*)p)->ef->getFerns();
  Img<Eigen::Vector4f> verts(f.height, f.width, (Eigen::Vector4f*)verts4);
  Img<Eigen::Matrix<uint8_t, 3, 1>> img(f.height, f.width, (Eigen::Matrix<uint8_t, 3, 1>*)rgb3);
  const Ferns::Frame* s = f.frames.at(id);
  return f.photometricCheck(verts, img, Sophus::SE3d(efo::se3_from_matrix(T_est16)), s->T_wc, s->initRgb);
}
}  // extern "C"
