// libefusion.so: class ElasticFusion (include/ElasticFusion.h) over the C ABI of libefusion_hip.so.
// Host code only; the per-frame script, kernels and device state are behind ef_process_frame (ef_context.hip).
#include "../../include/ElasticFusion.h"

#include <cassert>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "../../include/ef_hip.h"
#include "../../include/efusion_klg.hpp"
#include "ef_linalg_dev.hpp"

// ---- singletons (Core/Utils/Resolution.h, Intrinsics.h): first call fixes the values for the process ----
Resolution::Resolution(int width, int height) : imgWidth(width), imgHeight(height), imgNumPixels(width * height) {
  if (!(width > 0 && height > 0)) throw std::runtime_error("You haven't initialised the Resolution class!");
}
const Resolution& Resolution::getInstance(int width, int height) {
  static const Resolution instance(width, height);
  return instance;
}
Intrinsics::Intrinsics(float fx, float fy, float cx, float cy) : fx_(fx), fy_(fy), cx_(cx), cy_(cy) {
  if (!(fx != 0 && fy != 0)) throw std::runtime_error("You haven't initialised the Intrinsics class!");
}
const Intrinsics& Intrinsics::getInstance(float fx, float fy, float cx, float cy) {
  static const Intrinsics instance(fx, fy, cx, cy);
  return instance;
}

namespace efusion {

namespace {
ef_ctx* C(void* p) { return static_cast<ef_ctx*>(p); }
void chk(int rc, void* ctx, const char* what) {
  if (rc != EF_OK) {
    const char* m = ef_last_error(C(ctx));
    throw std::runtime_error(std::string(what) + ": libefusion_hip error " + std::to_string(rc) + (m ? std::string(": ") + m : ""));
  }
}
}  // namespace

void ef_ctx_deleter::operator()(void* p) const { if (p) ef_destroy(C(p)); }

SE3d SE3d::fromMatrix(const double* M) {
  const efl::SE3 T = efl::se3_from_matrix(M);   // Sophus::SE3d(Matrix4d): quaternion from the rotation block, normalised
  SE3d r;
  std::memcpy(r.q, T.q, sizeof(r.q));
  std::memcpy(r.t, T.t, sizeof(r.t));
  return r;
}
void SE3d::matrix(double* out16) const {
  efl::SE3 T;
  std::memcpy(T.q, q, sizeof(q));
  std::memcpy(T.t, t, sizeof(t));
  efl::se3_matrix(T, out16);
}

ElasticFusion::ElasticFusion(const int timeDelta_, const int countThresh, const float errThresh, const float covThresh,
                             const bool closeLoops, const bool iclnuim_, const bool reloc, const float photoThresh,
                             const float confidence, const float depthCut, const float icpThresh, const bool fastOdom,
                             const float fernThresh, const bool so3, const bool frameToFrameRGB, const std::string fileName,
                             const int device)
    : saveFilename(fileName), timeDelta(timeDelta_), confidenceThreshold(confidence), closeLoops(closeLoops), iclnuim(iclnuim_) {
  ef_config cfg;
  ef_default_config(&cfg);
  cfg.width = Resolution::getInstance().width();
  cfg.height = Resolution::getInstance().height();
  cfg.fx = Intrinsics::getInstance().fx();
  cfg.fy = Intrinsics::getInstance().fy();
  cfg.cx = Intrinsics::getInstance().cx();
  cfg.cy = Intrinsics::getInstance().cy();
  cfg.time_delta = timeDelta_;
  cfg.confidence = confidence;
  cfg.depth_cut = depthCut;
  cfg.icp_weight = icpThresh;
  cfg.fast_odom = fastOdom;
  cfg.so3 = so3;
  cfg.frame_to_frame_rgb = frameToFrameRGB;
  cfg.close_loops = closeLoops;
  cfg.device = device;
  ef_ctx* c = nullptr;
  chk(ef_create(&cfg, &c), nullptr, "ElasticFusion::ElasticFusion");
  ctx.reset(c);
  if (closeLoops) {   // the reference's closed-loop mode: global (fern) closure, then local closure, built-in optimiser for both
    chk(ef_set_loop_thresholds(c, countThresh, errThresh, covThresh), c, "ElasticFusion::ElasticFusion");
    chk(ef_use_builtin_loop_solver(c, 1), c, "ElasticFusion::ElasticFusion");
    // Ferns(500, depthCut * 1000, photoThresh), :53.  The 1/8-resolution registration needs width and height to be multiples of 32
    // (include/ef_hip.h); at other sizes (320x240: 40x30 views) the constructor does not throw: the context closes LOCAL loops only and
    // keeps no fern database (getFerns() stays empty, reloc has nothing to relocalise against)
    if (cfg.width % 32 == 0 && cfg.height % 32 == 0)
      chk(ef_enable_global_closure(c, 500, photoThresh, fernThresh, 0u), c, "ElasticFusion::ElasticFusion");
  }
  if (reloc) chk(ef_set_relocalisation(c, 1), c, "ElasticFusion::ElasticFusion");   // :326-366, 411-413: lost / found through the fern database
  // drop-in: getGlobalModel().downloadMap() and savePly() return what the reference's return (the pre-clean buffer, quirk Q14)
  chk(ef_set_reference_download(c, 1), c, "ElasticFusion::ElasticFusion");
  indexMap.ctx = globalModel.ctx = localDeformation.ctx = c;
  localDeformation.closeLoops = closeLoops;
  indexMap.w = cfg.width;
  indexMap.h = cfg.height;
  if (!saveFilename.empty()) {  // the reference truncates <file>.freiburg in its constructor (ElasticFusion.cpp:97-102)
    if (FILE* f = std::fopen((saveFilename + ".freiburg").c_str(), "w")) std::fclose(f);
  }
}

ElasticFusion::~ElasticFusion() {
  if (!ctx) return;
  try {
    if (iclnuim) savePly();   // ElasticFusion.cpp:108-110
  } catch (...) {
  }
  if (saveFilename.empty()) return;
  if (!iclnuim) {
    (void)ef_save_freiburg(C(ctx.get()), (saveFilename + ".freiburg").c_str());
    return;
  }
  // iclnuim: the timestamps are written as they came, not as microseconds / 1e6 (ElasticFusion.cpp:124-128)
  int n = 0;
  if (ef_get_trajectory(C(ctx.get()), nullptr, nullptr, 0x7fffffff, &n) != EF_OK) return;   // one logged pose per processed frame
  std::vector<double> T((size_t)(n > 0 ? n : 1) * 16);
  std::vector<int64_t> ts((size_t)(n > 0 ? n : 1));
  if (ef_get_trajectory(C(ctx.get()), T.data(), ts.data(), n, &n) != EF_OK) return;
  for (int i = 0; i < n; ++i) ts[i] *= 1000000;   // ef_write_freiburg divides by 1e6 and prints six decimals, which is the reference's format here
  (void)ef_write_freiburg((saveFilename + ".freiburg").c_str(), T.data(), ts.data(), n);
}

void ElasticFusion::processFrame(const uint8_t* rgb, const uint16_t* depth, const int64_t& timestamp, const float weightMultiplier,
                                 const SE3d* in_T_wc) {
  double M[16];
  if (in_T_wc) in_T_wc->matrix(M);
  chk(ef_process_frame(C(ctx.get()), rgb, depth, timestamp, weightMultiplier, in_T_wc ? M : nullptr), ctx.get(), "processFrame");
  ef_reloc_state rs;
  chk(ef_get_relocalisation(C(ctx.get()), &rs), ctx.get(), "processFrame");
  lost = rs.lost != 0;
  if (closeLoops) {
    ef_ctx* c = C(ctx.get());
    ef_closure* cl = ef_get_closure(c);
    ef_global_loop G;
    chk(ef_get_global_loop(c, &G), c, "processFrame");
    const ef_local_loop& L = getLocalLoop();
    const int n_frames = cl ? ef_ferns_count(ef_closure_ferns(cl)) : 0;
    auto rows_to_constraints = [&](bool fern) {
      std::vector<SurfaceConstraint> out;
      if (fern) {
        const int n = ef_closure_last_rows(cl, nullptr, 0, nullptr, nullptr);
        std::vector<ef_graph_constraint> rows((size_t)(n > 0 ? n : 1));
        ef_closure_last_rows(cl, rows.data(), n, nullptr, nullptr);
        for (int i = 0; i < n; ++i)
          if (!rows[i].relative && !rows[i].pin) out.push_back(SurfaceConstraint{{rows[i].src[0], rows[i].src[1], rows[i].src[2]}, {rows[i].target[0], rows[i].target[1], rows[i].target[2]}});
      } else {
        std::vector<double> cons((size_t)(L.n_constraints > 0 ? L.n_constraints : 1) * 8);
        int n = 0;
        ef_local_loop tmp;
        chk(ef_get_local_loop(c, &tmp, cons.data(), L.n_constraints, &n), c, "processFrame");
        for (int i = 0; i < n; ++i) out.push_back(SurfaceConstraint{{cons[i * 8], cons[i * 8 + 1], cons[i * 8 + 2]}, {cons[i * 8 + 3], cons[i * 8 + 4], cons[i * 8 + 5]}});
      }
      return out;
    };
    if (G.accepted) {   // ElasticFusion.cpp:428-441
      double Tf[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
      if (cl && G.closest >= 0) ef_ferns_get_frame(ef_closure_ferns(cl), G.closest, nullptr, nullptr, nullptr, Tf, nullptr, nullptr, nullptr);
      poseMatches.push_back(PoseMatch{G.closest, n_frames, SE3d::fromMatrix(Tf), SE3d::fromMatrix(G.T_wc_recovery), rows_to_constraints(true), true});
      fernDeforms += G.graph_nodes > 0 ? 1 : 0;
    } else if (L.applied) {   // :514-526
      poseMatches.push_back(PoseMatch{n_frames - 1, n_frames, SE3d::fromMatrix(L.T_wc_est), SE3d::fromMatrix(L.T_wc_curr), rows_to_constraints(false), false});
      deforms += L.graph_nodes > 0 ? 1 : 0;
    }
  }
}

const FernsView& ElasticFusion::getFerns() {
  fernsView.frames.clear();
  fernsView.lastClosest = -1;
  ef_closure* cl = ef_get_closure(C(ctx.get()));
  if (!cl) return fernsView;
  ef_ferns* f = ef_closure_ferns(cl);
  const int n = ef_ferns_count(f);
  for (int i = 0; i < n; ++i) {
    double T[16];
    int t = 0;
    ef_ferns_get_frame(f, i, nullptr, nullptr, &t, T, nullptr, nullptr, nullptr);
    fernsView.frames.push_back(FernFrame{i, t, SE3d::fromMatrix(T)});
  }
  fernsView.lastClosest = ef_ferns_last_closest(f);
  return fernsView;
}

void DeformationView::getRawGraph(std::vector<float>& nodes4) {
  nodes4.assign((size_t)1024 * 4, 0.f);
  int n = 0;
  chk(ef_sample_graph(C(ctx), nodes4.data(), 1023, &n), ctx, "getRawGraph");
  nodes4.resize((size_t)n * 4);
}

const std::vector<GraphNode*>& DeformationView::getGraph() {
  constexpr int K = 4;   // DeformationGraph's k (Deformation.cpp:23)
  nodes.clear();
  node_ptrs.clear();
  if (!closeLoops) return node_ptrs;            // sampleGraphModel only runs in closed-loop mode (ElasticFusion.cpp:592-594)
  std::vector<float> raw;
  getRawGraph(raw);
  const int n = (int)(raw.size() / 4);
  if (n <= K) return node_ptrs;                 // Deformation.cpp:283: no graph from k samples or fewer
  nodes.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    GraphNode& g = nodes[(size_t)i];
    g.id = i;
    g.enabled = true;
    for (int r = 0; r < 3; ++r) { g.position(r) = raw[(size_t)i * 4 + r]; g.translation(r) = 0.0; }
    for (int r = 0; r < 9; ++r) g.rotation.m[r] = (r % 4 == 0) ? 1.0 : 0.0;
    // sequence neighbours (DeformationGraph.cpp:239-266): the first K/2 and last K/2 nodes take the K others of the first / last K + 1,
    // everyone else K/2 on either side
    if (i < K / 2) { for (int q = 0; q < K + 1; ++q) if (q != i) g.neighbours.push_back(q); }
    else if (i >= n - K / 2) { for (int q = n - (K + 1); q < n; ++q) if (q != i) g.neighbours.push_back(q); }
    else { for (int q = 0; q < K / 2; ++q) { g.neighbours.push_back(i - (q + 1)); g.neighbours.push_back(i + (q + 1)); } }
  }
  for (GraphNode& g : nodes) node_ptrs.push_back(&g);
  return node_ptrs;
}

void ElasticFusion::predict() { chk(ef_predict(C(ctx.get())), ctx.get(), "predict"); }

void ElasticFusion::setLoopSolver(ef_loop_solver fn, void* user) { chk(ef_set_loop_solver(C(ctx.get()), fn, user), ctx.get(), "setLoopSolver"); }
void ElasticFusion::useBuiltinLoopSolver(bool on) { chk(ef_use_builtin_loop_solver(C(ctx.get()), on), ctx.get(), "useBuiltinLoopSolver"); }
const ef_local_loop& ElasticFusion::getLocalLoop() {
  chk(ef_get_local_loop(C(ctx.get()), &localLoop, nullptr, 0, nullptr), ctx.get(), "getLocalLoop");
  return localLoop;
}

const OdometryStats& ElasticFusion::getModelToModel() {
  float s[6];
  if (closeLoops) {   // the model-to-model tracker of the local loop closure (ElasticFusion.h:280)
    const ef_local_loop& L = getLocalLoop();
    stats.lastICPError = L.stats[0]; stats.lastICPCount = L.stats[1]; stats.lastRGBError = L.stats[2];
    stats.lastRGBCount = L.stats[3]; stats.lastSO3Error = L.stats[4]; stats.lastSO3Count = L.stats[5];
    return stats;
  }
  chk(ef_get_tracking_stats(C(ctx.get()), s, stats.lastA, stats.lastb), ctx.get(), "getModelToModel");
  stats.lastICPError = s[0]; stats.lastICPCount = s[1]; stats.lastRGBError = s[2];
  stats.lastRGBCount = s[3]; stats.lastSO3Error = s[4]; stats.lastSO3Count = s[5];
  return stats;
}

void ElasticFusion::setRgbOnly(const bool& v) { chk(ef_set_rgb_only(C(ctx.get()), v), ctx.get(), "setRgbOnly"); }
void ElasticFusion::setIcpWeight(const float& v) { chk(ef_set_icp_weight(C(ctx.get()), v), ctx.get(), "setIcpWeight"); }
void ElasticFusion::setPyramid(const bool& v) { chk(ef_set_pyramid(C(ctx.get()), v), ctx.get(), "setPyramid"); }
void ElasticFusion::setFastOdom(const bool& v) { chk(ef_set_fast_odom(C(ctx.get()), v), ctx.get(), "setFastOdom"); }
void ElasticFusion::setSo3(const bool& v) { chk(ef_set_so3(C(ctx.get()), v), ctx.get(), "setSo3"); }
void ElasticFusion::setFrameToFrameRGB(const bool& v) { chk(ef_set_frame_to_frame_rgb(C(ctx.get()), v), ctx.get(), "setFrameToFrameRGB"); }
void ElasticFusion::setConfidenceThreshold(const float& v) {
  chk(ef_set_confidence_threshold(C(ctx.get()), v), ctx.get(), "setConfidenceThreshold");
  confidenceThreshold = v;
}
void ElasticFusion::setFernThresh(const float& v) {
  if (ef_closure* cl = ef_get_closure(C(ctx.get()))) ef_closure_set_fern_thresh(cl, v);
}
void ElasticFusion::setDepthCutoff(const float& v) { chk(ef_set_depth_cutoff(C(ctx.get()), v), ctx.get(), "setDepthCutoff"); }

const int& ElasticFusion::getTick() {
  chk(ef_get_tick(C(ctx.get()), &tick), ctx.get(), "getTick");
  return tick;
}
void ElasticFusion::setTick(const int& val) { chk(ef_set_tick(C(ctx.get()), val), ctx.get(), "setTick"); }

const SE3d& ElasticFusion::get_T_wc_pod() {
  double M[16];
  chk(ef_get_pose(C(ctx.get()), M), ctx.get(), "get_T_wc");
  T_wc = SE3d::fromMatrix(M);
  return T_wc;
}

void ElasticFusion::savePly() { chk(ef_save_ply(C(ctx.get()), (saveFilename + ".ply").c_str()), ctx.get(), "savePly"); }
void ElasticFusion::setReferenceDownload(bool on) { chk(ef_set_reference_download(C(ctx.get()), on), ctx.get(), "setReferenceDownload"); }
void ElasticFusion::synchronize() { chk(ef_synchronize(C(ctx.get())), ctx.get(), "synchronize"); }

unsigned int GlobalModelView::lastCount() {
  uint32_t n = 0;
  chk(ef_map_count(C(ctx), &n), ctx, "lastCount");
  return n;
}
std::vector<float> GlobalModelView::downloadMap() {
  uint32_t n = 0;
  chk(ef_map_count(C(ctx), &n), ctx, "downloadMap");
  std::vector<float> m((size_t)n * 12);
  if (n) chk(ef_map_download(C(ctx), m.data(), n, &n), ctx, "downloadMap");
  m.resize((size_t)n * 12);
  return m;
}

std::vector<uint8_t> IndexMapView::image() {
  std::vector<uint8_t> v((size_t)w * h * 4);
  chk(ef_get_image(C(ctx), EF_IMG_PREDICT_IMAGE, v.data(), v.size()), ctx, "imageTex");
  return v;
}
std::vector<float> IndexMapView::vertex() {
  std::vector<float> v((size_t)w * h * 4);
  chk(ef_get_image(C(ctx), EF_IMG_PREDICT_VERTEX, v.data(), v.size() * 4), ctx, "vertexTex");
  return v;
}
std::vector<float> IndexMapView::normal() {
  std::vector<float> v((size_t)w * h * 4);
  chk(ef_get_image(C(ctx), EF_IMG_PREDICT_NORMAL, v.data(), v.size() * 4), ctx, "normalTex");
  return v;
}
std::vector<uint16_t> IndexMapView::time() {
  std::vector<uint16_t> v((size_t)w * h);
  chk(ef_get_image(C(ctx), EF_IMG_PREDICT_TIME, v.data(), v.size() * 2), ctx, "timeTex");
  return v;
}

}  // namespace efusion

// ---- C API of the .klg reader (include/efusion_klg.hpp) ----
namespace { thread_local std::string g_klg_error; }
extern "C" {
void* efk_open(const char* file, int width, int height, int deliver_last_frame, int flip_colors) {
  try {
    auto* r = new efusion::KlgReader(file, width, height);
    r->deliverLastFrame = deliver_last_frame != 0;
    r->flipColors = flip_colors != 0;
    return r;
  } catch (const std::exception& e) { g_klg_error = e.what(); return nullptr; }
}
void efk_close(void* reader) { delete (efusion::KlgReader*)reader; }
int efk_num_frames(void* reader) { return ((efusion::KlgReader*)reader)->getNumFrames(); }
int efk_has_more(void* reader) { return ((efusion::KlgReader*)reader)->hasMore() ? 1 : 0; }
int efk_next(void* reader, int64_t* timestamp, uint16_t* depth, uint8_t* rgb) {
  auto* r = (efusion::KlgReader*)reader;
  try { r->getNext(); } catch (const std::exception& e) { g_klg_error = e.what(); return 0; }
  if (timestamp) *timestamp = r->timestamp;
  if (depth) std::memcpy(depth, r->depth.data(), r->depth.size());
  if (rgb) std::memcpy(rgb, r->rgb.data(), r->rgb.size());
  return 1;
}
const char* efk_last_error(void) { return g_klg_error.c_str(); }
}
