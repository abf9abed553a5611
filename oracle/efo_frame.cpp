// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// CPU restatement of ElasticFusion::processFrame (Core/ElasticFusion.cpp:270-607) and predict()
// (:621-653) with reloc = false.  Open loop (closeLoops = false) by default; efo_fusion_set_close_loops adds the LOCAL loop
// closure block (:447-527) with a solver callback where Deformation::constrain stands.  The fern branch (:391-444),
// Ferns::addFrame (:601-619) and the deformation-graph sampling (:593-595) are omitted.
#include "efo_common.h"
#include "efo_linalg.h"
#include "efo_api.h"
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

using namespace efo;

// The surface constraints of the local loop closure, ElasticFusion.cpp:485-509: Resize::vertex / Resize::time (Resize.cpp:85-159) =
// NEAREST sample of texel (20a+10, 20b+10) like Resize::image (G8), walked column by column (:488-489); a sample counts when
// 0 < z < maxDepth and the inactive surface has a time there; both world points are T * Vector4d(x, y, z, 1), a 4x4 matrix product
// evaluated left to right.  rows: n x {vert_w_curr xyz, vert_w_est xyz, inactive time, pin}
extern "C" int efo_loop_constraints(const float* vertex4, const uint16_t* oldTime, int width, int height, int consSample, const double* M,
                                    const double* E, float maxDepth, int pin, double* rows) {
  const int cw = width / consSample, ch = height / consSample;
  int n = 0;
  for (int i = 0; i < cw; ++i)
    for (int j = 0; j < ch; ++j) {
      const size_t texel = (size_t)(j * consSample + consSample / 2) * width + (i * consSample + consSample / 2);
      const float* v = &vertex4[texel * 4];
      const uint16_t tm = oldTime[texel];
      if (v[2] > 0 && v[2] < maxDepth && tm > 0) {                                                           // :490-492
        double* row = rows + (size_t)n * 8;
        for (int r = 0; r < 3; ++r) {
          row[r] = ((M[r * 4] * (double)v[0] + M[r * 4 + 1] * (double)v[1]) + M[r * 4 + 2] * (double)v[2]) + M[r * 4 + 3] * 1.0;
          row[3 + r] = ((E[r * 4] * (double)v[0] + E[r * 4 + 1] * (double)v[1]) + E[r * 4 + 2] * (double)v[2]) + E[r * 4 + 3] * 1.0;
        }
        row[6] = (double)tm;
        row[7] = pin ? 1.0 : 0.0;                                                                             // :507-508
        ++n;
      }
    }
  return n;
}

struct efo_fusion {
  efo_fusion_params p;
  efo_cam cam;
  efo_odometry* frameToModel;
  efo_odometry* modelToModel;
  int tick = 1;
  // local loop closure (ElasticFusion.h:288-305)
  int closeLoops = 0, icpCountThresh = 35000, deforms = 0;
  float icpErrThresh = 5e-05f, covThresh = 1e-05f;
  const int consSample = 20;             // ElasticFusion.cpp:62
  efo_loop_solver solver = nullptr;
  void* solverUser = nullptr;
  efo_local_loop loop{};
  // optional trace of the frame loop (efo_fusion_trace): one line per step with its parameters, in the vocabulary
  // tests/test_oracle_vs_reference_frame.py also derives from the compiled reference's transcript
  bool tracing = false;
  std::string trace;
  void tr(const char* fmt, ...) {
    if (!tracing) return;
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    trace += buf;
    trace += '\n';
  }
  void tr_pose(const char* what, const double* M) {
    if (!tracing) return;
    std::string t;
    char b[40];
    for (int i = 0; i < 12; ++i) { snprintf(b, sizeof(b), " %.17g", M[i]); t += b; }
    tr("%s%s", what, t.c_str());
  }
  std::vector<double> loopConstraints;   // n x 8
  std::vector<uint8_t> oldImage;
  std::vector<float> oldVertex, oldNormal;
  std::vector<uint16_t> oldTime;
  std::vector<float> pendingGraph;   // nodes x 16, applied by the next frame's clean (efo_fusion_set_deformation)
  int pendingFern = 0;
  SE3 T_wc = se3_identity();
  const float maxDepthProcessed = 20.0f;  // ElasticFusion.cpp:83
  // "textures"
  std::vector<uint8_t> rgb;              // RGB8 as uploaded
  std::vector<uint8_t> rgba;             // the GL_RGBA texture contents (alpha 255)
  std::vector<uint16_t> depthRaw, depthFiltered;
  std::vector<float> depthMetric, depthMetricFiltered;
  // IndexMap
  std::vector<uint32_t> indexMap;
  std::vector<float> vertConf, colorTime, normRad;
  std::vector<uint8_t> image;
  std::vector<float> vertex, normal;
  std::vector<uint16_t> timeMap;
  // FillIn
  std::vector<uint8_t> fimage;
  std::vector<float> fvertex, fnormal;
  // GlobalModel
  std::vector<float> surfels, surfelsTmp, newUnstable;
  int count = 0;
  float lastWeighting = 0;

  explicit efo_fusion(const efo_fusion_params& pp) : p(pp) {
    cam = efo_cam{p.width, p.height, p.fx, p.fy, p.cx, p.cy};
    frameToModel = efo_odom_create(p.width, p.height, p.cx, p.cy, p.fx, p.fy);
    modelToModel = efo_odom_create(p.width, p.height, p.cx, p.cy, p.fx, p.fy);
    size_t P = (size_t)p.width * p.height;
    rgb.assign(P * 3, 0); rgba.assign(P * 4, 0);
    depthRaw.assign(P, 0); depthFiltered.assign(P, 0);
    depthMetric.assign(P, 0.f); depthMetricFiltered.assign(P, 0.f);
    indexMap.assign(P, 0); vertConf.assign(P * 4, 0.f); colorTime.assign(P * 4, 0.f); normRad.assign(P * 4, 0.f);
    image.assign(P * 4, 0); vertex.assign(P * 4, 0.f); normal.assign(P * 4, 0.f); timeMap.assign(P, 0);
    oldImage.assign(P * 4, 0); oldVertex.assign(P * 4, 0.f); oldNormal.assign(P * 4, 0.f); oldTime.assign(P, 0);
    fimage.assign(P * 4, 0); fvertex.assign(P * 4, 0.f); fnormal.assign(P * 4, 0.f);
    surfels.assign((size_t)p.maxSurfels * 12, 0.f);
    surfelsTmp.assign((size_t)p.maxSurfels * 12, 0.f);
    newUnstable.assign(P * 12, 0.f);
  }
  ~efo_fusion() { efo_odom_destroy(frameToModel); efo_odom_destroy(modelToModel); }

  // ElasticFusion.cpp:447-527 without the fern branch (:391-444, out of scope): the view of the INACTIVE part of the model is
  // registered against the ACTIVE prediction made by predict() at :387; returns true when a deformation was accepted
  void localLoopClosure() {
    loop = efo_local_loop{};
    loopConstraints.clear();
    loop.attempted = 1;
    double M[16];
    pose16(M);
    std::memcpy(loop.T_wc_curr, M, sizeof(M));
    tr("combinedPredict INACTIVE maxDepth=%g conf=%g time=%d maxTime=%d timeDelta=%d", maxDepthProcessed, p.confidence, 0, tick - p.timeDelta, p.timeDelta);
    tr_pose("  pose", M);
    tr("modelToModel.initICPModel vertices=old normals=old");
    tr_pose("  pose", M);
    tr("modelToModel.initRGBModel image=old");
    tr("modelToModel.initICP vertices=pred normals=pred");
    tr("modelToModel.initRGB image=pred");
    tr("modelToModel.track rgbOnly=0 icpWeight=10 pyramid=%d fastOdom=%d so3=0", p.pyramid, p.fastOdom);
    tr("modelToModel.getCovariance");
    efo_combined_predict(&cam, M, surfels.data(), count, maxDepthProcessed, p.confidence, 0, tick - p.timeDelta, p.timeDelta,
                         oldImage.data(), oldVertex.data(), oldNormal.data(), oldTime.data());                 // :451-459, INACTIVE
    efo_odom_init_icp_model(modelToModel, oldVertex.data(), oldNormal.data(), M);                              // :463
    efo_odom_init_rgb_model(modelToModel, oldImage.data());                                                    // :464
    efo_odom_init_icp_maps(modelToModel, vertex.data(), normal.data());                                        // :466
    efo_odom_init_rgb(modelToModel, image.data());                                                             // :467
    double E[16];
    std::memcpy(E, M, sizeof(M));
    efo_odom_track(modelToModel, E, 0, 10.0f, p.pyramid, p.fastOdom, 0);                                       // :471
    std::memcpy(loop.T_wc_est, E, sizeof(E));
    double lastA[36], cov[36];
    efo_odom_stats(modelToModel, loop.stats, lastA, nullptr);
    lu_inverse<double, 6>(lastA, cov);                                                                         // :473, getCovariance
    bool covOk = true;
    for (int i = 0; i < 6; ++i) {
      loop.cov_diag[i] = cov[i * 6 + i];
      if (cov[i * 6 + i] > (double)covThresh) { covOk = false; break; }
    }
    loop.cov_ok = covOk;
    loop.gates_ok = covOk && loop.stats[1] > (float)icpCountThresh && loop.stats[0] < icpErrThresh;            // :483-484
    if (!loop.gates_ok) return;
    // Resize::vertex / Resize::time (Resize.cpp:85-159): NEAREST sample of texel (20a+10, 20b+10) like Resize::image (G8)
    loopConstraints.resize((size_t)(p.width / consSample) * (p.height / consSample) * 8);
    loop.n_constraints = efo_loop_constraints(vertex.data(), oldTime.data(), p.width, p.height, consSample, M, E, maxDepthProcessed, deforms == 0,
                                              loopConstraints.data());
    loopConstraints.resize((size_t)loop.n_constraints * 8);
    if (!solver) return;
    std::vector<float> graph((size_t)1024 * 16, 0.f);
    int nodes = 0;
    if (solver(solverUser, &loop, loopConstraints.data(), loop.n_constraints, graph.data(), &nodes)) {          // :513-514
      loop.applied = 1;
      loop.graph_nodes = nodes;
      deforms += nodes > 0;                                                                                     // :523
      T_wc = se3_from_matrix(E);                                                                                // :525
      pendingGraph.assign(graph.begin(), graph.begin() + (size_t)nodes * 16);
      pendingFern = 0;
    }
  }

  void pose16(double* M) const { M4d m = se3_matrix(T_wc); std::memcpy(M, m.m, sizeof(m.m)); }

  // ElasticFusion::predict(), ElasticFusion.cpp:621-653 (lost == false)
  void predict() {
    double M[16];
    pose16(M);
    tr("combinedPredict ACTIVE maxDepth=%g conf=%g time=%d maxTime=%d timeDelta=%d", maxDepthProcessed, p.confidence, tick, tick, p.timeDelta);
    tr_pose("  pose", M);
    tr("fillIn vertex passthrough=0; normal passthrough=0; image passthrough=%d", p.frameToFrameRGB ? 1 : 0);
    efo_combined_predict(&cam, M, surfels.data(), count, maxDepthProcessed, p.confidence, tick, tick, p.timeDelta,
                         image.data(), vertex.data(), normal.data(), timeMap.data());
    efo_fill_in(&cam, image.data(), vertex.data(), normal.data(), depthFiltered.data(), rgb.data(), 0,
                p.frameToFrameRGB ? 1 : 0, fimage.data(), fvertex.data(), fnormal.data());
  }

  void processFrame(const uint8_t* rgb_in, const uint16_t* depth_in, int64_t, float weightMultiplier, const double* in_T_wc) {
    const size_t P = (size_t)p.width * p.height;
    std::memcpy(depthRaw.data(), depth_in, P * 2);       // :278-280
    std::memcpy(rgb.data(), rgb_in, P * 3);
    for (size_t i = 0; i < P; ++i) { rgba[i * 4] = rgb[i * 3]; rgba[i * 4 + 1] = rgb[i * 3 + 1]; rgba[i * 4 + 2] = rgb[i * 3 + 2]; rgba[i * 4 + 3] = 255; }
    tr("filterDepth cols=%d rows=%d maxD=%g", p.width, p.height, p.depthCut);
    tr("metriciseDepth raw maxD=%g; filtered maxD=%g", p.depthCut, p.depthCut);
    efo_filter_depth(depthRaw.data(), p.width, p.height, p.depthCut, depthFiltered.data());                 // :284
    efo_metricise_depth(depthRaw.data(), p.width, p.height, p.depthCut, depthMetric.data());                // :285
    efo_metricise_depth(depthFiltered.data(), p.width, p.height, p.depthCut, depthMetricFiltered.data());

    if (tick == 1) {  // :290-296
      tr("feedback raw+filtered time=%d maxDepth=%g; initialise", tick, maxDepthProcessed);
      tr("frameToModel.initFirstRGB image=rgb");
      count = efo_seed_map(&cam, rgb.data(), depthMetric.data(), depthMetricFiltered.data(), tick, maxDepthProcessed, surfels.data());
      efo_odom_init_first_rgb(frameToModel, rgba.data());
    } else {
      const SE3 T_prev = T_wc;
      if (!in_T_wc) {
        bool shouldFillIn = !efo_dense_enough(&cam, image.data());  // :304-305
        double M[16];
        pose16(M);
        tr("denseEnough -> %s", shouldFillIn ? "fill" : "model");
        tr("frameToModel.initICPModel vertices=%s normals=%s", shouldFillIn ? "fill" : "pred", shouldFillIn ? "fill" : "pred");
        tr_pose("  pose", M);
        tr("frameToModel.initRGBModel image=%s", (shouldFillIn || p.frameToFrameRGB) ? "fill" : "pred");
        tr("frameToModel.initICP depth=filtered cutoff=%g", maxDepthProcessed);
        tr("frameToModel.initRGB image=rgb");
        tr("frameToModel.track rgbOnly=%d icpWeight=%g pyramid=%d fastOdom=%d so3=%d", p.rgbOnly, p.icpWeight, p.pyramid, p.fastOdom, p.so3);
        efo_odom_init_icp_model(frameToModel, shouldFillIn ? fvertex.data() : vertex.data(),
                                shouldFillIn ? fnormal.data() : normal.data(), M);                           // :310-313
        efo_odom_init_rgb_model(frameToModel, (shouldFillIn || p.frameToFrameRGB) ? fimage.data() : image.data());
        efo_odom_init_icp(frameToModel, depthFiltered.data(), maxDepthProcessed);                            // :317
        efo_odom_init_rgb(frameToModel, rgba.data());                                                        // :318
        efo_odom_track(frameToModel, M, p.rgbOnly, p.icpWeight, p.pyramid, p.fastOdom, p.so3);               // :322-323
        T_wc = se3_from_matrix(M);
      } else {
        T_wc = se3_from_matrix(in_T_wc);
      }
      // velocity weighting, :369-383
      SE3 T_curr_prev = se3_mul(se3_inverse(T_wc), T_prev);
      double tn = std::sqrt(T_curr_prev.t[0] * T_curr_prev.t[0] + T_curr_prev.t[1] * T_curr_prev.t[1] + T_curr_prev.t[2] * T_curr_prev.t[2]);
      float weighting = (float)std::max(tn, se3_log_norm(T_curr_prev));
      float largest = 0.01f, minWeight = 0.5f;
      if (weighting > largest) weighting = largest;
      weighting = std::max(1.0f - (weighting / largest), minWeight) * weightMultiplier;
      lastWeighting = weighting;

      predict();  // :387 (result unused when closeLoops == false; kept for fidelity)
      if (closeLoops) localLoopClosure();
      if (!p.rgbOnly) {
        double Mt[16];
        pose16(Mt);
        tr("predictIndices time=%d maxDepth=%g timeDelta=%d", tick, maxDepthProcessed, p.timeDelta);
        tr("fuse time=%d maxDepth=%g weighting=%.9g", tick, maxDepthProcessed, weighting);
        tr_pose("  pose", Mt);
        tr("predictIndices time=%d maxDepth=%g timeDelta=%d", tick, maxDepthProcessed, p.timeDelta);
        if (!pendingGraph.empty() && !pendingFern)
          tr("synthesizeDepth maxDepth=%g conf=%g time=%d maxTime=%d timeDelta=%d", maxDepthProcessed, p.confidence, tick, tick - p.timeDelta, 65535);
        tr("clean time=%d conf=%g nodes=%d timeDelta=%d maxDepth=%g isFern=%d", tick, p.confidence, (int)(pendingGraph.size() / 16), p.timeDelta,
           maxDepthProcessed, pendingFern);
      }

      if (!p.rgbOnly) {  // :536-585 (trackingOk && !lost always hold without reloc)
        double M[16];
        pose16(M);
        efo_predict_indices(&cam, M, tick, surfels.data(), count, maxDepthProcessed, p.timeDelta, indexMap.data(),
                            vertConf.data(), colorTime.data(), normRad.data());
        int nNew = efo_fuse(&cam, M, tick, rgb.data(), depthMetric.data(), depthMetricFiltered.data(), indexMap.data(),
                            vertConf.data(), colorTime.data(), normRad.data(), maxDepthProcessed, weighting,
                            surfels.data(), count, newUnstable.data());
        efo_predict_indices(&cam, M, tick, surfels.data(), count, maxDepthProcessed, p.timeDelta, indexMap.data(),
                            vertConf.data(), colorTime.data(), normRad.data());
        int room = p.maxSurfels;
        (void)room;
        // a deformation handed over for this frame (rawGraph of ElasticFusion.cpp:558-585; loop-closure detection itself is
        // out of scope): synthesizeDepth of the surfels outside the time window unless a fern was accepted, then clean with the graph
        std::vector<float> synth;
        const int nodes = (int)(pendingGraph.size() / 16);
        if (nodes > 0 && !pendingFern) {
          synth.resize((size_t)cam.cols * cam.rows);
          efo_synthesize_depth(&cam, M, surfels.data(), count, maxDepthProcessed, p.confidence, tick, tick - p.timeDelta, 65535, synth.data());
        }
        count = efo_clean_deform(&cam, M, tick, indexMap.data(), vertConf.data(), colorTime.data(), normRad.data(),
                                 p.confidence, p.timeDelta, maxDepthProcessed, surfels.data(), count, newUnstable.data(), nNew,
                                 nodes > 0 ? pendingGraph.data() : nullptr, nodes, synth.empty() ? nullptr : synth.data(), pendingFern,
                                 surfelsTmp.data());
        pendingGraph.clear();
        surfels.swap(surfelsTmp);
      }
    }
    predict();  // :599
    tick++;     // :603 (lost is never set without reloc)
  }
};

extern "C" {

void efo_fusion_default_params(efo_fusion_params* p) {
  // MainController.cpp:37-43,69-104 front-end defaults, with -o (open loop): timeDelta = INT_MAX/2 (:179-183)
  p->width = 640; p->height = 480;
  p->fx = 528; p->fy = 528; p->cx = 320; p->cy = 240;
  p->timeDelta = 2147483647 / 2;
  p->confidence = 10.0f; p->depthCut = 3.0f; p->icpWeight = 10.0f;
  p->fastOdom = 0; p->so3 = 1; p->frameToFrameRGB = 0; p->pyramid = 1; p->rgbOnly = 0;
  p->maxSurfels = 2 * 1024 * 1024;
}
efo_fusion* efo_fusion_create(const efo_fusion_params* p) { return new efo_fusion(*p); }
void efo_fusion_destroy(efo_fusion* f) { delete f; }
void efo_fusion_process_frame(efo_fusion* f, const uint8_t* rgb, const uint16_t* depth, int64_t ts, float wm, const double* T) {
  f->processFrame(rgb, depth, ts, wm, T);
}
void efo_fusion_get_pose(const efo_fusion* f, double* T) { f->pose16(T); }
int efo_fusion_map_count(const efo_fusion* f) { return f->count; }
void efo_fusion_map_download(const efo_fusion* f, float* s) { std::memcpy(s, f->surfels.data(), (size_t)f->count * 48); }
int efo_fusion_tick(const efo_fusion* f) { return f->tick; }
void efo_set_threads(int n) { efo::threads() = n < 1 ? 1 : n; }
void efo_fusion_set_deformation(efo_fusion* f, const float* graph, int nodes, int isFern) {
  f->pendingGraph.assign(graph, graph + (size_t)nodes * 16);
  f->pendingFern = isFern;
}
void efo_fusion_trace(efo_fusion* f, int on) { f->tracing = on != 0; f->trace.clear(); }
const char* efo_fusion_take_trace(efo_fusion* f) {
  static thread_local std::string out;
  out.swap(f->trace);
  f->trace.clear();
  return out.c_str();
}
void efo_fusion_set_close_loops(efo_fusion* f, int on, int icpCountThresh, float icpErrThresh, float covThresh) {
  f->closeLoops = on; f->icpCountThresh = icpCountThresh; f->icpErrThresh = icpErrThresh; f->covThresh = covThresh;
}
void efo_fusion_set_loop_solver(efo_fusion* f, efo_loop_solver fn, void* user) { f->solver = fn; f->solverUser = user; }
int efo_fusion_local_loop(const efo_fusion* f, efo_local_loop* info, double* constraints, int max_constraints) {
  *info = f->loop;
  const int n = std::min(f->loop.n_constraints, max_constraints);
  if (constraints && n > 0) std::memcpy(constraints, f->loopConstraints.data(), (size_t)n * 8 * sizeof(double));
  return n;
}
// Deformation::sampleGraphModel (Deformation.cpp:232-306, sample.vert + sample.geom): every 5000th surfel -> {position, initTime}
int efo_sample_graph(const float* surfels, int count, float* out4) {
  int n = 0;
  for (int id = 0; id < count; ++id)
    if (id % 5000 == 0) {
      const float* s = surfels + (size_t)id * 12;
      out4[n * 4] = s[0]; out4[n * 4 + 1] = s[1]; out4[n * 4 + 2] = s[2]; out4[n * 4 + 3] = s[6];
      ++n;
    }
  return n;
}
const void* efo_fusion_old_buffer(const efo_fusion* f, int which) {
  switch (which) {
    case 0: return f->oldImage.data();
    case 1: return f->oldVertex.data();
    case 2: return f->oldNormal.data();
    case 3: return f->oldTime.data();
  }
  return nullptr;
}
void efo_fusion_stats(const efo_fusion* f, float* out6) {
  efo_odom_stats(f->frameToModel, out6, nullptr, nullptr);
}
const void* efo_fusion_buffer(const efo_fusion* f, int which) {
  switch (which) {
    case 0: return f->image.data();
    case 1: return f->vertex.data();
    case 2: return f->normal.data();
    case 3: return f->timeMap.data();
    case 4: return f->fimage.data();
    case 5: return f->fvertex.data();
    case 6: return f->fnormal.data();
    case 7: return f->indexMap.data();
    case 8: return f->vertConf.data();
    case 9: return f->colorTime.data();
    case 10: return f->normRad.data();
    case 11: return f->depthFiltered.data();
    case 12: return f->depthMetric.data();
    case 13: return f->depthMetricFiltered.data();
  }
  return nullptr;
}
efo_odometry* efo_fusion_odometry(efo_fusion* f) { return f->frameToModel; }

// ---- linalg exports for known-answer tests ----
void efo_ldlt6(const double* A, const double* b, double* x) { ldlt_solve<double, 6>(A, b, x); }
void efo_ldlt3f(const float* A, const float* b, float* x) { ldlt_solve<float, 3>(A, b, x); }
void efo_polar3(const double* A, double* R) { M3d a; std::memcpy(a.m, A, sizeof(a.m)); M3d r = polar3(a); std::memcpy(R, r.m, sizeof(r.m)); }
void efo_rodrigues(const double* v, double* R) { M3d r = rodrigues(V3d{{v[0], v[1], v[2]}}); std::memcpy(R, r.m, sizeof(r.m)); }
void efo_se3_inverse(const double* T, double* out) { M4d m = se3_matrix(se3_inverse(se3_from_matrix(T))); std::memcpy(out, m.m, sizeof(m.m)); }
double efo_se3_log_norm(const double* T, double* out6) { return se3_log_norm(se3_from_matrix(T), out6); }
float efo_expf_spec(float x) { return efo_expf(x); }
// RGBDOdometry::getCovariance (RGBDOdometry.cpp:573-575): lastA.cast<double>().lu().inverse()
void efo_covariance(const double* lastA36, double* cov36) { lu_inverse<double, 6>(lastA36, cov36); }

}  // extern "C"
