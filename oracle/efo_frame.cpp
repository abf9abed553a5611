// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// CPU restatement of ElasticFusion::processFrame (Core/ElasticFusion.cpp:270-607) and predict()
// (:621-653).  Relocalisation (reloc = true, :326-366, 402-413, 536, 601-604, 624-649) is off unless efo_fusion_set_reloc turns it
// on.  Open loop (closeLoops = false) by default; efo_fusion_set_close_loops adds the LOCAL loop
// closure block (:447-527) with a solver callback where Deformation::constrain stands; efo_fusion_enable_ferns adds the GLOBAL one:
// the fern branch (:391-444: Ferns::findFrame on the mid-frame fill-in, the global deformation with the relative constraints kept from
// local closures) and Ferns::addFrame at the end of the frame (:601-619), over efo_ferns.cpp.  The deformation-graph sampling
// (:593-595) is the solver callback's business (efo_sample_graph).
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
  // global loop closure (ElasticFusion.h:262-266,292-300)
  efo_ferns* ferns = nullptr;
  efo_odometry* fernOdom = nullptr;          // Ferns::rgbd, the 1/8-resolution tracker (Ferns.cpp:36-42)
  float fernThresh = 0.3095f;
  int fernDeforms = 0;
  efo_deform_solver deformSolver = nullptr;
  void* deformUser = nullptr;
  efo_global_loop gloop{};
  // relocalisation (ElasticFusion.h:283-286,311-312): the tracker's verdict on itself, the counter towards "lost", the one-frame
  // probation after a fern brought the pose back
  bool reloc = false, lost = false, lastFrameRecovery = false, trackingOk = true;
  int trackingCount = 0;
  std::vector<double> relativeCons;          // rows of 10, Deformation::Constraint with relative = true
  std::vector<double> trajectory;            // t_T_wc: 16 doubles per processed frame
  std::vector<int64_t> trajectoryTimes;
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
  std::vector<float> renderSource;   // the reference's second vertex buffer as its update pass leaves it (quirk Q14); zeros at first
  int count = 0;
  float lastWeighting = 0;

  explicit efo_fusion(const efo_fusion_params& pp) : p(pp) {
    cam = efo_cam{p.width, p.height, p.fx, p.fy, p.cx, p.cy};
    gloop.closest = -1;
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
  ~efo_fusion() {
    efo_odom_destroy(frameToModel);
    efo_odom_destroy(modelToModel);
    if (fernOdom) efo_odom_destroy(fernOdom);
    if (ferns) efo_ferns_destroy(ferns);
  }

  // the 1/8-resolution views of the fill-in the fern database works on (Ferns.cpp:91-93,178-180: Resize::image / vertex)
  struct FernView { std::vector<uint8_t> img; std::vector<float> verts, norms; };
  FernView lastViews[2];   // what findFrame (0) and addFrame (1) of the last frame saw (efo_fusion_fern_view)
  FernView fernView() const {
    FernView v;
    const size_t px = (size_t)(p.width / 8) * (p.height / 8);
    v.img.resize(px * 4); v.verts.resize(px * 4); v.norms.resize(px * 4);
    efo_resize_nearest(fimage.data(), p.width, p.height, 4, 8, v.img.data());
    efo_resize_nearest(fvertex.data(), p.width, p.height, 16, 8, v.verts.data());
    efo_resize_nearest(fnormal.data(), p.width, p.height, 16, 8, v.norms.data());
    return v;
  }
  // Ferns.cpp:243-258: the stored keyframe is the model, the current view the frame; ICP only in effect (no colour is initialised)
  static void fernTrack(void* user, const float* fv, const float* fn, const double* Tf, const float* cv, const float* cn, double* T, float* err, float* cnt) {
    efo_fusion* f = (efo_fusion*)user;
    f->tr("fernOdom.initICPModel vertices=fern normals=fern");
    f->tr_pose("  pose", Tf);
    f->tr("fernOdom.initICP vertices=view normals=view");
    f->tr("fernOdom.track rgbOnly=0 icpWeight=100 pyramid=0 fastOdom=0 so3=0");
    efo_odom_init_icp_model(f->fernOdom, fv, fn, Tf);
    efo_odom_init_icp_maps(f->fernOdom, cv, cn);
    efo_odom_track(f->fernOdom, T, 0, 100.0f, 0, 0, 0);
    float st[6];
    efo_odom_stats(f->fernOdom, st, nullptr, nullptr);
    *err = st[0];
    *cnt = st[1];
    f->gloop.icp_error = st[0];
    f->gloop.icp_count = st[1];
  }
  // every pose Deformation::constrain deforms along: the keyframes, and for a fern match the trajectory (Deformation.cpp:97-115)
  void gatherPoses(bool withTrajectory, std::vector<double>& poses, std::vector<int64_t>& times) const {
    const int nf = ferns ? efo_ferns_count(ferns) : 0;
    poses.resize((size_t)nf * 16);
    times.resize((size_t)nf);
    for (int i = 0; i < nf; ++i) {
      int t = 0;
      efo_ferns_get_frame(ferns, i, nullptr, nullptr, &t, &poses[(size_t)i * 16], nullptr, nullptr, nullptr);
      times[i] = t;
    }
    if (withTrajectory) {
      poses.insert(poses.end(), trajectory.begin(), trajectory.end());
      times.insert(times.end(), trajectoryTimes.begin(), trajectoryTimes.end());
    }
  }
  void scatterPoses(bool withTrajectory, const std::vector<double>& poses) {
    const int nf = ferns ? efo_ferns_count(ferns) : 0;
    for (int i = 0; i < nf; ++i) efo_ferns_set_frame_pose(ferns, i, &poses[(size_t)i * 16]);
    if (withTrajectory) std::copy(poses.begin() + (size_t)nf * 16, poses.end(), trajectory.begin());
  }
  static void push_row(std::vector<double>& rows, const double* src, const double* target, double srcTime, double targetTime, int relative, int pin) {
    const double r[10] = {src[0], src[1], src[2], target[0], target[1], target[2], srcTime, targetTime, (double)relative, (double)pin};
    rows.insert(rows.end(), r, r + 10);
  }

  // RGBDOdometry::getCovariance of the frame-to-model tracker and the gate of ElasticFusion.cpp:330-337,348-355
  bool covarianceOk() {
    tr("frameToModel.getCovariance");
    double lastA[36], cov[36];
    float st[6];
    efo_odom_stats(frameToModel, st, lastA, nullptr);
    lu_inverse<double, 6>(lastA, cov);
    for (int i = 0; i < 6; ++i)
      if (cov[i * 6 + i] > 1e-04) return false;
    return true;
  }

  // ElasticFusion.cpp:392-445: returns true when a fern was matched AND the global deformation accepted; a lost camera takes the
  // matched keyframe's registration as its pose instead (:411-413)
  bool fernClosure() {
    gloop = efo_global_loop{};
    gloop.attempted = 1;
    gloop.closest = -1;
    double M[16], E[16];
    pose16(M);
    const FernView v = fernView();
    lastViews[0] = v;
    std::vector<double> cons((size_t)128 * 6);
    int n = 0;
    tr("ferns.findFrame time=%d lost=%d", tick, lost ? 1 : 0);
    const int closest = efo_ferns_find_frame(ferns, v.img.data(), 4, v.verts.data(), v.norms.data(), M, tick, lost ? 1 : 0, &fernTrack, this, E, cons.data(), 128,
                                             &n);   // :395-402
    tr("ferns.findFrame -> closest=%d constraints=%d", closest, n);
    gloop.closest = closest;
    gloop.n_constraints = n;
    std::memcpy(gloop.T_wc_recovery, E, sizeof(E));
    if (closest == -1) return false;                                                                          // :410
    if (lost) {                                                                                               // :411-413
      gloop.n_constraints = 0;   // (findFrame's surface constraints are not used on this branch)
      T_wc = se3_from_matrix(E);
      lastFrameRecovery = true;
      return false;
    }
    int fernTime = 0;
    efo_ferns_get_frame(ferns, closest, nullptr, nullptr, &fernTime, nullptr, nullptr, nullptr, nullptr);
    std::vector<double> rows;
    for (int i = 0; i < n; ++i) {                                                                             // :415-422, pinConstraints = true
      push_row(rows, &cons[(size_t)i * 6], &cons[(size_t)i * 6 + 3], tick, fernTime, 0, 0);
      push_row(rows, &cons[(size_t)i * 6 + 3], &cons[(size_t)i * 6 + 3], fernTime, fernTime, 0, 1);
    }
    rows.insert(rows.end(), relativeCons.begin(), relativeCons.end());                                        // :424-426
    tr("global.constrain fernMatch=1 constraints=%d relative=%d", 2 * n, (int)(relativeCons.size() / 10));
    if (!deformSolver) return false;
    std::vector<double> poses;
    std::vector<int64_t> times;
    gatherPoses(true, poses, times);
    std::vector<float> graph((size_t)1024 * 16, 0.f);
    int nodes = 0;
    if (!deformSolver(deformUser, 1, rows.data(), (int)(rows.size() / 10), poses.data(), times.data(), (int)times.size(), graph.data(), &nodes, nullptr, nullptr))
      return false;                                                                                           // :428
    scatterPoses(true, poses);
    T_wc = se3_from_matrix(E);                                                                                // :429
    fernDeforms += nodes > 0;                                                                                 // :439
    gloop.accepted = 1;
    gloop.graph_nodes = nodes;
    pendingGraph.assign(graph.begin(), graph.begin() + (size_t)nodes * 16);
    pendingFern = 1;                                                                                          // fernAccepted, :441,584
    return true;
  }

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
    std::vector<float> graph((size_t)1024 * 16, 0.f);
    int nodes = 0;
    if (deformSolver) {   // Deformation::constrain in full: the keyframe poses follow, relative constraints are left behind (:511-526)
      std::vector<double> rows, poses, rel((size_t)loop.n_constraints * 10);
      for (int i = 0; i < loop.n_constraints; ++i) {                                                           // Deformation.cpp:73-86
        const double* c = &loopConstraints[(size_t)i * 8];
        push_row(rows, c, c + 3, tick, c[6], 0, 0);
        if (c[7] != 0) push_row(rows, c + 3, c + 3, c[6], c[6], 0, 1);
      }
      std::vector<int64_t> times;
      gatherPoses(false, poses, times);
      int nrel = 0;
      tr("local.constrain fernMatch=0 constraints=%d", (int)(rows.size() / 10));
      if (deformSolver(deformUser, 0, rows.data(), (int)(rows.size() / 10), poses.data(), times.data(), (int)times.size(), graph.data(), &nodes, rel.data(), &nrel)) {
        scatterPoses(false, poses);
        loop.applied = 1;
        loop.graph_nodes = nodes;
        deforms += nodes > 0;
        T_wc = se3_from_matrix(E);
        for (int i = 0; i < nrel && nrel >= 3; i += nrel / 3) relativeCons.insert(relativeCons.end(), &rel[(size_t)i * 10], &rel[(size_t)i * 10] + 10);   // :522-524
        pendingGraph.assign(graph.begin(), graph.begin() + (size_t)nodes * 16);
        pendingFern = 0;
      }
      return;
    }
    if (!solver) return;
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

  // ElasticFusion::predict(), ElasticFusion.cpp:621-653: right after a recovery the whole model is rendered (time = 0: nothing is
  // "inactive"); while lost the fill-in passes the raw frame through
  void predict() {
    double M[16];
    pose16(M);
    const int time = lastFrameRecovery ? 0 : tick;
    tr("combinedPredict ACTIVE maxDepth=%g conf=%g time=%d maxTime=%d timeDelta=%d", maxDepthProcessed, p.confidence, time, tick, p.timeDelta);
    tr_pose("  pose", M);
    tr("fillIn vertex passthrough=%d; normal passthrough=%d; image passthrough=%d", lost ? 1 : 0, lost ? 1 : 0, (lost || p.frameToFrameRGB) ? 1 : 0);
    efo_combined_predict(&cam, M, surfels.data(), count, maxDepthProcessed, p.confidence, time, tick, p.timeDelta,
                         image.data(), vertex.data(), normal.data(), timeMap.data());
    efo_fill_in(&cam, image.data(), vertex.data(), normal.data(), depthFiltered.data(), rgb.data(), lost ? 1 : 0,
                (lost || p.frameToFrameRGB) ? 1 : 0, fimage.data(), fvertex.data(), fnormal.data());
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
      trackingOk = true;                                                                                     // :300
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
        float st[6];
        efo_odom_stats(frameToModel, st, nullptr, nullptr);
        trackingOk = !reloc || st[0] < 1e-04;                                                                // :326, lastICPError
        if (reloc) {                                                                                         // :328-366
          if (!lost) {
            if (!covarianceOk()) trackingOk = false;
            if (!trackingOk) {
              trackingCount++;
              if (trackingCount > 10) lost = true;
            } else {
              trackingCount = 0;
            }
          } else if (lastFrameRecovery) {
            if (!covarianceOk()) trackingOk = false;
            if (trackingOk) {
              lost = false;
              trackingCount = 0;
            }
            lastFrameRecovery = false;
          }
        }
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
      bool fernAccepted = false;
      if (closeLoops) lastFrameRecovery = false;                                                             // :393
      if (closeLoops && ferns) fernAccepted = fernClosure();                                                 // :392-445
      if (!lost && closeLoops && !(fernAccepted && !pendingGraph.empty())) localLoopClosure();               // :447: rawGraph.size() == 0
      else loop = efo_local_loop{};
      const bool fuseThis = !p.rgbOnly && trackingOk && !lost;                                               // :536
      if (fuseThis) {
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

      if (fuseThis) {  // :536-585
        double M[16];
        pose16(M);
        efo_predict_indices(&cam, M, tick, surfels.data(), count, maxDepthProcessed, p.timeDelta, indexMap.data(),
                            vertConf.data(), colorTime.data(), normRad.data());
        int nNew = efo_fuse(&cam, M, tick, rgb.data(), depthMetric.data(), depthMetricFiltered.data(), indexMap.data(),
                            vertConf.data(), colorTime.data(), normRad.data(), maxDepthProcessed, weighting,
                            surfels.data(), count, newUnstable.data());
        // the update pass writes every surfel into the OTHER vertex buffer (GlobalModel.cpp:458-524), which clean never touches:
        // that buffer is what GlobalModel::downloadMap reads afterwards (quirk Q14)
        if (renderSource.size() < (size_t)p.maxSurfels * 12) renderSource.assign((size_t)p.maxSurfels * 12, 0.f);
        std::memcpy(renderSource.data(), surfels.data(), (size_t)count * 48);
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
      } else {
        pendingGraph.clear();   // rawGraph is a local of processFrame: a deformation accepted in a frame that does not fuse is never applied
      }
    }
    {           // :588-589
      double Mt[16];
      pose16(Mt);
      trajectory.insert(trajectory.end(), Mt, Mt + 16);
      trajectoryTimes.push_back(tick);
    }
    predict();  // :599
    if (lost) return;   // :601-604: neither a keyframe nor a tick while the camera is lost
    if (ferns) {   // processFerns, :609-618
      double Mt[16];
      pose16(Mt);
      const FernView v = fernView();
      lastViews[1] = v;
      const int kept = efo_ferns_add_frame(ferns, v.img.data(), 4, v.verts.data(), v.norms.data(), Mt, tick, fernThresh);
      tr("ferns.addFrame time=%d -> %d", tick, kept);
    }
    tick++;     // :603
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
void efo_fusion_get_pose_qt(const efo_fusion* f, double* q4_t3) {   // the pose as held: quaternion xyzw + translation
  for (int i = 0; i < 4; ++i) q4_t3[i] = f->T_wc.q[i];
  for (int i = 0; i < 3; ++i) q4_t3[4 + i] = f->T_wc.t[i];
}
int efo_fusion_map_count(const efo_fusion* f) { return f->count; }
void efo_fusion_map_download(const efo_fusion* f, float* s) { std::memcpy(s, f->surfels.data(), (size_t)f->count * 48); }
// GlobalModel::downloadMap as the reference has it (GlobalModel.cpp:673-706): vbos[renderSource] truncated to the post-clean count
void efo_fusion_map_download_reference(const efo_fusion* f, float* s) {
  std::memset(s, 0, (size_t)f->count * 48);
  if (!f->renderSource.empty()) std::memcpy(s, f->renderSource.data(), (size_t)f->count * 48);
}
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
void efo_fusion_set_tick(efo_fusion* f, int tick) { f->tick = tick; }
// TEST HOOK — the oracle's side of the engine's checkpoint / resume pair (ef_map_upload + ef_restore_state, csrc/ef_context.hip): a map brought
// in from outside (12 floats per surfel), the tick, the pose as held (quaternion xyzw + translation) and the frame processed last — its
// pre-processing and SO(3) reference image are re-done (what processFrame left of it: filtered depth for the fill-in, initFirstRGB's target),
// then predict() as at the end of a frame.  What bench.py --preseed (BASELINE configs[2]) and the one-frame harness start from.
void efo_fusion_restore(efo_fusion* f, const float* surfels12, int count, int tick, const double* q4_t3, const uint8_t* rgb_prev, const uint16_t* depth_prev) {
  const efo_fusion_params& p = f->p;
  const size_t P = (size_t)p.width * p.height;
  std::memcpy(f->surfels.data(), surfels12, (size_t)count * 48);
  f->count = count;
  std::memcpy(f->depthRaw.data(), depth_prev, P * 2);
  std::memcpy(f->rgb.data(), rgb_prev, P * 3);
  for (size_t i = 0; i < P; ++i) { f->rgba[i * 4] = f->rgb[i * 3]; f->rgba[i * 4 + 1] = f->rgb[i * 3 + 1]; f->rgba[i * 4 + 2] = f->rgb[i * 3 + 2]; f->rgba[i * 4 + 3] = 255; }
  efo_filter_depth(f->depthRaw.data(), p.width, p.height, p.depthCut, f->depthFiltered.data());
  efo_metricise_depth(f->depthRaw.data(), p.width, p.height, p.depthCut, f->depthMetric.data());
  efo_metricise_depth(f->depthFiltered.data(), p.width, p.height, p.depthCut, f->depthMetricFiltered.data());
  efo_odom_init_first_rgb(f->frameToModel, f->rgba.data());
  for (int i = 0; i < 4; ++i) f->T_wc.q[i] = q4_t3[i];
  for (int i = 0; i < 3; ++i) f->T_wc.t[i] = q4_t3[4 + i];
  f->tick = tick;
  f->lost = f->lastFrameRecovery = false;
  f->trackingOk = true;
  f->trackingCount = 0;
  f->predict();
}
void efo_fusion_set_reloc(efo_fusion* f, int on) { f->reloc = on != 0; }
// {lost, trackingOk of the last frame, trackingCount, lastFrameRecovery}
void efo_fusion_reloc_state(const efo_fusion* f, int* out4) {
  out4[0] = f->lost; out4[1] = f->trackingOk; out4[2] = f->trackingCount; out4[3] = f->lastFrameRecovery;
}
void efo_fusion_enable_ferns(efo_fusion* f, int num, float photoThresh, float fernThresh, unsigned seed) {
  const efo_fusion_params& p = f->p;
  f->ferns = efo_ferns_create(num, (int)(p.depthCut * 1000), photoThresh, p.width, p.height, p.fx, p.fy, p.cx, p.cy, seed);   // ElasticFusion.cpp:53
  f->fernOdom = efo_odom_create(p.width / 8, p.height / 8, p.cx / 8, p.cy / 8, p.fx / 8, p.fy / 8);                          // Ferns.cpp:36-42
  f->fernThresh = fernThresh;
}
efo_ferns* efo_fusion_ferns(efo_fusion* f) { return f->ferns; }
int efo_fusion_fern_view(const efo_fusion* f, int which, uint8_t* rgba, float* verts4, float* norms4) {
  const efo_fusion::FernView& v = f->lastViews[which ? 1 : 0];
  if (v.img.empty()) return 0;
  std::memcpy(rgba, v.img.data(), v.img.size());
  std::memcpy(verts4, v.verts.data(), v.verts.size() * 4);
  std::memcpy(norms4, v.norms.data(), v.norms.size() * 4);
  return 1;
}
void efo_fusion_set_deform_solver(efo_fusion* f, efo_deform_solver fn, void* user) { f->deformSolver = fn; f->deformUser = user; }
void efo_fusion_global_loop(const efo_fusion* f, efo_global_loop* info) { *info = f->gloop; }
int efo_fusion_relative_constraints(const efo_fusion* f, double* rows10, int max_rows) {
  const int n = std::min((int)(f->relativeCons.size() / 10), max_rows);
  if (rows10 && n > 0) std::memcpy(rows10, f->relativeCons.data(), (size_t)n * 10 * sizeof(double));
  return (int)(f->relativeCons.size() / 10);
}
int efo_fusion_trajectory(const efo_fusion* f, double* poses16, int max_poses) {
  const int n = std::min((int)f->trajectoryTimes.size(), max_poses);
  if (poses16 && n > 0) std::memcpy(poses16, f->trajectory.data(), (size_t)n * 16 * sizeof(double));
  return (int)f->trajectoryTimes.size();
}
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
