// libefusion.so — the reference's C++ API surface (class ElasticFusion, Core/ElasticFusion.h:40-255) re-implemented
// over the C ABI of libefusion_hip.so (include/ef_hip.h).  Written from scratch for this repository; only the PUBLIC
// names, argument meanings and defaults follow the reference so that its front-end (MainController.cpp:178-194,
// 222-254,262-500,520) compiles against it with the GL/Pangolin-typed getters removed.
//
// What is different, and why:
//  * Sophus::SE3d / Eigen are not vendored by the reference checkout (third-party/* empty) and are absent here, so
//    poses cross this boundary as Sophus-layout PODs (unit quaternion x,y,z,w + translation) with a row-major
//    matrix() accessor.  When <sophus/se3.hpp> is on the include path, define EFUSION_USE_SOPHUS and processFrame / get_T_wc
//    take / return real Sophus::SE3d, as in the reference (header-inline conversions; the library itself never sees Sophus).
//  * getIndexMap()/getGlobalModel()/getModelToModel() return small HBM-backed facades with the members the
//    front-end actually reads (lastCount, lastICPError, lastICPCount, downloadMap, host copies of the predicted
//    images); there is no GL texture or VBO behind them.
//  * closeLoops = true is the reference's closed-loop mode: every frame the fern-based GLOBAL closure (ElasticFusion.cpp:392-445:
//    fern match on the mid-frame fill-in view, 1/8-resolution registration on the device, global deformation) and, when that
//    does not fire, the LOCAL one (:447-527), both optimised by the built-in deformation-graph solver (no CHOLMOD); Ferns::addFrame
//    at the end of the frame.  The fern table's seed is fixed (the reference uses time(0)); setLoopSolver() replaces the local
//    optimiser.  reloc = true judges every tracked frame by its own statistics (:326-366): frames that are not ok are not fused, more
//    than ten in a row and the camera is lost (getLost()) until a fern match brings the pose back (:411-413; needs closeLoops = true).
//    The global closure's 1/8-resolution registration needs width and height to be multiples of 32 (640x480, 1280x960, ...); at other
//    sizes closeLoops = true closes local loops only (no fern database) instead of refusing to construct.
//  * getTextures / getFeedbackBuffers / computeFeedbackBuffers / normaliseDepth are OpenGL objects and display passes in the
//    reference and have no counterpart here.
//  * errors throw std::runtime_error instead of assert()/exit(0).
#ifndef EFUSION_ELASTICFUSION_H_
#define EFUSION_ELASTICFUSION_H_
// This header stands where the reference's Core/ElasticFusion.h stands.  It takes that header's include guard, so a front end that
// still has the old file on its include path (MainController.h:20 includes "Core/ElasticFusion.h" next to itself) gets this class and
// an inert old header when this one comes first (g++ -include ElasticFusion.h: tests/test_front_end_compiles.py compiles the
// reference's MainController.cpp where it lies that way), and a loud error instead of two classes when it comes second.
#ifdef ELASTICFUSION_H_
#error "the reference's Core/ElasticFusion.h was included before include/ElasticFusion.h: include this one first, or remove the old one"
#endif
#define ELASTICFUSION_H_

// what the reference's header brings along and its front end relies on (MainController.cpp: std::stringstream, std::isnan,
// std::numeric_limits, std::map, memcpy, std::setprecision)
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "ef_hip.h"   // ef_local_loop, ef_loop_solver

#ifdef EFUSION_USE_SOPHUS
#include <sophus/se3.hpp>
#endif

// ---- process-wide singletons of the reference (Core/Utils/Resolution.h:25-58, Intrinsics.h:25-51), under the reference's own include
// guards: the front end's headers include those files themselves (Tools/GUI.h:29, LogReader.h:27), and whichever definition comes first is
// the one the translation unit sees — same members, same layout, getInstance() in libefusion.so either way ----
#ifndef RESOLUTION_H_
#define RESOLUTION_H_
class Resolution {
 public:
  static const Resolution& getInstance(int width = 0, int height = 0);
  const int& width() const { return imgWidth; }
  const int& height() const { return imgHeight; }
  const int& cols() const { return imgWidth; }
  const int& rows() const { return imgHeight; }
  const int& numPixels() const { return imgNumPixels; }

 private:
  Resolution(int width, int height);
  const int imgWidth, imgHeight, imgNumPixels;
};
#endif  // RESOLUTION_H_

#ifndef INTRINSICS_H_
#define INTRINSICS_H_
class Intrinsics {
 public:
  static const Intrinsics& getInstance(float fx = 0, float fy = 0, float cx = 0, float cy = 0);
  const float& fx() const { return fx_; }
  const float& fy() const { return fy_; }
  const float& cx() const { return cx_; }
  const float& cy() const { return cy_; }

 private:
  Intrinsics(float fx, float fy, float cx, float cy);
  const float fx_, fy_, cx_, cy_;
};
#endif  // INTRINSICS_H_

// Core/Shaders/Vertex.h: the stride of one surfel in bytes — three float4 rows, the layout downloadMap() returns and a viewer's vertex
// attribute pointers step by (Tools/GUI.h:325-343)
#ifndef VERTEX_H_
#define VERTEX_H_
class Vertex {
 public:
  static constexpr int SIZE = 12 * (int)sizeof(float);
};
#endif  // VERTEX_H_

namespace efusion {

// small fixed-size values with Eigen's element access — v(i), v[i], m(r, c), data() — for the members the front end reads coefficient by
// coefficient (MainController.cpp:389-441: graph.at(i)->position(0), constraints.at(j).sourcePoint(1), ...); plain arrays underneath
struct Vec3d {
  double v[3];
  double& operator()(int i) { return v[i]; }
  const double& operator()(int i) const { return v[i]; }
  double& operator[](int i) { return v[i]; }
  const double& operator[](int i) const { return v[i]; }
  double* data() { return v; }
  const double* data() const { return v; }
};
struct Mat3d {       // row major
  double m[9];
  double& operator()(int r, int c) { return m[r * 3 + c]; }
  const double& operator()(int r, int c) const { return m[r * 3 + c]; }
};
template <typename S>
struct Mat4 {        // column major like Eigen's default, so that data() can be handed to whatever takes Eigen::Matrix4f::data()
  S m[16];
  S& operator()(int r, int c) { return m[c * 4 + r]; }
  const S& operator()(int r, int c) const { return m[c * 4 + r]; }
  S* data() { return m; }
  const S* data() const { return m; }
};

template <typename S> struct SE3Cast;
// Sophus::SE3d's data layout: Eigen::Quaterniond (x, y, z, w) followed by Eigen::Vector3d
struct SE3d {
  double q[4] = {0, 0, 0, 1};
  double t[3] = {0, 0, 0};
  SE3d() = default;
  static SE3d fromMatrix(const double* T_wc16_rowmajor);
  void matrix(double* out16_rowmajor) const;      // Sophus::SE3d::matrix()
  const double* translation() const { return t; } // Sophus::SE3d::translation()
  // Sophus::SE3d::cast<float>(), as far as the front end uses it: frames.at(i)->T_wc.cast<float>().matrix() (MainController.cpp:383,411)
  template <typename S> SE3Cast<S> cast() const;
};
template <typename S>
struct SE3Cast {
  SE3d pose;
#ifdef EFUSION_USE_SOPHUS
  Eigen::Matrix<S, 4, 4> matrix() const {
    double M[16];
    pose.matrix(M);
    Eigen::Matrix<S, 4, 4> r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r(i, j) = (S)M[i * 4 + j];
    return r;
  }
#else
  Mat4<S> matrix() const {
    double M[16];
    pose.matrix(M);
    Mat4<S> r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r(i, j) = (S)M[i * 4 + j];
    return r;
  }
#endif
};
template <typename S> SE3Cast<S> SE3d::cast() const { return SE3Cast<S>{*this}; }

struct ef_ctx_deleter { void operator()(void* p) const; };

// what MainController reads from getModelToModel() (RGBDOdometry.h:74-82)
struct OdometryStats {
  float lastICPError = 0, lastICPCount = 0, lastRGBError = 0, lastRGBCount = 0, lastSO3Error = 0, lastSO3Count = 0;
  double lastA[36] = {0}, lastb[6] = {0};
};

class ElasticFusion;

// GlobalModel facade (Core/GlobalModel.h:46-58): the surfel map lives in HBM inside the context
class GlobalModelView {
 public:
  unsigned int lastCount();                 // GlobalModel::lastCount()
  // GlobalModel::downloadMap(): count x 12 floats {x,y,z,conf} {colour,0,initTime,lastTime} {nx,ny,nz,radius}
  std::vector<float> downloadMap();
 private:
  friend class ::efusion::ElasticFusion;
  void* ctx = nullptr;
};

// IndexMap facade (Core/IndexMap.h): host copies of the predicted-surface images of the last predict()
class IndexMapView {
 public:
  std::vector<uint8_t> image();             // imageTex:  W*H*4 u8
  std::vector<float> vertex();              // vertexTex: W*H*4 f32
  std::vector<float> normal();              // normalTex: W*H*4 f32
  std::vector<uint16_t> time();             // timeTex:   W*H u16
 private:
  friend class ::efusion::ElasticFusion;
  void* ctx = nullptr;
  int w = 0, h = 0;
};

// Ferns::SurfaceConstraint (Core/Ferns.h) and PoseMatch (Core/PoseMatch.h): what getPoseMatches() hands to the front-end's drawing code
struct SurfaceConstraint {
  Vec3d sourcePoint, targetPoint;
};
struct PoseMatch {
  int firstId, secondId;
  SE3d T_wc_first, T_wc_second;
  std::vector<SurfaceConstraint> constraints;
  bool fern;
};
// Ferns facade: what MainController reads (frames.size(), lastClosest, the keyframe poses it draws)
struct FernFrame {
  int id, srcTime;
  SE3d T_wc;
  const FernFrame* operator->() const { return this; }   // the reference's frames hold Ferns::Frame*: frames.at(i)->T_wc compiles as written
};
struct FernsView {
  std::vector<FernFrame> frames;
  int lastClosest = -1;
};
// Core/Utils/GraphNode.h: a node of the embedded deformation graph as the front end draws it (MainController.cpp:388-404)
struct GraphNode {
  int id;
  Vec3d position;
  Mat3d rotation;
  Vec3d translation;
  std::vector<int> neighbours;
  bool enabled;
};
// Deformation facade.  getGraph() (Deformation.cpp:65-67): the graph Deformation::sampleGraphModel builds at the end of every closed-loop
// frame (ElasticFusion.cpp:593) — every 5000th surfel of the current model in time order, identity rotation, zero translation, each node
// joined to its four sequence neighbours (DeformationGraph.cpp:239-266) — sampled from the map as it stands when the call is made;
// empty while the map yields no more than four nodes (Deformation.cpp:283) and in open loop.  getRawGraph(): the same nodes as rows
// {x, y, z, time}.
class DeformationView {
 public:
  const std::vector<GraphNode*>& getGraph();
  void getRawGraph(std::vector<float>& nodes4);
 private:
  friend class ::efusion::ElasticFusion;
  void* ctx = nullptr;
  bool closeLoops = false;
  std::vector<GraphNode> nodes;
  std::vector<GraphNode*> node_ptrs;
};

class ElasticFusion {
 public:
  // same parameters, order and defaults as Core/ElasticFusion.h:42-58; `device` selects the HIP device
  ElasticFusion(const int timeDelta = 200, const int countThresh = 35000, const float errThresh = 5e-05,
                const float covThresh = 1e-05, const bool closeLoops = true, const bool iclnuim = false,
                const bool reloc = false, const float photoThresh = 115, const float confidence = 10,
                const float depthCut = 3, const float icpThresh = 10, const bool fastOdom = false,
                const float fernThresh = 0.3095, const bool so3 = true, const bool frameToFrameRGB = false,
                const std::string fileName = "", const int device = 0);
  virtual ~ElasticFusion();   // writes <fileName>.freiburg like the reference's destructor (ElasticFusion.cpp:107-139)

  // rgb: W*H*3 u8 row major; depth: W*H u16 millimetres, 0 invalid; borrowed for the duration of the call.
  // All GPU work is only enqueued; getters that return results synchronise.
  void processFrame(const uint8_t* rgb, const uint16_t* depth, const int64_t& timestamp, const float weightMultiplier = 1.f,
                    const SE3d* in_T_wc = 0);
#ifdef EFUSION_USE_SOPHUS
  // the reference's own signature (Core/ElasticFusion.h:70-75); header-inline, so that the library itself never needs Sophus.  The pose
  // crosses as rotation matrix + translation (Sophus::SE3d::rotationMatrix() / translation(), the two accessors every Sophus has).
  void processFrame(const uint8_t* rgb, const uint16_t* depth, const int64_t& timestamp, const float weightMultiplier,
                    const Sophus::SE3d* in_T_wc) {
    if (!in_T_wc) {
      processFrame(rgb, depth, timestamp, weightMultiplier, (const SE3d*)0);
      return;
    }
    const auto R = in_T_wc->rotationMatrix();
    const auto& t = in_T_wc->translation();
    const double M[16] = {R(0, 0), R(0, 1), R(0, 2), t(0), R(1, 0), R(1, 1), R(1, 2), t(1), R(2, 0), R(2, 1), R(2, 2), t(2), 0, 0, 0, 1};
    const SE3d T = SE3d::fromMatrix(M);
    processFrame(rgb, depth, timestamp, weightMultiplier, &T);
  }
#endif
  void predict();

  IndexMapView& getIndexMap() { return indexMap; }
  GlobalModelView& getGlobalModel() { return globalModel; }
  const FernsView& getFerns();                           // refreshed on each call (empty in open loop)
  DeformationView& getLocalDeformation() { return localDeformation; }
  const std::vector<PoseMatch>& getPoseMatches() { return poseMatches; }   // one entry per accepted closure (ElasticFusion.cpp:431,517)
  // closeLoops: the model-to-model tracker of the local loop closure; open loop: the frame-to-model tracker's statistics
  const OdometryStats& getModelToModel();   // refreshed from the device on each call
  // local loop closure (closeLoops = true): the solver standing where Deformation::constrain stands (include/ef_hip.h), and
  // the gates / statistics / poses of the last frame's attempt
  void setLoopSolver(ef_loop_solver fn, void* user);
  void useBuiltinLoopSolver(bool on = true);   // the built-in deformation-graph optimiser where Deformation::constrain stands
  const ef_local_loop& getLocalLoop();

  const float& getConfidenceThreshold() { return confidenceThreshold; }
  void setRgbOnly(const bool& val);
  void setIcpWeight(const float& val);
  void setPyramid(const bool& val);
  void setFastOdom(const bool& val);
  void setSo3(const bool& val);
  void setFrameToFrameRGB(const bool& val);
  void setConfidenceThreshold(const float& val);
  void setFernThresh(const float& val);     // Ferns::addFrame's dissimilarity threshold from now on
  void setDepthCutoff(const float& val);

  const bool& getLost() { return lost; }
  const int& getTick();
  const int& getTimeDelta() { return timeDelta; }
  void setTick(const int& val);
  const float& getMaxDepthProcessed() { return maxDepthProcessed; }
  const SE3d& get_T_wc_pod();               // the pose as the engine holds it (quaternion x,y,z,w + translation)
#ifdef EFUSION_USE_SOPHUS
  Sophus::SE3d get_T_wc() {                  // Core/ElasticFusion.h:196 (by value here: the class keeps no Sophus member, its layout is the library's)
    double M[16];
    get_T_wc_pod().matrix(M);
    Sophus::SE3d T;
    decltype(T.rotationMatrix()) R;
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) R(i, j) = M[i * 4 + j];
      T.translation()(i) = M[i * 4 + 3];
    }
    T.setRotationMatrix(R);
    return T;
  }
#else
  const SE3d& get_T_wc() { return get_T_wc_pod(); }
#endif
  const int& getDeforms() { return deforms; }
  const int& getFernDeforms() { return fernDeforms; }
  void savePly();                            // <fileName>.ply, binary little endian (ElasticFusion.cpp:684-781)

  // getGlobalModel().downloadMap() / savePly(): true (default) = the buffer GlobalModel::downloadMap reads in the reference (the map
  // BEFORE the frame's clean pass, truncated to the count after it: GlobalModel.cpp:673-706); false = model(), the map as it stands
  void setReferenceDownload(bool on);
  void synchronize();                        // wait for everything enqueued so far
  void* context() { return ctx.get(); }      // the ef_ctx* underneath (include/ef_hip.h)

 private:
  std::unique_ptr<void, ef_ctx_deleter> ctx;
  IndexMapView indexMap;
  GlobalModelView globalModel;
  DeformationView localDeformation;
  FernsView fernsView;
  std::vector<PoseMatch> poseMatches;
  bool iclnuim = false;
  OdometryStats stats;
  SE3d T_wc;
  std::string saveFilename;
  int tick = 1;
  int timeDelta;
  float confidenceThreshold;
  float maxDepthProcessed = 20.0f;
  bool lost = false;
  int deforms = 0, fernDeforms = 0;
  bool closeLoops = false;
  ef_local_loop localLoop{};
};

}  // namespace efusion

// the reference's classes live in the global namespace
using efusion::ElasticFusion;
using efusion::GraphNode;
using efusion::PoseMatch;
#ifndef EFUSION_USE_SOPHUS
namespace Sophus { using SE3d = efusion::SE3d; }
#endif

#endif  // EFUSION_ELASTICFUSION_H_
