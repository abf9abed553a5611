// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// CPU restatement of the tracking driver Core/Utils/RGBDOdometry.{h,cpp} +
// Core/Utils/OdometryProvider.h.  "Textures" are plain host images:
//   vertex / normal image : float4 [rows][cols]       (GL_RGBA32F, IndexMap / FillIn outputs)
//   colour image          : uchar4 [rows][cols]       (GL_RGBA8; .x .y .z as bgr2Intensity reads them)
//   filtered depth        : u16    [rows][cols]
// parity of THIS file unpinned (host driver: Eigen/Sophus un-vendored, cannot be compiled; no reference vectors) — see
// efo_linalg.h; the device operators it calls are pinned against the compiled reference (efo_track.cpp).
#include "efo_common.h"
#include "efo_linalg.h"
#include "efo_api.h"
#include <vector>
#include <cassert>

using namespace efo;

struct efo_odometry {
  static constexpr int NUM_PYRS = 3;  // RGBDOdometry.h:114
  int width, height;
  float cx, cy, fx, fy;
  float distThres, angleThres;
  const float sobelScale = 1.0f / 8.0f;      // 1/2^sobelSize, RGBDOdometry.cpp:39-40
  const float maxDepthDeltaRGB = 0.07f;      // :41
  const float maxDepthRGB = 6.0f;            // :42
  float minimumGradientMagnitudes[3] = {5, 3, 1};  // :112-114

  std::vector<uint16_t> depth_tmp[NUM_PYRS];
  std::vector<float> vmaps_tmp;
  std::vector<float> vmaps_g_prev[NUM_PYRS], nmaps_g_prev[NUM_PYRS];
  std::vector<float> vmaps_curr[NUM_PYRS], nmaps_curr[NUM_PYRS];
  std::vector<float> lastDepth[NUM_PYRS], nextDepth[NUM_PYRS];
  std::vector<uint8_t> lastImage[NUM_PYRS], nextImage[NUM_PYRS], lastNextImage[NUM_PYRS];
  std::vector<int16_t> nextdIdx[NUM_PYRS], nextdIdy[NUM_PYRS];
  std::vector<float> pointClouds[NUM_PYRS];
  std::vector<DataTerm> corresImg[NUM_PYRS];

  float lastICPError = 0, lastICPCount, lastRGBError = 0, lastRGBCount, lastSO3Error = 0, lastSO3Count;
  double lastA[36] = {0}, lastb[6] = {0};
  int so3_iterations_run = 0;

  int W(int l) const { return width >> l; }
  int H(int l) const { return height >> l; }
  void intr(int l, float& fx_, float& fy_, float& cx_, float& cy_) const {  // CameraModel::operator(), types.cuh:92-95
    int div = 1 << l;
    fx_ = fx / div; fy_ = fy / div; cx_ = cx / div; cy_ = cy / div;
  }

  efo_odometry(int w, int h, float cx_, float cy_, float fx_, float fy_, float distThresh, float angleThresh)
      : width(w), height(h), cx(cx_), cy(cy_), fx(fx_), fy(fy_), distThres(distThresh), angleThres(angleThresh) {
    lastICPCount = lastRGBCount = lastSO3Count = (float)(w * h);
    for (int i = 0; i < NUM_PYRS; ++i) {
      size_t n = (size_t)W(i) * H(i);
      depth_tmp[i].assign(n, 0);
      // device buffers are cudaMalloc'ed (uninitialised) in the reference; we zero them so that the
      // stale y/z planes of Q3 are deterministic.  The HIP side zero-fills the same buffers at creation.
      vmaps_g_prev[i].assign(3 * n, 0.f); nmaps_g_prev[i].assign(3 * n, 0.f);
      vmaps_curr[i].assign(3 * n, 0.f); nmaps_curr[i].assign(3 * n, 0.f);
      lastDepth[i].assign(n, 0.f); nextDepth[i].assign(n, 0.f);
      lastImage[i].assign(n, 0); nextImage[i].assign(n, 0); lastNextImage[i].assign(n, 0);
      nextdIdx[i].assign(n, 0); nextdIdy[i].assign(n, 0);
      pointClouds[i].assign(3 * n, 0.f);
      corresImg[i].resize(n);
      std::memset(corresImg[i].data(), 0, n * sizeof(DataTerm));
    }
    vmaps_tmp.assign((size_t)4 * w * h, 0.f);
  }

  // RGBDOdometry::initICP(GPUTexture* filteredDepth, float depthCutoff), RGBDOdometry.cpp:121-147
  void initICP(const uint16_t* filteredDepth, float depthCutoff) {
    std::memcpy(depth_tmp[0].data(), filteredDepth, sizeof(uint16_t) * width * height);
    for (int i = 1; i < NUM_PYRS; ++i) efo_pyr_down_u16(depth_tmp[i - 1].data(), W(i - 1), H(i - 1), depth_tmp[i].data());
    for (int i = 0; i < NUM_PYRS; ++i) {
      float fx_, fy_, cx_, cy_;
      intr(i, fx_, fy_, cx_, cy_);
      efo_create_vmap(depth_tmp[i].data(), W(i), H(i), fx_, fy_, cx_, cy_, depthCutoff, vmaps_curr[i].data());
      efo_create_nmap(vmaps_curr[i].data(), W(i), H(i), nmaps_curr[i].data());
    }
  }

  // RGBDOdometry::initICP(GPUTexture* predictedVertices, GPUTexture* predictedNormals), RGBDOdometry.cpp:149-169:
  // the "current" side taken from a model prediction (model-to-model tracking of the local loop closure)
  void initICP(const float* vtex, const float* ntex) {
    efo_copy_maps(vtex, ntex, width, height, vmaps_tmp.data(), vmaps_curr[0].data(), nmaps_curr[0].data());
    for (int i = 1; i < NUM_PYRS; ++i) {
      efo_resize_map(vmaps_curr[i - 1].data(), W(i - 1), H(i - 1), vmaps_curr[i].data(), 0);
      efo_resize_map(nmaps_curr[i - 1].data(), W(i - 1), H(i - 1), nmaps_curr[i].data(), 1);
    }
  }

  // RGBDOdometry::initICPModel, RGBDOdometry.cpp:171-210
  void initICPModel(const float* vtex, const float* ntex, const SE3& T_wc) {
    efo_copy_maps(vtex, ntex, width, height, vmaps_tmp.data(), vmaps_g_prev[0].data(), nmaps_g_prev[0].data());
    for (int i = 1; i < NUM_PYRS; ++i) {
      efo_resize_map(vmaps_g_prev[i - 1].data(), W(i - 1), H(i - 1), vmaps_g_prev[i].data(), 0);
      efo_resize_map(nmaps_g_prev[i - 1].data(), W(i - 1), H(i - 1), nmaps_g_prev[i].data(), 1);
    }
    M3d Rd = se3_rotation(T_wc);
    float R[9], t[3];
    for (int i = 0; i < 9; ++i) R[i] = (float)Rd.m[i];
    for (int i = 0; i < 3; ++i) t[i] = (float)T_wc.t[i];
    for (int i = 0; i < NUM_PYRS; ++i) efo_transform_maps(vmaps_g_prev[i].data(), nmaps_g_prev[i].data(), W(i), H(i), R, t);
  }

  // RGBDOdometry::populateRGBDData, RGBDOdometry.cpp:212-234 (Q1: depth always comes from vmaps_tmp)
  void populateRGBDData(const uint8_t* rgba, std::vector<float>* destDepths, std::vector<uint8_t>* destImages) {
    efo_vertices_to_depth(vmaps_tmp.data(), width, height, maxDepthRGB, destDepths[0].data());
    for (int i = 0; i + 1 < NUM_PYRS; ++i) efo_pyr_down_gauss_f(destDepths[i].data(), W(i), H(i), destDepths[i + 1].data());
    efo_bgr_to_intensity(rgba, width, height, destImages[0].data());
    for (int i = 0; i + 1 < NUM_PYRS; ++i) efo_pyr_down_uchar_gauss(destImages[i].data(), W(i), H(i), destImages[i + 1].data());
  }
  void initRGBModel(const uint8_t* rgba) { populateRGBDData(rgba, lastDepth, lastImage); }  // :236-239
  void initRGB(const uint8_t* rgba) { populateRGBDData(rgba, nextDepth, nextImage); }       // :241-244
  void initFirstRGB(const uint8_t* rgba) {                                                    // :246-257
    efo_bgr_to_intensity(rgba, width, height, lastNextImage[0].data());
    for (int i = 0; i + 1 < NUM_PYRS; ++i)
      efo_pyr_down_uchar_gauss(lastNextImage[i].data(), W(i), H(i), lastNextImage[i + 1].data());
  }

  static void to_f33(const M3d& m, float* o) { for (int i = 0; i < 9; ++i) o[i] = (float)m.m[i]; }

  // RGBDOdometry::getIncrementalTransformation, RGBDOdometry.cpp:259-571
  void getIncrementalTransformation(SE3& T_wc, bool rgbOnly, float icpWeight, bool pyramid, bool fastOdom, bool so3) {
    const bool icp = !rgbOnly && icpWeight > 0;
    const bool rgb = rgbOnly || icpWeight < 100;

    float Rprev[9], tprev[3], Rcurr[9], tcurr[3];
    to_f33(se3_rotation(T_wc), Rprev);
    for (int i = 0; i < 3; ++i) tprev[i] = (float)T_wc.t[i];
    std::memcpy(Rcurr, Rprev, sizeof(Rcurr));
    std::memcpy(tcurr, tprev, sizeof(tcurr));

    if (rgb)
      for (int i = 0; i < NUM_PYRS; ++i)
        efo_derivative_images(nextImage[i].data(), W(i), H(i), nextdIdx[i].data(), nextdIdy[i].data());

    M3d resultR = m3_identity();
    so3_iterations_run = 0;

    if (so3) {  // :284-369
      const int lvl = 2;
      float R_lr[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      float fx_, fy_, cx_, cy_;
      intr(lvl, fx_, fy_, cx_, cy_);
      M3d K{};
      K.m[0] = fx_; K.m[4] = fy_; K.m[2] = cx_; K.m[5] = cy_; K.m[8] = 1;
      float lastError = std::numeric_limits<float>::max() / 2;
      float lastCount = std::numeric_limits<float>::max() / 2;
      M3d lastResultR = m3_identity();
      for (int i = 0; i < 10; ++i) {
        float jtj[9], jtr[3];
        M3d Kinv = m3_inverse(K);
        M3d homography = m3_mul(m3_mul(K, resultR), Kinv);
        M3d K_R_lr = m3_mul(K, resultR);
        float imageBasis[9], kinv[9], krlr[9];
        to_f33(homography, imageBasis); to_f33(Kinv, kinv); to_f33(K_R_lr, krlr);
        float residual[2];
        efo_so3_step(lastNextImage[lvl].data(), nextImage[lvl].data(), imageBasis, kinv, krlr, W(lvl), H(lvl), jtj, jtr, residual);
        ++so3_iterations_run;
        lastSO3Error = sqrtf(residual[0]) / residual[1];
        lastSO3Count = residual[1];
        if (lastSO3Error < lastError && lastCount == lastSO3Count) break;             // converged
        else if (lastSO3Error > lastError + 0.001) {                                    // diverging
          lastSO3Error = lastError; lastSO3Count = lastCount; resultR = lastResultR; break;
        }
        lastError = lastSO3Error; lastCount = lastSO3Count; lastResultR = resultR;
        float delta[3];
        ldlt_solve<float, 3>(jtj, jtr, delta);
        M3d rotUpdate = rodrigues(V3d{{(double)delta[0], (double)delta[1], (double)delta[2]}});
        float ru[9], nR[9];
        to_f33(rotUpdate, ru);
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) {
            float s = 0;
            for (int k = 0; k < 3; ++k) s += ru[r * 3 + k] * R_lr[k * 3 + c];
            nR[r * 3 + c] = s;
          }
        std::memcpy(R_lr, nR, sizeof(nR));
        for (int k = 0; k < 9; ++k) resultR.m[k] = R_lr[k];
      }
    }

    int iterations[3];
    iterations[0] = fastOdom ? 3 : 10;   // :371-373
    iterations[1] = pyramid ? 5 : 0;
    iterations[2] = pyramid ? 4 : 0;

    float Rprev_inv[9];
    m3f_inverse(Rprev, Rprev_inv);

    M4d resultRt = m4_identity();
    if (so3)
      for (int x = 0; x < 3; ++x)
        for (int y = 0; y < 3; ++y) resultRt.m[x * 4 + y] = resultR.m[x * 3 + y];

    float residual[2] = {0, 0};
    for (int i = NUM_PYRS - 1; i >= 0; --i) {
      float fx_, fy_, cx_, cy_;
      intr(i, fx_, fy_, cx_, cy_);
      if (rgb) efo_project_to_point_cloud(lastDepth[i].data(), W(i), H(i), fx_, fy_, cx_, cy_, pointClouds[i].data());
      M3d K{};
      K.m[0] = fx_; K.m[4] = fy_; K.m[2] = cx_; K.m[5] = cy_; K.m[8] = 1;
      lastRGBError = std::numeric_limits<float>::max();

      for (int j = 0; j < iterations[i]; ++j) {
        M4d Rt = m4_affine_inverse(resultRt);
        M3d R;
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) R.m[r * 3 + c] = Rt.m[r * 4 + c];
        M3d KRK_inv = m3_mul(m3_mul(K, R), m3_inverse(K));
        float krkInv[9];
        to_f33(KRK_inv, krkInv);
        V3d Kt = m3_mulv(K, V3d{{Rt.m[3], Rt.m[7], Rt.m[11]}});
        float kt[3] = {(float)Kt.v[0], (float)Kt.v[1], (float)Kt.v[2]};

        int sigma = 0, rgbSize = 0;
        if (rgb) {
          float ms = (float)(std::pow((double)minimumGradientMagnitudes[i], 2.0) / std::pow((double)sobelScale, 2.0));
          efo_rgb_residual(ms, nextdIdx[i].data(), nextdIdy[i].data(), lastDepth[i].data(), nextDepth[i].data(),
                           lastImage[i].data(), nextImage[i].data(), corresImg[i].data(), maxDepthDeltaRGB, kt, krkInv,
                           W(i), H(i), &sigma, &rgbSize);
        }
        // Q2: precedence makes this sqrt(rgbSize) unless (float)sigma/rgbSize == 0 (RGBDOdometry.cpp:442)
        float sigmaVal = std::sqrt((float)sigma / rgbSize == 0 ? 1 : rgbSize);
        float rgbError = (float)(std::sqrt((double)sigma) / (rgbSize == 0 ? 1 : rgbSize));  // std::sqrt(int) is double
        if (rgbOnly && rgbError > lastRGBError) break;
        lastRGBError = rgbError;
        lastRGBCount = (float)rgbSize;
        if (rgbOnly) sigmaVal = -1;

        float A_icp[36] = {0}, b_icp[6] = {0}, A_rgbd[36] = {0}, b_rgbd[6] = {0};
        if (icp)
          efo_icp_step(Rcurr, tcurr, vmaps_curr[i].data(), nmaps_curr[i].data(), Rprev_inv, tprev, fx_, fy_, cx_, cy_,
                       vmaps_g_prev[i].data(), nmaps_g_prev[i].data(), distThres, angleThres, W(i), H(i), A_icp, b_icp, residual);
        lastICPError = sqrtf(residual[0]) / residual[1];
        lastICPCount = residual[1];
        if (rgb)
          efo_rgb_step(corresImg[i].data(), sigmaVal, pointClouds[i].data(), fx_, fy_, nextdIdx[i].data(), nextdIdy[i].data(),
                       sobelScale, W(i), H(i), A_rgbd, b_rgbd);

        double result[6];
        if (icp && rgb) {
          double w = icpWeight;
          for (int k = 0; k < 36; ++k) lastA[k] = (double)A_rgbd[k] + w * w * (double)A_icp[k];
          for (int k = 0; k < 6; ++k) lastb[k] = (double)b_rgbd[k] + w * (double)b_icp[k];
        } else if (icp) {
          for (int k = 0; k < 36; ++k) lastA[k] = A_icp[k];
          for (int k = 0; k < 6; ++k) lastb[k] = b_icp[k];
        } else {
          for (int k = 0; k < 36; ++k) lastA[k] = A_rgbd[k];
          for (int k = 0; k < 6; ++k) lastb[k] = b_rgbd[k];
        }
        ldlt_solve<double, 6>(lastA, lastb, result);

        // OdometryProvider::computeUpdateSE3, OdometryProvider.h:73-96
        M4d upd = m4_identity();
        M3d Rr = rodrigues(V3d{{result[3], result[4], result[5]}});
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) upd.m[r * 4 + c] = Rr.m[r * 3 + c];
          upd.m[r * 4 + 3] = result[r];
        }
        resultRt = m4_mul(upd, resultRt);
        // rgbOdom (Isometry3f) = float cast of resultRt;  currentT = [Rprev|tprev] * rgbOdom^-1, with
        // Isometry inverse = (R^T, -R^T t) and Isometry::rotation() = linear()  (Q13)
        float oR[9], ot[3];
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c) oR[r * 3 + c] = (float)resultRt.m[r * 4 + c];
          ot[r] = (float)resultRt.m[r * 4 + 3];
        }
        float iR[9], it[3];
        for (int r = 0; r < 3; ++r)
          for (int c = 0; c < 3; ++c) iR[r * 3 + c] = oR[c * 3 + r];
        for (int r = 0; r < 3; ++r) it[r] = -(iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1] + iR[r * 3 + 2] * ot[2]);
        for (int r = 0; r < 3; ++r) {
          for (int c = 0; c < 3; ++c)
            Rcurr[r * 3 + c] = Rprev[r * 3] * iR[c] + Rprev[r * 3 + 1] * iR[3 + c] + Rprev[r * 3 + 2] * iR[6 + c];
          tcurr[r] = (Rprev[r * 3] * it[0] + Rprev[r * 3 + 1] * it[1] + Rprev[r * 3 + 2] * it[2]) + tprev[r];
        }
      }
    }

    if (rgb) {  // :555-558
      float d0 = tcurr[0] - tprev[0], d1 = tcurr[1] - tprev[1], d2 = tcurr[2] - tprev[2];
      if (sqrtf(d0 * d0 + d1 * d1 + d2 * d2) > 0.3) {
        std::memcpy(Rcurr, Rprev, sizeof(Rcurr));
        std::memcpy(tcurr, tprev, sizeof(tcurr));
      }
    }
    if (so3)
      for (int i = 0; i < NUM_PYRS; ++i) std::swap(lastNextImage[i], nextImage[i]);

    M3d Rc;
    for (int k = 0; k < 9; ++k) Rc.m[k] = (double)Rcurr[k];
    for (int k = 0; k < 3; ++k) T_wc.t[k] = (double)tcurr[k];
    se3_set_rotation(T_wc, polar3(Rc));
  }
};

extern "C" {

efo_odometry* efo_odom_create(int w, int h, float cx, float cy, float fx, float fy) {
  // defaults of RGBDOdometry.h:41-42
  return new efo_odometry(w, h, cx, cy, fx, fy, 0.10f, sinf(20.f * 3.14159254f / 180.f));
}
void efo_odom_destroy(efo_odometry* o) { delete o; }
void efo_odom_init_icp(efo_odometry* o, const uint16_t* filteredDepth, float depthCutoff) { o->initICP(filteredDepth, depthCutoff); }
void efo_odom_init_icp_maps(efo_odometry* o, const float* vtex, const float* ntex) { o->initICP(vtex, ntex); }
void efo_odom_init_icp_model(efo_odometry* o, const float* vtex, const float* ntex, const double* T_wc16) {
  o->initICPModel(vtex, ntex, se3_from_matrix(T_wc16));
}
void efo_odom_init_rgb_model(efo_odometry* o, const uint8_t* rgba) { o->initRGBModel(rgba); }
void efo_odom_init_rgb(efo_odometry* o, const uint8_t* rgba) { o->initRGB(rgba); }
void efo_odom_init_first_rgb(efo_odometry* o, const uint8_t* rgba) { o->initFirstRGB(rgba); }
void efo_odom_track(efo_odometry* o, double* T_wc16, int rgbOnly, float icpWeight, int pyramid, int fastOdom, int so3) {
  SE3 T = se3_from_matrix(T_wc16);
  o->getIncrementalTransformation(T, rgbOnly != 0, icpWeight, pyramid != 0, fastOdom != 0, so3 != 0);
  M4d M = se3_matrix(T);
  std::memcpy(T_wc16, M.m, sizeof(M.m));
}
void efo_odom_stats(const efo_odometry* o, float* out6, double* lastA36, double* lastb6) {
  out6[0] = o->lastICPError; out6[1] = o->lastICPCount; out6[2] = o->lastRGBError;
  out6[3] = o->lastRGBCount; out6[4] = o->lastSO3Error; out6[5] = o->lastSO3Count;
  if (lastA36) std::memcpy(lastA36, o->lastA, sizeof(o->lastA));
  if (lastb6) std::memcpy(lastb6, o->lastb, sizeof(o->lastb));
}
// Buffer access for kernel-level tests. which: 0 vmap_curr 1 nmap_curr 2 vmap_g_prev 3 nmap_g_prev
// 4 lastDepth 5 nextDepth 6 lastImage 7 nextImage 8 lastNextImage 9 dIdx 10 dIdy 11 depth_tmp 12 vmaps_tmp(level ignored)
const void* efo_odom_buffer(const efo_odometry* o, int which, int level) {
  switch (which) {
    case 0: return o->vmaps_curr[level].data();
    case 1: return o->nmaps_curr[level].data();
    case 2: return o->vmaps_g_prev[level].data();
    case 3: return o->nmaps_g_prev[level].data();
    case 4: return o->lastDepth[level].data();
    case 5: return o->nextDepth[level].data();
    case 6: return o->lastImage[level].data();
    case 7: return o->nextImage[level].data();
    case 8: return o->lastNextImage[level].data();
    case 9: return o->nextdIdx[level].data();
    case 10: return o->nextdIdy[level].data();
    case 11: return o->depth_tmp[level].data();
    case 12: return o->vmaps_tmp.data();
    case 13: return o->pointClouds[level].data();
    case 14: return o->corresImg[level].data();
  }
  return nullptr;
}

}  // extern "C"
