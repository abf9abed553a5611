// TEST INFRASTRUCTURE — drives the REFERENCE's own tracking driver, Core/Utils/RGBDOdometry.cpp (+ OdometryProvider.h), compiled
// from /root/reference where it lies against host_on_cpu/ (Eigen / Sophus / Pangolin / GL-interop in miniature) and linked with the
// reference's own CUDA operator sources compiled by cuda_on_cpu/ (oracle/Makefile, target `refdriver` -> _ref/libefr_driver.so).
// efd_odom_<x> has the signature of efo_odom_<x> (efo_api.h), so tests/efo.py can route the oracle's driver calls here.
// This file is ours; it defines GPUTexture's constructor itself (Core/GPUTexture.cpp registers a GL texture with CUDA) as a
// holder of host texels.
#include <cstdint>
#include <cstring>

#include "cuda_runtime.h"
#include "Utils/RGBDOdometry.h"
#include "efo_linalg.h"

const std::string GPUTexture::RGB = "RGB";
const std::string GPUTexture::DEPTH_RAW = "DEPTH";
const std::string GPUTexture::DEPTH_FILTERED = "DEPTH_FILTERED";
const std::string GPUTexture::DEPTH_METRIC = "DEPTH_METRIC";
const std::string GPUTexture::DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
const std::string GPUTexture::DEPTH_NORM = "DEPTH_NORM";

GPUTexture::GPUTexture(const int w, const int h, const GLenum internalFormat_, const GLenum format_, const GLenum dataType_, const bool draw_)
    : texture(nullptr), cudaRes(new cudaGraphicsResource{}), draw(draw_), width(w), height(h), internalFormat(internalFormat_),
      format(format_), dataType(dataType_) {}
GPUTexture::~GPUTexture() { delete cudaRes; }

namespace {
struct Driver {
  int w, h;
  RGBDOdometry odom;
  GPUTexture a, b;
  Driver(int w_, int h_, float cx, float cy, float fx, float fy) : w(w_), h(h_), odom(w_, h_, cx, cy, fx, fy), a(w_, h_, 0, 0, 0, false), b(w_, h_, 0, 0, 0, false) {}
  GPUTexture* tex(GPUTexture& t, const void* data, size_t elem) {
    t.cudaRes->array = cudaArray{const_cast<void*>(data), w, h, elem};
    return &t;
  }
};
Sophus::SE3d pose_of(const double* T16) { return Sophus::SE3d(efo::se3_from_matrix(T16)); }
}  // namespace

extern "C" {

void* efd_odom_create(int w, int h, float cx, float cy, float fx, float fy) { return new Driver(w, h, cx, cy, fx, fy); }
void efd_odom_destroy(void* p) { delete (Driver*)p; }
void efd_odom_init_icp(void* p, const uint16_t* filteredDepth, float depthCutoff) {
  Driver* d = (Driver*)p;
  d->odom.initICP(d->tex(d->a, filteredDepth, sizeof(uint16_t)), depthCutoff);
}
void efd_odom_init_icp_model(void* p, const float* vtex, const float* ntex, const double* T_wc16) {
  Driver* d = (Driver*)p;
  d->odom.initICPModel(d->tex(d->a, vtex, sizeof(float4)), d->tex(d->b, ntex, sizeof(float4)), pose_of(T_wc16));
}
void efd_odom_init_icp_maps(void* p, const float* vtex, const float* ntex) {
  Driver* d = (Driver*)p;
  d->odom.initICP(d->tex(d->a, vtex, sizeof(float4)), d->tex(d->b, ntex, sizeof(float4)));
}
void efd_odom_init_rgb_model(void* p, const uint8_t* rgba) { Driver* d = (Driver*)p; d->odom.initRGBModel(d->tex(d->a, rgba, sizeof(uchar4))); }
void efd_odom_init_rgb(void* p, const uint8_t* rgba) { Driver* d = (Driver*)p; d->odom.initRGB(d->tex(d->a, rgba, sizeof(uchar4))); }
void efd_odom_init_first_rgb(void* p, const uint8_t* rgba) { Driver* d = (Driver*)p; d->odom.initFirstRGB(d->tex(d->a, rgba, sizeof(uchar4))); }
void efd_odom_track(void* p, double* T_wc16, int rgbOnly, float icpWeight, int pyramid, int fastOdom, int so3) {
  Driver* d = (Driver*)p;
  Sophus::SE3d T = pose_of(T_wc16);
  d->odom.getIncrementalTransformation(T, rgbOnly != 0, icpWeight, pyramid != 0, fastOdom != 0, so3 != 0);
  const efo::M4d M = efo::se3_matrix(T.value());
  std::memcpy(T_wc16, M.m, sizeof(M.m));
}
void efd_odom_stats(const void* p, float* out6, double* lastA36, double* lastb6) {
  const RGBDOdometry& o = ((const Driver*)p)->odom;
  out6[0] = o.lastICPError; out6[1] = o.lastICPCount; out6[2] = o.lastRGBError;
  out6[3] = o.lastRGBCount; out6[4] = o.lastSO3Error; out6[5] = o.lastSO3Count;
  if (lastA36)
    for (int i = 0; i < 6; ++i)
      for (int j = 0; j < 6; ++j) lastA36[i * 6 + j] = o.lastA(i, j);
  if (lastb6)
    for (int i = 0; i < 6; ++i) lastb6[i] = o.lastb(i);
}
void efd_odom_covariance(void* p, double* cov36) {
  const Eigen::MatrixXd c = ((Driver*)p)->odom.getCovariance();
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) cov36[i * 6 + j] = c(i, j);
}
const char* efd_about() {
  return "reference Core/Utils/RGBDOdometry.cpp + OdometryProvider.h compiled by g++ against oracle/host_on_cpu (Eigen / Sophus / "
         "Pangolin / GL interop in miniature) over the reference's Core/Cuda operators compiled through oracle/cuda_on_cpu";
}

}  // extern "C"
