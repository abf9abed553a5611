// TEST INFRASTRUCTURE — host-pointer C entry points over the REFERENCE's own operator functions
// (Core/Cuda/cudafuncs.cuh:61-169), which oracle/Makefile compiles from /root/reference with the CUDA-on-CPU shim
// (oracle/cuda_on_cpu/).  efr_<op> has exactly the signature of the oracle's efo_<op> (oracle/efo_api.h), so a test
// can run the same inputs through both and compare bit patterns (tests/test_oracle_vs_reference.py).
// This file is ours; it only CALLS the reference (declarations come from the reference's header, included where it lies).
#include "cudafuncs.cuh"   // -I/root/reference/Core/Cuda

#include <vector>

namespace {
template <typename T>
void up2d(DeviceArray2D<T>& d, const void* host, int rows, int cols) { d.upload(host, (size_t)cols * sizeof(T), rows, cols); }
template <typename T>
void down2d(const DeviceArray2D<T>& d, void* host, int cols) { d.download(host, (size_t)cols * sizeof(T)); }
mat33 m33(const float* p) {
  mat33 m;
  for (int r = 0; r < 3; ++r) m.data[r] = make_float3(p[3 * r], p[3 * r + 1], p[3 * r + 2]);
  return m;
}
float3 f3(const float* p) { return make_float3(p[0], p[1], p[2]); }
}  // namespace

extern "C" {

void efr_pyr_down_u16(const uint16_t* src, int scols, int srows, uint16_t* dst) {
  DeviceArray2D<uint16_t> s, d(srows / 2, scols / 2);
  up2d(s, src, srows, scols);
  pyrDown(s, d);
  down2d(d, dst, scols / 2);
}
void efr_create_vmap(const uint16_t* depth, int cols, int rows, float fx, float fy, float cx, float cy, float depthCutoff, float* vmap) {
  DeviceArray2D<uint16_t> d;
  up2d(d, depth, rows, cols);
  DeviceArray2D<float> v;   // createVMap creates it (rows*3 x cols)
  // pre-fill: the reference leaves the y/z planes of invalid pixels untouched (quirk Q3); give them a known value
  std::vector<float> zeros((size_t)3 * rows * cols, 0.0f);
  up2d(v, zeros.data(), 3 * rows, cols);
  createVMap(CameraModel(fx, fy, cx, cy), d, v, depthCutoff);
  down2d(v, vmap, cols);
}
void efr_create_nmap(const float* vmap, int cols, int rows, float* nmap) {
  DeviceArray2D<float> v, n;
  up2d(v, vmap, 3 * rows, cols);
  std::vector<float> zeros((size_t)3 * rows * cols, 0.0f);
  up2d(n, zeros.data(), 3 * rows, cols);
  createNMap(v, n);
  down2d(n, nmap, cols);
}
void efr_transform_maps(float* vmap, float* nmap, int cols, int rows, const float* R9, const float* t3) {
  DeviceArray2D<float> v, n;
  up2d(v, vmap, 3 * rows, cols);
  up2d(n, nmap, 3 * rows, cols);
  tranformMaps(v, n, m33(R9), f3(t3), v, n);   // in place, as RGBDOdometry.cpp:206 calls it
  down2d(v, vmap, cols);
  down2d(n, nmap, cols);
}
void efr_copy_maps(const float* vtex, const float* ntex, int cols, int rows, float* vmaps_tmp, float* vmap, float* nmap) {
  cudaArray va{(void*)vtex, cols, rows, sizeof(float4)}, na{(void*)ntex, cols, rows, sizeof(float4)};
  DeviceArray<float> tmp((size_t)rows * cols * 4);
  DeviceArray2D<float> v(3 * rows, cols), n(3 * rows, cols);
  std::vector<float> zeros((size_t)4 * rows * cols, 0.0f);
  tmp.upload(zeros.data(), zeros.size());
  up2d(v, zeros.data(), 3 * rows, cols);
  up2d(n, zeros.data(), 3 * rows, cols);
  copyMaps(&va, &na, (size_t)cols, (size_t)rows, tmp, v, n);
  tmp.download(vmaps_tmp);
  down2d(v, vmap, cols);
  down2d(n, nmap, cols);
}
void efr_resize_map(const float* in, int scols, int srows, float* out, int normalize) {
  DeviceArray2D<float> i, o;
  up2d(i, in, 3 * srows, scols);
  std::vector<float> zeros((size_t)3 * (srows / 2) * (scols / 2), 0.0f);
  up2d(o, zeros.data(), 3 * (srows / 2), scols / 2);
  if (normalize) resizeNMap(i, o); else resizeVMap(i, o);
  down2d(o, out, scols / 2);
}
void efr_pyr_down_gauss_f(const float* src, int scols, int srows, float* dst) {
  DeviceArray2D<float> s, d(srows / 2, scols / 2);
  up2d(s, src, srows, scols);
  pyrDownGaussF(s, d);
  down2d(d, dst, scols / 2);
}
void efr_pyr_down_uchar_gauss(const uint8_t* src, int scols, int srows, uint8_t* dst) {
  DeviceArray2D<uint8_t> s, d(srows / 2, scols / 2);
  up2d(s, src, srows, scols);
  pyrDownUcharGauss(s, d);
  down2d(d, dst, scols / 2);
}
void efr_vertices_to_depth(const float* vmaps_tmp, int cols, int rows, float cutOff, float* dst) {
  DeviceArray<float> tmp;
  tmp.upload(vmaps_tmp, (size_t)rows * cols * 4);
  DeviceArray2D<float> d(rows, cols);
  verticesToDepth(tmp, d, cutOff);
  down2d(d, dst, cols);
}
void efr_bgr_to_intensity(const uint8_t* rgba, int cols, int rows, uint8_t* dst) {
  cudaArray a{(void*)rgba, cols, rows, sizeof(uchar4)};
  DeviceArray2D<uint8_t> d(rows, cols);
  imageBGRToIntensity(&a, d);
  down2d(d, dst, cols);
}
void efr_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy) {
  DeviceArray2D<uint8_t> s;
  up2d(s, src, rows, cols);
  DeviceArray2D<int16_t> x(rows, cols), y(rows, cols);
  computeDerivativeImages(s, x, y);
  down2d(x, dx, cols);
  down2d(y, dy, cols);
}
void efr_project_to_point_cloud(const float* depth, int cols, int rows, float fx, float fy, float cx, float cy, float* cloud) {
  DeviceArray2D<float> d;
  up2d(d, depth, rows, cols);
  DeviceArray2D<float3> c(rows, cols);
  CameraModel k(fx, fy, cx, cy);
  projectToPointCloud(d, c, k, 0);   // level 0: intrinsics / 1
  down2d(c, cloud, cols);
}
void efr_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr, const float* Rprev_inv,
                  const float* tprev, float fx, float fy, float cx, float cy, const float* vmap_g_prev, const float* nmap_g_prev,
                  float distThres, float angleThres, int cols, int rows, float* A36, float* b6, float* residual2) {
  DeviceArray2D<float> vc, nc, vp, np_;
  up2d(vc, vmap_curr, 3 * rows, cols);
  up2d(nc, nmap_curr, 3 * rows, cols);
  up2d(vp, vmap_g_prev, 3 * rows, cols);
  up2d(np_, nmap_g_prev, 3 * rows, cols);
  DeviceArray<JtJJtrSE3> sum(MAX_THREADS), out(1);   // RGBDOdometry.cpp:53-54
  icpStep(m33(Rcurr), f3(tcurr), vc, nc, m33(Rprev_inv), f3(tprev), CameraModel(fx, fy, cx, cy), vp, np_, distThres, angleThres, sum, out,
          A36, b6, residual2);
}
void efr_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth, const float* nextDepth,
                      const uint8_t* lastImage, const uint8_t* nextImage, void* corres_out, float maxDepthDelta, const float* kt3,
                      const float* krkinv9, int cols, int rows, int* sigmaSum, int* count) {
  DeviceArray2D<int16_t> dx, dy;
  up2d(dx, dIdx, rows, cols);
  up2d(dy, dIdy, rows, cols);
  DeviceArray2D<float> ld, nd;
  up2d(ld, lastDepth, rows, cols);
  up2d(nd, nextDepth, rows, cols);
  DeviceArray2D<uint8_t> li, ni;
  up2d(li, lastImage, rows, cols);
  up2d(ni, nextImage, rows, cols);
  DeviceArray2D<DataTerm> corres;
  std::vector<DataTerm> zeros((size_t)rows * cols);
  memset(zeros.data(), 0, zeros.size() * sizeof(DataTerm));
  up2d(corres, zeros.data(), rows, cols);
  DeviceArray<int2> sumResidual(MAX_THREADS);
  int s = 0, c = 0;
  computeRgbResidual(minScale, dx, dy, ld, nd, li, ni, corres, sumResidual, maxDepthDelta, f3(kt3), m33(krkinv9), s, c);
  *sigmaSum = s;
  *count = c;
  down2d(corres, corres_out, cols);
}
void efr_rgb_step(const void* corres_in, float sigma, const float* cloud, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                  float sobelScale, int cols, int rows, float* A36, float* b6) {
  DeviceArray2D<DataTerm> corres;
  up2d(corres, corres_in, rows, cols);
  DeviceArray2D<float3> cl;
  up2d(cl, cloud, rows, cols);
  DeviceArray2D<int16_t> dx, dy;
  up2d(dx, dIdx, rows, cols);
  up2d(dy, dIdy, rows, cols);
  DeviceArray<JtJJtrSE3> sum(MAX_THREADS), out(1);
  rgbStep(corres, sigma, cl, fx, fy, dx, dy, sobelScale, sum, out, A36, b6);
}
void efr_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis9, const float* kinv9, const float* krlr9,
                  int cols, int rows, float* A9, float* b3, float* residual2) {
  DeviceArray2D<uint8_t> li, ni;
  up2d(li, lastImage, rows, cols);
  up2d(ni, nextImage, rows, cols);
  DeviceArray<JtJJtrSO3> sum(MAX_THREADS), out(1);
  so3Step(li, ni, m33(imageBasis9), m33(kinv9), m33(krlr9), sum, out, A9, b3, residual2);
}
const char* efr_about() {
  return "reference Core/Cuda/{reduce,cudafuncs}.cu + containers/device_memory.cpp compiled by g++ with oracle/cuda_on_cpu "
         "(-ffp-contract=off, warpSize 32, rsqrtf = 1/sqrtf)";
}

}  // extern "C"
