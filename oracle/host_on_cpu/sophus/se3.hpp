// TEST INFRASTRUCTURE — the part of Sophus::SE3d the reference's tracking driver uses (see ../mini_eigen.h), over the oracle's
// own SE(3) arithmetic (efo_linalg.h: unit quaternion x,y,z,w + translation; rotationMatrix / setRotationMatrix as restated there).
#pragma once
#include "../mini_eigen.h"
#include "../../efo_pose.h"

namespace Sophus {
// T.cast<float>(): only its matrix() is used (GlobalModel.cpp:403): quaternion cast to float, renormalised, rotation in float
class SE3f {
 public:
  explicit SE3f(const efo::Mat4f& m) : M(m) {}
  Eigen::Matrix4f matrix() const {
    Eigen::Matrix4f r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r(i, j) = M.m[i * 4 + j];
    return r;
  }
  Eigen::Vector4f operator*(const Eigen::Vector4f& p) const { return matrix() * p; }

 private:
  efo::Mat4f M;
};
class SE3d {
 public:
  SE3d() : T(efo::se3_identity()) { sync_t(); }
  explicit SE3d(const efo::SE3& s) : T(s) { sync_t(); }
  template <int O>
  explicit SE3d(const Eigen::Matrix<double, 4, 4, O>& M) {   // Sophus::SE3d(Matrix4d): MainController.cpp:238-239 (ground-truth poses)
    double m[16];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) m[i * 4 + j] = M(i, j);
    T = efo::se3_from_matrix(m);
    sync_t();
  }
  Eigen::Matrix3d rotationMatrix() const {
    const efo::M3d R = efo::se3_rotation(T);
    Eigen::Matrix3d r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r(i, j) = R.m[i * 3 + j];
    return r;
  }
  Eigen::Vector3d& translation() { return t; }
  const Eigen::Vector3d& translation() const { return t; }
  template <int O>
  void setRotationMatrix(const Eigen::Matrix<double, 3, 3, O>& R) {
    efo::M3d m;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) m.m[i * 3 + j] = R(i, j);
    efo::se3_set_rotation(T, m);
  }
  SE3d inverse() const { return SE3d(efo::se3_inverse(value())); }
  SE3d operator*(const SE3d& o) const { return SE3d(efo::se3_mul(value(), o.value())); }
  // homogeneous point: 4x4 matrix product evaluated left to right (the specification of ElasticFusion.cpp:493-503)
  Eigen::Vector4d operator*(const Eigen::Vector4d& p) const {
    const efo::M4d M = efo::se3_matrix(value());
    Eigen::Vector4d r;
    for (int i = 0; i < 4; ++i) r(i) = ((M.m[i * 4] * p(0) + M.m[i * 4 + 1] * p(1)) + M.m[i * 4 + 2] * p(2)) + M.m[i * 4 + 3] * p(3);
    return r;
  }
  Eigen::Matrix<double, 6, 1> log() const {
    Eigen::Matrix<double, 6, 1> r;
    efo::se3_log_norm(value(), r.data());
    return r;
  }
  Eigen::Matrix4d matrix() const {
    const efo::M4d M = efo::se3_matrix(value());
    Eigen::Matrix4d r;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r(i, j) = M.m[i * 4 + j];
    return r;
  }
  template <typename U>
  SE3f cast() const {
    const efo::M4d M = efo::se3_matrix(value());
    return SE3f(efo::pose_castf(M.m));
  }
  efo::SE3 value() const {
    efo::SE3 r = T;
    for (int i = 0; i < 3; ++i) r.t[i] = t(i);
    return r;
  }

 private:
  void sync_t() { for (int i = 0; i < 3; ++i) t(i) = T.t[i]; }
  efo::SE3 T;
  Eigen::Vector3d t;
};
}  // namespace Sophus
