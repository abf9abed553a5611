// TEST INFRASTRUCTURE — the part of Sophus::SE3d the reference's tracking driver uses (see ../mini_eigen.h), over the oracle's
// own SE(3) arithmetic (efo_linalg.h: unit quaternion x,y,z,w + translation; rotationMatrix / setRotationMatrix as restated there).
#pragma once
#include "../mini_eigen.h"

namespace Sophus {
class SE3d {
 public:
  SE3d() : T(efo::se3_identity()) { sync_t(); }
  explicit SE3d(const efo::SE3& s) : T(s) { sync_t(); }
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
