// TEST INFRASTRUCTURE — the two float pose matrices the reference's host code hands to its shaders, shared by the
// oracle's map passes (efo_map.cpp) and the shader bridge (ref_glsl_bridge.cpp).
#pragma once
#include "efo_common.h"
#include "efo_linalg.h"

namespace efo {

struct Mat4f { float m[16]; };  // row-major
// T_wc.inverse().matrix().cast<float>()  (IndexMap.cpp:208, GlobalModel.cpp:567)
inline Mat4f T_cw_float(const double* T_wc16) {
  SE3 T = se3_from_matrix(T_wc16);
  M4d Mi = se3_matrix(se3_inverse(T));
  Mat4f r;
  for (int i = 0; i < 16; ++i) r.m[i] = (float)Mi.m[i];
  return r;
}
// T_wc.cast<float>().matrix()  (GlobalModel.cpp:403): quaternion cast to float, renormalised in float,
// rotation matrix evaluated in float (Sophus SO3 ctor + Eigen toRotationMatrix).
inline Mat4f pose_castf(const double* T_wc16) {
  SE3 T = se3_from_matrix(T_wc16);
  float q[4] = {(float)T.q[0], (float)T.q[1], (float)T.q[2], (float)T.q[3]};
  float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
  float x = q[0], y = q[1], z = q[2], w = q[3];
  float tx = 2 * x, ty = 2 * y, tz = 2 * z;
  float twx = tx * w, twy = ty * w, twz = tz * w;
  float txx = tx * x, txy = ty * x, txz = tz * x;
  float tyy = ty * y, tyz = tz * y, tzz = tz * z;
  Mat4f r{};
  r.m[0] = 1 - (tyy + tzz); r.m[1] = txy - twz;       r.m[2] = txz + twy;       r.m[3] = (float)T.t[0];
  r.m[4] = txy + twz;       r.m[5] = 1 - (txx + tzz); r.m[6] = tyz - twx;       r.m[7] = (float)T.t[1];
  r.m[8] = txz - twy;       r.m[9] = tyz + twx;       r.m[10] = 1 - (txx + tyy); r.m[11] = (float)T.t[2];
  r.m[15] = 1;
  return r;
}

}  // namespace efo
