// TEST INFRASTRUCTURE — CPU oracle (see oracle/README.md). Not part of the product path.
//
// Small fixed-size linear algebra restating the Eigen / Sophus arithmetic that sits on the
// reference's hot path.  Eigen and Sophus are un-vendored submodules in the reference checkout
// (.gitmodules:1-12, third-party/* empty, SHA unpinned) => "parity unpinned" for these ops; we
// restate the published algorithms and pin them with known-answer tests (tests/test_oracle_linalg.py).
//
//   ldlt6_solve        <- Eigen::LDLT (diagonal pivoting) as used at RGBDOdometry.cpp:526,530,534
//   ldlt3f_solve       <- Eigen::LDLT<float 3x3>          RGBDOdometry.cpp:356
//   polar3             <- JacobiSVD U*V^T                  RGBDOdometry.cpp:566-570
//   rodrigues          <- OdometryProvider.h:34-71
//   mat_to_quat / quat_to_mat / se3_* <- Sophus::SE3d (setRotationMatrix, rotationMatrix, inverse,
//                         operator*, log) as used at ElasticFusion.cpp:371-374, IndexMap.cpp:208
#pragma once
#include <cmath>
#include <cstring>
#include <cfloat>
#include <algorithm>
#include <limits>

namespace efo {

struct M3d { double m[9]; };   // row-major
struct V3d { double v[3]; };
struct M4d { double m[16]; };  // row-major

inline M3d m3_identity() { M3d r{}; r.m[0] = r.m[4] = r.m[8] = 1.0; return r; }
inline M4d m4_identity() { M4d r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0; return r; }

inline M3d m3_mul(const M3d& a, const M3d& b) {
  M3d r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += a.m[i * 3 + k] * b.m[k * 3 + j];
      r.m[i * 3 + j] = s;
    }
  return r;
}
inline M3d m3_transpose(const M3d& a) {
  M3d r;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r.m[i * 3 + j] = a.m[j * 3 + i];
  return r;
}
inline V3d m3_mulv(const M3d& a, const V3d& x) {
  V3d r;
  for (int i = 0; i < 3; ++i) r.v[i] = a.m[i * 3] * x.v[0] + a.m[i * 3 + 1] * x.v[1] + a.m[i * 3 + 2] * x.v[2];
  return r;
}
inline M4d m4_mul(const M4d& a, const M4d& b) {
  M4d r;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0;
      for (int k = 0; k < 4; ++k) s += a.m[i * 4 + k] * b.m[k * 4 + j];
      r.m[i * 4 + j] = s;
    }
  return r;
}

// General 3x3 inverse by cofactors (Eigen's fixed-size 3x3 inverse is cofactor based).
inline M3d m3_inverse(const M3d& a) {
  const double* m = a.m;
  double c00 = m[4] * m[8] - m[5] * m[7];
  double c01 = m[5] * m[6] - m[3] * m[8];
  double c02 = m[3] * m[7] - m[4] * m[6];
  double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  double id = 1.0 / det;
  M3d r;
  r.m[0] = c00 * id;
  r.m[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  r.m[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r.m[3] = c01 * id;
  r.m[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r.m[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r.m[6] = c02 * id;
  r.m[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r.m[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  return r;
}
inline void m3f_inverse(const float* m, float* r) {
  float c00 = m[4] * m[8] - m[5] * m[7];
  float c01 = m[5] * m[6] - m[3] * m[8];
  float c02 = m[3] * m[7] - m[4] * m[6];
  float det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  float id = 1.0f / det;
  r[0] = c00 * id;
  r[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  r[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  r[3] = c01 * id;
  r[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  r[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  r[6] = c02 * id;
  r[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  r[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// Inverse of a 4x4 whose last row is [0 0 0 1] with a general (not nec. orthonormal) 3x3 block.
// resultRt.inverse() at RGBDOdometry.cpp:407 is such a matrix.
inline M4d m4_affine_inverse(const M4d& a) {
  M3d A;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) A.m[i * 3 + j] = a.m[i * 4 + j];
  M3d Ai = m3_inverse(A);
  V3d t{{a.m[3], a.m[7], a.m[11]}};
  V3d ti = m3_mulv(Ai, t);
  M4d r = m4_identity();
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) r.m[i * 4 + j] = Ai.m[i * 3 + j];
    r.m[i * 4 + 3] = -ti.v[i];
  }
  return r;
}

// Symmetric N x N solve by LDL^T with diagonal (symmetric) pivoting, the scheme Eigen::LDLT uses:
// at step k pick the largest |diagonal| of the trailing block, swap rows+cols, eliminate.
template <typename T, int N>
inline void ldlt_solve(const T* A_in, const T* b_in, T* x) {
  T A[N * N];
  int perm[N];
  for (int i = 0; i < N * N; ++i) A[i] = A_in[i];
  for (int i = 0; i < N; ++i) perm[i] = i;
  for (int k = 0; k < N; ++k) {
    int p = k;
    T best = std::fabs(A[k * N + k]);
    for (int i = k + 1; i < N; ++i) {
      T v = std::fabs(A[i * N + i]);
      if (v > best) { best = v; p = i; }
    }
    if (p != k) {
      for (int j = 0; j < N; ++j) std::swap(A[k * N + j], A[p * N + j]);
      for (int i = 0; i < N; ++i) std::swap(A[i * N + k], A[i * N + p]);
      std::swap(perm[k], perm[p]);
    }
    T d = A[k * N + k];
    if (d == T(0)) continue;  // singular direction: leave (Eigen zeroes the remaining factor)
    T colk[N];
    for (int i = k + 1; i < N; ++i) colk[i] = A[i * N + k];
    for (int i = k + 1; i < N; ++i) {
      T l = colk[i] / d;
      for (int j = k + 1; j <= i; ++j) {
        A[i * N + j] -= l * colk[j];
        A[j * N + i] = A[i * N + j];
      }
      A[i * N + k] = l;  // store L below the diagonal
    }
    for (int j = k + 1; j < N; ++j) A[k * N + j] = T(0);
  }
  // Solve P^T L D L^T P x = b
  T y[N];
  for (int i = 0; i < N; ++i) y[i] = b_in[perm[i]];
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < i; ++j) y[i] -= A[i * N + j] * y[j];
  for (int i = 0; i < N; ++i) {
    T d = A[i * N + i];
    // Eigen: entries whose |d| is below a tolerance are treated as zero -> solution component 0
    y[i] = (std::fabs(d) > std::numeric_limits<T>::min()) ? y[i] / d : T(0);
  }
  for (int i = N - 1; i >= 0; --i)
    for (int j = i + 1; j < N; ++j) y[i] -= A[j * N + i] * y[j];
  for (int i = 0; i < N; ++i) x[perm[i]] = y[i];
}

// OdometryProvider::rodrigues (OdometryProvider.h:34-71)
inline M3d rodrigues(const V3d& src) {
  M3d dst = m3_identity();
  double rx = src.v[0], ry = src.v[1], rz = src.v[2];
  double theta = std::sqrt(rx * rx + ry * ry + rz * rz);
  if (theta >= DBL_EPSILON) {
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    double c = std::cos(theta), s = std::sin(theta), c1 = 1. - c;
    double itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) dst.m[k] = c * I[k] + c1 * rrt[k] + s * rx_[k];
  }
  return dst;
}

// Orthogonal polar factor U*V^T of a 3x3 (what JacobiSVD U*V^T yields), by one-sided Jacobi
// (Hestenes) sweeps: A*V = U*S  =>  U V^T = (A V) S^-1 V^T.
// one-sided Jacobi SVD of a 3x3 matrix: U (the rotated, column-normalised input) and V; singular values are not needed
inline void svd3_uv(const M3d& Ain, M3d& Uout, M3d& Vout) {
  double A[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  std::memcpy(A, Ain.m, sizeof(A));
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 3; ++i) {
          alpha += A[i * 3 + p] * A[i * 3 + p];
          beta += A[i * 3 + q] * A[i * 3 + q];
          gamma += A[i * 3 + p] * A[i * 3 + q];
        }
        off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
        if (gamma == 0.0) continue;
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = c * ap - s * aq;
          A[i * 3 + q] = s * ap + c * aq;
          double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = c * vp - s * vq;
          V[i * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  // normalise columns of A -> U
  for (int j = 0; j < 3; ++j) {
    double n = 0;
    for (int i = 0; i < 3; ++i) n += A[i * 3 + j] * A[i * 3 + j];
    n = std::sqrt(n);
    for (int i = 0; i < 3; ++i) A[i * 3 + j] /= n;
  }
  std::memcpy(Uout.m, A, sizeof(A));
  std::memcpy(Vout.m, V, sizeof(V));
}
inline M3d polar3(const M3d& Ain) {
  M3d Um, Vm;
  svd3_uv(Ain, Um, Vm);
  const double* A = Um.m;
  const double* V = Vm.m;
  M3d R;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * V[j * 3 + k];
      R.m[i * 3 + j] = s;
    }
  return R;
}

// ---- Sophus::SE3d stand-in: unit quaternion (x,y,z,w) + translation -------------------------
struct SE3 {
  double q[4];  // x y z w
  double t[3];
};
inline SE3 se3_identity() { return SE3{{0, 0, 0, 1}, {0, 0, 0}}; }

// Eigen::Quaternion(Matrix3) — the branchy trace method (Eigen/src/Geometry/Quaternion.h).
inline void mat_to_quat(const M3d& R, double q[4]) {
  const double* m = R.m;
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}
// Eigen::Quaternion::toRotationMatrix
inline M3d quat_to_mat(const double q[4]) {
  double x = q[0], y = q[1], z = q[2], w = q[3];
  double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  double twx = tx * w, twy = ty * w, twz = tz * w;
  double txx = tx * x, txy = ty * x, txz = tz * x;
  double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  M3d R;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
  R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
  return R;
}
inline void se3_set_rotation(SE3& T, const M3d& R) {
  mat_to_quat(R, T.q);
  double n = std::sqrt(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
  for (int i = 0; i < 4; ++i) T.q[i] /= n;  // Sophus normalises on set
}
inline M3d se3_rotation(const SE3& T) { return quat_to_mat(T.q); }
inline void quat_mul(const double a[4], const double b[4], double r[4]) {
  r[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  r[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  r[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  r[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
inline SE3 se3_inverse(const SE3& T) {
  SE3 r;
  r.q[0] = -T.q[0]; r.q[1] = -T.q[1]; r.q[2] = -T.q[2]; r.q[3] = T.q[3];
  M3d Ri = quat_to_mat(r.q);
  V3d t{{T.t[0], T.t[1], T.t[2]}};
  V3d ti = m3_mulv(Ri, t);
  for (int i = 0; i < 3; ++i) r.t[i] = -ti.v[i];
  return r;
}
inline SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r;
  quat_mul(a.q, b.q, r.q);
  double sn = r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3];
  if (sn != 1.0) {  // Sophus SO3 product: first-order renormalisation
    double sc = 2.0 / (1.0 + sn);
    for (int i = 0; i < 4; ++i) r.q[i] *= sc;
  }
  M3d Ra = quat_to_mat(a.q);
  V3d tb{{b.t[0], b.t[1], b.t[2]}};
  V3d rt = m3_mulv(Ra, tb);
  for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] + rt.v[i];
  return r;
}
inline M4d se3_matrix(const SE3& T) {
  M3d R = quat_to_mat(T.q);
  M4d r = m4_identity();
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) r.m[i * 4 + j] = R.m[i * 3 + j];
    r.m[i * 4 + 3] = T.t[i];
  }
  return r;
}
inline SE3 se3_from_matrix(const double M[16]) {
  SE3 T;
  M3d R;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R.m[i * 3 + j] = M[i * 4 + j];
    T.t[i] = M[i * 4 + 3];
  }
  se3_set_rotation(T, R);
  return T;
}
// Sophus::SE3::log() -> (upsilon, omega); returns the 6-vector norm and fills out[6] if given.
inline double se3_log_norm(const SE3& T, double* out6 = nullptr) {
  const double eps = 1e-10;  // Sophus::Constants<double>::epsilon()
  double vx = T.q[0], vy = T.q[1], vz = T.q[2], w = T.q[3];
  double sqn = vx * vx + vy * vy + vz * vz;
  double two_atan;
  double theta;
  if (sqn < eps * eps) {
    double sw = w * w;
    two_atan = 2.0 / w - (2.0 / 3.0) * sqn / (w * sw);
    theta = 2.0 * sqn / w;
  } else {
    // newer Sophus form (SO3::logAndTheta): atan2 with the sign of w folded in
    double n = std::sqrt(sqn);
    double at = (w < 0.0) ? std::atan2(-n, -w) : std::atan2(n, w);
    two_atan = 2.0 * at / n;
    theta = two_atan * n;
  }
  double om[3] = {two_atan * vx, two_atan * vy, two_atan * vz};
  // Omega = hat(om)
  double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
      O2[i * 3 + j] = s;
    }
  double Vi[9];
  double coef;
  if (std::fabs(theta) < eps) {
    coef = 1.0 / 12.0;
  } else {
    double ht = 0.5 * theta;
    coef = (1.0 - theta * std::cos(ht) / (2.0 * std::sin(ht))) / (theta * theta);
  }
  for (int i = 0; i < 9; ++i) Vi[i] = ((i % 4 == 0) ? 1.0 : 0.0) - 0.5 * O[i] + coef * O2[i];
  double up[3];
  for (int i = 0; i < 3; ++i) up[i] = Vi[i * 3] * T.t[0] + Vi[i * 3 + 1] * T.t[1] + Vi[i * 3 + 2] * T.t[2];
  if (out6) {
    for (int i = 0; i < 3; ++i) { out6[i] = up[i]; out6[3 + i] = om[i]; }
  }
  return std::sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2] + om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
}


// Eigen::PartialPivLU<Matrix<double,6,6>>::inverse() as RGBDOdometry::getCovariance uses it (RGBDOdometry.cpp:573-575):
// row-pivoted Doolittle LU (pivot = largest |entry| of the column at or below the diagonal, first one on ties), then
// the inverse column by column from P, L (unit lower) and U by forward / back substitution.
template <typename T, int N>
inline void lu_inverse(const T* A_in, T* inv) {
  T lu[N * N];
  int perm[N];
  for (int i = 0; i < N * N; ++i) lu[i] = A_in[i];
  for (int i = 0; i < N; ++i) perm[i] = i;
  for (int k = 0; k < N; ++k) {
    int piv = k;
    T best = lu[k * N + k] < 0 ? -lu[k * N + k] : lu[k * N + k];
    for (int r = k + 1; r < N; ++r) {
      const T a = lu[r * N + k] < 0 ? -lu[r * N + k] : lu[r * N + k];
      if (a > best) { best = a; piv = r; }
    }
    if (piv != k) {
      for (int c = 0; c < N; ++c) { const T tmp = lu[k * N + c]; lu[k * N + c] = lu[piv * N + c]; lu[piv * N + c] = tmp; }
      const int ti = perm[k]; perm[k] = perm[piv]; perm[piv] = ti;
    }
    for (int r = k + 1; r < N; ++r) {
      lu[r * N + k] = lu[r * N + k] / lu[k * N + k];
      for (int c = k + 1; c < N; ++c) lu[r * N + c] = lu[r * N + c] - lu[r * N + k] * lu[k * N + c];
    }
  }
  for (int col = 0; col < N; ++col) {
    T y[N];
    for (int r = 0; r < N; ++r) {        // L y = P e_col
      T v = perm[r] == col ? T(1) : T(0);
      for (int c = 0; c < r; ++c) v = v - lu[r * N + c] * y[c];
      y[r] = v;
    }
    for (int r = N - 1; r >= 0; --r) {   // U x = y
      T v = y[r];
      for (int c = r + 1; c < N; ++c) v = v - lu[r * N + c] * inv[c * N + col];
      inv[r * N + col] = v / lu[r * N + r];
    }
  }
}
}  // namespace efo
