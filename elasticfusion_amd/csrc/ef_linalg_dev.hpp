// Small dense linear algebra that the reference runs on the host between kernel launches
// (Eigen LDLT / inverse / JacobiSVD, Sophus SE3, OdometryProvider::rodrigues), re-expressed as
// single-lane __device__ code so the Gauss-Newton loop never leaves the GPU
// (reference call sites: Core/Utils/RGBDOdometry.cpp:309-367,407-417,516-551,566-570;
//  Core/Utils/OdometryProvider.h:34-96; Core/ElasticFusion.cpp:369-383).
// Usable from host code too (the C-ABI operator entry points call the same functions).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <float.h>

namespace efl {

#define EFL_HD __host__ __device__ __forceinline__

EFL_HD void m3_identity(double* r) { for (int i = 0; i < 9; ++i) r[i] = (i % 4 == 0) ? 1.0 : 0.0; }
EFL_HD void m4_identity(double* r) { for (int i = 0; i < 16; ++i) r[i] = (i % 5 == 0) ? 1.0 : 0.0; }

EFL_HD void m3_mul(const double* a, const double* b, double* r) {
  double o[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) s += a[i * 3 + k] * b[k * 3 + j];
      o[i * 3 + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 9; ++i) r[i] = o[i];
}
EFL_HD void m3_mulv(const double* a, const double* x, double* r) {
  double o[3];
  for (int i = 0; i < 3; ++i) o[i] = a[i * 3] * x[0] + a[i * 3 + 1] * x[1] + a[i * 3 + 2] * x[2];
  for (int i = 0; i < 3; ++i) r[i] = o[i];
}
EFL_HD void m4_mul(const double* a, const double* b, double* r) {
  double o[16];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += a[i * 4 + k] * b[k * 4 + j];
      o[i * 4 + j] = s;
    }
#pragma unroll
  for (int i = 0; i < 16; ++i) r[i] = o[i];
}

template <typename T>
EFL_HD void m3_inverse(const T* m, T* r) {
  T c00 = m[4] * m[8] - m[5] * m[7];
  T c01 = m[5] * m[6] - m[3] * m[8];
  T c02 = m[3] * m[7] - m[4] * m[6];
  T det = m[0] * c00 + m[1] * c01 + m[2] * c02;
  T id = T(1) / det;
  T o[9];
  o[0] = c00 * id;
  o[1] = (m[2] * m[7] - m[1] * m[8]) * id;
  o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
  o[3] = c01 * id;
  o[4] = (m[0] * m[8] - m[2] * m[6]) * id;
  o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
  o[6] = c02 * id;
  o[7] = (m[1] * m[6] - m[0] * m[7]) * id;
  o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
  for (int i = 0; i < 9; ++i) r[i] = o[i];
}

// inverse of [A t; 0 1] with general 3x3 A
EFL_HD void m4_affine_inverse(const double* a, double* r) {
  double A[9], Ai[9], t[3], ti[3];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) A[i * 3 + j] = a[i * 4 + j];
    t[i] = a[i * 4 + 3];
  }
  m3_inverse<double>(A, Ai);
  m3_mulv(Ai, t, ti);
  m4_identity(r);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) r[i * 4 + j] = Ai[i * 3 + j];
    r[i * 4 + 3] = -ti[i];
  }
}

// LDL^T with symmetric (diagonal) pivoting, as Eigen::LDLT does; A is N x N row-major symmetric.
// Written so that every array index is a compile-time constant after unrolling (the data-dependent pivot
// row/column swap is a chain of predicated swaps): the whole factorisation lives in registers — the
// dynamically indexed version spilled to scratch and cost ~20 us per solve on gfx950.
template <typename T>
EFL_HD void cswap(bool c, T& a, T& b) { const T x = a, y = b; a = c ? y : x; b = c ? x : y; }

template <typename T, int N>
EFL_HD void ldlt_solve(const T* A_in, const T* b_in, T* x) {
  T A[N][N];
  int perm[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    perm[i] = i;
#pragma unroll
    for (int j = 0; j < N; ++j) A[i][j] = A_in[i * N + j];
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    int p = k;
    T best = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const T v = fabs(A[i][i]);
      const bool g = v > best;
      best = g ? v : best;
      p = g ? i : p;
    }
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const bool sw = (p == i);
#pragma unroll
      for (int j = 0; j < N; ++j) cswap(sw, A[k][j], A[i][j]);
#pragma unroll
      for (int r = 0; r < N; ++r) cswap(sw, A[r][k], A[r][i]);
      cswap(sw, perm[k], perm[i]);
    }
    const T d = A[k][k];
    const bool live = !(d == T(0));
    T colk[N];
#pragma unroll
    for (int i = k + 1; i < N; ++i) colk[i] = A[i][k];
#pragma unroll
    for (int i = k + 1; i < N; ++i) {
      const T l = colk[i] / d;
#pragma unroll
      for (int j = k + 1; j <= i; ++j) {
        const T v = A[i][j] - l * colk[j];
        A[i][j] = live ? v : A[i][j];
        A[j][i] = A[i][j];
      }
      A[i][k] = live ? l : A[i][k];
    }
#pragma unroll
    for (int j = k + 1; j < N; ++j) A[k][j] = live ? T(0) : A[k][j];
  }
  T y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    T v = T(0);
#pragma unroll
    for (int j = 0; j < N; ++j) v = (perm[i] == j) ? b_in[j] : v;
    y[i] = v;
  }
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int j = 0; j < i; ++j) y[i] -= A[i][j] * y[j];
  const T tiny = sizeof(T) == 8 ? (T)DBL_MIN : (T)FLT_MIN;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const T d = A[i][i];
    y[i] = (fabs(d) > tiny) ? y[i] / d : T(0);
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i)
#pragma unroll
    for (int j = i + 1; j < N; ++j) y[i] -= A[j][i] * y[j];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    T v = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) v = (perm[i] == j) ? y[i] : v;
    x[j] = v;
  }
}

// OdometryProvider::rodrigues
EFL_HD void rodrigues(const double* src, double* dst) {
  m3_identity(dst);
  double rx = src[0], ry = src[1], rz = src[2];
  const double theta = sqrt(rx * rx + ry * ry + rz * rz);
  if (theta >= DBL_EPSILON) {
    const double c = cos(theta), s = sin(theta), c1 = 1. - c;
    const double itheta = theta ? 1. / theta : 0.;
    rx *= itheta; ry *= itheta; rz *= itheta;
    const double rrt[9] = {rx * rx, rx * ry, rx * rz, rx * ry, ry * ry, ry * rz, rx * rz, ry * rz, rz * rz};
    const double rx_[9] = {0, -rz, ry, rz, 0, -rx, -ry, rx, 0};
    for (int k = 0; k < 9; ++k) dst[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx_[k];
  }
}

// nearest rotation U V^T of a 3x3 via one-sided Jacobi sweeps
EFL_HD void polar3(const double* Ain, double* R) {
  double A[9], V[9];
  for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
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
        const double rel = fabs(gamma) / sqrt(alpha * beta);
        off = rel > off ? rel : off;
        if (gamma == 0.0) continue;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double ap = A[i * 3 + p], aq = A[i * 3 + q];
          A[i * 3 + p] = c * ap - s * aq;
          A[i * 3 + q] = s * ap + c * aq;
          const double vp = V[i * 3 + p], vq = V[i * 3 + q];
          V[i * 3 + p] = c * vp - s * vq;
          V[i * 3 + q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j) {
    double n = 0;
    for (int i = 0; i < 3; ++i) n += A[i * 3 + j] * A[i * 3 + j];
    n = sqrt(n);
    for (int i = 0; i < 3; ++i) A[i * 3 + j] /= n;
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double s = 0;
      for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * V[j * 3 + k];
      R[i * 3 + j] = s;
    }
}

// ---- Sophus::SE3d stand-in: unit quaternion (x,y,z,w) + translation ----
struct SE3 { double q[4]; double t[3]; };

// the branch of Eigen::Quaternion(Matrix3) for a non-positive trace, largest diagonal entry I: every index a compile-time constant (a
// run-time i put q[] into scratch memory on gfx950 — 64 bytes per lane, the only private segment of the whole library — and a kernel
// with a private segment pays for its set-up at every dispatch)
template <int I>
EFL_HD void mat_to_quat_diag(const double* m, double* q) {
  constexpr int J = (I + 1) % 3, K = (J + 1) % 3;
  double t = sqrt(m[I * 3 + I] - m[J * 3 + J] - m[K * 3 + K] + 1.0);
  q[I] = 0.5 * t;
  t = 0.5 / t;
  q[3] = (m[K * 3 + J] - m[J * 3 + K]) * t;
  q[J] = (m[J * 3 + I] + m[I * 3 + J]) * t;
  q[K] = (m[K * 3 + I] + m[I * 3 + K]) * t;
}
EFL_HD void mat_to_quat(const double* m, double* q) {  // Eigen::Quaternion(Matrix3)
  double t = m[0] + m[4] + m[8];
  if (t > 0.0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[7] - m[5]) * t;
    q[1] = (m[2] - m[6]) * t;
    q[2] = (m[3] - m[1]) * t;
  } else {
    const bool one = m[4] > m[0];
    const bool two = m[8] > (one ? m[4] : m[0]);
    if (two) mat_to_quat_diag<2>(m, q);
    else if (one) mat_to_quat_diag<1>(m, q);
    else mat_to_quat_diag<0>(m, q);
  }
}
template <typename T>
EFL_HD void quat_to_mat(const T* q, T* R) {  // Eigen::Quaternion::toRotationMatrix
  const T x = q[0], y = q[1], z = q[2], w = q[3];
  const T tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const T twx = tx * w, twy = ty * w, twz = tz * w;
  const T txx = tx * x, txy = ty * x, txz = tz * x;
  const T tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
EFL_HD void se3_set_rotation(SE3& T, const double* R) {
  mat_to_quat(R, T.q);
  const double n = sqrt(T.q[0] * T.q[0] + T.q[1] * T.q[1] + T.q[2] * T.q[2] + T.q[3] * T.q[3]);
  for (int i = 0; i < 4; ++i) T.q[i] /= n;
}
EFL_HD SE3 se3_from_matrix(const double* M) {
  SE3 T;
  double R[9];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = M[i * 4 + j];
    T.t[i] = M[i * 4 + 3];
  }
  se3_set_rotation(T, R);
  return T;
}
EFL_HD void se3_matrix(const SE3& T, double* M) {
  double R[9];
  quat_to_mat<double>(T.q, R);
  m4_identity(M);
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M[i * 4 + j] = R[i * 3 + j];
    M[i * 4 + 3] = T.t[i];
  }
}
EFL_HD SE3 se3_inverse(const SE3& T) {
  SE3 r;
  r.q[0] = -T.q[0]; r.q[1] = -T.q[1]; r.q[2] = -T.q[2]; r.q[3] = T.q[3];
  double Ri[9], ti[3];
  quat_to_mat<double>(r.q, Ri);
  m3_mulv(Ri, T.t, ti);
  for (int i = 0; i < 3; ++i) r.t[i] = -ti[i];
  return r;
}
EFL_HD SE3 se3_mul(const SE3& a, const SE3& b) {
  SE3 r;
  r.q[3] = a.q[3] * b.q[3] - a.q[0] * b.q[0] - a.q[1] * b.q[1] - a.q[2] * b.q[2];
  r.q[0] = a.q[3] * b.q[0] + a.q[0] * b.q[3] + a.q[1] * b.q[2] - a.q[2] * b.q[1];
  r.q[1] = a.q[3] * b.q[1] + a.q[1] * b.q[3] + a.q[2] * b.q[0] - a.q[0] * b.q[2];
  r.q[2] = a.q[3] * b.q[2] + a.q[2] * b.q[3] + a.q[0] * b.q[1] - a.q[1] * b.q[0];
  const double sn = r.q[0] * r.q[0] + r.q[1] * r.q[1] + r.q[2] * r.q[2] + r.q[3] * r.q[3];
  if (sn != 1.0) {
    const double sc = 2.0 / (1.0 + sn);
    for (int i = 0; i < 4; ++i) r.q[i] *= sc;
  }
  double Ra[9], rt[3];
  quat_to_mat<double>(a.q, Ra);
  m3_mulv(Ra, b.t, rt);
  for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] + rt[i];
  return r;
}
// || SE3::log() ||  (6-vector norm)
EFL_HD double se3_log_norm(const SE3& T) {
  const double eps = 1e-10;
  const double vx = T.q[0], vy = T.q[1], vz = T.q[2], w = T.q[3];
  const double sqn = vx * vx + vy * vy + vz * vz;
  double two_atan, theta;
  if (sqn < eps * eps) {
    two_atan = 2.0 / w - (2.0 / 3.0) * sqn / (w * (w * w));
    theta = 2.0 * sqn / w;
  } else {
    const double n = sqrt(sqn);
    const double at = (w < 0.0) ? atan2(-n, -w) : atan2(n, w);
    two_atan = 2.0 * at / n;
    theta = two_atan * n;
  }
  const double om[3] = {two_atan * vx, two_atan * vy, two_atan * vz};
  const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
  double O2[9];
  m3_mul(O, O, O2);
  double coef;
  if (fabs(theta) < eps) coef = 1.0 / 12.0;
  else {
    const double ht = 0.5 * theta;
    coef = (1.0 - theta * cos(ht) / (2.0 * sin(ht))) / (theta * theta);
  }
  double up[3];
  for (int i = 0; i < 3; ++i) {
    double s = 0;
    for (int j = 0; j < 3; ++j) {
      const double vi = ((i == j) ? 1.0 : 0.0) - 0.5 * O[i * 3 + j] + coef * O2[i * 3 + j];
      s += vi * T.t[j];
    }
    up[i] = s;
  }
  return sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2] + om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
}
// T.inverse().matrix().cast<float>()
EFL_HD void se3_inverse_matrix_f(const SE3& T, float* M16) {
  double M[16];
  se3_matrix(se3_inverse(T), M);
  for (int i = 0; i < 16; ++i) M16[i] = (float)M[i];
}
// T.cast<float>().matrix(): float quaternion, renormalised in float, rotation evaluated in float
EFL_HD void se3_castf_matrix(const SE3& T, float* M16) {
  float q[4] = {(float)T.q[0], (float)T.q[1], (float)T.q[2], (float)T.q[3]};
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; ++i) q[i] /= n;
  float R[9];
  quat_to_mat<float>(q, R);
  for (int i = 0; i < 16; ++i) M16[i] = 0.f;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) M16[i * 4 + j] = R[i * 3 + j];
    M16[i * 4 + 3] = (float)T.t[i];
  }
  M16[15] = 1.f;
}


// RGBDOdometry::getCovariance (RGBDOdometry.cpp:573-575): Eigen's PartialPivLU inverse of the 6x6 normal matrix.  Row-pivoted
// Doolittle LU (largest |entry| at or below the diagonal, first on ties), inverse by forward / back substitution per column.
template <typename T, int N>
EFL_HD void lu_inverse(const T* A_in, T* inv) {
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
    for (int r = 0; r < N; ++r) {
      T v = perm[r] == col ? T(1) : T(0);
      for (int c = 0; c < r; ++c) v = v - lu[r * N + c] * y[c];
      y[r] = v;
    }
    for (int r = N - 1; r >= 0; --r) {
      T v = y[r];
      for (int c = r + 1; c < N; ++c) v = v - lu[r * N + c] * inv[c * N + col];
      inv[r * N + col] = v / lu[r * N + r];
    }
  }
}
}  // namespace efl
