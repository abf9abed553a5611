// TEST INFRASTRUCTURE — "the host libraries on the CPU, in miniature": the part of the Eigen API that the reference's
// Core/Utils/RGBDOdometry.{h,cpp} and OdometryProvider.h use, so that THOSE SOURCE FILES can be compiled where they lie
// (oracle/Makefile, target refdriver) although Eigen is not vendored by the reference checkout and absent from this image.
// Written from scratch for this repository; nothing here comes from Eigen.  What it pins and what it does not:
//   * compiled from the reference: the whole control flow of getIncrementalTransformation — which operators run in which
//     order on which buffers, the iteration schedule, the convergence / divergence / break tests, every scalar expression
//     the source spells out (sigmaVal's precedence quirk Q2, pow(...)/pow(...), the 0.3 m guard, ...);
//   * supplied by this header, NOT pinned: the small dense primitives behind the Eigen calls (matrix products evaluated
//     left to right without fused multiply-add, 3x3 / 4x4 inverse, LDL^T solve, one-sided Jacobi SVD, partial-pivot LU,
//     Isometry inverse / product).  They are the oracle's own restatements (efo_linalg.h), pinned by known-answer tests.
// No expression templates: every operation returns a plain value.
#pragma once
#include <cassert>
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <memory>
#include <vector>
#include "../efo_linalg.h"

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

enum { ColMajor = 0, RowMajor = 1 };
enum { ComputeFullU = 4, ComputeFullV = 16 };
enum { Isometry = 1 };
constexpr int Dynamic = -1;

template <typename T, int R, int C, int O = ColMajor>
class Matrix;

// a view of a rectangular part of a fixed-size matrix (topLeftCorner / topRightCorner)
template <typename M>
struct Block {
  M& m;
  int r0, c0, nr, nc;
  template <typename U, int R2, int C2, int O2>
  Block& operator=(const Matrix<U, R2, C2, O2>& o) {
    for (int i = 0; i < nr; ++i)
      for (int j = 0; j < nc; ++j) m(r0 + i, c0 + j) = o(i, j);
    return *this;
  }
  auto operator()(int i, int j) const { return m(r0 + i, c0 + j); }
};

template <typename T, int N>
struct LDLT {
  T a[N * N];   // row-major copy
  template <int O>
  Matrix<T, N, 1> solve(const Matrix<T, N, 1, O>& b) const;
};
template <typename T, int N, int O>
struct PartialPivLU {
  T a[N * N];
  Matrix<T, N, N, O> inverse() const;
};

template <typename T, int R, int C, int O>
class Matrix {
 public:
  T m[R * C];

  Matrix() {}
  template <typename U, int O2>
  Matrix(const Matrix<U, R, C, O2>& o) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) (*this)(i, j) = o(i, j);
  }
  template <typename M2>
  Matrix(const Block<M2>& b) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) (*this)(i, j) = b(i, j);
  }
  Matrix(T x, T y) {
    static_assert(R * C == 2, "two-coefficient constructor");
    m[0] = x; m[1] = y;
  }
  Matrix(T x, T y, T z) {
    static_assert(R * C == 3, "three-coefficient constructor");
    m[0] = x; m[1] = y; m[2] = z;
  }
  Matrix(T x, T y, T z, T w) {
    static_assert(R * C == 4, "four-coefficient constructor");
    m[0] = x; m[1] = y; m[2] = z; m[3] = w;
  }
  template <typename U, int O2>
  Matrix& operator=(const Matrix<U, R, C, O2>& o) {
    Matrix t;   // the right-hand side may alias *this
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) t(i, j) = o(i, j);
    std::memcpy(m, t.m, sizeof(m));
    return *this;
  }
  template <typename M2>
  Matrix& operator=(const Block<M2>& b) {
    Matrix t;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) t(i, j) = b(i, j);
    std::memcpy(m, t.m, sizeof(m));
    return *this;
  }

  static Matrix Zero() {
    Matrix z;
    for (int i = 0; i < R * C; ++i) z.m[i] = T(0);
    return z;
  }
  static Matrix Identity() {
    Matrix z = Zero();
    for (int i = 0; i < (R < C ? R : C); ++i) z(i, i) = T(1);
    return z;
  }

  static constexpr int idx(int i, int j) { return O == RowMajor ? i * C + j : j * R + i; }
  T& operator()(int i, int j) { return m[idx(i, j)]; }
  const T& operator()(int i, int j) const { return m[idx(i, j)]; }
  T& operator()(int i) { return m[i]; }
  const T& operator()(int i) const { return m[i]; }
  T& operator[](int i) { return m[i]; }
  const T& operator[](int i) const { return m[i]; }
  T* data() { return m; }
  const T* data() const { return m; }
  int rows() const { return R; }
  int cols() const { return C; }

  template <typename U>
  Matrix<U, R, C, O> cast() const {
    Matrix<U, R, C, O> r;
    for (int i = 0; i < R * C; ++i) r.m[i] = (U)m[i];
    return r;
  }
  Matrix eval() const { return *this; }
  void setIdentity() { *this = Identity(); }
  void setZero() { *this = Zero(); }
  template <int O2>
  Matrix& operator+=(const Matrix<T, R, C, O2>& o) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) (*this)(i, j) += o(i, j);
    return *this;
  }
  template <int O2>
  Matrix& operator-=(const Matrix<T, R, C, O2>& o) {
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) (*this)(i, j) -= o(i, j);
    return *this;
  }
  // ---- only used by the reference's display code (MainController.cpp:262-286, compiled but never run: tests/test_front_end_compiles.py) ----
  Matrix normalized() const {
    Matrix r;
    const T n = norm();
    for (int i = 0; i < R * C; ++i) r.m[i] = m[i] / n;
    return r;
  }
  template <int O2>
  Matrix cross(const Matrix<T, R, C, O2>& o) const {
    static_assert(R * C == 3, "3-vectors");
    return Matrix(m[1] * o.m[2] - m[2] * o.m[1], m[2] * o.m[0] - m[0] * o.m[2], m[0] * o.m[1] - m[1] * o.m[0]);
  }
  // m << a, b, c, ...: coefficients in ROW order
  struct CommaInit {
    Matrix& dst;
    int n;
    template <typename U>
    CommaInit& operator,(const U& v) { dst(n / C, n % C) = (T)v; ++n; return *this; }
  };
  template <typename U>
  CommaInit operator<<(const U& v) { (*this)(0, 0) = (T)v; return CommaInit{*this, 1}; }
  Matrix<T, R, 1> col(int j) const {
    Matrix<T, R, 1> r;
    for (int i = 0; i < R; ++i) r.m[i] = (*this)(i, j);
    return r;
  }
  template <int O2>
  T dot(const Matrix<T, R, C, O2>& o) const {   // coefficient products summed in storage order
    T s = m[0] * o.m[0];
    for (int i = 1; i < R * C; ++i) s += m[i] * o.m[i];
    return s;
  }
  T squaredNorm() const {
    T s = m[0] * m[0];
    for (int i = 1; i < R * C; ++i) s += m[i] * m[i];
    return s;
  }
  Matrix<T, C, R, O> transpose() const {
    Matrix<T, C, R, O> r;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) r(j, i) = (*this)(i, j);
    return r;
  }
  T norm() const {   // sqrt of the coefficients' squares summed in storage order
    T s = m[0] * m[0];
    for (int i = 1; i < R * C; ++i) s += m[i] * m[i];
    return std::sqrt(s);
  }
  template <int N>
  Matrix<T, N, 1> head() const {
    Matrix<T, N, 1> r;
    for (int i = 0; i < N; ++i) r.m[i] = m[i];
    return r;
  }
  Matrix<T, 3, 1> head(int n) const { assert(n == 3); return head<3>(); }
  Block<Matrix> topLeftCorner(int nr, int nc) { return Block<Matrix>{*this, 0, 0, nr, nc}; }
  Block<Matrix> topRightCorner(int nr, int nc) { return Block<Matrix>{*this, 0, C - nc, nr, nc}; }

  Matrix inverse() const;                    // 3x3 and 4x4 (efo_linalg.h)
  LDLT<T, R> ldlt() const {
    static_assert(R == C, "square");
    LDLT<T, R> d;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) d.a[i * C + j] = (*this)(i, j);
    return d;
  }
  PartialPivLU<T, R, O> lu() const {
    PartialPivLU<T, R, O> d;
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) d.a[i * C + j] = (*this)(i, j);
    return d;
  }
};

// products: coefficient (i, j) = a(i,0) b(0,j) + a(i,1) b(1,j) + ... summed left to right, each operation IEEE (no contraction)
template <typename T, int R, int K, int C, int O1, int O2>
Matrix<T, R, C, O1> operator*(const Matrix<T, R, K, O1>& a, const Matrix<T, K, C, O2>& b) {
  Matrix<T, R, C, O1> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) {
      T s = a(i, 0) * b(0, j);
      for (int k = 1; k < K; ++k) s += a(i, k) * b(k, j);
      r(i, j) = s;
    }
  return r;
}
template <typename T, int R, int C, int O>
Matrix<T, R, C, O> operator*(T s, const Matrix<T, R, C, O>& a) {
  Matrix<T, R, C, O> r;
  for (int i = 0; i < R * C; ++i) r.m[i] = s * a.m[i];
  return r;
}
template <typename T, int R, int C, int O>
Matrix<T, R, C, O> operator*(const Matrix<T, R, C, O>& a, T s) {
  Matrix<T, R, C, O> r;
  for (int i = 0; i < R * C; ++i) r.m[i] = a.m[i] * s;
  return r;
}
template <typename T, int R, int C, int O, int O2>
Matrix<T, R, C, O> operator+(const Matrix<T, R, C, O>& a, const Matrix<T, R, C, O2>& b) {
  Matrix<T, R, C, O> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r(i, j) = a(i, j) + b(i, j);
  return r;
}
template <typename T, int R, int C, int O, int O2>
Matrix<T, R, C, O> operator-(const Matrix<T, R, C, O>& a, const Matrix<T, R, C, O2>& b) {
  Matrix<T, R, C, O> r;
  for (int i = 0; i < R; ++i)
    for (int j = 0; j < C; ++j) r(i, j) = a(i, j) - b(i, j);
  return r;
}

template <typename T, int R, int C, int O>
Matrix<T, R, C, O> Matrix<T, R, C, O>::inverse() const {
  static_assert(R == C && (R == 3 || R == 4), "3x3 or 4x4");
  Matrix r;
  if constexpr (R == 3 && sizeof(T) == sizeof(double)) {
    efo::M3d a;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) a.m[i * 3 + j] = (*this)(i, j);
    const efo::M3d b = efo::m3_inverse(a);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r(i, j) = b.m[i * 3 + j];
  } else if constexpr (R == 3) {
    float a[9], b[9];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) a[i * 3 + j] = (*this)(i, j);
    efo::m3f_inverse(a, b);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) r(i, j) = b[i * 3 + j];
  } else {
    // the only 4x4 matrix the driver inverts is the rigid resultRt (RGBDOdometry.cpp:405): specified as the affine inverse
    efo::M4d a;
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) a.m[i * 4 + j] = (double)(*this)(i, j);
    const efo::M4d b = efo::m4_affine_inverse(a);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) r(i, j) = (T)b.m[i * 4 + j];
  }
  return r;
}
template <typename T, int N>
template <int O>
Matrix<T, N, 1> LDLT<T, N>::solve(const Matrix<T, N, 1, O>& b) const {
  Matrix<T, N, 1> x;
  efo::ldlt_solve<T, N>(a, b.m, x.m);
  return x;
}
template <typename T, int N, int O>
Matrix<T, N, N, O> PartialPivLU<T, N, O>::inverse() const {
  T inv[N * N];
  efo::lu_inverse<T, N>(a, inv);
  Matrix<T, N, N, O> r;
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) r(i, j) = inv[i * N + j];
  return r;
}

typedef Matrix<int, 2, 1> Vector2i;
typedef Matrix<int, 4, 1> Vector4i;
typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<float, 4, 4> Matrix4f;
typedef Matrix<double, 4, 4> Matrix4d;

// Quaternion<double>(rotation matrix): the branchy trace method, as restated in efo_linalg.h (mat_to_quat)
class Quaterniond {
 public:
  template <int O>
  explicit Quaterniond(const Matrix<double, 3, 3, O>& R) {
    efo::M3d m;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) m.m[i * 3 + j] = R(i, j);
    efo::mat_to_quat(m, q);
  }
  double x() const { return q[0]; }
  double y() const { return q[1]; }
  double z() const { return q[2]; }
  double w() const { return q[3]; }

 private:
  double q[4];
};
// display code only (MainController.cpp:265-270): the camera's rotation applied to the view axes
class Quaternionf {
 public:
  template <int O>
  explicit Quaternionf(const Matrix<float, 3, 3, O>& R) : Rm(R) {}
  Matrix<float, 3, 1> operator*(const Matrix<float, 3, 1>& v) const { return Rm * v; }

 private:
  Matrix<float, 3, 3> Rm;
};
// std::map<..., Eigen::aligned_allocator<...>> (Tools/GroundTruthOdometry.h:50)
template <typename T>
using aligned_allocator = std::allocator<T>;
// run-time sized vector (only declared by the reference's deformation-graph headers here)
class VectorXd {
 public:
  std::vector<double> v;
  VectorXd() {}
  explicit VectorXd(int n) : v((size_t)n, 0.0) {}
  double& operator()(int i) { return v[(size_t)i]; }
  const double& operator()(int i) const { return v[(size_t)i]; }
  int rows() const { return (int)v.size(); }
  double* data() { return v.data(); }
  const double* data() const { return v.data(); }
  double squaredNorm() const { double s = 0; for (double x : v) s += x * x; return s; }
  double norm() const { return std::sqrt(squaredNorm()); }
  VectorXd operator-() const { VectorXd r((int)v.size()); for (size_t i = 0; i < v.size(); ++i) r.v[i] = -v[i]; return r; }
  void conservativeResize(int n) { v.resize((size_t)n, 0.0); }
  struct Segment {
    VectorXd& o;
    int start, len;
    template <int R, int C, int O>
    Segment& operator=(const Matrix<double, R, C, O>& m) { for (int i = 0; i < len; ++i) o.v[(size_t)(start + i)] = m.m[i]; return *this; }
  };
  Segment segment(int start, int len) { return Segment{*this, start, len}; }
};

// what getCovariance() returns: a run-time sized copy of a fixed-size result
class MatrixXd {
 public:
  int r = 0, c = 0;
  double v[36];
  MatrixXd() {}
  template <int R, int C, int O>
  MatrixXd(const Matrix<double, R, C, O>& o) : r(R), c(C) {
    static_assert(R * C <= 36, "at most 6x6");
    for (int i = 0; i < R; ++i)
      for (int j = 0; j < C; ++j) v[i * C + j] = o(i, j);
  }
  double operator()(int i, int j) const { return v[i * c + j]; }
  int rows() const { return r; }
  int cols() const { return c; }
};

template <typename MatrixType>
class JacobiSVD {
 public:
  template <int O>
  JacobiSVD(const Matrix<double, 3, 3, O>& a, unsigned) {
    efo::M3d A, Um, Vm;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) A.m[i * 3 + j] = a(i, j);
    efo::svd3_uv(A, Um, Vm);
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) { U(i, j) = Um.m[i * 3 + j]; V(i, j) = Vm.m[i * 3 + j]; }
  }
  const Matrix3d& matrixU() const { return U; }
  const Matrix3d& matrixV() const { return V; }

 private:
  Matrix3d U, V;
};

// Transform<float, 3, Isometry>: linear part + translation; inverse = (L^T, -L^T t); product = (L1 L2, L1 t2 + t1); rotation() = L
template <typename T, int Dim, int Mode>
class Transform {
 public:
  Matrix<T, 3, 3> L;
  Matrix<T, 3, 1> t;
  void setIdentity() { L = Matrix<T, 3, 3>::Identity(); t = Matrix<T, 3, 1>::Zero(); }
  template <int O>
  void rotate(const Matrix<T, 3, 3, O>& R) { L = L * R; }
  Matrix<T, 3, 1>& translation() { return t; }
  const Matrix<T, 3, 1>& translation() const { return t; }
  Matrix<T, 3, 3> rotation() const { return L; }
  Transform inverse() const {
    Transform r;
    r.L = L.transpose();
    const Matrix<T, 3, 1> p = r.L * t;
    for (int i = 0; i < 3; ++i) r.t(i) = -p(i);
    return r;
  }
  Transform operator*(const Transform& o) const {
    Transform r;
    r.L = L * o.L;
    r.t = (L * o.t) + t;
    return r;
  }
};
typedef Transform<float, 3, Isometry> Isometry3f;

}  // namespace Eigen
