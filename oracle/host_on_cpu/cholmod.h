/* TEST INFRASTRUCTURE — the handful of CHOLMOD entry points Core/Utils/CholeskyDecomp.cpp uses, implemented with a DENSE Cholesky so that
 * the reference's own sparse-system builder and Gauss-Newton driver (DeformationGraph.cpp, CholeskyDecomp.cpp) can be compiled where they
 * lie and run (SuiteSparse is absent from this image).  Written from scratch.  Semantics kept: cholmod_factorize on an UNSYMMETRIC matrix
 * At factorises At * At' (= J'J here); the fill-reducing permutation is the identity; solves are by phase (P, L, Lt). */
#ifndef EFR_CHOLMOD_STUB_H_
#define EFR_CHOLMOD_STUB_H_
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHOLMOD_REAL 1
#define CHOLMOD_P 7
#define CHOLMOD_L 4
#define CHOLMOD_Lt 5
struct cholmod_common { int status; };
struct cholmod_sparse { size_t nrow, ncol, nzmax; void *p, *i, *x; };
struct cholmod_dense { size_t nrow, ncol; void* x; };
struct cholmod_factor { size_t n; void* Perm; std::vector<double>* Ldense; };   /* row-major lower-triangular factor of At*At' */
static inline int cholmod_start(cholmod_common* c) { c->status = 0; return 1; }
static inline int cholmod_finish(cholmod_common*) { return 1; }
static inline cholmod_sparse* cholmod_allocate_sparse(size_t nrow, size_t ncol, size_t nzmax, int, int, int, int, cholmod_common*) {
  cholmod_sparse* s = new cholmod_sparse{nrow, ncol, nzmax, calloc(ncol + 1, sizeof(int)), calloc(nzmax ? nzmax : 1, sizeof(int)), calloc(nzmax ? nzmax : 1, sizeof(double))};
  return s;
}
static inline int cholmod_free_sparse(cholmod_sparse** s, cholmod_common*) { free((*s)->p); free((*s)->i); free((*s)->x); delete *s; *s = 0; return 1; }
static inline cholmod_dense* cholmod_zeros(size_t nrow, size_t ncol, int, cholmod_common*) { return new cholmod_dense{nrow, ncol, calloc(nrow * ncol ? nrow * ncol : 1, sizeof(double))}; }
static inline int cholmod_free_dense(cholmod_dense** d, cholmod_common*) { free((*d)->x); delete *d; *d = 0; return 1; }
static inline cholmod_factor* cholmod_analyze(cholmod_sparse* A, cholmod_common*) {
  cholmod_factor* f = new cholmod_factor{A->nrow, malloc(A->nrow * sizeof(int)), nullptr};
  for (size_t k = 0; k < A->nrow; ++k) ((int*)f->Perm)[k] = (int)k;
  return f;
}
static inline cholmod_factor* cholmod_copy_factor(cholmod_factor* L, cholmod_common*) {
  cholmod_factor* f = new cholmod_factor{L->n, malloc(L->n * sizeof(int)), nullptr};
  memcpy(f->Perm, L->Perm, L->n * sizeof(int));
  return f;
}
static inline int cholmod_free_factor(cholmod_factor** f, cholmod_common*) { free((*f)->Perm); delete (*f)->Ldense; delete *f; *f = 0; return 1; }
static inline int cholmod_factorize(cholmod_sparse* At, cholmod_factor* L, cholmod_common*) {
  const size_t n = At->nrow;
  std::vector<double>* M = new std::vector<double>(n * n, 0.0);
  const int* p = (const int*)At->p; const int* idx = (const int*)At->i; const double* x = (const double*)At->x;
  for (size_t c = 0; c < At->ncol; ++c)        /* At * At' = sum over columns (= rows of J) of outer products */
    for (int a = p[c]; a < p[c + 1]; ++a)
      for (int b = p[c]; b < p[c + 1]; ++b) (*M)[(size_t)idx[a] * n + idx[b]] += x[a] * x[b];
  for (size_t j = 0; j < n; ++j) {              /* Cholesky, lower, in place */
    double d = (*M)[j * n + j];
    for (size_t k = 0; k < j; ++k) d -= (*M)[j * n + k] * (*M)[j * n + k];
    d = std::sqrt(d);
    (*M)[j * n + j] = d;
    for (size_t i = j + 1; i < n; ++i) {
      double s = (*M)[i * n + j];
      for (size_t k = 0; k < j; ++k) s -= (*M)[i * n + k] * (*M)[j * n + k];
      (*M)[i * n + j] = s / d;
    }
  }
  delete L->Ldense;
  L->Ldense = M;
  return 1;
}
static inline int cholmod_change_factor(int, int, int, int, int, cholmod_factor*, cholmod_common*) { return 1; }
static inline int cholmod_sdmult(cholmod_sparse* A, int transpose, double* alpha, double* beta, cholmod_dense* X, cholmod_dense* Y, cholmod_common*) {
  double* y = (double*)Y->x; const double* xv = (const double*)X->x;
  for (size_t r = 0; r < Y->nrow; ++r) y[r] *= beta[0];
  const int* p = (const int*)A->p; const int* idx = (const int*)A->i; const double* x = (const double*)A->x;
  for (size_t c = 0; c < A->ncol; ++c)
    for (int a = p[c]; a < p[c + 1]; ++a) y[idx[a]] += alpha[0] * x[a] * xv[c];
  (void)transpose;
  return 1;
}
static inline cholmod_dense* cholmod_solve(int sys, cholmod_factor* L, cholmod_dense* B, cholmod_common* c) {
  const size_t n = L->n;
  cholmod_dense* X = cholmod_zeros(n, 1, CHOLMOD_REAL, c);
  double* x = (double*)X->x; const double* b = (const double*)B->x;
  const std::vector<double>& M = *L->Ldense;
  if (sys == CHOLMOD_P) {
    for (size_t k = 0; k < n; ++k) x[k] = b[((int*)L->Perm)[k]];
  } else if (sys == CHOLMOD_L) {
    for (size_t i = 0; i < n; ++i) { double s = b[i]; for (size_t k = 0; k < i; ++k) s -= M[i * n + k] * x[k]; x[i] = s / M[i * n + i]; }
  } else {
    for (size_t ii = n; ii-- > 0;) { double s = b[ii]; for (size_t k = ii + 1; k < n; ++k) s -= M[k * n + ii] * x[k]; x[ii] = s / M[ii * n + ii]; }
  }
  return X;
}
#endif
