// Wave-parallel Gauss-Newton update: everything RGBDOdometry.cpp:440-551 does on the host between two reductions
// (A = A_rgb + w^2 A_icp, Eigen LDL^T 6x6 solve in double, computeUpdateSE3, composition with Tprev in float, the
// next iteration's K R K^-1 / K t), evaluated by ONE wavefront with one matrix element per lane.
//
// Why: run on a single lane this is ~2700 dependent, mostly fp64, instructions — 10+ us on a machine whose wave64
// issues an fp64 op every 8 cycles no matter how many lanes are live — and it sits between every pair of reduction
// kernels, 19 times a frame.  Spread over lanes (36 lanes hold the 6x6 matrix, 16 a 4x4, 9 a 3x3 ...) the same
// arithmetic is ~700 instructions.  Every element still sees exactly the scalar sequence of IEEE operations of
// ef_linalg_dev.hpp (the Eigen/Sophus restatement), so results are bit-identical to the single-lane version; lanes
// exchange values through ds_bpermute shuffles and a small LDS scratch, ordered by wave_sync().
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include "ef_linalg_dev.hpp"
#include "ef_track.hpp"

namespace efs {

// developer instrumentation: -DEF_STAGE_CLOCKS stamps wall_clock64() (100 MHz) into TrackState::dbg_clock
#ifdef EF_STAGE_CLOCKS
#define EF_STAMP(st, i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) (st)->dbg_clock[i] = wall_clock64(); } while (0)
#else
#define EF_STAMP(st, i) do { } while (0)
#endif

// all lanes of the calling wave are converged here; LDS operations of one wave execute in issue order, so a
// compiler-level fence is all that separates a lane's write from another lane's read
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }
// broadcast from a lane known at compile time (the diagonal of the 6x6 system): v_readlane (a few cycles) instead of the LDS crossbar
// (ds_bpermute, a round trip of >100 cycles, 27 of them on the factorisation's critical path).  Validated on the GPU in round 3 (full
// -m gpu suite) and adopted: profiles/r03a_ab.log.
__device__ __forceinline__ double bcast_d(double v, int src) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

struct SolveScratch {     // LDS, one per workgroup that runs a solve
  double A[36];           // factorised matrix: D on the diagonal, L below
  double b[6];
  double x[6];            // solution
  double upd[16];         // computeUpdateSE3 increment
  double Rt[16];          // resultRt (new)
  double inv[2][9];       // inverse of resultRt's 3x3 block, inverse of K
  double ti[3];
  double KR[9];
  float pose[12];         // scratch for the float composition
  // state of the previous iteration, fetched by the first wave while the partial sums are still in flight: resultRt, Rprev | tprev
  double prevRt[16];
  float prevPose[12];
  // what the step leaves for the workgroup that evaluated it (every workgroup evaluates it; workgroup 0 also publishes it)
  float krkinv[9], kt[3];   // K R K^-1, K t of the coming iteration
  float Rcurr[9], tcurr[3]; // the new pose
  int broken;               // rgbOnly "break" flag after this step
  int tail_pending;         // SPLIT_TAIL: 1 = resultRt is in Rt and the two tails (gn_tail_pose, gn_tail_krk) are still to be evaluated
};
// the per-lane part of that prefetch (registers until the update starts)
struct SolvePrefetch {
  double rt;      // resultRt[lane & 15]
  float pose;     // lane < 9: Rprev[lane]; 9..11: tprev[lane - 9]
  int slot_a, slot_b;
  float lastRGBErrorLevel;
  int broken;
};
// prev: the state the last update left; slots: the {count, sum diff^2} slots of the residual pass that update belongs to
__device__ __forceinline__ SolvePrefetch solve_prefetch(const eft::TrackState* st, const eft::GNState* prev, const int* slots) {
  const int lane = threadIdx.x & 63;
  SolvePrefetch P;
  P.rt = prev->resultRt[lane & 15];
  P.pose = lane < 9 ? st->Rprev[lane] : st->tprev[lane < 12 ? lane - 9 : 0];
  P.slot_a = slots[lane * 16];
  P.slot_b = slots[lane * 16 + 1];
  P.lastRGBErrorLevel = prev->lastRGBErrorLevel;
  P.broken = prev->rgb_broken;
  return P;
}
__device__ __forceinline__ void solve_prefetch_publish(const SolvePrefetch& P, SolveScratch& S) {
  const int lane = threadIdx.x & 63;
  if (lane < 16) S.prevRt[lane] = P.rt;
  if (lane < 12) S.prevPose[lane] = P.pose;
}

// index of member (i,j), i <= j <= 6, in the JtJJtrSE3 order (types.cuh:98-143): rows 0..5 start at 0,7,13,18,22,25
__device__ __forceinline__ int se3_member_of(int i, int j) {
  const int lo = i < j ? i : j, hi = i < j ? j : i;
  const int start = lo * 7 - (lo * (lo - 1)) / 2;   // 0 7 13 18 22 25
  return start + (hi - lo);
}

// Eigen::LDLT-style solve of the 6x6 system held one element per lane (lane e < 36 holds A[e/6][e%6], bitwise
// symmetric); b in S.b.  Mirrors efl::ldlt_solve<double,6> operation for operation.  Result in S.x (all 6 entries).
// (Rounds 2-4's version, kept for the A/B and the host emulation — tests/test_wave_emulation.py runs both against the scalar statement;
// the update step runs ldlt6_every_lane below.)
__device__ __forceinline__ void ldlt6_wave(double a, SolveScratch& S) {
  const int lane = threadIdx.x & 63;
  const int e = lane < 36 ? lane : 0;
  const int i = e / 6, j = e - i * 6;
  const int hi = i > j ? i : j, lo = i > j ? j : i;
  int perm[6] = {0, 1, 2, 3, 4, 5};
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    // pivot: largest |diagonal| of the trailing block, first one wins
    int p = k;
    double best = fabs(bcast_d(a, k * 7));
#pragma unroll
    for (int m = k + 1; m < 6; ++m) {
      const double v = fabs(bcast_d(a, m * 7));
      const bool gt = v > best;
      best = gt ? v : best;
      p = gt ? m : p;
    }
    // symmetric row/column swap k <-> p
    const int pi = (i == k) ? p : ((i == p) ? k : i);
    const int pj = (j == k) ? p : ((j == p) ? k : j);
    a = shfl_d(a, pi * 6 + pj);
#pragma unroll
    for (int m = k + 1; m < 6; ++m) {   // swap(perm[k], perm[p]) with p > k
      const bool sw = (p == m);
      const int pk = perm[k], pm = perm[m];
      perm[k] = sw ? pm : pk;
      perm[m] = sw ? pk : pm;
    }
    const double d = bcast_d(a, k * 7);
    const double c_hi = shfl_d(a, hi * 6 + k), c_lo = shfl_d(a, lo * 6 + k);
    if (!(d == 0.0)) {
      const double l = c_hi / d;
      if (i > k && j > k) a = a - l * c_lo;          // trailing update (both triangles, kept bitwise symmetric)
      else if (j == k && i > k) a = l;               // L below the diagonal (c_hi == a here)
      else if (i == k && j > k) a = 0.0;
    }
  }
  if (lane < 36) S.A[lane] = a;
  wave_sync();
  // substitution, evaluated redundantly by every lane on register copies (static indices)
  double F[36];
#pragma unroll
  for (int q = 0; q < 36; ++q) F[q] = S.A[q];
  double y[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) y[q] = S.b[perm[q]];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) y[r] -= F[r * 6 + c] * y[c];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double d = F[r * 7];
    y[r] = (fabs(d) > DBL_MIN) ? y[r] / d : 0.0;
  }
#pragma unroll
  for (int r = 5; r >= 0; --r)
#pragma unroll
    for (int c = r + 1; c < 6; ++c) y[r] -= F[c * 6 + r] * y[c];
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) S.x[perm[q]] = y[q];
  }
  wave_sync();
}

// entry e (0..8) of the inverse of the row-major 3x3 m: efl::m3_inverse<double> evaluated in full (45 flops) and one
// entry selected, so that no register array is indexed dynamically
__device__ __forceinline__ double m3_inverse_entry(const double* m, int e) {
  double o[9];
  efl::m3_inverse<double>(m, o);
  double r = o[0];
#pragma unroll
  for (int q = 1; q < 9; ++q) r = (e == q) ? o[q] : r;
  return r;
}

// The same factorisation with the WHOLE matrix in every lane's registers (round 5; what the update step runs) — no shuffles, no read-lanes
// on the six pivots' critical path; the pivot index is the same in every lane, so the symmetric swap is a uniform branch over static register
// indices instead of 13 conditional swaps per candidate.  The matrix is bitwise symmetric and stays so (ldlt6_wave keeps both triangles equal), so only the lower
// triangle is held: 21 doubles.  The same IEEE operations on every element, in efl::ldlt_solve's order.
__device__ __forceinline__ constexpr int ldlt6_at(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }
template <int K, int M_>
__device__ __forceinline__ void ldlt6_swap(double (&A)[21], unsigned& perm /* entry q in bits 3 q .. 3 q + 2 */) {
  // new(i, j) = old(pi(i), pi(j)), pi = the transposition (K M_), on the lower triangle
  double B[21];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int pi = (i == K) ? M_ : ((i == M_) ? K : i), pj = (j == K) ? M_ : ((j == M_) ? K : j);
      B[ldlt6_at(i, j)] = A[ldlt6_at(pi, pj)];
    }
#pragma unroll
  for (int q = 0; q < 21; ++q) A[q] = B[q];
  const unsigned pk = (perm >> (3 * K)) & 7u, pm = (perm >> (3 * M_)) & 7u;
  perm = (perm & ~((7u << (3 * K)) | (7u << (3 * M_)))) | (pm << (3 * K)) | (pk << (3 * M_));
}
template <int K>
__device__ __forceinline__ void ldlt6_step(double (&A)[21], unsigned& perm) {
  int p = K;
  double best = fabs(A[ldlt6_at(K, K)]);
#pragma unroll
  for (int m = K + 1; m < 6; ++m) {
    const double v = fabs(A[ldlt6_at(m, m)]);
    const bool gt = v > best;
    best = gt ? v : best;
    p = gt ? m : p;
  }
  p = __builtin_amdgcn_readfirstlane(p);   // (every lane holds the same matrix)
  if (K + 1 < 6 && p == K + 1) ldlt6_swap<K, (K + 1 < 6 ? K + 1 : K)>(A, perm);
  else if (K + 2 < 6 && p == K + 2) ldlt6_swap<K, (K + 2 < 6 ? K + 2 : K)>(A, perm);
  else if (K + 3 < 6 && p == K + 3) ldlt6_swap<K, (K + 3 < 6 ? K + 3 : K)>(A, perm);
  else if (K + 4 < 6 && p == K + 4) ldlt6_swap<K, (K + 4 < 6 ? K + 4 : K)>(A, perm);
  else if (K + 5 < 6 && p == K + 5) ldlt6_swap<K, (K + 5 < 6 ? K + 5 : K)>(A, perm);
  const double d = A[ldlt6_at(K, K)];
  if (!(d == 0.0)) {
    double colk[6];
#pragma unroll
    for (int i = K + 1; i < 6; ++i) colk[i] = A[ldlt6_at(i, K)];
#pragma unroll
    for (int i = K + 1; i < 6; ++i) {
      const double l = colk[i] / d;
#pragma unroll
      for (int j = K + 1; j <= i; ++j) A[ldlt6_at(i, j)] = A[ldlt6_at(i, j)] - l * colk[j];
      A[ldlt6_at(i, K)] = l;
    }
  }
}
__device__ __forceinline__ void ldlt6_every_lane(double a, SolveScratch& S) {
  const int lane = threadIdx.x & 63;
  if (lane < 36) S.A[lane] = a;
  wave_sync();
  double A[21];
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = 0; j <= i; ++j) A[ldlt6_at(i, j)] = S.A[i * 6 + j];
  unsigned perm = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12) | (5u << 15);
  ldlt6_step<0>(A, perm);
  ldlt6_step<1>(A, perm);
  ldlt6_step<2>(A, perm);
  ldlt6_step<3>(A, perm);
  ldlt6_step<4>(A, perm);
  ldlt6_step<5>(A, perm);
  double y[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) y[q] = S.b[(perm >> (3 * q)) & 7u];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c < r; ++c) y[r] -= A[ldlt6_at(r, c)] * y[c];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    const double d = A[ldlt6_at(r, r)];
    y[r] = (fabs(d) > DBL_MIN) ? y[r] / d : 0.0;
  }
#pragma unroll
  for (int r = 5; r >= 0; --r)
#pragma unroll
    for (int c = r + 1; c < 6; ++c) y[r] -= A[ldlt6_at(c, r)] * y[c];
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 6; ++q) S.x[(perm >> (3 * q)) & 7u] = y[q];
  }
  wave_sync();
}

struct SolveInputs {
  bool icp, rgb, rgbOnly;
  float icpWeight;
  eft::Intr knext;
  bool level_changes;
};

// The two tails of the update step behind resultRt (S.Rt): the float pose of the coming iteration, and its K R K^-1 / K t.  Each is called by ONE
// converged wavefront; they read S.Rt / S.prevPose and write disjoint members, so two wavefronts may run them at the same time.
__device__ __forceinline__ void gn_tail_pose(SolveScratch& S, eft::GNState* next, bool publish) {
  const int lane = threadIdx.x & 63;
  // ---- currentT = [Rprev|tprev] * rgbOdom^-1 in float, Isometry inverse = (R^T, -R^T t) (quirk Q13) ----
  if (lane < 12) {
    // iR = oR^T with oR = float(resultRt 3x3), ot = float(resultRt translation)
    float iR[9], ot[3], it[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) iR[r * 3 + c] = (float)S.Rt[c * 4 + r];
      ot[r] = (float)S.Rt[r * 4 + 3];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) it[r] = -(iR[r * 3] * ot[0] + iR[r * 3 + 1] * ot[1] + iR[r * 3 + 2] * ot[2]);
    const float* Rp = S.prevPose;
    if (lane < 9) {
      const int r = lane / 3, c = lane - r * 3;
      // column c of iR = row c of float(resultRt), read from LDS with the lane's own index (selecting among the register copies
      // makes the compiler index a private array, which lands in scratch when the alloca is not promoted to LDS)
      const float i0 = (float)S.Rt[c * 4], i1 = (float)S.Rt[c * 4 + 1], i2 = (float)S.Rt[c * 4 + 2];
      const float v = Rp[r * 3] * i0 + Rp[r * 3 + 1] * i1 + Rp[r * 3 + 2] * i2;
      S.Rcurr[lane] = v;
      if (publish) next->Rcurr[lane] = v;
    } else {
      const int r = lane - 9;
      const float v = (Rp[r * 3] * it[0] + Rp[r * 3 + 1] * it[1] + Rp[r * 3 + 2] * it[2]) + S.prevPose[9 + r];
      S.tcurr[r] = v;
      if (publish) next->tcurr[r] = v;
    }
  }
}
__device__ __forceinline__ void gn_tail_krk(const eft::Intr knext, SolveScratch& S, eft::GNState* next, bool publish) {
  const int lane = threadIdx.x & 63;
  // ---- next iteration's K R K^-1 and K t (RGBDOdometry.cpp:395-417) ----
  {
    const eft::Intr k = knext;
    const double K[9] = {k.fx, 0, k.cx, 0, k.fy, k.cy, 0, 0, 1};
    // lanes 0..8: inverse of resultRt's 3x3 block; lanes 16..24: inverse of K (same code, different operand)
    const bool second = lane >= 16;
    double m[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) m[q] = second ? K[q] : S.Rt[(q / 3) * 4 + (q % 3)];
    const int e = second ? lane - 16 : lane;
    const double inv = m3_inverse_entry(m, (e >= 0 && e < 9) ? e : 0);
    if (lane < 9) S.inv[0][lane] = inv;
    else if (lane >= 16 && lane < 25) S.inv[1][lane - 16] = inv;
    wave_sync();
    // translation of the inverse: -(Ai * t)  (m4_affine_inverse)
    if (lane < 3) {
      const double* Ai = S.inv[0];
      S.ti[lane] = Ai[lane * 3] * S.Rt[3] + Ai[lane * 3 + 1] * S.Rt[7] + Ai[lane * 3 + 2] * S.Rt[11];
    }
    // K * R
    if (lane >= 16 && lane < 25) {
      const int q = lane - 16, r = q / 3, c = q - r * 3;
      // row r of K, selected without indexing a register array by a lane-dependent r
      const double k0 = (r == 0) ? K[0] : 0.0, k1 = (r == 1) ? K[4] : 0.0, k2 = (r == 0) ? K[2] : (r == 1 ? K[5] : 1.0);
      double s = 0;
      s += k0 * S.inv[0][c];
      s += k1 * S.inv[0][3 + c];
      s += k2 * S.inv[0][6 + c];
      S.KR[q] = s;
    }
    wave_sync();
    if (lane < 9) {
      const int r = lane / 3, c = lane - r * 3;
      double s = 0;
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) s += S.KR[r * 3 + kk] * S.inv[1][kk * 3 + c];
      S.krkinv[lane] = (float)s;
      if (publish) next->krkinv[lane] = (float)s;
    } else if (lane >= 16 && lane < 19) {
      const int r = lane - 16;
      const double t0 = -S.ti[0], t1 = -S.ti[1], t2 = -S.ti[2];
      const double k0 = (r == 0) ? K[0] : 0.0, k1 = (r == 1) ? K[4] : 0.0, k2 = (r == 0) ? K[2] : (r == 1 ? K[5] : 1.0);
      const float v = (float)(k0 * t0 + k1 * t1 + k2 * t2);
      S.kt[r] = v;
      if (publish) next->kt[r] = v;
    }
  }
  wave_sync();
}
// The update step proper.  sums: 58 floats in LDS (ICP members 0..28, RGB members 29..57).  Called by ONE converged
// wavefront (lanes 0..63).  Leaves resultRt, Rcurr/tcurr, krkinv/kt in S; with `publish` also writes them into `next`
// and lastA/lastb into st.
// `stats`: also leave lastA / lastb in st (one workgroup does; in the persistent small-level kernel every workgroup publishes into its own
// LDS copy of the state but only workgroup 0 writes the TrackState)
template <bool SPLIT_TAIL = false>
__device__ __forceinline__ void gauss_newton_update_wave(eft::TrackState* st, eft::GNState* next, bool publish, const float* sums,
                                                         const SolveInputs in, SolveScratch& S, bool stats) {
  const int lane = threadIdx.x & 63;
  // ---- A = A_rgb + w^2 A_icp, b = b_rgb + w b_icp (RGBDOdometry.cpp:522-534), one element per lane ----
  double a = 0.0;
  {
    const int e = lane < 36 ? lane : 0;
    const int i = e / 6, j = e - i * 6;
    const int mA = se3_member_of(i, j);
    const int bi = lane >= 36 && lane < 42 ? lane - 36 : 0;
    const int mb = se3_member_of(bi, 6);
    const int m = lane < 36 ? mA : mb;
    double v;
    if (in.icp && in.rgb) {
      const double w = in.icpWeight;
      if (lane < 36) v = (double)sums[eft::SE3_ACCS + m] + w * w * (double)sums[m];
      else v = (double)sums[eft::SE3_ACCS + m] + w * (double)sums[m];
    } else if (in.icp) {
      v = (double)sums[m];
    } else {
      v = (double)sums[eft::SE3_ACCS + m];
    }
    if (lane < 36) { a = v; if (stats) st->lastA[lane] = v; }
    else if (lane < 42) { S.b[lane - 36] = v; if (stats) st->lastb[lane - 36] = v; }
  }
  wave_sync();
  EF_STAMP(st, 4);
#ifdef EF_LDLT_WAVE
  ldlt6_wave(a, S);         // rounds 2-4 (A/B: python -m elasticfusion_amd.build --variant ldlt_wave)
#else
  ldlt6_every_lane(a, S);   // round 5: 2.14 -> 1.88 us per update step (profiles/r05l_clocks_ldlt.jsonl), 16 registers fewer in k_track_ref
#endif
  EF_STAMP(st, 5);
  // ---- computeUpdateSE3 (OdometryProvider.h:73-96): rodrigues(result[3..5]) and the 4x4 increment ----
  {
    double rx = S.x[3], ry = S.x[4], rz = S.x[5];
    const double theta = sqrt(rx * rx + ry * ry + rz * rz);
    double val;
    const int e = lane < 16 ? lane : 0;
    const int r = e >> 2, c = e & 3;
    if (r < 3 && c < 3) {
      const int k = r * 3 + c;
      val = (r == c) ? 1.0 : 0.0;
      if (theta >= DBL_EPSILON) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(EF_SEPARATE_SIN_COS)
        // one sincos instead of cos and sin apart (155 against 294 VALU instructions of one dependent double-precision chain on the critical path of
        // every iteration): ocml's sincos returns, bit for bit, what its sin and its cos return (tools/probe/sincos_probe.hip on MI355X: 2^24 arguments from
        // subnormal to 1e300, zero differences — profiles/r06x_sincos_probe.json)
        double sn, cs;
        sincos(theta, &sn, &cs);
        const double c1 = 1. - cs;
#else
        const double cs = cos(theta), sn = sin(theta), c1 = 1. - cs;
#endif
        const double itheta = theta ? 1. / theta : 0.;
        rx *= itheta; ry *= itheta; rz *= itheta;
        const double u = (r == 0) ? rx : (r == 1 ? ry : rz), v = (c == 0) ? rx : (c == 1 ? ry : rz);
        // rrt is symmetric and built from the products rx*rx, rx*ry, rx*rz, ry*ry, ry*rz, rz*rz (first factor = lower index)
        const double rrt = (r <= c) ? u * v : v * u;
        // [r]_x = {0,-rz,ry, rz,0,-rx, -ry,rx,0}
        double cross = 0.0;
        if (k == 1) cross = -rz; else if (k == 2) cross = ry; else if (k == 3) cross = rz;
        else if (k == 5) cross = -rx; else if (k == 6) cross = -ry; else if (k == 7) cross = rx;
        val = cs * ((r == c) ? 1.0 : 0.0) + c1 * rrt + sn * cross;
      }
    } else if (r < 3) {
      val = S.x[r];
    } else {
      val = (c == 3) ? 1.0 : 0.0;
    }
    if (lane < 16) S.upd[lane] = val;
  }
  wave_sync();
  EF_STAMP(st, 6);
  // ---- resultRt = upd * resultRt (RGBDOdometry.cpp:537) ----
  if (lane < 16) {
    const int r = lane >> 2, c = lane & 3;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) s += S.upd[r * 4 + k] * S.prevRt[k * 4 + c];
    S.Rt[lane] = s;
  }
  wave_sync();
  if (publish && lane < 16) next->resultRt[lane] = S.Rt[lane];
  if (SPLIT_TAIL) {   // the caller evaluates the two tails on two wavefronts side by side (round 6: they only share their input, Rt)
    if (lane == 0) S.tail_pending = 1;
    wave_sync();
    return;
  }
  gn_tail_pose(S, next, publish);
  EF_STAMP(st, 7);
  gn_tail_krk(in.knext, S, next, publish);
  wave_sync();
  EF_STAMP(st, 8);
}

}  // namespace efs
