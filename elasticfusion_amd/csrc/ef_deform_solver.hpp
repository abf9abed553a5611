// Built-in optimiser of the LOCAL deformation graph: what Deformation::constrain(..., fernMatch = false) computes in the reference
// (Core/Deformation.cpp:88-215 over Core/Utils/DeformationGraph.cpp and a CHOLMOD sparse Cholesky), written from scratch as a
// host-side banded Gauss-Newton solver.  Embedded deformation (Sumner et al.): every graph node carries an affine 3x3 + translation;
//   E = E_rot (columns orthonormal, 6 rows per node) + 10 E_reg (a node predicts its sequence neighbours, 3 rows per pair)
//       + 100 E_con (constraint sources land on their targets, 3 rows per constraint)
// over the nodes younger than the last deformation.  Nodes are the model samples in time order, connected to their +-2 sequence
// neighbours (k = 4); a surface point is carried by the 4 nearest of the <= 20 nodes around its time, weights (1 - d / d_5th)^2.
// The normal equations are banded (node span of any row <= 20): banded Cholesky, O(n * band^2).  Host code only (no GPU work:
// <= 1024 nodes); checked against the reference's own optimiser compiled where it lies (tests/test_deform_solver_vs_reference.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

namespace efd {

constexpr int K = 4;            // Deformation.cpp:23: def(4, ...)
constexpr int LOOK_BACK = 20;   // DeformationGraph.cpp:129,256
constexpr double W_REG = 10.0, W_CON = 100.0;   // DeformationGraph.cpp:26-28 (wRot = 1)

struct Node {
  double g[3];       // position
  double R[9];       // column-major: R(i, j) = R[j * 3 + i]
  double t[3];
  uint64_t time;
  bool enabled;
  int nb[K];
};
struct Carrier { double w; int node; };
struct Vertex { double p[3]; Carrier c[K]; };
struct Result { bool ok; int iterations; float error, meanConsErr; };

// index of the node whose time is closest to `t` among the candidates a bisection ends on (DeformationGraph.cpp:259-291)
inline int nearest_in_time(const std::vector<Node>& nodes, uint64_t t) {
  int lo = 0, hi = (int)nodes.size() - 1, mid = (lo + hi) / 2;
  while (hi >= lo) {
    mid = (lo + hi) / 2;
    if (nodes[mid].time < t) lo = mid + 1;
    else if (nodes[mid].time > t) hi = mid - 1;
    else break;
  }
  lo = std::min(lo, (int)nodes.size() - 1);
  auto gap = [&](int i) { return std::llabs((long long)nodes[i].time - (long long)t); };
  const int hi_c = hi < 0 ? 0 : hi;   // the bisection can leave hi = -1: that candidate never wins unless it ties with the others
  if (hi >= 0) {
    if (gap(lo) <= gap(mid) && gap(lo) <= gap(hi)) return lo;
    if (gap(mid) <= gap(lo) && gap(mid) <= gap(hi)) return mid;
    return hi;
  }
  return gap(lo) <= gap(mid) ? lo : (gap(mid) <= gap(lo) ? mid : hi_c);
}

inline void carriers_of(const std::vector<Node>& nodes, const double* p, uint64_t time, Carrier (&out)[K]) {
  const int n = (int)nodes.size();
  int found = nearest_in_time(nodes, time);
  std::vector<std::pair<float, int>> near;
  auto dist = [&](int j) { const double dx = nodes[j].g[0] - p[0], dy = nodes[j].g[1] - p[1], dz = nodes[j].g[2] - p[2]; return std::sqrt(dx * dx + dy * dy + dz * dz); };
  int taken = 0;
  for (int j = found; j >= 0 && taken < LOOK_BACK; --j, ++taken) near.emplace_back((float)dist(j), j);
  for (int j = found + 1; j < n && taken < LOOK_BACK; ++j, ++taken) near.emplace_back((float)dist(j), j);
  std::sort(near.begin(), near.end(), [](const std::pair<float, int>& a, const std::pair<float, int>& b) { return a.first < b.first; });
  const double dMax = near[K].first;
  double sum = 0;
  for (int j = 0; j < K; ++j) {
    const double w = 1.0 - dist(near[j].second) / dMax;
    out[j] = Carrier{w * w, near[j].second};
    sum += out[j].w;
  }
  for (int j = 0; j < K; ++j) out[j].w /= sum;
  std::sort(out, out + K, [](const Carrier& a, const Carrier& b) { return a.node < b.node; });
}

inline void deformed(const std::vector<Node>& nodes, const Vertex& v, double* out) {
  out[0] = out[1] = out[2] = 0;
  for (int i = 0; i < K; ++i) {
    const Node& nd = nodes[v.c[i].node];
    const double d[3] = {v.p[0] - nd.g[0], v.p[1] - nd.g[1], v.p[2] - nd.g[2]};
    for (int r = 0; r < 3; ++r)
      out[r] += v.c[i].w * (nd.R[r] * d[0] + nd.R[3 + r] * d[1] + nd.R[6 + r] * d[2] + nd.g[r] + nd.t[r]);
  }
}

// lower-banded symmetric positive definite system, row-major band storage: A(i, j) for i - band <= j <= i at a[i * (band + 1) + (j - i + band)]
struct Banded {
  int n, band;
  std::vector<double> a, rhs;
  Banded(int n_, int band_) : n(n_), band(band_), a((size_t)n_ * (band_ + 1), 0.0), rhs((size_t)n_, 0.0) {}
  double& at(int i, int j) { return a[(size_t)i * (band + 1) + (j - i + band)]; }
  // one Jacobian row with `cnt` non-zeros (columns ascending) and residual r: accumulate J'J and J'r
  void add_row(const int* cols, const double* vals, int cnt, double r) {
    for (int x = 0; x < cnt; ++x) {
      rhs[cols[x]] += vals[x] * r;
      for (int y = 0; y <= x; ++y) at(cols[x], cols[y]) += vals[x] * vals[y];
    }
  }
  bool solve_negative(std::vector<double>& delta) {   // delta = -(J'J)^-1 J'r
    for (int j = 0; j < n; ++j) {
      double d = at(j, j);
      for (int k = std::max(0, j - band); k < j; ++k) d -= at(j, k) * at(j, k);
      if (!(d > 0)) return false;
      d = std::sqrt(d);
      at(j, j) = d;
      for (int i = j + 1; i <= std::min(n - 1, j + band); ++i) {
        double s = at(i, j);
        for (int k = std::max(0, i - band); k < j; ++k) s -= at(i, k) * at(j, k);
        at(i, j) = s / d;
      }
    }
    delta.assign((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) {
      double s = -rhs[i];
      for (int k = std::max(0, i - band); k < i; ++k) s -= at(i, k) * delta[k];
      delta[i] = s / at(i, i);
    }
    for (int i = n - 1; i >= 0; --i) {
      double s = delta[i];
      for (int k = i + 1; k <= std::min(n - 1, i + band); ++k) s -= at(k, i) * delta[k];
      delta[i] = s / at(i, i);
    }
    return true;
  }
};

// nodes4: n x {x, y, z, time} (ef_sample_graph); constraints: m x {src xyz, target xyz, target time, pin} (ef_get_local_loop), all with
// source time `src_time`; graph16: n x {position 3, rotation 9 column-major, translation 3, time} (the layout ef_set_deformation takes).
inline Result solve_local(const float* nodes4, int n, const double* constraints, int m, uint64_t src_time, uint64_t last_deform_time, float* graph16) {
  Result res{false, 0, 0.f, 0.f};
  if (n <= K || m <= 0) return res;   // Deformation::sampleGraphModel only builds a graph for more than k samples (Deformation.cpp:283)
  std::vector<Node> nodes((size_t)n);
  for (int i = 0; i < n; ++i) {
    Node& nd = nodes[i];
    for (int r = 0; r < 3; ++r) { nd.g[r] = nodes4[i * 4 + r]; nd.t[r] = 0; }
    for (int r = 0; r < 9; ++r) nd.R[r] = (r % 4 == 0) ? 1.0 : 0.0;
    nd.time = (uint64_t)nodes4[i * 4 + 3];
    nd.enabled = nd.time > last_deform_time;
    int c = 0;   // sequence neighbours (DeformationGraph.cpp:226-251)
    if (i < K / 2) { for (int q = 0; q < K + 1; ++q) if (q != i) nd.nb[c++] = q; }
    else if (i >= n - K / 2) { for (int q = n - (K + 1); q < n; ++q) if (q != i) nd.nb[c++] = q; }
    else { for (int q = 0; q < K / 2; ++q) { nd.nb[c++] = i - (q + 1); nd.nb[c++] = i + (q + 1); } }
  }
  int first = n;
  for (int i = 0; i < n; ++i) if (nodes[i].enabled) { first = i; break; }
  const int unknowns = (n - first) * 12;
  // the vertices the constraints act on: every source, and for pinned constraints the target as a second vertex held in place
  std::vector<Vertex> verts;
  std::vector<std::pair<int, const double*>> cons;   // (vertex, target xyz)
  for (int i = 0; i < m; ++i) {
    const double* c = constraints + (size_t)i * 8;
    Vertex v{{c[0], c[1], c[2]}, {}};
    carriers_of(nodes, v.p, src_time, v.c);
    cons.emplace_back((int)verts.size(), c + 3);
    verts.push_back(v);
    if (c[7] != 0) {
      Vertex p{{c[3], c[4], c[5]}, {}};
      carriers_of(nodes, p.p, (uint64_t)c[6], p.c);
      cons.emplace_back((int)verts.size(), c + 3);
      verts.push_back(p);
    }
  }
  auto mean_error = [&]() {
    float e = 0;
    for (auto& cn : cons) {
      double q[3];
      deformed(nodes, verts[cn.first], q);
      e += (float)std::sqrt((q[0] - cn.second[0]) * (q[0] - cn.second[0]) + (q[1] - cn.second[1]) * (q[1] - cn.second[1]) + (q[2] - cn.second[2]) * (q[2] - cn.second[2]));
    }
    return e / (float)cons.size();
  };
  res.meanConsErr = mean_error();
  res.ok = true;
  if (unknowns == 0) { res.error = 0; goto emit; }
  {
    int span = 4;   // node span of the regularisation rows at the ends of the sequence
    for (auto& v : verts) span = std::max(span, v.c[K - 1].node - v.c[0].node);
    const int band = std::min(unknowns - 1, (span + 1) * 12 - 1);
    const double sr = std::sqrt(W_REG), sc = std::sqrt(W_CON);
    auto col = [&](int node) { return (node - first) * 12; };
    // one pass over all residual rows; with `sys` the normal equations are accumulated too
    auto pass = [&](Banded* sys) {
      double err = 0;
      int cols[12 * K];
      double vals[12 * K];
      for (int j = first; j < n; ++j) {   // E_rot
        const double* R = nodes[j].R;
        auto dotc = [&](int a, int b) { return R[a * 3] * R[b * 3] + R[a * 3 + 1] * R[b * 3 + 1] + R[a * 3 + 2] * R[b * 3 + 2]; };
        const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
        for (int q = 0; q < 3; ++q) {
          const int a = pairs[q][0], b = pairs[q][1];
          const double r = dotc(a, b);
          err += r * r;
          if (sys) {
            for (int i = 0; i < 3; ++i) { cols[i] = col(j) + a * 3 + i; vals[i] = R[b * 3 + i]; cols[3 + i] = col(j) + b * 3 + i; vals[3 + i] = R[a * 3 + i]; }
            sys->add_row(cols, vals, 6, r);
          }
        }
        for (int a = 0; a < 3; ++a) {
          const double r = dotc(a, a) - 1.0;
          err += r * r;
          if (sys) {
            for (int i = 0; i < 3; ++i) { cols[i] = col(j) + a * 3 + i; vals[i] = 2 * R[a * 3 + i]; }
            sys->add_row(cols, vals, 3, r);
          }
        }
      }
      for (int j = 0; j < n; ++j)   // E_reg
        for (int q = 0; q < K; ++q) {
          const int nbj = nodes[j].nb[q];
          if (!nodes[nbj].enabled && !nodes[j].enabled) continue;
          const Node &a = nodes[j], &b = nodes[nbj];
          const double d[3] = {b.g[0] - a.g[0], b.g[1] - a.g[1], b.g[2] - a.g[2]};
          for (int r = 0; r < 3; ++r) {
            const double res_r = (a.R[r] * d[0] + a.R[3 + r] * d[1] + a.R[6 + r] * d[2] + a.g[r] + a.t[r] - (b.g[r] + b.t[r])) * sr;
            err += res_r * res_r;
            if (!sys) continue;
            int cnt = 0;
            auto put = [&](int c, double v) { cols[cnt] = c; vals[cnt] = v; ++cnt; };
            if (nbj < j && b.enabled) put(col(nbj) + 9 + r, -sr);
            if (a.enabled) { put(col(j) + r, d[0] * sr); put(col(j) + 3 + r, d[1] * sr); put(col(j) + 6 + r, d[2] * sr); put(col(j) + 9 + r, sr); }
            if (nbj > j && b.enabled) put(col(nbj) + 9 + r, -sr);
            sys->add_row(cols, vals, cnt, res_r);
          }
        }
      for (auto& cn : cons) {   // E_con
        const Vertex& v = verts[cn.first];
        bool any = false;
        for (int i = 0; i < K; ++i) any = any || nodes[v.c[i].node].enabled;
        if (!any) continue;
        double q[3];
        deformed(nodes, v, q);
        for (int r = 0; r < 3; ++r) {
          const double res_r = (q[r] - cn.second[r]) * sc;
          err += res_r * res_r;
          if (!sys) continue;
          int cnt = 0;
          for (int i = 0; i < K; ++i) {
            const Node& nd = nodes[v.c[i].node];
            if (!nd.enabled) continue;
            const double w = v.c[i].w;
            cols[cnt] = col(v.c[i].node) + r; vals[cnt++] = (v.p[0] - nd.g[0]) * w * sc;
            cols[cnt] = col(v.c[i].node) + 3 + r; vals[cnt++] = (v.p[1] - nd.g[1]) * w * sc;
            cols[cnt] = col(v.c[i].node) + 6 + r; vals[cnt++] = (v.p[2] - nd.g[2]) * w * sc;
            cols[cnt] = col(v.c[i].node) + 9 + r; vals[cnt++] = w * sc;
          }
          sys->add_row(cols, vals, cnt, res_r);
        }
      }
      return err;
    };
    Banded sys(unknowns, band);
    double last = pass(&sys);
    res.error = (float)last;
    std::vector<double> delta;
    for (int it = 1; it <= 3; ++it) {   // DeformationGraph.cpp:456-481
      if (!sys.solve_negative(delta)) break;
      res.iterations = it;
      double dn = 0;
      for (int j = first; j < n; ++j) {
        const double* d = &delta[(size_t)col(j)];
        for (int i = 0; i < 9; ++i) nodes[j].R[i] += d[i];
        for (int i = 0; i < 3; ++i) nodes[j].t[i] += d[9 + i];
      }
      for (double x : delta) dn += x * x;
      sys = Banded(unknowns, band);
      const double err = pass(&sys);
      res.error = (float)err;
      if ((float)err > last || std::sqrt(dn) < 1e-2 || (float)err < 1e-3 || std::fabs((float)err - last) < 1e-5 * (float)err) break;
      last = (float)err;
    }
  }
  res.meanConsErr = mean_error();
emit:
  for (int i = 0; i < n; ++i) {
    float* g = graph16 + (size_t)i * 16;
    for (int r = 0; r < 3; ++r) g[r] = (float)nodes[i].g[r];
    for (int r = 0; r < 9; ++r) g[3 + r] = (float)nodes[i].R[r];
    for (int r = 0; r < 3; ++r) g[12 + r] = (float)nodes[i].t[r];
    g[15] = (float)nodes[i].time;
  }
  return res;
}

}  // namespace efd
