// Built-in optimiser of the deformation graphs: what Deformation::constrain computes in the reference (Core/Deformation.cpp:88-215
// over Core/Utils/DeformationGraph.cpp and a CHOLMOD sparse Cholesky) — the LOCAL graph (fernMatch = false, source-to-target
// constraints of one frame, nodes older than the last deformation fixed) and the GLOBAL one (fernMatch = true: all nodes free,
// relative constraints kept from earlier local closures, acceptance thresholds, keyframe / trajectory poses carried along) —
// written from scratch as a host-side Gauss-Newton solver on envelope (skyline) normal equations.  Embedded deformation (Sumner et al.): every graph node carries an affine 3x3 + translation;
//   E = E_rot (columns orthonormal, 6 rows per node) + 10 E_reg (a node predicts its sequence neighbours, 3 rows per pair)
//       + 100 E_con (constraint sources land on their targets, 3 rows per constraint)
// over the nodes younger than the last deformation.  Nodes are the model samples in time order, connected to their +-2 sequence
// neighbours (k = 4); a surface point is carried by the 4 nearest of the <= 20 nodes around its time, weights (1 - d / d_5th)^2.
// A row of the local problem spans <= 20 nodes, so its normal equations are banded; a relative constraint couples two times, which
// adds an arrow of long rows: both fit an envelope Cholesky (each row stored from its first non-zero), O(sum of envelope^2).
// Host code only (no GPU work: <= 1024 nodes); checked against the reference's own optimiser compiled where it lies
// (tests/test_deform_solver_vs_reference.py).
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

// symmetric positive definite system in envelope storage: row i holds A(i, first[i] .. i).  Built in two passes over the Jacobian
// rows: note() learns the envelope, add_row() accumulates J'J and J'r (columns of a row ascending).
struct Skyline {
  int n;
  std::vector<int> first;
  std::vector<size_t> off;
  std::vector<double> a, rhs;
  explicit Skyline(int n_) : n(n_), first((size_t)n_), rhs((size_t)n_, 0.0) { for (int i = 0; i < n; ++i) first[i] = i; }
  void note(const int* cols, int cnt) { for (int x = 1; x < cnt; ++x) first[cols[x]] = std::min(first[cols[x]], cols[0]); }
  void allocate() {
    off.assign((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) off[i + 1] = off[i] + (size_t)(i - first[i] + 1);
    a.assign(off[n], 0.0);
  }
  void clear() { std::fill(a.begin(), a.end(), 0.0); std::fill(rhs.begin(), rhs.end(), 0.0); }
  double& at(int i, int j) { return a[off[i] + (size_t)(j - first[i])]; }
  void add_row(const int* cols, const double* vals, int cnt, double r) {
    for (int x = 0; x < cnt; ++x) {
      rhs[cols[x]] += vals[x] * r;
      for (int y = 0; y <= x; ++y) at(cols[x], cols[y]) += vals[x] * vals[y];
    }
  }
  bool solve_negative(std::vector<double>& delta) {   // delta = -(J'J)^-1 J'r; the factor overwrites the matrix row by row
    for (int i = 0; i < n; ++i) {
      double* Li = &a[off[i]] - first[i];
      for (int j = first[i]; j < i; ++j) {
        const double* Lj = &a[off[j]] - first[j];
        double s = Li[j];
        for (int k = std::max(first[i], first[j]); k < j; ++k) s -= Li[k] * Lj[k];
        Li[j] = s / Lj[j];
      }
      double d = Li[i];
      for (int k = first[i]; k < i; ++k) d -= Li[k] * Li[k];
      if (!(d > 0)) return false;
      Li[i] = std::sqrt(d);
    }
    delta.assign((size_t)n, 0.0);
    for (int i = 0; i < n; ++i) {
      const double* Li = &a[off[i]] - first[i];
      double s = -rhs[i];
      for (int k = first[i]; k < i; ++k) s -= Li[k] * delta[k];
      delta[i] = s / Li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
      const double* Li = &a[off[i]] - first[i];
      delta[i] /= Li[i];
      for (int k = first[i]; k < i; ++k) delta[k] -= Li[k] * delta[i];
    }
    return true;
  }
};

// one entry of Deformation::constraints (Deformation.h:57-90): the point `src` seen at srcTime has to land on `target`; relative: on
// wherever the graph carries `target` (seen at targetTime) to; pin marks the constraints that hold a target in place (src == target)
struct Constraint {
  double src[3], target[3];
  uint64_t srcTime, targetTime;
  bool relative, pin;
};

struct Graph {
  std::vector<Node> nodes;
  bool build(const float* nodes4, int n, uint64_t last_deform_time) {
    if (n <= K) return false;   // Deformation::sampleGraphModel / sampleGraphFrom only build a graph for more than k samples (Deformation.cpp:219,283)
    nodes.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
      Node& nd = nodes[i];
      for (int r = 0; r < 3; ++r) { nd.g[r] = nodes4[i * 4 + r]; nd.t[r] = 0; }
      for (int r = 0; r < 9; ++r) nd.R[r] = (r % 4 == 0) ? 1.0 : 0.0;
      nd.time = (uint64_t)nodes4[i * 4 + 3];
      nd.enabled = nd.time > last_deform_time;
      int c = 0;   // sequence neighbours (DeformationGraph.cpp:239-266)
      if (i < K / 2) { for (int q = 0; q < K + 1; ++q) if (q != i) nd.nb[c++] = q; }
      else if (i >= n - K / 2) { for (int q = n - (K + 1); q < n; ++q) if (q != i) nd.nb[c++] = q; }
      else { for (int q = 0; q < K / 2; ++q) { nd.nb[c++] = i - (q + 1); nd.nb[c++] = i + (q + 1); } }
    }
    return true;
  }
  void emit(float* graph16) const {   // Deformation.cpp:176-190
    for (size_t i = 0; i < nodes.size(); ++i) {
      float* g = graph16 + i * 16;
      for (int r = 0; r < 3; ++r) g[r] = (float)nodes[i].g[r];
      for (int r = 0; r < 9; ++r) g[3 + r] = (float)nodes[i].R[r];
      for (int r = 0; r < 3; ++r) g[12 + r] = (float)nodes[i].t[r];
      g[15] = (float)nodes[i].time;
    }
  }
};

// DeformationGraph::optimiseGraphSparse (DeformationGraph.cpp:416-492) on a built graph.  Result.ok is its return value.
// The three gates of a GLOBAL closure (fernMatch): nothing to do below `entry` metres of mean constraint error (DeformationGraph.cpp:425);
// accepted only when the optimised mean constraint error is below `meanConsErr` and the energy below `energy` (Deformation.cpp:154).
// The defaults are the reference's hard-coded constants.
struct Gates {
  float entry = 0.06f, meanConsErr = 0.0003f, energy = 0.12f;
};
inline Result optimise(Graph& G, const Constraint* constraints, int m, bool fernMatch, const Gates& gates = Gates()) {
  std::vector<Node>& nodes = G.nodes;
  const int n = (int)nodes.size();
  Result res{false, 0, 0.f, 0.f};
  int first = n;
  for (int i = 0; i < n; ++i) if (nodes[i].enabled) { first = i; break; }
  const int unknowns = (n - first) * 12;
  // the vertices the constraints act on (Deformation.cpp:121-133): every source, and the target of a relative constraint
  struct Con { int v, tv; const double* target; };
  std::vector<Vertex> verts;
  std::vector<Con> cons;
  for (int i = 0; i < m; ++i) {
    const Constraint& c = constraints[i];
    Vertex v{{c.src[0], c.src[1], c.src[2]}, {}};
    carriers_of(nodes, v.p, c.srcTime, v.c);
    Con cn{(int)verts.size(), -1, c.target};
    verts.push_back(v);
    if (c.relative) {
      Vertex t{{c.target[0], c.target[1], c.target[2]}, {}};
      carriers_of(nodes, t.p, c.targetTime, t.c);
      cn.tv = (int)verts.size();
      verts.push_back(t);
    }
    cons.push_back(cn);
  }
  auto mean_error = [&]() {   // nonRelativeConstraintError: relative constraints count in the divisor only
    float e = 0;
    for (auto& cn : cons) {
      if (cn.tv >= 0) continue;
      double q[3];
      deformed(nodes, verts[cn.v], q);
      e += (float)std::sqrt((q[0] - cn.target[0]) * (q[0] - cn.target[0]) + (q[1] - cn.target[1]) * (q[1] - cn.target[1]) + (q[2] - cn.target[2]) * (q[2] - cn.target[2]));
    }
    return e / (float)cons.size();
  };
  res.meanConsErr = mean_error();
  if (fernMatch && res.meanConsErr < gates.entry) return res;   // the keyframe already agrees with the map: nothing to close
  res.ok = true;
  if (unknowns == 0) return res;
  const double sr = std::sqrt(W_REG), sc = std::sqrt(W_CON);
  auto col = [&](int node) { return (node - first) * 12; };
  enum Mode { RESIDUAL, PATTERN, ASSEMBLE };
  Skyline sys(unknowns);
  // one pass over all residual rows: the error, and either the envelope or the normal equations along with it
  auto pass = [&](Mode mode) {
    double err = 0;
    int cols[12 * K * 2];
    double vals[12 * K * 2];
    auto row = [&](int cnt, double r) { if (mode == PATTERN) sys.note(cols, cnt); else sys.add_row(cols, vals, cnt, r); };
    for (int j = first; j < n; ++j) {   // E_rot
      const double* R = nodes[j].R;
      auto dotc = [&](int a, int b) { return R[a * 3] * R[b * 3] + R[a * 3 + 1] * R[b * 3 + 1] + R[a * 3 + 2] * R[b * 3 + 2]; };
      const int pairs[3][2] = {{0, 1}, {0, 2}, {1, 2}};
      for (int q = 0; q < 3; ++q) {
        const int a = pairs[q][0], b = pairs[q][1];
        const double r = dotc(a, b);
        err += r * r;
        if (mode != RESIDUAL) {
          for (int i = 0; i < 3; ++i) { cols[i] = col(j) + a * 3 + i; vals[i] = R[b * 3 + i]; cols[3 + i] = col(j) + b * 3 + i; vals[3 + i] = R[a * 3 + i]; }
          row(6, r);
        }
      }
      for (int a = 0; a < 3; ++a) {
        const double r = dotc(a, a) - 1.0;
        err += r * r;
        if (mode != RESIDUAL) {
          for (int i = 0; i < 3; ++i) { cols[i] = col(j) + a * 3 + i; vals[i] = 2 * R[a * 3 + i]; }
          row(3, r);
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
          if (mode == RESIDUAL) continue;
          int cnt = 0;
          auto put = [&](int c, double v) { cols[cnt] = c; vals[cnt] = v; ++cnt; };
          if (nbj < j && b.enabled) put(col(nbj) + 9 + r, -sr);
          if (a.enabled) { put(col(j) + r, d[0] * sr); put(col(j) + 3 + r, d[1] * sr); put(col(j) + 6 + r, d[2] * sr); put(col(j) + 9 + r, sr); }
          if (nbj > j && b.enabled) put(col(nbj) + 9 + r, -sr);
          row(cnt, res_r);
        }
      }
    for (auto& cn : cons) {   // E_con
      const Vertex& v = verts[cn.v];
      bool any = false;
      for (int i = 0; i < K; ++i) any = any || nodes[v.c[i].node].enabled;
      if (cn.tv >= 0)
        for (int i = 0; i < K; ++i) any = any || nodes[verts[cn.tv].c[i].node].enabled;
      if (!any) continue;
      double q[3], tq[3] = {cn.target[0], cn.target[1], cn.target[2]};
      deformed(nodes, v, q);
      if (cn.tv >= 0) deformed(nodes, verts[cn.tv], tq);
      // the carriers of the row in node order; a node carrying both ends of a relative constraint gets the sum (DeformationGraph.cpp:655-749)
      struct Term { int node; double w; const double* p; double sign; };
      Term terms[2 * K];
      int nt = 0;
      for (int i = 0; i < K; ++i) terms[nt++] = Term{v.c[i].node, v.c[i].w, v.p, 1.0};
      if (cn.tv >= 0)
        for (int i = 0; i < K; ++i) terms[nt++] = Term{verts[cn.tv].c[i].node, verts[cn.tv].c[i].w, verts[cn.tv].p, -1.0};
      std::stable_sort(terms, terms + nt, [](const Term& x, const Term& y) { return x.node < y.node; });
      for (int r = 0; r < 3; ++r) {
        const double res_r = (q[r] - tq[r]) * sc;
        err += res_r * res_r;
        if (mode == RESIDUAL) continue;
        int cnt = 0, last = -1;
        for (int i = 0; i < nt; ++i) {
          const Node& nd = nodes[terms[i].node];
          if (!nd.enabled) continue;
          const double w = terms[i].w * terms[i].sign;
          const double e[4] = {(terms[i].p[0] - nd.g[0]) * w * sc, (terms[i].p[1] - nd.g[1]) * w * sc, (terms[i].p[2] - nd.g[2]) * w * sc, w * sc};
          if (terms[i].node == last) {
            for (int x = 0; x < 4; ++x) vals[cnt - 4 + x] += e[x];
          } else {
            for (int x = 0; x < 4; ++x) { cols[cnt] = col(terms[i].node) + 3 * x + r; vals[cnt++] = e[x]; }
            last = terms[i].node;
          }
        }
        row(cnt, res_r);
      }
    }
    return err;
  };
  pass(PATTERN);
  sys.allocate();
  double last = pass(ASSEMBLE);
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
    const double err_now = pass(RESIDUAL);
    res.error = (float)err_now;
    if ((float)err_now > last || std::sqrt(dn) < 1e-2 || (float)err_now < 1e-3 || std::fabs((float)err_now - last) < 1e-5 * (float)err_now ||
        (it == 1 && fernMatch && (float)err_now > 10.0f))
      break;
    last = (float)err_now;
    sys.clear();
    pass(ASSEMBLE);
  }
  res.meanConsErr = mean_error();
  return res;
}

// DeformationGraph::setPosesSeq + applyGraphToPoses (DeformationGraph.cpp:98-237): a camera pose is carried like a surface point at its
// translation.  The blended, re-orthonormalised rotation the reference computes is assigned to the temporary that
// Sophus::SE3d::rotationMatrix() returns by value (DeformationGraph.cpp:124), so only the translation moves — reproduced as observed.
// carriers: weights taken on the undeformed graph, before the optimisation, as setPosesSeq does.
inline void pose_carriers(const Graph& G, const double* T16, uint64_t time, Carrier (&out)[K]) {
  const double p[3] = {T16[3], T16[7], T16[11]};
  carriers_of(G.nodes, p, time, out);
}
inline void apply_to_pose(const Graph& G, const Carrier (&c)[K], double* T16) {
  Vertex v{{T16[3], T16[7], T16[11]}, {c[0], c[1], c[2], c[3]}};
  double q[3];
  deformed(G.nodes, v, q);
  T16[3] = q[0]; T16[7] = q[1]; T16[11] = q[2];
}

// Deformation::constrain (Deformation.cpp:88-215) on explicit inputs.  poses16 (n_poses x 16, row-major, in/out) with pose_times: the
// keyframe poses — and with fernMatch the trajectory — that are deformed along when the result is accepted.  Returns poseUpdated.
// new_relative: what a local closure leaves behind for later global ones (Deformation.cpp:160-173) — per plain constraint the
// DEFORMED source against its target, relative.
inline bool constrain(const float* nodes4, int n, const Constraint* constraints, int m, bool fernMatch, uint64_t last_deform_time, double* poses16,
                      const int64_t* pose_times, int n_poses, float* graph16, Result* result, std::vector<Constraint>* new_relative = nullptr,
                      const Gates& gates = Gates()) {
  Graph G;
  Result res{false, 0, 0.f, 0.f};
  if (result) *result = res;
  if (m <= 0 || !G.build(nodes4, n, last_deform_time)) return false;
  std::vector<Carrier> pc((size_t)n_poses * K);
  for (int i = 0; i < n_poses; ++i) {
    Carrier c[K];
    pose_carriers(G, poses16 + (size_t)i * 16, (uint64_t)pose_times[i], c);
    for (int j = 0; j < K; ++j) pc[(size_t)i * K + j] = c[j];
  }
  res = optimise(G, constraints, m, fernMatch, gates);
  if (result) *result = res;
  if (!(!fernMatch || (res.ok && res.meanConsErr < gates.meanConsErr && res.error < gates.energy))) return false;
  for (int i = 0; i < n_poses; ++i) {
    Carrier c[K];
    for (int j = 0; j < K; ++j) c[j] = pc[(size_t)i * K + j];
    apply_to_pose(G, c, poses16 + (size_t)i * 16);
  }
  if (!fernMatch && new_relative) {
    new_relative->clear();
    for (int i = 0; i < m; ++i) {
      const Constraint& c = constraints[i];
      if (c.relative || c.pin) continue;
      Vertex v{{c.src[0], c.src[1], c.src[2]}, {}};
      carriers_of(G.nodes, v.p, c.srcTime, v.c);
      Constraint r = c;
      deformed(G.nodes, v, r.src);
      r.relative = true;
      new_relative->push_back(r);
    }
  }
  G.emit(graph16);
  return true;
}

// the local case as ElasticFusion.cpp:488-516 uses it.  nodes4: n x {x, y, z, time} (ef_sample_graph); constraints: m x {src xyz, target
// xyz, target time, pin} (ef_get_local_loop), all with source time `src_time`; graph16: n x {position 3, rotation 9 column-major,
// translation 3, time} (the layout ef_set_deformation takes).
inline Result solve_local(const float* nodes4, int n, const double* constraints, int m, uint64_t src_time, uint64_t last_deform_time, float* graph16) {
  std::vector<Constraint> cons;
  for (int i = 0; i < m; ++i) {   // Deformation::addConstraint(src, target, srcTime, targetTime, pin) (Deformation.cpp:73-86)
    const double* c = constraints + (size_t)i * 8;
    cons.push_back(Constraint{{c[0], c[1], c[2]}, {c[3], c[4], c[5]}, src_time, (uint64_t)c[6], false, false});
    if (c[7] != 0) cons.push_back(Constraint{{c[3], c[4], c[5]}, {c[3], c[4], c[5]}, (uint64_t)c[6], (uint64_t)c[6], false, true});
  }
  Result res{false, 0, 0.f, 0.f};
  constrain(nodes4, n, cons.data(), (int)cons.size(), false, last_deform_time, nullptr, nullptr, 0, graph16, &res);
  return res;
}

}  // namespace efd
