// TEST INFRASTRUCTURE — runs the REFERENCE's own deformation-graph optimiser: Core/Deformation.cpp (constrain), Core/Utils/DeformationGraph.cpp
// (graph connectivity, vertex weights, residual, sparse Jacobian, Gauss-Newton loop) and Core/Utils/CholeskyDecomp.cpp, compiled from
// /root/reference where they lie — the latter over host_on_cpu/cholmod.h, a dense stand-in for the CHOLMOD calls it makes (SuiteSparse is
// absent here) — so that the product's built-in solver (elasticfusion_amd/csrc/ef_deform_solver.hpp) has the reference's to be compared
// with.  The graph nodes reach Deformation::sampleGraphModel through the tape-recorder GL's scripted read-backs.  This file is ours.
// oracle/Makefile, target `refsolver` -> _ref/libefr_solver.so.
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

#include "Ferns.h"
// Deformation::lastDeformTime (Deformation.h:126) is private and only ever set by an accepted local closure: efs_constrain sets it directly
#define private public
#include "Deformation.h"
#undef private
#include "efo_linalg.h"

const std::string GPUTexture::RGB = "RGB";
const std::string GPUTexture::DEPTH_RAW = "DEPTH";
const std::string GPUTexture::DEPTH_FILTERED = "DEPTH_FILTERED";
const std::string GPUTexture::DEPTH_METRIC = "DEPTH_METRIC";
const std::string GPUTexture::DEPTH_METRIC_FILTERED = "DEPTH_METRIC_FILTERED";
const std::string GPUTexture::DEPTH_NORM = "DEPTH_NORM";
GPUTexture::GPUTexture(const int w, const int h, const GLenum a, const GLenum b, const GLenum c, const bool d)
    : texture(nullptr), cudaRes(nullptr), draw(d), width(w), height(h), internalFormat(a), format(b), dataType(c) {}
GPUTexture::~GPUTexture() {}

extern "C" {
// nodes4: n x {x, y, z, time} as Deformation::sampleGraphModel reads them back; constraints: m x {src xyz, target xyz, target time, pin}
// (the rows of ElasticFusion.cpp:488-509, source time = `time`).  graph_out: n x 16 floats (Deformation.cpp:176-190).  Returns what constrain() returns.
// prior_time > 0: a deformation is first carried out at that time (same nodes and constraints), so that Deformation::lastDeformTime ==
// prior_time and only younger nodes are optimised by the one that is returned.
int efs_local_constrain(const float* nodes4, int n, const double* constraints, int m, int time, int prior_time, float* graph_out, int* nodes_out) {
  Resolution::getInstance(640, 480);
  Intrinsics::getInstance(528, 528, 320, 240);
  Deformation d;
  glrec::S().buffer_data.assign((const unsigned char*)nodes4, (const unsigned char*)nodes4 + (size_t)n * 16);
  const std::pair<GLuint, GLuint> model(1, 2);
  if (prior_time > 0) {
    glrec::S().query_queue.assign(1, n);
    d.sampleGraphModel(model);
    for (int i = 0; i < m; ++i) {
      const double* c = constraints + (size_t)i * 8;
      d.addConstraint(Eigen::Vector4d(c[0], c[1], c[2], 1.0), Eigen::Vector4d(c[3], c[4], c[5], 1.0), (uint64_t)prior_time, (uint64_t)c[6], c[7] != 0);
    }
    std::vector<Ferns::Frame*> f0;
    std::vector<float> r0;
    std::vector<std::pair<uint64_t, Sophus::SE3d>> p0;
    d.constrain(f0, r0, prior_time, false, p0, false, nullptr);
  }
  glrec::S().query_queue.assign(1, n);
  d.sampleGraphModel(model);
  for (int i = 0; i < m; ++i) {
    const double* c = constraints + (size_t)i * 8;
    d.addConstraint(Eigen::Vector4d(c[0], c[1], c[2], 1.0), Eigen::Vector4d(c[3], c[4], c[5], 1.0), (uint64_t)time, (uint64_t)c[6], c[7] != 0);
  }
  std::vector<Ferns::Frame*> ferns;
  std::vector<float> raw;
  std::vector<std::pair<uint64_t, Sophus::SE3d>> t_T_wc;
  std::vector<Deformation::Constraint> rel;
  const bool ok = d.constrain(ferns, raw, time, false, t_T_wc, false, &rel);
  *nodes_out = (int)(raw.size() / 16);
  if (!raw.empty()) std::memcpy(graph_out, raw.data(), raw.size() * sizeof(float));
  glrec::S().buffer_data.clear();
  glrec::S().log.clear();
  return ok ? 1 : 0;
}
// the general form: explicit Deformation::Constraint entries (relative / pin included), fernMatch / relaxGraph as ElasticFusion.cpp:428,515
// pass them, the keyframe poses (Ferns::Frame::T_wc) and the trajectory (t_T_wc) that constrain() deforms along.  Returns poseUpdated;
// rel_out: the relative constraints a local closure leaves behind (newRelativeCons), n_rel x {src xyz, target xyz, srcTime, targetTime}.
struct efs_constraint { double src[3], target[3]; long long src_time, target_time; int relative, pin; };
int efs_constrain(const float* nodes4, int n, const efs_constraint* cons, int m, int time, int fernMatch, int relaxGraph, int lastDeformTime, double* fern_poses16,
                  const long long* fern_times, int nf, double* traj_poses16, const long long* traj_times, int nt, float* graph_out, int* nodes_out,
                  double* rel_out, int* n_rel) {
  Resolution::getInstance(640, 480);
  Intrinsics::getInstance(528, 528, 320, 240);
  Deformation d;
  glrec::S().buffer_data.assign((const unsigned char*)nodes4, (const unsigned char*)nodes4 + (size_t)n * 16);
  glrec::S().query_queue.assign(1, n);
  d.sampleGraphModel(std::pair<GLuint, GLuint>(1, 2));
  d.lastDeformTime = lastDeformTime;
  for (int i = 0; i < m; ++i)
    d.addConstraint(Deformation::Constraint(Eigen::Vector3d(cons[i].src[0], cons[i].src[1], cons[i].src[2]),
                                            Eigen::Vector3d(cons[i].target[0], cons[i].target[1], cons[i].target[2]), (uint64_t)cons[i].src_time,
                                            (uint64_t)cons[i].target_time, cons[i].relative != 0, cons[i].pin != 0));
  std::vector<Ferns::Frame*> ferns;
  for (int i = 0; i < nf; ++i) ferns.push_back(new Ferns::Frame(1, i, Sophus::SE3d(efo::se3_from_matrix(fern_poses16 + i * 16)), (int)fern_times[i], 1));
  std::vector<std::pair<uint64_t, Sophus::SE3d>> traj;
  for (int i = 0; i < nt; ++i) traj.emplace_back((uint64_t)traj_times[i], Sophus::SE3d(efo::se3_from_matrix(traj_poses16 + i * 16)));
  std::vector<float> raw;
  std::vector<Deformation::Constraint> rel;
  const bool ok = d.constrain(ferns, raw, time, fernMatch != 0, traj, relaxGraph != 0, &rel);
  *nodes_out = (int)(raw.size() / 16);
  if (!raw.empty()) std::memcpy(graph_out, raw.data(), raw.size() * sizeof(float));
  auto put = [](const Sophus::SE3d& T, double* o) {
    const Eigen::Matrix4d M = T.matrix();
    for (int r = 0; r < 4; ++r)
      for (int c = 0; c < 4; ++c) o[r * 4 + c] = M(r, c);
  };
  for (int i = 0; i < nf; ++i) { put(ferns[i]->T_wc, fern_poses16 + i * 16); delete ferns[i]; }
  for (int i = 0; i < nt; ++i) put(traj[i].second, traj_poses16 + i * 16);
  *n_rel = (int)rel.size();
  for (size_t i = 0; i < rel.size(); ++i) {
    for (int k = 0; k < 3; ++k) { rel_out[i * 8 + k] = rel[i].src(k); rel_out[i * 8 + 3 + k] = rel[i].target(k); }
    rel_out[i * 8 + 6] = (double)rel[i].srcTime;
    rel_out[i * 8 + 7] = (double)rel[i].targetTime;
  }
  glrec::S().buffer_data.clear();
  glrec::S().log.clear();
  return ok ? 1 : 0;
}
}
