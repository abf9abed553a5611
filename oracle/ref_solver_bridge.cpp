// TEST INFRASTRUCTURE — runs the REFERENCE's own deformation-graph optimiser: Core/Deformation.cpp (constrain), Core/Utils/DeformationGraph.cpp
// (graph connectivity, vertex weights, residual, sparse Jacobian, Gauss-Newton loop) and Core/Utils/CholeskyDecomp.cpp, compiled from
// /root/reference where they lie — the latter over host_on_cpu/cholmod.h, a dense stand-in for the CHOLMOD calls it makes (SuiteSparse is
// absent here) — so that the product's built-in solver (elasticfusion_amd/csrc/ef_deform_solver.hpp) has the reference's to be compared
// with.  The graph nodes reach Deformation::sampleGraphModel through the tape-recorder GL's scripted read-backs.  This file is ours.
// oracle/Makefile, target `refsolver` -> _ref/libefr_solver.so.
#include <cstring>
#include <vector>

#include "Deformation.h"

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
}
