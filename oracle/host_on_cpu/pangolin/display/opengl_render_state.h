// TEST INFRASTRUCTURE — pangolin::OpenGlMatrix, only so that the reference's draw-only functions compile (they are never called)
#pragma once
#include <Eigen/Core>
namespace pangolin {
struct OpenGlMatrix {
  double m[16];
  operator Eigen::Matrix4f() const {
    Eigen::Matrix4f r;
    for (int i = 0; i < 16; ++i) r.data()[i] = (float)m[i];
    return r;
  }
};
}  // namespace pangolin
