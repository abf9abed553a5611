// TEST INFRASTRUCTURE — the pangolin::glDraw* helpers Tools/GUI.h and MainController.cpp call (see ../pangolin.h)
#pragma once
#include <Eigen/Core>
namespace pangolin {
inline void glDrawLine(float, float, float, float, float, float) {}
inline void glDrawCross(float, float, float, float = 0.1f) {}
template <typename K, typename P> inline void glDrawFrustum(const K&, int, int, const P&, float) {}
}  // namespace pangolin
