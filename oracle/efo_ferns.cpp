// TEST INFRASTRUCTURE — CPU oracle. Not part of the product path (see oracle/README.md).
//
// CPU restatement of the fern keyframe database, Core/Ferns.cpp:22-393 (Ferns.h:35-184), which the GLOBAL loop closure and relocalisation
// of ElasticFusion::processFrame consult (ElasticFusion.cpp:392-404, 609-618).  Follows the reference function by function; the images
// are what Resize::image / Resize::vertex read back (80x60 at the default resolution), the fern-to-view registration (Ferns.cpp:243-258)
// is a callback so that the frame loop can put its own tracker there.  Checked against the compiled Ferns.cpp
// (tests/test_ferns_vs_reference.py, backend "oracle").
#include <cmath>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include "efo_api.h"
#include "efo_linalg.h"
#include "efo_pose.h"

using namespace efo;

namespace {
struct FernTest { int px, py, r, g, b, d; std::vector<int> ids[16]; };   // Ferns.h:72-80
struct KeyFrame {                                                        // Ferns.h:84-145
  std::vector<uint8_t> codes;
  int goodCodes, id, srcTime;
  SE3 T_wc;
  std::vector<uint8_t> initRgb;
  std::vector<float> initVerts, initNorms;
};
}  // namespace

struct efo_ferns {
  int num, factor, width, height, maxDepth;
  float photoThresh, fx, fy, cx, cy;
  int lastClosest;
  std::vector<FernTest> conservatory;
  std::vector<KeyFrame> frames;

  // the loop of Ferns.cpp:97-118 (addFrame) and :186-208 (findFrame)
  int codesOf(const uint8_t* rgb, int ch, const float* verts, std::vector<uint8_t>& codes, std::vector<int>& coOccurrences) const {
    int goodCodes = 0;
    codes.assign(num, 255);
    coOccurrences.assign(frames.size(), 0);
    for (int i = 0; i < num; ++i) {
      const FernTest& f = conservatory[i];
      const size_t at = (size_t)f.py * width + f.px;
      if (verts[at * 4 + 2] > 0) {
        const uint8_t* pix = rgb + at * ch;
        const uint8_t code = (pix[0] > f.r) << 3 | (pix[1] > f.g) << 2 | (pix[2] > f.b) << 1 | (int(verts[at * 4 + 2] * 1000.0f) > f.d);
        goodCodes++;
        for (size_t j = 0; j < f.ids[code].size(); ++j) coOccurrences[f.ids[code][j]]++;
        codes[i] = code;
      }
    }
    return goodCodes;
  }
};

extern "C" {

efo_ferns* efo_ferns_create(int num, int max_depth_mm, float photo_thresh, int width, int height, float fx, float fy, float cx, float cy, unsigned seed) {
  efo_ferns* f = new efo_ferns();
  f->num = num; f->factor = 8; f->width = width / 8; f->height = height / 8; f->maxDepth = max_depth_mm; f->photoThresh = photo_thresh;
  f->fx = fx; f->fy = fy; f->cx = cx; f->cy = cy; f->lastClosest = -1;
  std::mt19937 random;
  random.seed(seed);                                                      // Ferns.cpp:52 (time(0) there)
  std::uniform_int_distribution<int32_t> widthDist(0, f->width - 1), heightDist(0, f->height - 1), rgbDist(0, 255), dDist(400, max_depth_mm);
  for (int i = 0; i < num; ++i) {                                         // generateFerns, Ferns.cpp:62-77
    FernTest t;
    t.px = widthDist(random);
    t.py = heightDist(random);
    t.r = rgbDist(random);
    t.g = rgbDist(random);
    t.b = rgbDist(random);
    t.d = dDist(random);
    f->conservatory.push_back(t);
  }
  return f;
}
void efo_ferns_destroy(efo_ferns* f) { delete f; }
int efo_ferns_get_table(const efo_ferns* f, int* t) {
  for (int i = 0; i < f->num; ++i) {
    const FernTest& e = f->conservatory[i];
    const int row[6] = {e.px, e.py, e.r, e.g, e.b, e.d};
    std::memcpy(t + i * 6, row, sizeof(row));
  }
  return 0;
}
int efo_ferns_set_table(efo_ferns* f, const int* t) {
  if (!f->frames.empty()) return -4;
  for (int i = 0; i < f->num; ++i) {
    FernTest& e = f->conservatory[i];
    e.px = t[i * 6]; e.py = t[i * 6 + 1]; e.r = t[i * 6 + 2]; e.g = t[i * 6 + 3]; e.b = t[i * 6 + 4]; e.d = t[i * 6 + 5];
  }
  return 0;
}

// Ferns::addFrame, Ferns.cpp:78-160
int efo_ferns_add_frame(efo_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int src_time, float threshold) {
  KeyFrame frame;
  std::vector<int> coOccurrences;
  frame.goodCodes = f->codesOf(rgb, ch, verts4, frame.codes, coOccurrences);
  float minimum = std::numeric_limits<float>::max();
  if (frame.goodCodes > 0) {
    for (size_t i = 0; i < f->frames.size(); ++i) {
      float maxCo = std::min(frame.goodCodes, f->frames[i].goodCodes);
      float dissim = (float)(maxCo - coOccurrences[i]) / (float)maxCo;
      if (dissim < minimum) minimum = dissim;
    }
  }
  if ((minimum > threshold || f->frames.size() == 0) && frame.goodCodes > 0) {
    frame.id = (int)f->frames.size();
    frame.srcTime = src_time;
    frame.T_wc = se3_from_matrix(T_wc16);
    const size_t px = (size_t)f->width * f->height;
    frame.initRgb.resize(px * 3);
    for (size_t i = 0; i < px; ++i) { frame.initRgb[i * 3] = rgb[i * ch]; frame.initRgb[i * 3 + 1] = rgb[i * ch + 1]; frame.initRgb[i * 3 + 2] = rgb[i * ch + 2]; }
    frame.initVerts.assign(verts4, verts4 + px * 4);
    frame.initNorms.assign(norms4, norms4 + px * 4);
    for (int i = 0; i < f->num; ++i)
      if (frame.codes[i] != 255) f->conservatory[i].ids[frame.codes[i]].push_back(frame.id);
    f->frames.push_back(frame);
    return 1;
  }
  return 0;
}

// Ferns::blockHDAware, Ferns.cpp:378-393
static float blockHDAware(const efo_ferns* f, const std::vector<uint8_t>& a, const std::vector<uint8_t>& b) {
  int count = 0;
  float val = 0;
  for (int i = 0; i < f->num; ++i) {
    if (a[i] != 255 && b[i] != 255) {
      count++;
      if (a[i] == b[i]) val += 1.0f;
    }
  }
  return val / (float)count;
}

// Ferns::photometricCheck, Ferns.cpp:301-383
static float photometricCheck(const efo_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const SE3& T_wc_est, const KeyFrame& k) {
  const float cx = f->cx / f->factor, cy = f->cy / f->factor;
  const float invfx = 1.0f / float(f->fx / f->factor), invfy = 1.0f / float(f->fy / f->factor);
  float photoSum = 0;
  int photoCount = 0;
  const M4d Td = se3_matrix(se3_mul(se3_inverse(k.T_wc), T_wc_est));
  const Mat4f T = pose_castf(Td.m);                                       // .cast<float>()
  for (int i = 0; i < f->num; ++i) {
    const FernTest& t = f->conservatory[i];
    const size_t at = (size_t)t.py * f->width + t.px;
    const float* v = verts4 + at * 4;
    if (v[2] > 0 && int(v[2] * 1000.0f) < f->maxDepth) {
      float q[3];
      for (int r = 0; r < 3; ++r) q[r] = ((T.m[r * 4] * v[0] + T.m[r * 4 + 1] * v[1]) + T.m[r * 4 + 2] * v[2]) + T.m[r * 4 + 3] * 1.0f;
      const int u = (int)(q[0] * (1 / invfx) / q[2] + cx);
      const int w = (int)(q[1] * (1 / invfy) / q[2] + cy);
      if (u >= 0 && w >= 0 && u < f->width && w < f->height) {
        const uint8_t* fern = &k.initRgb[((size_t)w * f->width + u) * 3];
        if (fern[0] > 0 || fern[1] > 0 || fern[2] > 0) {
          const uint8_t* pix = rgb + at * ch;
          photoSum += std::abs((int)fern[0] - (int)pix[0]);
          photoSum += std::abs((int)fern[1] - (int)pix[1]);
          photoSum += std::abs((int)fern[2] - (int)pix[2]);
          photoCount++;
        }
      }
    }
  }
  return photoSum / float(photoCount);
}

// Ferns::findFrame, Ferns.cpp:162-298
int efo_ferns_find_frame(efo_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int time, int lost,
                         efo_fern_tracker tracker, void* user, double* T_est16_out, double* cons, int max_cons, int* n_out) {
  f->lastClosest = -1;
  if (n_out) *n_out = 0;
  std::vector<uint8_t> codes;
  std::vector<int> coOccurrences;
  const int goodCodes = f->codesOf(rgb, ch, verts4, codes, coOccurrences);
  float minimum = std::numeric_limits<float>::max();
  int minId = -1;
  for (size_t i = 0; i < f->frames.size(); ++i) {
    float maxCo = std::min(goodCodes, f->frames[i].goodCodes);
    float dissim = (float)(maxCo - coOccurrences[i]) / (float)maxCo;
    if (dissim < minimum && time - f->frames[i].srcTime > 300) {
      minimum = dissim;
      minId = (int)i;
    }
  }
  SE3 T_wc_est = se3_identity();
  if (minId != -1 && blockHDAware(f, codes, f->frames[minId].codes) > 0.3) {
    const KeyFrame& k = f->frames[minId];
    const M4d Tf = se3_matrix(k.T_wc);
    double T[16];
    std::memcpy(T, Tf.m, sizeof(T));
    float lastICPError = 0, lastICPCount = 0;
    tracker(user, k.initVerts.data(), k.initNorms.data(), Tf.m, verts4, norms4, T, &lastICPError, &lastICPCount);   // :243-258
    T_wc_est = se3_from_matrix(T);
    const float photoError = photometricCheck(f, rgb, ch, verts4, T_wc_est, k);
    const int icpCountThresh = lost ? 1400 : 2400;
    if (lastICPError < 0.0003 && lastICPCount > icpCountThresh && photoError < f->photoThresh) {
      f->lastClosest = minId;
      int n = 0;
      const M4d E = se3_matrix(T_wc_est);
      const int step = f->num / 50;   // the reference never terminates below 50 ferns; nothing is sampled here
      for (int i = 0; step > 0 && i < f->num; i += step) {
        const FernTest& t = f->conservatory[i];
        const float* v = verts4 + ((size_t)t.py * f->width + t.px) * 4;
        if (v[2] > 0 && int(v[2] * 1000.0f) < f->maxDepth) {
          if (n < max_cons)
            for (int r = 0; r < 3; ++r) {
              cons[n * 6 + r] = ((T_wc16[r * 4] * (double)v[0] + T_wc16[r * 4 + 1] * (double)v[1]) + T_wc16[r * 4 + 2] * (double)v[2]) + T_wc16[r * 4 + 3] * 1.0;
              cons[n * 6 + 3 + r] = ((E.m[r * 4] * (double)v[0] + E.m[r * 4 + 1] * (double)v[1]) + E.m[r * 4 + 2] * (double)v[2]) + E.m[r * 4 + 3] * 1.0;
            }
          ++n;
        }
      }
      if (n_out) *n_out = n;
    }
  }
  const M4d E = se3_matrix(T_wc_est);
  std::memcpy(T_est16_out, E.m, sizeof(E.m));
  return f->lastClosest;
}

int efo_ferns_count(const efo_ferns* f) { return (int)f->frames.size(); }
int efo_ferns_last_closest(const efo_ferns* f) { return f->lastClosest; }
int efo_ferns_get_frame(const efo_ferns* f, int id, uint8_t* codes, int* good, int* src_time, double* T_wc16, uint8_t* rgb3, float* verts4, float* norms4) {
  if (id < 0 || id >= (int)f->frames.size()) return -1;
  const KeyFrame& k = f->frames[id];
  if (codes) std::memcpy(codes, k.codes.data(), k.codes.size());
  if (good) *good = k.goodCodes;
  if (src_time) *src_time = k.srcTime;
  if (T_wc16) { const M4d M = se3_matrix(k.T_wc); std::memcpy(T_wc16, M.m, sizeof(M.m)); }
  if (rgb3) std::memcpy(rgb3, k.initRgb.data(), k.initRgb.size());
  if (verts4) std::memcpy(verts4, k.initVerts.data(), k.initVerts.size() * 4);
  if (norms4) std::memcpy(norms4, k.initNorms.data(), k.initNorms.size() * 4);
  return 0;
}
int efo_ferns_set_frame_pose(efo_ferns* f, int id, const double* T_wc16) {
  if (id < 0 || id >= (int)f->frames.size()) return -1;
  f->frames[id].T_wc = se3_from_matrix(T_wc16);
  return 0;
}
float efo_ferns_block_hd_aware(const efo_ferns* f, int a, int b) { return blockHDAware(f, f->frames[a].codes, f->frames[b].codes); }
float efo_ferns_photometric_check(const efo_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const double* T_est16, int id) {
  return photometricCheck(f, rgb, ch, verts4, se3_from_matrix(T_est16), f->frames[id]);
}

}  // extern "C"
