// Fern database: the keyframe store behind the reference's GLOBAL loop closure and relocalisation (Core/Ferns.h:35-184,
// Core/Ferns.cpp:22-393).  Everything in here is host-side bookkeeping on 1/8-resolution images the device already produced
// (ef_get_image_resized): `num` random ferns each make four binary tests on one pixel (r, g, b against a byte threshold, depth in
// mm against a threshold), a frame is the vector of those 4-bit codes, two frames are compared through the inverted lists the
// ferns keep per code value.  The one piece of per-pixel arithmetic the reference does between a stored frame and a new view — an
// 80x60 ICP — is not done here: ef_ferns_find_frame calls the caller's ef_fern_tracker for it.
//
// Layout choices (not the reference's): frames own flat std::vector storage instead of new[]'d Eigen arrays; a fern's inverted
// lists are 16 vectors of frame ids as in the reference because the co-occurrence count walks exactly those.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <memory>
#include <random>
#include <vector>

#include "../../include/ef_hip.h"
#include "ef_deform_solver.hpp"
#include "ef_linalg_dev.hpp"

namespace {
constexpr uint8_t BAD_CODE = 255;   // Ferns.cpp:35 badCode

struct Fern {
  int x, y;              // Fern::pos
  int rgbd[4];           // thresholds: r, g, b (0..255), depth in mm (400..maxDepth)
  std::vector<int> ids[16];
};
struct StoredFrame {
  std::vector<uint8_t> codes;
  int goodCodes = 0;
  int id = 0;
  efl::SE3 T_wc;
  int srcTime = 0;
  std::vector<uint8_t> rgb;     // 3 bytes per pixel
  std::vector<float> verts, norms;
};
// one view as the caller hands it over
struct View {
  const uint8_t* rgb;
  int ch;
  const float* verts;
  int w;
  float z(int x, int y) const { return verts[((size_t)y * w + x) * 4 + 2]; }
  const float* v(int x, int y) const { return verts + ((size_t)y * w + x) * 4; }
  const uint8_t* px(int x, int y) const { return rgb + ((size_t)y * w + x) * ch; }
};
void identity16(double* M) {
  for (int i = 0; i < 16; ++i) M[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
}  // namespace

struct ef_ferns {
  int num, factor, width, height, maxDepth;
  float photoThresh;
  float fx, fy, cx, cy;   // full-resolution intrinsics
  int lastClosest = -1;
  int tableVersion = 0;   // bumped by ef_ferns_set_table (a device-side copy of the table knows when to refresh)
  std::vector<Fern> conservatory;
  std::vector<std::unique_ptr<StoredFrame>> frames;

  // Ferns.cpp:97-118 / :186-208: the codes of one view and, through the inverted lists, how many codes it shares with each stored frame
  void encode(const View& v, std::vector<uint8_t>& codes, int& good, std::vector<int>& coOccurrences) const {
    codes.assign(num, BAD_CODE);
    good = 0;
    for (int i = 0; i < num; ++i) {
      const Fern& f = conservatory[i];
      const float z = v.z(f.x, f.y);
      if (z > 0) {
        const uint8_t* p = v.px(f.x, f.y);
        codes[i] = (uint8_t)((p[0] > f.rgbd[0]) << 3 | (p[1] > f.rgbd[1]) << 2 | (p[2] > f.rgbd[2]) << 1 | (int(z * 1000.0f) > f.rgbd[3]));
        ++good;
      }
    }
    cooccur(codes, coOccurrences);
  }
  // the inverted-list walk alone, for codes that were computed elsewhere (on the device: k_fern_codes, ef_context.hip)
  void cooccur(const std::vector<uint8_t>& codes, std::vector<int>& coOccurrences) const {
    coOccurrences.assign(frames.size(), 0);
    for (int i = 0; i < num; ++i)
      if (codes[i] != BAD_CODE)
        for (int id : conservatory[i].ids[codes[i]]) ++coOccurrences[id];
  }
  float dissimilarity(int good, const StoredFrame& s, int co) const {
    const float maxCo = (float)(good < s.goodCodes ? good : s.goodCodes);
    return (maxCo - (float)co) / maxCo;
  }
  // Ferns.cpp:378-393
  float blockHDAware(const std::vector<uint8_t>& a, const std::vector<uint8_t>& b) const {
    int count = 0;
    float val = 0;
    for (int i = 0; i < num; ++i)
      if (a[i] != BAD_CODE && b[i] != BAD_CODE) {
        ++count;
        if (a[i] == b[i]) val += 1.0f;
      }
    return val / (float)count;
  }
  bool usable(const View& v, const Fern& f) const {   // the depth test shared by the photometric check and the constraints
    const float z = v.z(f.x, f.y);
    return z > 0 && int(z * 1000.0f) < maxDepth;
  }
  // Ferns.cpp:301-383: mean absolute colour difference between the view and the stored frame it was registered to
  float photometricCheck(const View& v, const efl::SE3& T_wc_est, const StoredFrame& s) const {
    const float cxs = cx / factor, cys = cy / factor;
    const float invfx = 1.0f / float(fx / factor), invfy = 1.0f / float(fy / factor);
    float M[16];
    efl::se3_castf_matrix(efl::se3_mul(efl::se3_inverse(s.T_wc), T_wc_est), M);
    float photoSum = 0;
    int photoCount = 0;
    for (int i = 0; i < num; ++i) {
      const Fern& f = conservatory[i];
      if (!usable(v, f)) continue;
      const float* p = v.v(f.x, f.y);
      float q[3];
      for (int r = 0; r < 3; ++r) q[r] = ((M[r * 4] * p[0] + M[r * 4 + 1] * p[1]) + M[r * 4 + 2] * p[2]) + M[r * 4 + 3] * 1.0f;
      const int u = (int)(q[0] * (1 / invfx) / q[2] + cxs);
      const int w = (int)(q[1] * (1 / invfy) / q[2] + cys);
      if (u >= 0 && w >= 0 && u < width && w < height) {
        const uint8_t* a = s.rgb.data() + ((size_t)w * width + u) * 3;
        if (a[0] > 0 || a[1] > 0 || a[2] > 0) {
          const uint8_t* b = v.px(f.x, f.y);
          photoSum += std::abs((int)a[0] - (int)b[0]);
          photoSum += std::abs((int)a[1] - (int)b[1]);
          photoSum += std::abs((int)a[2] - (int)b[2]);
          ++photoCount;
        }
      }
    }
    return photoSum / float(photoCount);
  }
};

extern "C" {

ef_ferns* ef_ferns_create(int num, int max_depth_mm, float photo_thresh, int width, int height, float fx, float fy, float cx, float cy,
                          unsigned seed) {
  if (num <= 0 || width < 8 || height < 8 || max_depth_mm < 400) return nullptr;
  ef_ferns* f = new ef_ferns();
  f->num = num;
  f->factor = 8;                       // Ferns.cpp:25
  f->width = width / f->factor;
  f->height = height / f->factor;
  f->maxDepth = max_depth_mm;
  f->photoThresh = photo_thresh;
  f->fx = fx; f->fy = fy; f->cx = cx; f->cy = cy;
  // Ferns::generateFerns (Ferns.cpp:62-77): position, then the three colour thresholds, then the depth threshold, fern by fern
  std::mt19937 random(seed);
  std::uniform_int_distribution<int32_t> widthDist(0, f->width - 1), heightDist(0, f->height - 1), rgbDist(0, 255), dDist(400, max_depth_mm);
  f->conservatory.resize(num);
  for (Fern& e : f->conservatory) {
    e.x = widthDist(random);
    e.y = heightDist(random);
    e.rgbd[0] = rgbDist(random);
    e.rgbd[1] = rgbDist(random);
    e.rgbd[2] = rgbDist(random);
    e.rgbd[3] = dDist(random);
  }
  return f;
}
void ef_ferns_destroy(ef_ferns* f) { delete f; }

int ef_ferns_get_table(const ef_ferns* f, int* t) {
  if (!f || !t) return EF_EINVAL;
  for (int i = 0; i < f->num; ++i) {
    const Fern& e = f->conservatory[i];
    t[i * 6] = e.x; t[i * 6 + 1] = e.y;
    for (int k = 0; k < 4; ++k) t[i * 6 + 2 + k] = e.rgbd[k];
  }
  return EF_OK;
}
int ef_ferns_set_table(ef_ferns* f, const int* t) {
  if (!f || !t) return EF_EINVAL;
  if (!f->frames.empty()) return EF_ESTATE;
  for (int i = 0; i < f->num; ++i)
    if (t[i * 6] < 0 || t[i * 6] >= f->width || t[i * 6 + 1] < 0 || t[i * 6 + 1] >= f->height) return EF_EINVAL;
  for (int i = 0; i < f->num; ++i) {
    Fern& e = f->conservatory[i];
    e.x = t[i * 6]; e.y = t[i * 6 + 1];
    for (int k = 0; k < 4; ++k) e.rgbd[k] = t[i * 6 + 2 + k];
  }
  ++f->tableVersion;
  return EF_OK;
}
int ef_ferns_table_version(const ef_ferns* f) { return f ? f->tableVersion : EF_EINVAL; }

}  // extern "C"

namespace {
// where a view comes from: handed over by the caller, or fetched on demand (a device read-back) the first time it is needed
struct ViewSource {
  View v{nullptr, 0, nullptr, 0};
  const float* norms = nullptr;
  ef_view_fetch fetch = nullptr;
  void* user = nullptr;
  bool have = false;
  bool get(int width) {
    if (have) return true;
    if (!fetch) return false;
    const uint8_t* rgb = nullptr;
    const float *verts = nullptr, *nrm = nullptr;
    int ch = 0;
    if (fetch(user, &rgb, &ch, &verts, &nrm) != EF_OK || !rgb || !verts || !nrm || (ch != 3 && ch != 4)) return false;
    v = View{rgb, ch, verts, width};
    norms = nrm;
    have = true;
    return true;
  }
};

// Ferns::addFrame from the codes on (Ferns.cpp:120-159)
int add_frame_core(ef_ferns* f, std::unique_ptr<StoredFrame>& frame, const std::vector<int>& co, ViewSource& src, const double* T_wc16, int src_time,
                   float threshold) {
  float minimum = std::numeric_limits<float>::max();
  if (frame->goodCodes > 0)
    for (size_t i = 0; i < f->frames.size(); ++i) {
      const float d = f->dissimilarity(frame->goodCodes, *f->frames[i], co[i]);
      if (d < minimum) minimum = d;
    }
  if (!((minimum > threshold || f->frames.empty()) && frame->goodCodes > 0)) return 0;
  if (!src.get(f->width)) return EF_EINVAL;   // the frame is kept: now its images are needed
  const size_t px = (size_t)f->width * f->height;
  frame->id = (int)f->frames.size();
  frame->T_wc = efl::se3_from_matrix(T_wc16);
  frame->srcTime = src_time;
  frame->rgb.resize(px * 3);
  for (size_t i = 0; i < px; ++i)
    for (int k = 0; k < 3; ++k) frame->rgb[i * 3 + k] = src.v.rgb[i * src.v.ch + k];
  frame->verts.assign(src.v.verts, src.v.verts + px * 4);
  frame->norms.assign(src.norms, src.norms + px * 4);
  for (int i = 0; i < f->num; ++i)
    if (frame->codes[i] != BAD_CODE) f->conservatory[i].ids[frame->codes[i]].push_back(frame->id);
  f->frames.push_back(std::move(frame));
  return 1;
}

// Ferns::findFrame from the codes on (Ferns.cpp:210-299)
int find_frame_core(ef_ferns* f, const std::vector<uint8_t>& codes, int good, const std::vector<int>& co, ViewSource& src, const double* T_wc16, int time,
                    int lost, ef_fern_tracker tracker, void* user, double* T_est16_out, double* cons, int max_cons, int* n_out) {
  float minimum = std::numeric_limits<float>::max();
  int minId = -1;
  for (size_t i = 0; i < f->frames.size(); ++i) {
    const float d = f->dissimilarity(good, *f->frames[i], co[i]);
    if (d < minimum && time - f->frames[i]->srcTime > 300) {   // Ferns.cpp:218: only frames seen a while ago can close a loop
      minimum = d;
      minId = (int)i;
    }
  }
  if (minId == -1 || !(f->blockHDAware(codes, f->frames[minId]->codes) > 0.3)) return -1;
  if (!tracker) return EF_EINVAL;
  if (!src.get(f->width)) return EF_EINVAL;   // a candidate passed the code gates: the registration needs the view itself
  const View& v = src.v;
  const StoredFrame& s = *f->frames[minId];
  double T_fern[16];
  efl::se3_matrix(s.T_wc, T_fern);
  std::memcpy(T_est16_out, T_fern, sizeof(T_fern));
  float icpError = 0, icpCount = 0;
  // rgbd.initICPModel(fern) / initICP(current) / getIncrementalTransformation(T, false, 100, false, false, false) (Ferns.cpp:243-258)
  tracker(user, s.verts.data(), s.norms.data(), T_fern, v.verts, src.norms, T_est16_out, &icpError, &icpCount);
  const efl::SE3 T_est = efl::se3_from_matrix(T_est16_out);
  const float photoError = f->photometricCheck(v, T_est, s);
  const int icpCountThresh = lost ? 1400 : 2400;
  if (icpError < 0.0003 && icpCount > icpCountThresh && photoError < f->photoThresh) {
    f->lastClosest = minId;
    int n = 0;
    const int step = f->num / 50;
    for (int i = 0; step > 0 && i < f->num; i += step) {
      const Fern& e = f->conservatory[i];
      if (!f->usable(v, e)) continue;
      if (cons && n < max_cons) {
        const float* p = v.v(e.x, e.y);
        const double h[4] = {p[0], p[1], p[2], 1.0};
        for (int r = 0; r < 3; ++r) {
          cons[n * 6 + r] = ((T_wc16[r * 4] * h[0] + T_wc16[r * 4 + 1] * h[1]) + T_wc16[r * 4 + 2] * h[2]) + T_wc16[r * 4 + 3] * h[3];
          cons[n * 6 + 3 + r] = ((T_est16_out[r * 4] * h[0] + T_est16_out[r * 4 + 1] * h[1]) + T_est16_out[r * 4 + 2] * h[2]) + T_est16_out[r * 4 + 3] * h[3];
        }
      }
      ++n;
    }
    if (n_out) *n_out = n;
  }
  return f->lastClosest;
}
bool codes_ok(const ef_ferns* f, const uint8_t* codes, int good) {
  if (!f || !codes || good < 0 || good > f->num) return false;
  int g = 0;
  for (int i = 0; i < f->num; ++i) {
    if (codes[i] != BAD_CODE && codes[i] > 15) return false;
    g += codes[i] != BAD_CODE;
  }
  return g == good;
}
}  // namespace

extern "C" {

int ef_ferns_add_frame(ef_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int src_time,
                       float threshold) {
  if (!f || !rgb || !verts4 || !norms4 || !T_wc16 || (ch != 3 && ch != 4)) return EF_EINVAL;
  ViewSource src;
  src.v = View{rgb, ch, verts4, f->width};
  src.norms = norms4;
  src.have = true;
  std::unique_ptr<StoredFrame> frame(new StoredFrame());
  std::vector<int> co;
  f->encode(src.v, frame->codes, frame->goodCodes, co);
  return add_frame_core(f, frame, co, src, T_wc16, src_time, threshold);
}
int ef_ferns_add_frame_coded(ef_ferns* f, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int src_time,
                             float threshold) {
  if (!f || !fetch || !T_wc16 || !codes_ok(f, codes, good_codes)) return EF_EINVAL;
  ViewSource src;
  src.fetch = fetch;
  src.user = fetch_user;
  std::unique_ptr<StoredFrame> frame(new StoredFrame());
  frame->codes.assign(codes, codes + f->num);
  frame->goodCodes = good_codes;
  std::vector<int> co;
  f->cooccur(frame->codes, co);
  return add_frame_core(f, frame, co, src, T_wc16, src_time, threshold);
}

int ef_ferns_find_frame(ef_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int time, int lost,
                        ef_fern_tracker tracker, void* user, double* T_est16_out, double* cons, int max_cons, int* n_out) {
  if (!f || !rgb || !verts4 || !norms4 || !T_wc16 || !T_est16_out || (ch != 3 && ch != 4)) return EF_EINVAL;
  f->lastClosest = -1;
  if (n_out) *n_out = 0;
  identity16(T_est16_out);                       // Sophus::SE3d T_wc_est; (Ferns.cpp:236)
  ViewSource src;
  src.v = View{rgb, ch, verts4, f->width};
  src.norms = norms4;
  src.have = true;
  std::vector<uint8_t> codes;
  std::vector<int> co;
  int good = 0;
  f->encode(src.v, codes, good, co);
  return find_frame_core(f, codes, good, co, src, T_wc16, time, lost, tracker, user, T_est16_out, cons, max_cons, n_out);
}
int ef_ferns_find_frame_coded(ef_ferns* f, const uint8_t* codes_in, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int time,
                              int lost, ef_fern_tracker tracker, void* user, double* T_est16_out, double* cons, int max_cons, int* n_out) {
  if (!f || !fetch || !T_wc16 || !T_est16_out || !codes_ok(f, codes_in, good_codes)) return EF_EINVAL;
  f->lastClosest = -1;
  if (n_out) *n_out = 0;
  identity16(T_est16_out);
  ViewSource src;
  src.fetch = fetch;
  src.user = fetch_user;
  const std::vector<uint8_t> codes(codes_in, codes_in + f->num);
  std::vector<int> co;
  f->cooccur(codes, co);
  return find_frame_core(f, codes, good_codes, co, src, T_wc16, time, lost, tracker, user, T_est16_out, cons, max_cons, n_out);
}
// can any stored frame close a loop at `time` (Ferns.cpp:218: only frames stored more than 300 ticks ago are candidates)?  0 => findFrame
// returns -1 whatever the view looks like, and a caller may skip bringing the view (or its codes) to the host: lastClosest is reset here
// as findFrame would have reset it
int ef_ferns_candidate_possible(ef_ferns* f, int time) {
  if (!f) return EF_EINVAL;
  for (const auto& s : f->frames)
    if (time - s->srcTime > 300) return 1;
  f->lastClosest = -1;   // what findFrame would have left (Ferns.cpp:169)
  return 0;
}

int ef_ferns_count(const ef_ferns* f) { return f ? (int)f->frames.size() : EF_EINVAL; }
int ef_ferns_last_closest(const ef_ferns* f) { return f ? f->lastClosest : EF_EINVAL; }

int ef_ferns_get_frame(const ef_ferns* f, int id, uint8_t* codes, int* good, int* src_time, double* T_wc16, uint8_t* rgb3, float* verts4, float* norms4) {
  if (!f || id < 0 || id >= (int)f->frames.size()) return EF_EINVAL;
  const StoredFrame& s = *f->frames[id];
  if (codes) std::memcpy(codes, s.codes.data(), s.codes.size());
  if (good) *good = s.goodCodes;
  if (src_time) *src_time = s.srcTime;
  if (T_wc16) efl::se3_matrix(s.T_wc, T_wc16);
  if (rgb3) std::memcpy(rgb3, s.rgb.data(), s.rgb.size());
  if (verts4) std::memcpy(verts4, s.verts.data(), s.verts.size() * sizeof(float));
  if (norms4) std::memcpy(norms4, s.norms.data(), s.norms.size() * sizeof(float));
  return EF_OK;
}
int ef_ferns_set_frame_pose(ef_ferns* f, int id, const double* T_wc16) {
  if (!f || !T_wc16 || id < 0 || id >= (int)f->frames.size()) return EF_EINVAL;
  f->frames[id]->T_wc = efl::se3_from_matrix(T_wc16);
  return EF_OK;
}
float ef_ferns_block_hd_aware(const ef_ferns* f, int a, int b) {
  if (!f || a < 0 || b < 0 || a >= (int)f->frames.size() || b >= (int)f->frames.size()) return -1.f;
  return f->blockHDAware(f->frames[a]->codes, f->frames[b]->codes);
}
float ef_ferns_photometric_check(const ef_ferns* f, const uint8_t* rgb, int ch, const float* verts4, const double* T_est16, int id) {
  if (!f || !rgb || !verts4 || !T_est16 || id < 0 || id >= (int)f->frames.size() || (ch != 3 && ch != 4)) return -1.f;
  const View v{rgb, ch, verts4, f->width};
  return f->photometricCheck(v, efl::se3_from_matrix(T_est16), *f->frames[id]);
}

}  // extern "C"

// ---- host-side state and decisions of the loop closures around the fern database (ElasticFusion.cpp:392-445, 511-526, 588-589,
// 609-618): which constraints reach which deformation graph, what an accepted closure changes.  No device work: the caller brings the
// 1/8-resolution views, the pose, the sampled graph nodes and (through ef_fern_tracker) the fern-to-view registration.
struct ef_closure {
  ef_ferns* ferns = nullptr;
  float fernThresh = 0.3095f;
  efd::Gates gates;   // ef_closure_set_gates; the reference's constants by default
  int deforms = 0, fernDeforms = 0;
  int64_t lastDeformTime = 0;                 // Deformation::lastDeformTime of the LOCAL deformation
  std::vector<efd::Constraint> relativeCons;  // ElasticFusion::relativeCons
  std::vector<double> trajectory;             // t_T_wc
  std::vector<int64_t> trajectoryTimes;
  std::vector<efd::Constraint> lastRows;      // what the last closure handed to the optimiser
  float lastError = 0, lastMeanConsErr = 0;

  void poses(bool withTrajectory, std::vector<double>& P, std::vector<int64_t>& t) const {
    const size_t nf = ferns->frames.size();
    P.resize(nf * 16);
    t.resize(nf);
    for (size_t i = 0; i < nf; ++i) { efl::se3_matrix(ferns->frames[i]->T_wc, &P[i * 16]); t[i] = ferns->frames[i]->srcTime; }
    if (withTrajectory) { P.insert(P.end(), trajectory.begin(), trajectory.end()); t.insert(t.end(), trajectoryTimes.begin(), trajectoryTimes.end()); }
  }
  void adopt(bool withTrajectory, const std::vector<double>& P) {
    const size_t nf = ferns->frames.size();
    for (size_t i = 0; i < nf; ++i) ferns->frames[i]->T_wc = efl::se3_from_matrix(&P[i * 16]);
    if (withTrajectory) std::copy(P.begin() + nf * 16, P.end(), trajectory.begin());
  }
};

extern "C" {

ef_closure* ef_closure_create(int num_ferns, float depth_cut, float photo_thresh, float fern_thresh, int width, int height, float fx, float fy, float cx,
                              float cy, unsigned seed) {
  ef_ferns* f = ef_ferns_create(num_ferns, (int)(depth_cut * 1000), photo_thresh, width, height, fx, fy, cx, cy, seed);   // ElasticFusion.cpp:53
  if (!f) return nullptr;
  ef_closure* c = new ef_closure();
  c->ferns = f;
  c->fernThresh = fern_thresh;
  return c;
}
void ef_closure_destroy(ef_closure* c) {
  if (!c) return;
  ef_ferns_destroy(c->ferns);
  delete c;
}
ef_ferns* ef_closure_ferns(ef_closure* c) { return c ? c->ferns : nullptr; }

// ElasticFusion.cpp:410-445 once Ferns::findFrame has answered (closest = its return value, cons / n its constraints)
static int closure_global_after_find(ef_closure* c, int closest, const double* cons, int n, int tick, const float* nodes4, int n_nodes, float* graph16_out,
                                     int* nodes_out) {
  if (closest < -1) return closest;
  if (closest == -1) return 0;                                                                     // :410
  const int64_t fernTime = c->ferns->frames[closest]->srcTime;
  std::vector<efd::Constraint>& rows = c->lastRows;
  for (int i = 0; i < n && i < 128; ++i) {                                                          // :415-422: addConstraint(src, target, tick, srcTime, pin = true)
    const double* s = cons + i * 6;
    rows.push_back(efd::Constraint{{s[0], s[1], s[2]}, {s[3], s[4], s[5]}, (uint64_t)tick, (uint64_t)fernTime, false, false});
    rows.push_back(efd::Constraint{{s[3], s[4], s[5]}, {s[3], s[4], s[5]}, (uint64_t)fernTime, (uint64_t)fernTime, false, true});
  }
  rows.insert(rows.end(), c->relativeCons.begin(), c->relativeCons.end());                         // :424-426
  std::vector<float> global;                                                                       // Deformation::sampleGraphFrom: every 5th local sample
  if (n_nodes / 5 > efd::K)
    for (int i = 0; i < n_nodes; i += 5) global.insert(global.end(), nodes4 + (size_t)i * 4, nodes4 + (size_t)i * 4 + 4);
  std::vector<double> P;
  std::vector<int64_t> t;
  c->poses(true, P, t);
  efd::Result r{false, 0, 0.f, 0.f};
  const int gn = (int)(global.size() / 4);
  const bool ok = efd::constrain(global.data(), gn, rows.data(), (int)rows.size(), true, 0, P.data(), t.data(), (int)t.size(), graph16_out, &r, nullptr,
                                 c->gates);   // :428
  c->lastError = r.error;
  c->lastMeanConsErr = r.meanConsErr;
  if (!ok) return 0;
  c->adopt(true, P);
  c->fernDeforms += gn > 0;                                                                        // :439
  *nodes_out = gn;
  return 1;
}
// ElasticFusion.cpp:392-445 (lost == false; ef_closure_relocalise is the other branch)
int ef_closure_global(ef_closure* c, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int tick,
                      ef_fern_tracker tracker, void* user, const float* nodes4, int n_nodes, double* T_recovery16_out, float* graph16_out, int* nodes_out) {
  if (!c || !T_recovery16_out || !graph16_out || !nodes_out || n_nodes < 0 || (n_nodes > 0 && !nodes4)) return EF_EINVAL;
  *nodes_out = 0;
  c->lastRows.clear();
  double cons[128 * 6];   // at most 99 constraints: every (num / 50)-th fern, Ferns.cpp:268
  int n = 0;
  const int closest = ef_ferns_find_frame(c->ferns, rgb, ch, verts4, norms4, T_wc16, tick, 0, tracker, user, T_recovery16_out, cons, 128, &n);   // :395-402
  return closure_global_after_find(c, closest, cons, n, tick, nodes4, n_nodes, graph16_out, nodes_out);
}
// the same with the view's fern codes computed by the caller (on the device) and the view itself fetched only if a keyframe passes the
// code gates
int ef_closure_global_coded(ef_closure* c, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int tick,
                            ef_fern_tracker tracker, void* user, const float* nodes4, int n_nodes, double* T_recovery16_out, float* graph16_out,
                            int* nodes_out) {
  if (!c || !T_recovery16_out || !graph16_out || !nodes_out || n_nodes < 0 || (n_nodes > 0 && !nodes4)) return EF_EINVAL;
  *nodes_out = 0;
  c->lastRows.clear();
  double cons[128 * 6];
  int n = 0;
  const int closest = ef_ferns_find_frame_coded(c->ferns, codes, good_codes, fetch, fetch_user, T_wc16, tick, 0, tracker, user, T_recovery16_out, cons, 128, &n);
  return closure_global_after_find(c, closest, cons, n, tick, nodes4, n_nodes, graph16_out, nodes_out);
}

// ElasticFusion.cpp:488-526 once the gates are open: constraints8 as ef_get_local_loop returns them (pin flag = deforms == 0 is the caller's)
int ef_closure_local(ef_closure* c, const double* constraints8, int n, int tick, const float* nodes4, int n_nodes, float* graph16_out, int* nodes_out) {
  if (!c || !constraints8 || !graph16_out || !nodes_out || n < 0 || n_nodes < 0 || (n_nodes > 0 && !nodes4)) return EF_EINVAL;
  *nodes_out = 0;
  std::vector<efd::Constraint>& rows = c->lastRows;
  rows.clear();
  for (int i = 0; i < n; ++i) {                                                                    // Deformation.cpp:73-86
    const double* q = constraints8 + (size_t)i * 8;
    rows.push_back(efd::Constraint{{q[0], q[1], q[2]}, {q[3], q[4], q[5]}, (uint64_t)tick, (uint64_t)q[6], false, false});
    if (q[7] != 0) rows.push_back(efd::Constraint{{q[3], q[4], q[5]}, {q[3], q[4], q[5]}, (uint64_t)q[6], (uint64_t)q[6], false, true});
  }
  std::vector<double> P;
  std::vector<int64_t> t;
  c->poses(false, P, t);
  std::vector<efd::Constraint> rel;
  efd::Result r{false, 0, 0.f, 0.f};
  const bool ok = efd::constrain(nodes4, n_nodes, rows.data(), (int)rows.size(), false, (uint64_t)c->lastDeformTime, P.data(), t.data(), (int)t.size(),
                                 graph16_out, &r, &rel);                                             // :513-514
  c->lastError = r.error;
  c->lastMeanConsErr = r.meanConsErr;
  if (!ok) return 0;
  c->adopt(false, P);
  c->lastDeformTime = tick;                                                                        // Deformation.cpp:199-201
  c->deforms += n_nodes > 0;                                                                       // :523
  for (size_t i = 0; rel.size() >= 3 && i < rel.size(); i += rel.size() / 3) c->relativeCons.push_back(rel[i]);   // :522-524
  *nodes_out = n_nodes;
  return 1;
}

// ElasticFusion.cpp:588-589 and 609-618: the frame's final pose joins the trajectory, the final fill-in view may become a keyframe
int ef_closure_end_frame(ef_closure* c, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int tick) {
  if (!c || !T_wc16) return EF_EINVAL;
  c->trajectory.insert(c->trajectory.end(), T_wc16, T_wc16 + 16);
  c->trajectoryTimes.push_back(tick);
  return ef_ferns_add_frame(c->ferns, rgb, ch, verts4, norms4, T_wc16, tick, c->fernThresh);
}

// Can this frame's Ferns::findFrame match at all (ef_ferns_candidate_possible)?  Answering 0 it leaves the closure object as the
// ef_closure_global / ef_closure_relocalise that was not needed would have: no rows, lastClosest = -1.
int ef_closure_candidate_possible(ef_closure* c, int tick) {
  if (!c) return EF_EINVAL;
  const int r = ef_ferns_candidate_possible(c->ferns, tick);
  if (r == 0) c->lastRows.clear();
  return r;
}
int ef_closure_end_frame_coded(ef_closure* c, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int tick) {
  if (!c || !T_wc16) return EF_EINVAL;
  c->trajectory.insert(c->trajectory.end(), T_wc16, T_wc16 + 16);
  c->trajectoryTimes.push_back(tick);
  return ef_ferns_add_frame_coded(c->ferns, codes, good_codes, fetch, fetch_user, T_wc16, tick, c->fernThresh);
}
int ef_closure_relocalise_coded(ef_closure* c, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int tick,
                                ef_fern_tracker tracker, void* user, double* T_recovery16_out) {
  if (!c || !T_recovery16_out) return EF_EINVAL;
  c->lastRows.clear();
  double cons[128 * 6];
  int n = 0;
  const int closest = ef_ferns_find_frame_coded(c->ferns, codes, good_codes, fetch, fetch_user, T_wc16, tick, 1, tracker, user, T_recovery16_out, cons, 128, &n);
  if (closest < -1) return closest;
  return closest == -1 ? 0 : 1;
}

// ElasticFusion.cpp:395-413 for a lost camera: the match itself is the answer (no deformation)
int ef_closure_relocalise(ef_closure* c, const uint8_t* rgb, int ch, const float* verts4, const float* norms4, const double* T_wc16, int tick,
                          ef_fern_tracker tracker, void* user, double* T_recovery16_out) {
  if (!c || !T_recovery16_out) return EF_EINVAL;
  c->lastRows.clear();
  double cons[128 * 6];
  int n = 0;
  const int closest = ef_ferns_find_frame(c->ferns, rgb, ch, verts4, norms4, T_wc16, tick, 1, tracker, user, T_recovery16_out, cons, 128, &n);
  if (closest < -1) return closest;
  return closest == -1 ? 0 : 1;
}
int ef_closure_log_pose(ef_closure* c, const double* T_wc16, int tick) {
  if (!c || !T_wc16) return EF_EINVAL;
  c->trajectory.insert(c->trajectory.end(), T_wc16, T_wc16 + 16);
  c->trajectoryTimes.push_back(tick);
  return EF_OK;
}

int ef_closure_set_fern_thresh(ef_closure* c, float fern_thresh) {
  if (!c) return EF_EINVAL;
  c->fernThresh = fern_thresh;
  return EF_OK;
}
int ef_closure_set_gates(ef_closure* c, float entry_mean_error, float accept_mean_error, float accept_energy) {
  if (!c) return EF_EINVAL;
  c->gates = efd::Gates{entry_mean_error, accept_mean_error, accept_energy};
  return EF_OK;
}
int ef_closure_counts(const ef_closure* c, int* deforms, int* fern_deforms, int* relative, int* trajectory) {
  if (!c) return EF_EINVAL;
  if (deforms) *deforms = c->deforms;
  if (fern_deforms) *fern_deforms = c->fernDeforms;
  if (relative) *relative = (int)c->relativeCons.size();
  if (trajectory) *trajectory = (int)c->trajectoryTimes.size();
  return EF_OK;
}
static void put_rows(const std::vector<efd::Constraint>& v, ef_graph_constraint* out, int max_rows) {
  for (size_t i = 0; i < v.size() && (int)i < max_rows; ++i) {
    for (int k = 0; k < 3; ++k) { out[i].src[k] = v[i].src[k]; out[i].target[k] = v[i].target[k]; }
    out[i].src_time = (int64_t)v[i].srcTime; out[i].target_time = (int64_t)v[i].targetTime; out[i].relative = v[i].relative; out[i].pin = v[i].pin;
  }
}
int ef_closure_last_rows(const ef_closure* c, ef_graph_constraint* rows, int max_rows, float* error, float* mean_constraint_error) {
  if (!c) return EF_EINVAL;
  if (rows) put_rows(c->lastRows, rows, max_rows);
  if (error) *error = c->lastError;
  if (mean_constraint_error) *mean_constraint_error = c->lastMeanConsErr;
  return (int)c->lastRows.size();
}
int ef_closure_relative(const ef_closure* c, ef_graph_constraint* rows, int max_rows) {
  if (!c) return EF_EINVAL;
  if (rows) put_rows(c->relativeCons, rows, max_rows);
  return (int)c->relativeCons.size();
}
int ef_closure_trajectory(const ef_closure* c, double* poses16, int max_poses) {
  if (!c) return EF_EINVAL;
  const int n = (int)c->trajectoryTimes.size();
  if (poses16) memcpy(poses16, c->trajectory.data(), (size_t)(n < max_poses ? n : max_poses) * 16 * sizeof(double));
  return n;
}

}  // extern "C"

