// TEST INFRASTRUCTURE — host execution of the product's k_model_maps (a quad of lanes per 4x4 block; the harness is built with
// -DEF_MODEL_MAPS_QUAD for its grid) and of the archived one-thread-per-block form of rounds 1-2 (model_maps_block.inc), compiled from
// the text tests/test_wave_emulation.py cuts out of elasticfusion_amd/csrc/ef_track_kernels.hip (MODEL_MAPS_SOURCE) over the stand-in <hip/hip_runtime.h> beside this file.  Each lane of a quad is a host thread.
#include <hip/hip_runtime.h>
#include <thread>
#include <vector>
#include "ef_device.hpp"
#include "ef_track.hpp"
namespace eft {
using namespace ef;
namespace {
#include MODEL_MAPS_SOURCE
}
}  // namespace eft

extern "C" int run_model_maps(const float* pred_vertex, const float* pred_normal, const float* fill_vertex, const float* fill_normal, int cols, int rows,
                              float maxDepthRGB, int camera_frame, const float* R9, const float* t3, unsigned dense_count, int dense_samples,
                              float* vmap0, float* vmap1, float* vmap2, float* nmap0, float* nmap1, float* nmap2, float* depth0) {
  using namespace eft;
  ModelMapsArgs A{};   // (no level-0 intensity image here: last0 = null)
  A.pred_vertex = (const float4*)pred_vertex; A.pred_normal = (const float4*)pred_normal;
  A.fill_vertex = (const float4*)fill_vertex; A.fill_normal = (const float4*)fill_normal;
  A.vmap[0] = vmap0; A.vmap[1] = vmap1; A.vmap[2] = vmap2;
  A.nmap[0] = nmap0; A.nmap[1] = nmap1; A.nmap[2] = nmap2;
  A.depth0 = depth0; A.cols = cols; A.rows = rows; A.maxDepthRGB = maxDepthRGB; A.camera_frame = camera_frame != 0;
  static TrackState st;
  for (int i = 0; i < 9; ++i) st.R_wc_f[i] = R9[i];
  for (int i = 0; i < 3; ++i) st.t_wc_f[i] = t3[i];
  st.dense_count = dense_count; st.dense_samples = dense_samples;
  const unsigned bdx = 64, bdy = 4;
#ifdef EF_MODEL_MAPS_QUAD
  const unsigned gx = (cols + 63) / 64, gy = (rows / 4 + 3) / 4;
#else
  const unsigned gx = (cols / 4 + 63) / 64, gy = (rows / 4 + 3) / 4;
#endif
  for (unsigned by = 0; by < gy; ++by)
    for (unsigned bx = 0; bx < gx; ++bx)
      for (unsigned ty = 0; ty < bdy; ++ty)
        for (unsigned q = 0; q < bdx / 4; ++q) {   // one quad of lanes at a time, its four lanes in lockstep at every DPP operation
          int slots[4];
          std::barrier<> bar(4);
          std::vector<std::thread> th;
          for (int j = 0; j < 4; ++j)
            th.emplace_back([&, j] {
              emu::Lane& L = emu::lane;
              L.tid = {q * 4 + (unsigned)j, ty, 0}; L.bid = {bx, by, 0}; L.bdim = {bdx, bdy, 1}; L.gdim = {gx, gy, 1};
              L.quad_lane = j; L.quad_slots = slots; L.quad_barrier = &bar;
              k_model_maps(A, &st);
            });
          for (auto& t : th) t.join();
        }
  return 0;
}
