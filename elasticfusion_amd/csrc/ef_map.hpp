// Map side of libefusion_hip: depth pre-processing and surfel-map maintenance as HIP compute.
// Replaces the reference's GLSL transform-feedback / FBO passes (Core/IndexMap.cpp, Core/GlobalModel.cpp,
// Core/Shaders/{ComputePack,FillIn,FeedbackBuffer,Resize}.cpp and their shaders): no GL, no interop.
//
// HBM layout: surfels are three float4 streams (SoA) {x,y,z,conf} {colour,0,initTime,lastTime}
// {nx,ny,nz,radius}: every per-surfel pass issues perfectly coalesced 16-byte loads, and passes that
// cull on position/time never touch the normal stream.  The 48-byte AoS records of the reference
// (Core/Shaders/Vertex.cpp:21-41) exist only at the download / upload boundary.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ef_track.hpp"

namespace efm {

struct Cam { int cols, rows; float fx, fy, cx, cy; };

struct SurfelSoA {
  float4* pos_conf;
  float4* col_time;
  float4* nrm_rad;
};

constexpr unsigned long long ZBUF_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr unsigned WINNER_EMPTY = 0xFFFFFFFFu;
constexpr int CHUNK = 1024;  // elements per compaction chunk (256 threads x 4) of the seeding / candidate passes
constexpr int CLEAN_ROW = 256; // elements per compaction chunk of clean(): one per thread

// per-pixel model prediction products
struct IndexMaps {   // IndexMap::predictIndices outputs (IndexMap.h:74-88)
  uint32_t* index;
  float4* vert_conf;
  float4* color_time;
  float4* norm_rad;
  // Storage order of the four images (and of the z-buffer they are resolved from).  The frame tier keeps them
  // COLUMN-major (texel (x, y) at x * rows + y): surfels are created in column-major pixel order (the reference's draw
  // order, FeedbackBuffer.cpp:44-52) and move little between frames, so consecutive surfel ids project to vertically
  // adjacent pixels — with this layout the per-surfel taps of clean() and the per-pixel gathers of the resolve pass
  // touch consecutive addresses instead of one cache line per lane.  The operator tier passes row-major host images.
  int colmajor = 0;
};
__host__ __device__ inline int im_texel(const IndexMaps& im, const Cam& cam, int px, int py) {
  return im.colmajor ? px * cam.rows + py : py * cam.cols + px;
}
struct PredictMaps { // IndexMap::combinedPredict outputs (IndexMap.h:98-112)
  uchar4* image;
  float4* vertex;
  float4* normal;
  uint16_t* time;
};
struct FillMaps {    // FillIn textures (FillIn.h)
  uchar4* image;
  float4* vertex;
  float4* normal;
};
// association products of one frame: one slot per fused pixel (W/2 x H/2, quirk Q12), in draw order
struct Candidates {
  float4* pos_conf;
  float4* col_time;   // .w tag: -1 matched, -2 new unstable, 0 not emitted
  float4* nrm_rad;
  uint32_t* best;     // surfel id chosen by the association (tag -1)
  int n;              // (W/2)*(H/2)
};
struct CompactScratch {
  uint8_t* flags;          // capacity + n_candidates (or 2 x pixels for seeding)
  uint32_t* chunk_count;   // per chunk
  uint32_t* chunk_offset;  // exclusive scan
  uint32_t* totals;        // small scratch (>= 4 u32)
  int max_chunks;
  // clean() without its scan launch (round 6): kept elements per GROUP of CLEAN_GROUP rows, added up by k_clean_flags (integer atomics: exact
  // in any order) and read by k_clean_scatter, whose workgroups find their row's offset themselves.  Two halves of max_groups words: a call
  // adds into half `flip` and zeroes the other one for the call after it (the owner flips before every clean()).  Null: the scan launch.
  // One sum per 128-byte line (CLEAN_GSTRIDE words apart): with the sums side by side every row's atomic of a frame queued on one or two lines
  // (k_clean_flags 16.9 -> 20.9 us, profiles/r08a_bench_kernel_stats.csv).
  uint32_t* group_sum = nullptr;
  int max_groups = 0;
  int flip = 0;
};
constexpr int CLEAN_GROUP = 32;     // rows per group of CompactScratch::group_sum
constexpr int CLEAN_GSTRIDE = 32;   // words between two group sums

// ---- pre-processing (ComputePack FILTER / METRIC / METRIC_FILTERED, ElasticFusion.cpp:655-673) ----
// the bilateral filter's weight table of the current device (built on first use, synchronously; null on a HIP error).  ef_create asks
// for it so that no later call — possibly inside a stream capture — is the first
const float* bilateral_table();
bool filter_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered, hipStream_t s);   // false: no weight table on this device
void metricise_depth(const uint16_t* in, int cols, int rows, float maxD, float* out, hipStream_t s);
// fused: bilateral + both metric conversions in one pass over the raw depth
// extra_lds: unused dynamic LDS bytes added to the launch = an occupancy cap for when the kernel shares the GPU
// rgb3 given: the frame's level-0 intensity image (next0) and, with rgb_keep, a copy of the colour image are written by the same launch
// table: bilateral_table()'s pointer when the caller holds it (null: asked here); returns false when the device has no table
bool preprocess_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered, float* metric,
                      float* metric_filtered, hipStream_t s, unsigned extra_lds = 0, const uint8_t* rgb3 = nullptr, uint8_t* next0 = nullptr,
                      uint8_t* rgb_keep = nullptr, const float* table = nullptr);

// ---- layout conversion at the API boundary ----
void aos_to_soa(const float* aos, uint32_t count, SurfelSoA soa, hipStream_t s);
void soa_to_aos(SurfelSoA soa, uint32_t count, float* aos, hipStream_t s);
// dst[0, *count_dev) = src[0, *count_dev): the full-map copy the reference's update pass makes into its second vertex buffer
// (GlobalModel.cpp:458-524); only used to reproduce GlobalModel::downloadMap's buffer choice (ef_set_reference_download)
void copy_map(SurfelSoA src, const unsigned* count_dev, SurfelSoA dst, hipStream_t s);

// ---- first frame (vertex_feedback x2 + init_unstable) ----
void seed_map(const Cam& cam, const uint8_t* rgb3, const float* depth_metric, const float* depth_metric_filtered, int time,
              float maxDepth, SurfelSoA out, unsigned* count_dev, const CompactScratch& cs, hipStream_t s);

// ---- model prediction ----
// T_cw16_dev: device pointer to the float 4x4 T_wc^-1; count_dev: device surfel count
void predict_indices(const Cam& cam, const float* T_cw16_dev, int time, SurfelSoA map, const unsigned* count_dev, float maxDepth,
                     int timeDelta, unsigned long long* zbuf, IndexMaps out, hipStream_t s, eft::KernelProbe* probe = nullptr,
                     // merge_cand / merge_winner given (behind fuse(..., defer_merge = true)): the update pass of the fusion rides on this splat
                     const struct Candidates* merge_cand = nullptr, const uint32_t* merge_winner = nullptr);
void combined_predict(const Cam& cam, const float* T_cw16_dev, SurfelSoA map, const unsigned* count_dev, float maxDepth,
                      float confThreshold, int time, int maxTime, int timeDelta, unsigned long long* zbuf, PredictMaps out,
                      // optional fused fill-in + denseEnough sampling (null fill.image => skipped)
                      FillMaps fill, const uint16_t* depth_filtered, const uint8_t* rgb3, bool passthroughImage,
                      unsigned* dense_counter, hipStream_t s,
                      // optional: *nonempty_flag = nonempty_value when the view shows at least one surfel (the caller stamps a fresh value per use)
                      unsigned* nonempty_flag = nullptr, unsigned nonempty_value = 0,
                      // optional: *consumed_mark = consumed_value (system scope, e.g. host-mapped memory) as soon as the splat launch starts, i.e. once
                      // everything enqueued before it has finished
                      unsigned* consumed_mark = nullptr, unsigned consumed_value = 0,
                      // optional: build_ray_table's table for this camera (cols x rows float4, column-major): the splat loads a fragment's ray instead of
                      // evaluating it — the same value
                      const float* rays4 = nullptr);
void build_ray_table(const Cam& cam, float* rays4, hipStream_t s);
// IndexMap::synthesizeDepth (splat.vert + depth_splat.frag): float depth of the nearest splat per pixel, 0 = none
void synthesize_depth(const Cam& cam, const float* T_cw16_dev, SurfelSoA map, const unsigned* count_dev, float maxDepth,
                      float confThreshold, int time, int maxTime, int timeDelta, unsigned long long* zbuf, float* depth, hipStream_t s,
                      const float* rays4 = nullptr);
void fill_in(const Cam& cam, PredictMaps pred, const uint16_t* depth_filtered, const uint8_t* rgb3, bool passthrough,
             bool passthroughImage, FillMaps out, hipStream_t s);
// counts the (W/20)x(H/20) sample texels with r,g,b > 0 into *counter (Resize::image + denseEnough)
void dense_count(const Cam& cam, const uchar4* image, unsigned* counter, hipStream_t s);

// ---- fusion ----
// pose_f16_dev: float T_wc (cast<float>().matrix()); weighting_dev: device float
void fuse(const Cam& cam, const float* pose_f16_dev, int time, const uint8_t* rgb3, const float* depth_metric,
          const float* depth_metric_filtered, IndexMaps im, float maxDepth, const float* weighting_dev, SurfelSoA map,
          const unsigned* count_dev, Candidates cand, uint32_t* winner, hipStream_t s, bool defer_merge = false);
// deformation graph handed to clean() after a loop closure (copy_unstable.vert:128-322): nodes x 16 floats sorted by time
// {position 3, rotation 9 column-major, translation 3, time}; depth = synthesize_depth image (read unless is_fern)
struct Deformation {
  const float* graph_dev;
  int nodes;
  const float* depth_dev;
  int is_fern;
  float max_depth;
};
// clean + append; writes the compacted map to `out` and the new count (clamped to capacity) to *count_out_dev
void clean(const Cam& cam, const float* T_cw16_dev, int time, IndexMaps im, float confThreshold, int timeDelta, SurfelSoA map,
           const unsigned* count_dev, Candidates cand, uint32_t* winner, SurfelSoA out, unsigned* count_out_dev, uint32_t capacity,
           const CompactScratch& cs, int* overflow_flag, hipStream_t s, const Deformation* deform = nullptr);
// Deformation::sampleGraphModel: nodes {x, y, z, initTime} = every `stride`-th surfel (5000 in the reference); *n_out = node count
void sample_graph(SurfelSoA map, const unsigned* count_dev, int stride, int max_nodes, float* out4, unsigned* n_out, hipStream_t s);
// candidates -> AoS "newUnstable" list in draw order (operator tier / tests)
void candidates_to_aos(Candidates cand, float* aos, unsigned* count_dev, const CompactScratch& cs, hipStream_t s);

}  // namespace efm
