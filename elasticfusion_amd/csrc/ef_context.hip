// libefusion_hip: context, per-frame driver and the C ABI of include/ef_hip.h.
// The frame script follows ElasticFusion::processFrame (Core/ElasticFusion.cpp:270-607) for the open-loop
// configuration; every stage is only ENQUEUED on the context's stream — pose, surfel count, fill-in
// decision and fusion weight all live in a device-resident state block, so a frame costs zero
// host<->device round trips (the reference has ~70 blocking ones).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ef_hip.h"
#include "ef_linalg_dev.hpp"
#include "ef_solve_dev.hpp"
#include "ef_map.hpp"
#include "ef_deform_solver.hpp"
#include "ef_track.hpp"

namespace {

thread_local std::string g_create_error;

struct StageTimer {
  const char* name;
  hipEvent_t a, b;
  bool used;
};

}  // namespace

struct ef_ctx {
  ef_config cfg;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  efm::Cam cam;
  eft::Intr intr;
  const float maxDepthProcessed = 20.0f;  // ElasticFusion.cpp:83
  // frame images
  uint8_t* rgb = nullptr;
  uint16_t* depth_raw = nullptr;
  uint16_t* depth_filtered = nullptr;
  float* depth_metric = nullptr;
  float* depth_metric_filtered = nullptr;
  // second set of the five frame images: frame k+1's input stage (copy, bilateral filter, frame pyramids) runs on
  // in_stream while frame k is still being fused on `stream`, so the two frames must not share these buffers
  uint8_t* rgb_alt = nullptr;
  uint16_t* depth_raw_alt = nullptr;
  uint16_t* depth_filtered_alt = nullptr;
  float* depth_metric_alt = nullptr;
  float* depth_metric_filtered_alt = nullptr;
  hipStream_t in_stream = nullptr;
  hipEvent_t ev_input_done = nullptr, ev_track_done = nullptr, ev_staged = nullptr;
  hipEvent_t ev_frame_done[2] = {nullptr, nullptr};   // end of the frame that last used each set of frame images
  int frame_parity = 0;
  int overlap_mode = 1;          // ef_set_input_overlap: 1 = whole input stage after the previous tracker; 2 = copy + bilateral filter already during it
  bool overlap = false;
  bool events_live = false;      // the previous frame recorded ev_track_done (and, in mode 2, ev_frame_done): see process_frame
  // host-pointer frames (ef_process_frame, the reference's processFrame signature): a ring of pinned staging pairs and device landing pairs; the
  // upload runs on copy_stream, the frame script reads the landing pair in place (round 6)
  static constexpr int RING = 3;
  uint8_t* h_rgb_ring[RING] = {};
  uint16_t* h_depth_ring[RING] = {};
  uint8_t* d_rgb_ring[RING] = {};
  uint16_t* d_depth_ring[RING] = {};
  hipEvent_t ev_h2d[RING] = {};
  hipStream_t copy_stream = nullptr;
  unsigned host_seq = 0;           // host-pointer frames submitted so far
  unsigned mark_value = 0;         // what this frame's prediction writes into *h_consumed (host_seq + 1; 0: nothing)
  unsigned* h_consumed = nullptr;  // host-mapped: 1 + the index of the last host-pointer frame whose input images the GPU has finished reading
  unsigned* d_consumed = nullptr;
  unsigned* h_abort = nullptr;   // 4 words of host-mapped pinned memory: k_track_end copies a tracker instance's sticky abort flag here (d_abort: the device alias)
  unsigned* d_abort = nullptr;
  // tracker
  eft::Pyramid pyr{};
  eft::TrackState* st = nullptr;
  // model prediction
  efm::IndexMaps im{};
  efm::PredictMaps pm{};
  efm::FillMaps fm{};
  unsigned long long* zbuf = nullptr;
  // global model
  efm::SurfelSoA maps[2]{};
  int cur = 0;
  // ef_set_reference_download: the reference's vbos[renderSource] after a frame (GlobalModel.cpp:521,667,693: what its update pass
  // wrote, never overwritten by clean) — the buffer GlobalModel::downloadMap and savePly actually read (quirk Q14)
  efm::SurfelSoA shadow{};
  bool reference_download = false;
  uint32_t capacity = 0;
  uint32_t* winner = nullptr;
  efm::Candidates cand{};
  efm::CompactScratch cs{};
  int* overflow = nullptr;
  // trajectory (device log + host timestamps)
  double* traj = nullptr;  // 16 doubles per frame
  int traj_cap = 0;
  std::vector<int64_t> stamps;
  int tick = 1;
  std::vector<void*> allocs;
  // denseEnough()'s tally (ElasticFusion.cpp:256-268): taken by the model-map workgroups of the next tracked frame from the predicted image
  // (eft::ModelMapsArgs::tally_image) instead of one atomic per sample from the prediction's resolve pass
  // (set at ef_create: up to 1 024 samples — 640 x 480 has 768: the resolve pass 13.5 -> 10.3 us, 2072 -> 2091 frames/s; at 1280 x 960, 3 072
  // samples, every model-map workgroup reading them all costs more than the atomics, which hide behind that size's 139 MB: 1029 against 1034,
  // profiles/r08o_*, r08p_*.  -DEF_RESOLVE_TALLY, the A/B build "resolvetally": never)
  bool tally_by_consumer = false;
  bool tally_pending = false;
  // timing
  bool timing = false;
  std::vector<StageTimer> timers;
  // deformation graph to be applied by the next frame's clean() (ef_set_deformation): the device half of loop closure
  float* graph_dev = nullptr;
  int graph_nodes = 0, graph_is_fern = 0;
  float* synth_depth = nullptr;
  float* rays = nullptr;           // efm::build_ray_table: the ray of every pixel (float4, column-major), for the surface splat
  // local loop closure, front half (ElasticFusion.cpp:447-527): a second tracker instance registers the view of the INACTIVE
  // part of the model against the ACTIVE one; buffers exist only when cfg.close_loops is set
  int icp_count_thresh = 35000;            // ElasticFusion.h:44-46
  float icp_err_thresh = 5e-05f, cov_thresh = 1e-05f;
  int deforms = 0;
  const float* bil_table = nullptr;        // efm::bilateral_table() of this context's device, asked once at ef_create
  eft::Pyramid pyr2{};                     // RGBDOdometry modelToModel (ElasticFusion.h:280)
  eft::TrackState* st2 = nullptr;
  efm::PredictMaps old{};                  // IndexMap's oldImage/oldVertex/oldNormal/oldTime textures (IndexMap.h:114-128)
  float* cons_dev = nullptr;               // (W/20) x (H/20) x {x, y, z, inactive time}
  float* h_cons = nullptr;                 // pinned
  eft::TrackState* h_states = nullptr;     // pinned: [0] frame-to-model, [1] model-to-model / fern tracker, [2] frame-to-model at the end of the frame
  ef_loop_solver solver = nullptr;
  void* solver_user = nullptr;
  bool builtin_solver = false;             // ef_use_builtin_loop_solver: efd::solve_local where Deformation::constrain stands
  int64_t last_deform_time = 0;            // Deformation::lastDeformTime (Deformation.cpp:31,199-201)
  float* nodes_dev = nullptr;              // sampled graph nodes (Deformation::sampleGraphModel), 1024 x 4 + count
  std::vector<float> h_nodes;
  ef_local_loop loop{};
  std::vector<double> loop_constraints;    // n x 8
  std::vector<float> loop_graph;
  // global loop closure (ElasticFusion.cpp:392-445, 588-589, 609-618; ef_enable_global_closure): the host-side closure object (fern
  // database, relative constraints, trajectory; ef_ferns.hip), the 1/8-resolution fill-in views it works on, and a third tracker
  // instance at 1/8 resolution for the fern-to-view registration (Ferns.cpp:243-258: RGBDOdometry rgbd(w / 8, h / 8, ...))
  ef_closure* closure = nullptr;
  ef_global_loop gloop{};
  std::string fern_tracker_error;          // first HIP error inside the fern tracker callback (it cannot return one)
  int fern_w = 0, fern_h = 0;
  uchar4* view_img_dev = nullptr;          // Resize::image / vertex (x2) of the fill-in maps, factor 8
  float4* view_vert_dev = nullptr;
  float4* view_norm_dev = nullptr;
  uint8_t* h_view = nullptr;               // pinned: image | vertices | normals
  eft::Pyramid pyr3{};
  eft::TrackState* st3 = nullptr;
  eft::Intr intr3{};
  float4* fern_maps_dev = nullptr;         // fern vertices | fern normals | view vertices | view normals (+ a zero image)
  float* h_nodes_pinned = nullptr;         // graph nodes sampled at the end of the previous frame (Deformation::sampleGraphModel, :593)
  // fern coding on the device (k_fern_codes): the table, the codes of the view just coded (num bytes, padded to 512, + their count),
  // pinned landing zones for the mid-frame codes and for the END-of-frame record (codes, view, pose, nodes), which is only looked at
  // at the next frame's first synchronisation (Ferns::addFrame's verdict matters to nobody before the next findFrame)
  int* fern_table_dev = nullptr;
  int fern_num = 0, fern_table_version = -1;
  uint8_t* fern_codes_dev = nullptr;       // FERN_CODES_BYTES
  uint8_t* h_codes = nullptr;              // pinned, mid-frame
  uint8_t* h_codes_end = nullptr;          // pinned, end of frame
  uint8_t* h_view_end = nullptr;           // pinned, end of frame: image | vertices | normals
  hipEvent_t ev_end_record = nullptr;
  bool end_pending = false, end_lost = false;
  int end_tick = 0;
  int n_nodes_host = 0;
  double h_pose[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  // relocalisation (ef_set_relocalisation; ElasticFusion.h:283-286,311-312)
  bool reloc = false, lost = false, last_frame_recovery = false, tracking_ok = true;
  int tracking_count = 0;
  eft::TrackState h_reloc{};               // the frame-to-model tracker's state as read back for the verdict
  // hipGraph replay of the tracker (BASELINE.json configs[4]): the ~70 launches of getIncrementalTransformation are
  // captured once per pyramid parity (the SO(3) reference / frame intensity buffers swap every frame) and replayed
  bool use_graph = false;
  bool track_only = false;       // ef_set_track_only: odometry on a frozen map (BASELINE.json configs[4])
  bool no_resident = false;      // ef_set_resident_levels(ctx, 0): the persistent tracker launch streams every level's pixel data (round 5's kernel; A/B)
  bool fused_step = false;       // ef_set_fused_step: level-0 update step inside the correspondence-search launch
  int persistent = 1;            // ef_set_persistent_tracker: 1 = the whole tracker as one persistent launch of 256 workgroups, 0 = one launch per step,
                                 // 2 = (reference-order builds) round 3's launch of the small levels on 128 workgroups
  struct TrackGraph { hipGraphExec_t exec = nullptr; const void* key = nullptr; eft::TrackParams tp{}; eft::TrackTail tail{}; };
  TrackGraph tgraph[2];
  // HIP-event sampling of the dominant kernel (ef_kernel_timing)
  int ktime_every = 0;
  std::vector<hipEvent_t> kt_start, kt_stop;
  eft::KernelProbe probe{nullptr, nullptr, 0, 0};
  // second sampled kernel: the IndexMap point splat (k_index_splat of the first predictIndices of a frame)
  std::vector<hipEvent_t> ks_start, ks_stop;
  eft::KernelProbe probe_splat{nullptr, nullptr, 0, 0};
  std::vector<hipEvent_t> ka_start, ka_stop;   // the persistent tracker launch (fast order)
  eft::KernelProbe probe_all{nullptr, nullptr, 0, 0};
  hipStream_t debug_stream = nullptr;          // ef_debug_occupy
};

namespace {

#define EF_HIP(ctx, expr)                                                                       \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                           \
      return EF_EHIP;                                                                           \
    }                                                                                           \
  } while (0)

// Every entry point that takes a context runs on the CONTEXT's device, whatever device is current on the calling thread
// (two contexts on two GPUs in one process, or a context used from a thread that never called hipSetDevice), and
// leaves the thread's current device as it found it.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const ef_ctx* c) {
    if (c && hipGetDevice(&prev) == hipSuccess && prev != c->cfg.device) switched = hipSetDevice(c->cfg.device) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

template <typename T>
int dev_alloc(ef_ctx* c, T** p, size_t n, int fill = 0) {
  void* q = nullptr;
  hipError_t e = hipMalloc(&q, n * sizeof(T));
  if (e != hipSuccess) {
    c->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return EF_ENOMEM;
  }
  e = hipMemsetAsync(q, fill, n * sizeof(T), c->stream);
  if (e != hipSuccess) {
    c->err = std::string("hipMemset: ") + hipGetErrorString(e);
    return EF_EHIP;
  }
  c->allocs.push_back(q);
  *p = (T*)q;
  return EF_OK;
}
#define EF_ALLOC(c, p, n, ...)                                   \
  do {                                                           \
    int _r = dev_alloc((c), &(p), (size_t)(n), ##__VA_ARGS__);   \
    if (_r != EF_OK) return _r;                                  \
  } while (0)

// scalar per-frame bookkeeping kernels -----------------------------------------------------------
__global__ void k_init_state(eft::TrackState* st, int dense_samples, int pixels) {
  if (threadIdx.x != 0) return;
  st->dense_count = 0;
  st->dense_samples = dense_samples;
  st->map_counts[0] = st->map_counts[1] = 0;
  // RGBDOdometry's constructor (RGBDOdometry.cpp:31-36): errors 0, counts width * height until a step overwrites them
  st->lastICPError = st->lastRGBError = st->lastSO3Error = 0.f;
  st->lastICPCount = st->lastRGBCount = st->lastSO3Count = (float)pixels;
}
// API boundary: the frame tier's column-major index maps are handed out in the reference's row-major order
template <typename T>
__global__ void k_to_rowmajor(const T* __restrict__ src, int cols, int rows, T* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols * rows) return;
  const int y = i / cols, x = i - y * cols;
  dst[i] = src[x * rows + y];
}
// Resize::{image,vertex,time} (Resize.cpp:50-159): NEAREST downsample, destination (a, b) <- source texel (f a + f/2, f b + f/2)
template <typename T>
__global__ void k_resize_nearest(const T* __restrict__ src, int cols, int dw, int dh, int factor, T* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= dw * dh) return;
  const int b = i / dw, a = i - b * dw;
  dst[i] = src[(size_t)(b * factor + factor / 2) * cols + (a * factor + factor / 2)];
}
// Ferns.cpp:97-118 / :186-208 on the device: the 4-bit code of every fern on the view Resize::image / Resize::vertex would produce
// (NEAREST, texel (f x + f / 2, f y + f / 2) of the full-resolution fill-in maps), 255 where the depth is not positive, and the number
// of valid codes behind them.  One workgroup of FERN_CODES_PAD threads; table6 rows {x, y, r, g, b, d}.
constexpr int FERN_CODES_PAD = 512, FERN_CODES_BYTES = FERN_CODES_PAD + 16;
__global__ void __launch_bounds__(FERN_CODES_PAD) k_fern_codes(const uchar4* __restrict__ image, const float4* __restrict__ vertex, int cols, int factor,
                                                              const int* __restrict__ table6, int num, uint8_t* __restrict__ codes) {
  __shared__ int wsum[FERN_CODES_PAD / 64];
  const int i = threadIdx.x;
  uint8_t code = 255;
  int good = 0;
  if (i < num) {
    const int x = table6[i * 6], y = table6[i * 6 + 1];
    const size_t texel = (size_t)(y * factor + factor / 2) * cols + (x * factor + factor / 2);
    const float z = vertex[texel].z;
    if (z > 0) {
      const uchar4 p = image[texel];
      code = (uint8_t)((p.x > table6[i * 6 + 2]) << 3 | (p.y > table6[i * 6 + 3]) << 2 | (p.z > table6[i * 6 + 4]) << 1 | ((int)(z * 1000.0f) > table6[i * 6 + 5]));
      good = 1;
    }
  }
  codes[i] = code;
  for (int off = 32; off > 0; off >>= 1) good += __shfl_down(good, off, 64);
  if ((i & 63) == 0) wsum[i >> 6] = good;
  __syncthreads();
  if (i == 0) {
    int g = 0;
    for (int w = 0; w < FERN_CODES_PAD / 64; ++w) g += wsum[w];
    *(int*)(codes + FERN_CODES_PAD) = g;
  }
}
__global__ void k_set_count(unsigned* count_dev, unsigned v) {
  if (threadIdx.x == 0) *count_dev = v;
}

void timer_begin(ef_ctx* c, const char* name) {
  if (!c->timing) return;
  for (auto& t : c->timers)
    if (t.name == name || !strcmp(t.name, name)) { (void)hipEventRecord(t.a, c->stream); t.used = true; return; }
  StageTimer t{name, nullptr, nullptr, true};
  (void)hipEventCreate(&t.a);
  (void)hipEventCreate(&t.b);
  (void)hipEventRecord(t.a, c->stream);
  c->timers.push_back(t);
}
void timer_end(ef_ctx* c, const char* name) {
  if (!c->timing) return;
  for (auto& t : c->timers)
    if (!strcmp(t.name, name)) { (void)hipEventRecord(t.b, c->stream); return; }
}

int grow_trajectory(ef_ctx* c) {
  double* bigger = nullptr;
  EF_HIP(c, hipStreamSynchronize(c->stream));
  hipError_t e = hipMalloc((void**)&bigger, (size_t)c->traj_cap * 2 * 16 * sizeof(double));
  if (e != hipSuccess) { c->err = std::string("hipMalloc (trajectory log): ") + hipGetErrorString(e); return EF_ENOMEM; }
  e = hipMemcpy(bigger, c->traj, (size_t)c->traj_cap * 16 * sizeof(double), hipMemcpyDeviceToDevice);
  if (e != hipSuccess) { (void)hipFree(bigger); c->err = std::string("hipMemcpy (trajectory log): ") + hipGetErrorString(e); return EF_EHIP; }
  for (auto& p : c->allocs)
    if (p == (void*)c->traj) p = bigger;
  (void)hipFree(c->traj);
  c->traj = bigger;
  c->traj_cap *= 2;
  return EF_OK;
}

int do_predict(ef_ctx* c, bool count_dense = true) {
  // ElasticFusion::predict(), ElasticFusion.cpp:621-653: combinedPredict(ACTIVE) + FillIn (fused into the resolve).  Right after a
  // relocalisation the whole model is rendered (time = 0: no surfel is too old); while the camera is lost the fill-in passes the raw
  // frame through (a second, plain fill-in pass over the fused one: the rare path)
  // (a host-pointer frame: this launch, behind every reader of the frame's landing buffers, tells the host that their ring slot is free again)
  efm::combined_predict(c->cam, c->st->T_cw, c->maps[c->cur], &c->st->map_counts[c->cur], c->maxDepthProcessed, c->cfg.confidence,
                        c->last_frame_recovery ? 0 : c->tick, c->tick, c->cfg.time_delta, c->zbuf, c->pm, c->fm, c->depth_filtered, c->rgb,
                        c->cfg.frame_to_frame_rgb != 0, (count_dense && !c->tally_by_consumer) ? &c->st->dense_count : nullptr, c->stream, nullptr, 0u,
                        (count_dense && c->mark_value) ? c->d_consumed : nullptr, c->mark_value, c->rays);
  if (count_dense) c->mark_value = 0;
  if (count_dense && c->tally_by_consumer) c->tally_pending = true;   // (the next tracked frame's model maps count the samples of this prediction)
  if (c->lost) efm::fill_in(c->cam, c->pm, c->depth_filtered, c->rgb, true, true, c->fm, c->stream);
  return EF_OK;
}

// RGBDOdometry::getCovariance of the frame-to-model tracker against the gate of ElasticFusion.cpp:330-337,348-355
bool reloc_covariance_ok(const eft::TrackState& h) {
  double cov[36];
  efl::lu_inverse<double, 6>(h.lastA, cov);
  for (int i = 0; i < 6; ++i)
    if (cov[i * 6 + i] > 1e-04) return false;
  return true;
}

// The 1/8-resolution views of the fill-in maps (Ferns.cpp:91-93,178-180: Resize::image / Resize::vertex x2) into a pinned buffer
// (image | vertices | normals); enqueued only, the caller synchronises
int enqueue_fern_view(ef_ctx* c, uint8_t* h_dst) {
  hipStream_t s = c->stream;
  const int W = c->cam.cols, dw = c->fern_w, dh = c->fern_h, n = dw * dh;
  const dim3 g((unsigned)((n + 255) / 256));
  hipLaunchKernelGGL(k_resize_nearest<uint32_t>, g, dim3(256), 0, s, (const uint32_t*)c->fm.image, W, dw, dh, 8, (uint32_t*)c->view_img_dev);
  hipLaunchKernelGGL(k_resize_nearest<float4>, g, dim3(256), 0, s, (const float4*)c->fm.vertex, W, dw, dh, 8, c->view_vert_dev);
  hipLaunchKernelGGL(k_resize_nearest<float4>, g, dim3(256), 0, s, (const float4*)c->fm.normal, W, dw, dh, 8, c->view_norm_dev);
  EF_HIP(c, hipMemcpyAsync(h_dst, c->view_img_dev, (size_t)n * 4, hipMemcpyDeviceToHost, s));
  EF_HIP(c, hipMemcpyAsync(h_dst + (size_t)n * 4, c->view_vert_dev, (size_t)n * 16, hipMemcpyDeviceToHost, s));
  EF_HIP(c, hipMemcpyAsync(h_dst + (size_t)n * 20, c->view_norm_dev, (size_t)n * 16, hipMemcpyDeviceToHost, s));
  return EF_OK;
}
// the fern codes of the current fill-in maps (k_fern_codes) into a pinned buffer; enqueued only
int enqueue_fern_codes(ef_ctx* c, uint8_t* h_dst) {
  hipStream_t s = c->stream;
  ef_ferns* F = ef_closure_ferns(c->closure);
  if (ef_ferns_table_version(F) != c->fern_table_version) {   // first use, or ef_ferns_set_table since: (rare) synchronous upload
    std::vector<int> t((size_t)c->fern_num * 6);
    if (ef_ferns_get_table(F, t.data()) != EF_OK) { c->err = "ef_ferns_get_table failed"; return EF_EINVAL; }
    EF_HIP(c, hipStreamSynchronize(s));
    EF_HIP(c, hipMemcpy(c->fern_table_dev, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
    c->fern_table_version = ef_ferns_table_version(F);
  }
  hipLaunchKernelGGL(k_fern_codes, dim3(1), dim3(FERN_CODES_PAD), 0, s, (const uchar4*)c->fm.image, (const float4*)c->fm.vertex, c->cam.cols, 8,
                     (const int*)c->fern_table_dev, c->fern_num, c->fern_codes_dev);
  EF_HIP(c, hipMemcpyAsync(h_dst, c->fern_codes_dev, FERN_CODES_BYTES, hipMemcpyDeviceToHost, s));
  return EF_OK;
}
void pose_of_state(const eft::TrackState& h, double* T16) {
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = h.q[i];
  for (int i = 0; i < 3; ++i) T.t[i] = h.t[i];
  efl::se3_matrix(T, T16);
}
// ef_view_fetch of the mid-frame view: Ferns::findFrame asks for it only when a keyframe passed the code gates (one more synchronisation,
// in those frames only); the fill-in maps still hold the mid-frame prediction
int fetch_mid_view(void* user, const uint8_t** rgb, int* ch, const float** verts, const float** norms) {
  ef_ctx* c = (ef_ctx*)user;
  const int r = enqueue_fern_view(c, c->h_view);
  if (r != EF_OK) return r;
  EF_HIP(c, hipStreamSynchronize(c->stream));
  const size_t n = (size_t)c->fern_w * c->fern_h;
  *rgb = c->h_view; *ch = 4; *verts = (const float*)(c->h_view + n * 4); *norms = (const float*)(c->h_view + n * 20);
  return EF_OK;
}
// ... and of the end-of-frame view, which was copied with the end-of-frame record
int fetch_end_view(void* user, const uint8_t** rgb, int* ch, const float** verts, const float** norms) {
  ef_ctx* c = (ef_ctx*)user;
  const size_t n = (size_t)c->fern_w * c->fern_h;
  *rgb = c->h_view_end; *ch = 4; *verts = (const float*)(c->h_view_end + n * 4); *norms = (const float*)(c->h_view_end + n * 20);
  return EF_OK;
}
// End of a frame (ElasticFusion.cpp:588-589, 593, 609-618) — ENQUEUED: fern codes and 1/8 view of the final fill-in maps, the pose, a
// fresh sample of the graph nodes, all into pinned memory behind one event.  Nothing waits for them here.
int enqueue_end_record(ef_ctx* c) {
  hipStream_t s = c->stream;
  int r = enqueue_fern_codes(c, c->h_codes_end);
  if (r != EF_OK) return r;
  if (!c->lost) {   // a lost camera stores no keyframe (:601-604): its view is never asked for
    r = enqueue_fern_view(c, c->h_view_end);
    if (r != EF_OK) return r;
  }
  EF_HIP(c, hipMemcpyAsync(&c->h_states[2], c->st, sizeof(eft::TrackState), hipMemcpyDeviceToHost, s));
  unsigned* n_dev = (unsigned*)(c->nodes_dev + (size_t)1024 * 4);   // Deformation::sampleGraphModel (:593): every 5000th surfel of the new map
  efm::sample_graph(c->maps[c->cur], &c->st->map_counts[c->cur], 5000, 1023, c->nodes_dev, n_dev, s);
  EF_HIP(c, hipMemcpyAsync(c->h_nodes_pinned, c->nodes_dev, ((size_t)1024 * 4 + 1) * sizeof(float), hipMemcpyDeviceToHost, s));
  EF_HIP(c, hipEventRecord(c->ev_end_record, s));
  c->end_pending = true;
  c->end_lost = c->lost;
  c->end_tick = c->tick;
  return EF_OK;
}
// ... and looked at: pose -> trajectory, codes (+ view, if the frame is kept) -> Ferns::addFrame, the node count.  Called at the next
// point where the host waits for the stream anyway (the next frame's closures) and by every getter that shows closure state.
int flush_end_record(ef_ctx* c) {
  if (!c->closure || !c->end_pending) return EF_OK;
  EF_HIP(c, hipEventSynchronize(c->ev_end_record));
  c->end_pending = false;
  double T[16];
  pose_of_state(c->h_states[2], T);
  int good = 0;
  memcpy(&good, c->h_codes_end + FERN_CODES_PAD, sizeof(int));
  const int r = c->end_lost ? ef_closure_log_pose(c->closure, T, c->end_tick)
                            : ef_closure_end_frame_coded(c->closure, c->h_codes_end, good, &fetch_end_view, c, T, c->end_tick);
  unsigned nn = 0;
  memcpy(&nn, c->h_nodes_pinned + (size_t)1024 * 4, sizeof(unsigned));
  c->n_nodes_host = (int)nn;
  if (r < 0) { c->err = "ef_closure_end_frame failed"; return r; }
  return EF_OK;
}

// Ferns.cpp:243-258 on the device: the stored keyframe is the model (initICPModel with its pose), the current view the frame
// (initICP(vertices, normals)); getIncrementalTransformation(T, rgbOnly = false, icpWeight = 100, pyramid = false, fastOdom = false,
// so3 = false) = ten ICP-only iterations at the 1/8 resolution itself.  One synchronisation (pose + statistics back).
void fern_tracker_device(void* user, const float* fv, const float* fn, const double* Tf, const float* cv, const float* cn, double* T_io, float* err,
                         float* cnt) {
  ef_ctx* c = (ef_ctx*)user;
  hipStream_t s = c->stream;
  const size_t n = (size_t)c->fern_w * c->fern_h;
  float4* d_fv = c->fern_maps_dev;
  float4* d_fn = d_fv + n;
  float4* d_cv = d_fn + n;
  float4* d_cn = d_cv + n;
  const uint8_t* zero_image = (const uint8_t*)(d_cn + n);
  // a failed copy or launch must not hand a stale pose and inlier count to Ferns::findFrame's gates: the first HIP error is kept in the
  // context (global_loop_closure returns EF_EHIP for it) and the candidate is rejected (error = +inf, count = 0)
  hipError_t e = hipMemcpyAsync(d_fv, fv, n * 16, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_fn, fn, n * 16, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_cv, cv, n * 16, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) e = hipMemcpyAsync(d_cn, cn, n * 16, hipMemcpyHostToDevice, s);
  (void)Tf;   // the caller hands T_io = T_wc_fern in (Ferns.cpp:250); the model maps are transformed with it
  eft::pose_injected(c->st3, T_io, false, 1.0f, false, nullptr, 0, s);
  eft::init_icp_model(c->pyr3, (const float*)d_fv, (const float*)d_fn, (const float*)d_fv, (const float*)d_fn, c->st3, 6.0f, s);
  eft::init_icp_maps(c->pyr3, (const float*)d_cv, (const float*)d_cn, zero_image, c->st3, 6.0f, s);
  eft::TrackParams tp;
  tp.rgbOnly = false; tp.pyramid = false; tp.fastOdom = false; tp.so3 = false; tp.icpWeight = 100.f;
  tp.persistent = c->persistent;
  tp.fused_step = c->fused_step ? 1 : 0;
  tp.no_resident = c->no_resident ? 1 : 0;
  tp.distThres = 0.10f;
  tp.angleThres = sinf(20.f * 3.14159254f / 180.f);
  const eft::TrackTail tail = eft::track(c->pyr3, c->st3, c->intr3, tp, s, nullptr);
  eft::track_end(c->st3, tail, false, 1.0f, nullptr, -1, s, eft::tracker_abort_word(c->pyr3), c->d_abort + 2);
  if (e == hipSuccess) e = hipMemcpyAsync(&c->h_states[1], c->st3, sizeof(eft::TrackState), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e == hipSuccess) e = hipGetLastError();
  if (e != hipSuccess) {
    if (c->fern_tracker_error.empty()) c->fern_tracker_error = std::string("fern-to-view registration: ") + hipGetErrorString(e);
    *err = std::numeric_limits<float>::infinity();
    *cnt = 0.f;
    c->gloop.icp_error = *err;
    c->gloop.icp_count = 0.f;
    return;
  }
  const eft::TrackState& h = c->h_states[1];
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = h.q[i];
  for (int i = 0; i < 3; ++i) T.t[i] = h.t[i];
  efl::se3_matrix(T, T_io);
  *err = h.lastICPError;
  *cnt = h.lastICPCount;
  c->gloop.icp_error = h.lastICPError;
  c->gloop.icp_count = h.lastICPCount;
}

// ElasticFusion.cpp:392-445; *accepted_with_graph = 1 when a fern was matched AND the global deformation accepted with a graph.  A lost
// camera (relocalisation) takes the matched keyframe's registration as its pose instead (:411-413).
int global_loop_closure(ef_ctx* c, int log_slot, int* accepted_with_graph) {
  *accepted_with_graph = 0;
  ef_global_loop& G = c->gloop;
  memset(&G, 0, sizeof(G));
  G.attempted = 1;
  G.closest = -1;
  for (int i = 0; i < 16; ++i) G.T_wc_recovery[i] = (i % 5 == 0) ? 1.0 : 0.0;   // Sophus::SE3d T_wc_est; (Ferns.cpp:236)
  ef_ferns* F = ef_closure_ferns(c->closure);
  // Ferns::findFrame only considers keyframes stored more than 300 ticks ago (Ferns.cpp:218).  While there is none — the host knows: it
  // keeps the database — the answer is -1 whatever the view shows, and nothing has to come back from the device: no synchronisation.
  if (!ef_closure_candidate_possible(c->closure, c->tick)) return EF_OK;
  // otherwise: the view's fern codes, computed on the device, + the pose — one small read-back (0.5 KB + the state)
  int r0 = enqueue_fern_codes(c, c->h_codes);
  if (r0 != EF_OK) return r0;
  EF_HIP(c, hipMemcpyAsync(&c->h_states[0], c->st, sizeof(eft::TrackState), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  r0 = flush_end_record(c);   // the previous frame's keyframe decision first: the database findFrame walks must be complete
  if (r0 != EF_OK) return r0;
  pose_of_state(c->h_states[0], c->h_pose);
  int good = 0;
  memcpy(&good, c->h_codes + FERN_CODES_PAD, sizeof(int));
  if (c->lost) {
    const int r = ef_closure_relocalise_coded(c->closure, c->h_codes, good, &fetch_mid_view, c, c->h_pose, c->tick, &fern_tracker_device, c, G.T_wc_recovery);
    if (!c->fern_tracker_error.empty()) { c->err = c->fern_tracker_error; c->fern_tracker_error.clear(); return EF_EHIP; }
    if (r < 0) { c->err = "ef_closure_relocalise failed"; return r; }
    G.closest = ef_ferns_last_closest(F);
    if (r == 1) {
      eft::pose_injected(c->st, G.T_wc_recovery, false, 1.0f, false, log_slot >= 0 ? c->traj : nullptr, log_slot, c->stream);
      c->last_frame_recovery = true;
    }
    return EF_OK;
  }
  c->loop_graph.assign((size_t)1024 * 16, 0.f);
  int nodes = 0;
  const int r = ef_closure_global_coded(c->closure, c->h_codes, good, &fetch_mid_view, c, c->h_pose, c->tick, &fern_tracker_device, c, c->h_nodes_pinned,
                                        c->n_nodes_host, G.T_wc_recovery, c->loop_graph.data(), &nodes);
  if (!c->fern_tracker_error.empty()) { c->err = c->fern_tracker_error; c->fern_tracker_error.clear(); return EF_EHIP; }
  if (r < 0) { c->err = "ef_closure_global failed"; return r; }
  G.closest = ef_ferns_last_closest(ef_closure_ferns(c->closure));   // Ferns::lastClosest: -1 unless a keyframe passed every gate
  if (G.closest >= 0) {   // the rows handed to the optimiser: two per fern constraint (the constraint and its pin) + the kept relative ones
    const int rows = ef_closure_last_rows(c->closure, nullptr, 0, nullptr, nullptr), rel = ef_closure_relative(c->closure, nullptr, 0);
    G.n_constraints = rows > rel ? (rows - rel) / 2 : 0;
  }
  if (r != 1) return EF_OK;
  if (nodes < 0 || nodes >= 1024) { c->err = "global closure: 0..1023 graph nodes (GlobalModel::MAX_NODES)"; return EF_EINVAL; }
  G.accepted = 1;
  G.graph_nodes = nodes;
  // T_wc := the recovered pose (:429); the frame's logged pose follows (:588); the velocity weighting of :369-383 stays
  eft::pose_injected(c->st, G.T_wc_recovery, false, 1.0f, false, log_slot >= 0 ? c->traj : nullptr, log_slot, c->stream);
  if (nodes > 0) {
    EF_HIP(c, hipMemcpyAsync(c->graph_dev, c->loop_graph.data(), (size_t)nodes * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    EF_HIP(c, hipStreamSynchronize(c->stream));
    c->graph_nodes = nodes;
    c->graph_is_fern = 1;                                                                            // fernAccepted, :441,584
    *accepted_with_graph = 1;
  }
  return EF_OK;
}

// ElasticFusion.cpp:447-527.  The optimisation is the registered solver's, the built-in one's, or — with the global closure enabled —
// the closure object's (keyframe poses follow, relative constraints are kept).  Synchronises once, where the reference reads the
// constraint buffers back (Resize.cpp:108,146).  have_active: the ACTIVE prediction at the new pose (predict() of :387) was already made.
int local_loop_closure(ef_ctx* c, int log_slot, bool have_active) {
  hipStream_t s = c->stream;
  const int W = c->cam.cols, H = c->cam.rows, step = 20 /* consSample, ElasticFusion.cpp:62 */;
  const int cw = W / step, ch = H / step;
  const efm::FillMaps none{nullptr, nullptr, nullptr};
  const unsigned* count = &c->st->map_counts[c->cur];
  // predict() of :387: the ACTIVE view at the pose just estimated (its fill-in only feeds the fern database: made by the caller then)
  if (!have_active)
    efm::combined_predict(c->cam, c->st->T_cw, c->maps[c->cur], count, c->maxDepthProcessed, c->cfg.confidence, c->tick, c->tick, c->cfg.time_delta,
                          c->zbuf, c->pm, none, nullptr, nullptr, false, nullptr, s, nullptr, 0u, nullptr, 0u, c->rays);
  // :451-459, IndexMap::INACTIVE: surfels last seen at or before tick - timeDelta
  // (the prediction stamps st2->model_view_stamp with this frame's value when it shows at least one surfel: the model-to-model tracker's
  // persistent launch leaves at once otherwise — nothing can be registered against an empty view, and the reference's tracker, which runs
  // all the same, ends on zero sums: the stamp only says which frames those are)
  const unsigned view_stamp = (unsigned)c->tick * 2u + 1u;
  efm::combined_predict(c->cam, c->st->T_cw, c->maps[c->cur], count, c->maxDepthProcessed, c->cfg.confidence, 0, c->tick - c->cfg.time_delta,
                        c->cfg.time_delta, c->zbuf, c->old, none, nullptr, nullptr, false, nullptr, s, &c->st2->model_view_stamp, view_stamp, nullptr, 0u,
                        c->rays);
  eft::copy_pose(c->st2, c->st, s);                                                              // :469
  const float maxDepthRGB = 6.0f;                                                                // RGBDOdometry.cpp:42
  // :463 initICPModel(inactive view) + :464 initRGBModel(its image) + :466-467 initICP / initRGB(active view), fused (eft::init_model_pair)
  eft::init_model_pair(c->pyr2, (const float*)c->old.vertex, (const float*)c->old.normal, (const uint8_t*)c->old.image, (const float*)c->pm.vertex,
                       (const float*)c->pm.normal, (const uint8_t*)c->pm.image, c->st2, maxDepthRGB, s);
  eft::init_rgb_sobel(c->pyr2, s);
  eft::TrackParams tp;
  tp.rgbOnly = false; tp.pyramid = c->cfg.pyramid != 0; tp.fastOdom = c->cfg.fast_odom != 0; tp.so3 = false; tp.icpWeight = 10.f;   // :471
  tp.persistent = c->persistent;
  tp.fused_step = c->fused_step ? 1 : 0;
  tp.no_resident = c->no_resident ? 1 : 0;
  tp.distThres = 0.10f;
  tp.angleThres = sinf(20.f * 3.14159254f / 180.f);
  tp.empty_model_flag = &c->st2->model_view_stamp;
  tp.empty_model_value = view_stamp;
  const eft::TrackTail tail2 = eft::track(c->pyr2, c->st2, c->intr, tp, s, nullptr);
  eft::track_end(c->st2, tail2, true, 1.0f, nullptr, -1, s, eft::tracker_abort_word(c->pyr2), c->d_abort + 1);
  eft::sample_constraints((const float*)c->pm.vertex, c->old.time, W, H, step, c->cons_dev, s);  // :485-486
  EF_HIP(c, hipMemcpyAsync(&c->h_states[0], c->st, sizeof(eft::TrackState), hipMemcpyDeviceToHost, s));
  EF_HIP(c, hipMemcpyAsync(&c->h_states[1], c->st2, sizeof(eft::TrackState), hipMemcpyDeviceToHost, s));
  EF_HIP(c, hipMemcpyAsync(c->h_cons, c->cons_dev, (size_t)cw * ch * 4 * sizeof(float), hipMemcpyDeviceToHost, s));
  EF_HIP(c, hipStreamSynchronize(s));
  {
    const int rf = flush_end_record(c);   // the previous frame's end-of-frame record (keyframe decision, graph nodes) has landed by now
    if (rf != EF_OK) return rf;
  }
  ef_local_loop& L = c->loop;
  memset(&L, 0, sizeof(L));
  c->loop_constraints.clear();
  L.attempted = 1;
  L.graph_capacity = 1023;   // GlobalModel::MAX_NODES - 1 (GlobalModel.cpp:24): rows of loop_graph / graph_dev
  const eft::TrackState& hc = c->h_states[0];
  const eft::TrackState& he = c->h_states[1];
  efl::SE3 Tc, Te;
  for (int i = 0; i < 4; ++i) { Tc.q[i] = hc.q[i]; Te.q[i] = he.q[i]; }
  for (int i = 0; i < 3; ++i) { Tc.t[i] = hc.t[i]; Te.t[i] = he.t[i]; }
  efl::se3_matrix(Tc, L.T_wc_curr);
  efl::se3_matrix(Te, L.T_wc_est);
  L.stats[0] = he.lastICPError; L.stats[1] = he.lastICPCount; L.stats[2] = he.lastRGBError;
  L.stats[3] = he.lastRGBCount; L.stats[4] = he.lastSO3Error; L.stats[5] = he.lastSO3Count;
  double cov[36];
  efl::lu_inverse<double, 6>(he.lastA, cov);                                                     // :473, getCovariance
  bool covOk = true;
  for (int i = 0; i < 6; ++i) {
    L.cov_diag[i] = cov[i * 6 + i];
    if (cov[i * 6 + i] > (double)c->cov_thresh) { covOk = false; break; }
  }
  L.cov_ok = covOk;
  L.gates_ok = covOk && he.lastICPCount > (float)c->icp_count_thresh && he.lastICPError < c->icp_err_thresh;   // :483-484
  if (!L.gates_ok) return EF_OK;
  const double* M = L.T_wc_curr;
  const double* E = L.T_wc_est;
  for (int i = 0; i < cw; ++i)
    for (int j = 0; j < ch; ++j) {
      const float* v = c->h_cons + (size_t)(i * ch + j) * 4;
      const unsigned tm = (unsigned)v[3];
      if (v[2] > 0 && v[2] < c->maxDepthProcessed && tm > 0) {                                     // :490-492
        double row[8];
        for (int r = 0; r < 3; ++r) {   // T * Vector4d(x, y, z, 1), a 4x4 matrix product evaluated left to right
          row[r] = ((M[r * 4] * (double)v[0] + M[r * 4 + 1] * (double)v[1]) + M[r * 4 + 2] * (double)v[2]) + M[r * 4 + 3] * 1.0;
          row[3 + r] = ((E[r * 4] * (double)v[0] + E[r * 4 + 1] * (double)v[1]) + E[r * 4 + 2] * (double)v[2]) + E[r * 4 + 3] * 1.0;
        }
        row[6] = (double)tm;
        row[7] = c->deforms == 0 ? 1.0 : 0.0;                                                      // :507-508 pinConstraints
        c->loop_constraints.insert(c->loop_constraints.end(), row, row + 8);
      }
    }
  L.n_constraints = (int)(c->loop_constraints.size() / 8);
  if (!c->solver && !c->builtin_solver) return EF_OK;
  c->loop_graph.assign((size_t)1024 * 16, 0.f);
  int nodes = 0;
  bool accepted = false;
  if (c->solver) {
    accepted = c->solver(c->solver_user, &L, c->loop_constraints.data(), L.n_constraints, c->loop_graph.data(), &nodes) != 0;   // :513-514
  } else if (c->closure) {
    // Deformation::constrain in full (:511-526): graph sampled at the end of the previous frame, keyframe poses deformed along,
    // a third of the new relative constraints kept for later global closures
    const int r = ef_closure_local(c->closure, c->loop_constraints.data(), L.n_constraints, c->tick, c->h_nodes_pinned, c->n_nodes_host,
                                   c->loop_graph.data(), &nodes);
    if (r < 0) { c->err = "ef_closure_local failed"; return r; }
    accepted = r == 1;
  } else {
    // the built-in optimiser on the graph Deformation::sampleGraphModel would have sampled at the end of the previous frame
    // (ElasticFusion.cpp:593): every 5000th surfel of the map as it stands now
    const int max_nodes = 1023;
    unsigned* n_dev = (unsigned*)(c->nodes_dev + (size_t)1024 * 4);
    efm::sample_graph(c->maps[c->cur], count, 5000, max_nodes, c->nodes_dev, n_dev, s);
    c->h_nodes.resize((size_t)1024 * 4 + 4);
    EF_HIP(c, hipMemcpyAsync(c->h_nodes.data(), c->nodes_dev, ((size_t)1024 * 4 + 1) * sizeof(float), hipMemcpyDeviceToHost, s));
    EF_HIP(c, hipStreamSynchronize(s));
    unsigned n_nodes = 0;
    memcpy(&n_nodes, &c->h_nodes[(size_t)1024 * 4], sizeof(unsigned));
    const efd::Result r = efd::solve_local(c->h_nodes.data(), (int)n_nodes, c->loop_constraints.data(), L.n_constraints, (uint64_t)c->tick,
                                           (uint64_t)c->last_deform_time, c->loop_graph.data());
    accepted = r.ok;
    nodes = r.ok ? (int)n_nodes : 0;
    if (r.ok) c->last_deform_time = c->tick;   // Deformation.cpp:199-201
  }
  if (accepted) {
    if (nodes < 0 || nodes >= 1024) { c->err = "loop solver: 0..1023 graph nodes (GlobalModel::MAX_NODES)"; return EF_EINVAL; }
    L.applied = 1;
    L.graph_nodes = nodes;
    c->deforms += nodes > 0;                                                                       // :523
    eft::adopt_pose(c->st, c->st2, log_slot >= 0 ? c->traj : nullptr, log_slot, s);                // :525
    if (nodes > 0) {
      EF_HIP(c, hipMemcpyAsync(c->graph_dev, c->loop_graph.data(), (size_t)nodes * 16 * sizeof(float), hipMemcpyHostToDevice, s));
      EF_HIP(c, hipStreamSynchronize(s));
    }
    c->graph_nodes = nodes;
    c->graph_is_fern = 0;
  }
  return EF_OK;
}

// rgb_src / depth_src: host (pinned staging) or device pointers, `kind` says which
int process_frame(ef_ctx* c, const uint8_t* rgb_src, const uint16_t* depth_src, hipMemcpyKind kind, int64_t timestamp,
                  float weightMultiplier, const double* in_T_wc, hipEvent_t images_ready = nullptr) {
  hipStream_t s = c->stream;
  const int W = c->cam.cols, H = c->cam.rows;
  // a persistent tracker launch of an EARLIER frame gave up waiting after admission (a protocol failure, sticky: every later launch of that
  // instance returns at once): k_track_end has copied the flag into host-mapped memory; reported here, where the front end calls
  // (class ElasticFusion::processFrame throws), without synchronising — ef_synchronize reports the same condition for the frames in flight
  if ((c->h_abort && (c->h_abort[0] | c->h_abort[1] | c->h_abort[2])) || c->pyr.sticky_abort || c->pyr2.sticky_abort || c->pyr3.sticky_abort) {
    c->err = "a persistent tracker launch of an earlier frame timed out waiting for another workgroup after its whole grid had reported in (a protocol "
             "failure, not a busy chip: that case runs on one workgroup, ef_get_tracker_fallbacks): the poses and the map since then are invalid; "
             "recreate the context (ef_set_persistent_tracker(ctx, 0) selects the launch-per-step script)";
    return EF_EHIP;
  }
  // Input stage: everything that needs nothing but the new frame.  With overlap on it is enqueued on in_stream and
  // waits only for the previous frame's TRACKER (the last reader of the frame-side pyramids); it then runs concurrently
  // with the previous frame's fusion + prediction on `stream`, which read the other set of frame images.
  const bool overlap = c->overlap && !c->timing && c->in_stream != nullptr;
  hipStream_t sb = overlap ? c->in_stream : s;
  // Round 6: the events the second stream waits for are recorded only while the overlap is on — an event record between two kernels of a stream is
  // a barrier packet, measured as a 6 us bubble each (three per frame: profiles/r06f_timeline_single_stream.txt).  The first overlapped frame
  // after a frame that recorded none joins the streams on the host instead.
  const bool want_events = c->overlap && c->in_stream != nullptr;
  if (overlap && !c->events_live) EF_HIP(c, hipStreamSynchronize(s));
  std::swap(c->rgb, c->rgb_alt);
  std::swap(c->depth_raw, c->depth_raw_alt);
  std::swap(c->depth_filtered, c->depth_filtered_alt);
  std::swap(c->depth_metric, c->depth_metric_alt);
  std::swap(c->depth_metric_filtered, c->depth_metric_filtered_alt);
  const bool track_this = c->tick > 1 && !in_T_wc;
  c->frame_parity ^= 1;
  // (mode 2 would put the bilateral filter on the chip WHILE the persistent tracker launch wants all of its CUs: its admission would fail and the
  // frame would run on the one-workgroup fallback — mode 1 is what such a context gets)
  const int overlap_mode = (c->overlap_mode == 2 && c->persistent == 1 && !c->use_graph) ? 1 : c->overlap_mode;
  if (overlap) EF_HIP(c, hipStreamWaitEvent(sb, overlap_mode == 2 ? c->ev_frame_done[c->frame_parity] : c->ev_track_done, 0));
  if (images_ready) EF_HIP(c, hipStreamWaitEvent(sb, images_ready, 0));   // (the upload of a host-pointer frame, on copy_stream)
  // the frame images are referenced by later stages of this frame and by the next frame's tracker
  // (fill-in / predict read depth_filtered + rgb), so they are copied into context-owned buffers
  // Frames that are already in HBM are not copied by separate launches in the single-stream script: the bilateral filter
  // reads the caller's depth directly (nothing later needs the raw image) and the RGB copy rides on the intensity kernel.
  const bool fold_copies = kind == hipMemcpyDeviceToDevice && track_this;
  const uint16_t* depth_in = c->depth_raw;
  if (fold_copies) {
    depth_in = depth_src;
  } else {
    EF_HIP(c, hipMemcpyAsync(c->rgb, rgb_src, (size_t)W * H * 3, kind, sb));
    EF_HIP(c, hipMemcpyAsync(c->depth_raw, depth_src, (size_t)W * H * 2, kind, sb));
  }
  timer_begin(c, "Preprocess");
  // (a tracked frame: the level-0 intensity image of the frame and — folded copies — the context's copy of the colours ride on this launch)
  const uint8_t* rgb_in = fold_copies ? rgb_src : c->rgb;
  // Round 6: in the single-stream script of a tracked frame the pre-processing and the tracker's model-side maps are ONE launch (k_frame_inputs,
  // at init_icp_model below): two independent kernels, one bound by LDS look-ups, the other by HBM.
#ifdef EF_SEPARATE_INPUTS   // (A/B build "sepinputs")
  const bool joint_inputs = false;
#else
  const bool joint_inputs = track_this && !overlap && !c->timing;
#endif
  if (!joint_inputs && !efm::preprocess_depth(depth_in, W, H, c->cfg.depth_cut, c->depth_filtered, c->depth_metric, c->depth_metric_filtered, sb, 0u,
                             track_this ? rgb_in : nullptr, c->pyr.nextImage[0], fold_copies ? c->rgb : nullptr, c->bil_table)) {
    c->err = "bilateral weight table missing on this device";
    return EF_EHIP;
  }
  timer_end(c, "Preprocess");
  if (overlap && overlap_mode == 2) EF_HIP(c, hipStreamWaitEvent(sb, c->ev_track_done, 0));
  if (track_this && overlap)   // the single-stream script builds all pyramids together below (eft::build_pyramids)
    eft::build_pyramids_frame_side(c->pyr, c->depth_filtered, c->intr, c->maxDepthProcessed, nullptr, sb, nullptr);
  if (overlap) EF_HIP(c, hipEventRecord(c->ev_input_done, sb));

  const bool rgbOnly = c->cfg.rgb_only != 0;
  // t_T_wc.push_back / poseLogTimes.push_back, ElasticFusion.cpp:588-589: the pose is logged by the kernel that produces it
  if ((int)c->stamps.size() >= c->traj_cap) {   // t_T_wc grows without bound in the reference: double the device log (rare: every 2^16+ frames)
    const int r = grow_trajectory(c);
    if (r != EF_OK) return r;
  }
  const int log_slot = (int)c->stamps.size();
  c->stamps.push_back(timestamp);
  if (c->tick == 1) {  // ElasticFusion.cpp:290-296
    if (overlap) EF_HIP(c, hipStreamWaitEvent(s, c->ev_input_done, 0));
    timer_begin(c, "feedbackBuffers");
    efm::seed_map(c->cam, c->rgb, c->depth_metric, c->depth_metric_filtered, c->tick, c->maxDepthProcessed, c->maps[c->cur],
                  &c->st->map_counts[c->cur], c->cs, s);
    eft::init_first_rgb(c->pyr, c->rgb, s);
    if (log_slot >= 0) eft::log_pose(c->st, c->traj, log_slot, s);
    if (want_events) EF_HIP(c, hipEventRecord(c->ev_track_done, s));
    timer_end(c, "feedbackBuffers");
  } else {
    if (!in_T_wc) {
      eft::TrackParams tp;
      tp.rgbOnly = rgbOnly;
      tp.pyramid = c->cfg.pyramid != 0;
      tp.fastOdom = c->cfg.fast_odom != 0;
      tp.so3 = c->cfg.so3 != 0;
      tp.icpWeight = c->cfg.icp_weight;
      tp.persistent = c->persistent;
      tp.fused_step = c->fused_step ? 1 : 0;
  tp.no_resident = c->no_resident ? 1 : 0;
      tp.distThres = 0.10f;                                   // RGBDOdometry.h:41
      tp.angleThres = sinf(20.f * 3.14159254f / 180.f);       // RGBDOdometry.h:42
      const bool rgb = tp.rgbOnly || tp.icpWeight < 100;
      timer_begin(c, "odomInit");
      const eft::FramePreprocess fp{depth_in, c->cfg.depth_cut, c->bil_table, c->depth_filtered, c->depth_metric, c->depth_metric_filtered, rgb_in,
                                    fold_copies ? c->rgb : nullptr};
      eft::init_icp_model(c->pyr, (const float*)c->pm.vertex, (const float*)c->pm.normal, (const float*)c->fm.vertex,
                          (const float*)c->fm.normal, c->st, 6.0f /* maxDepthRGB, RGBDOdometry.cpp:42 */, s, (const uint8_t*)c->pm.image,
                          (const uint8_t*)c->fm.image, c->cfg.frame_to_frame_rgb != 0, joint_inputs ? &fp : nullptr, c->tally_pending);
      c->tally_pending = false;
      if (overlap) {
        eft::build_pyramids_model_side(c->pyr, nullptr, nullptr, false, c->st, s);
        EF_HIP(c, hipStreamWaitEvent(s, c->ev_input_done, 0));
      } else {
        eft::build_pyramids(c->pyr, c->depth_filtered, c->intr, c->maxDepthProcessed, nullptr, nullptr, false, nullptr, c->st, s, nullptr, rgb);
      }
      if (rgb && overlap) eft::init_rgb_sobel(c->pyr, s);
      timer_end(c, "odomInit");
      timer_begin(c, "odom");
      const bool sample = c->ktime_every > 0 && (c->tick % c->ktime_every) == 0;
      // BASELINE configs[4] is the hipGraph-captured launch-per-step script; the persistent launch takes a fresh exchange epoch
      // per launch as a kernel argument, which a replayed graph cannot give it
      if (c->use_graph && !sample && !c->timing) tp.persistent = 0;
      eft::TrackTail tail{};
      if (c->use_graph && !sample && !c->timing) {
        // key: which of the two intensity pyramids is "next" this frame + the knobs baked into the launch arguments
        const void* key = c->pyr.nextImage[0];
        ef_ctx::TrackGraph* g = nullptr;
        for (auto& cand : c->tgraph)
          if (cand.exec && cand.key == key && !memcmp(&cand.tp, &tp, sizeof(tp))) g = &cand;
        if (!g) {
          g = (c->tgraph[0].exec && c->tgraph[0].key != key) ? &c->tgraph[1] : &c->tgraph[0];
          if (g->exec) { (void)hipGraphExecDestroy(g->exec); g->exec = nullptr; }
          eft::track_prepare(c->pyr, tp, s);   // (a script switch clears the exchange areas: outside the capture, not replayed with it — ADVICE r5)
          eft::Pyramid pyr_copy = c->pyr;   // track() swaps the copy's pointers; the real swap is done below
          hipGraph_t graph = nullptr;
          EF_HIP(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
          g->tail = eft::track(pyr_copy, c->st, c->intr, tp, s, nullptr);
          hipError_t ce = hipStreamEndCapture(s, &graph);   // always ends the capture, whatever was recorded
          if (ce == hipSuccess) ce = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
          if (graph) (void)hipGraphDestroy(graph);
          if (ce != hipSuccess) {
            g->exec = nullptr;
            c->err = std::string("hipGraph capture of the tracker: ") + hipGetErrorString(ce);
            return EF_EHIP;
          }
          g->key = key;
          memcpy(&g->tp, &tp, sizeof(tp));
        }
        EF_HIP(c, hipGraphLaunch(g->exec, s));
        eft::track_swap(c->pyr, tp);
        tail = g->tail;
      } else {
        tail = eft::track(c->pyr, c->st, c->intr, tp, s, sample ? &c->probe : nullptr, sample && c->probe_all.start ? &c->probe_all : nullptr);
      }
      eft::track_end(c->st, tail, rgb, weightMultiplier, log_slot >= 0 ? c->traj : nullptr, log_slot, s, eft::tracker_abort_word(c->pyr), c->d_abort);
      timer_end(c, "odom");
      c->tracking_ok = true;
      if (c->reloc) {   // :326-366: the tracker's verdict on itself, read back where the reference reads lastICPError / getCovariance()
        EF_HIP(c, hipMemcpyAsync(&c->h_reloc, c->st, sizeof(eft::TrackState), hipMemcpyDeviceToHost, s));
        EF_HIP(c, hipStreamSynchronize(s));
        c->tracking_ok = c->h_reloc.lastICPError < 1e-04;
        if (!c->lost) {
          if (!reloc_covariance_ok(c->h_reloc)) c->tracking_ok = false;
          if (!c->tracking_ok) {
            if (++c->tracking_count > 10) c->lost = true;
          } else {
            c->tracking_count = 0;
          }
        } else if (c->last_frame_recovery) {
          if (!reloc_covariance_ok(c->h_reloc)) c->tracking_ok = false;
          if (c->tracking_ok) {
            c->lost = false;
            c->tracking_count = 0;
          }
          c->last_frame_recovery = false;
        }
      }
    } else {
      c->tracking_ok = true;   // :300: an injected pose is never judged
      if (overlap) EF_HIP(c, hipStreamWaitEvent(s, c->ev_input_done, 0));
      eft::pose_injected(c->st, in_T_wc, true, weightMultiplier, true, log_slot >= 0 ? c->traj : nullptr, log_slot, s);
    }
    // from here on nothing of this frame reads the frame-side pyramids: the next frame's input stage may start
    if (want_events) EF_HIP(c, hipEventRecord(c->ev_track_done, s));
    // mid-frame predict() of ElasticFusion.cpp:387 is dead work without loop closure: skipped
    if (c->cfg.close_loops) {
      int fern_graph = 0;
      if (c->closure) {   // :387-445: predict() with its fill-in at the new pose, then the fern database
        timer_begin(c, "globalLoop");
        do_predict(c, false);
        c->last_frame_recovery = false;                                                           // :393
        const int r = global_loop_closure(c, log_slot, &fern_graph);
        timer_end(c, "globalLoop");
        if (r != EF_OK) return r;
      }
      if (c->lost) {       // :447: a lost camera closes no local loop
        memset(&c->loop, 0, sizeof(c->loop));
      } else if (!fern_graph) {   // :447: rawGraph.size() == 0
        timer_begin(c, "localLoop");
        const int r = local_loop_closure(c, log_slot, c->closure != nullptr);
        timer_end(c, "localLoop");
        if (r != EF_OK) return r;
      } else {
        memset(&c->loop, 0, sizeof(c->loop));
      }
    }
    if (!rgbOnly && c->tracking_ok && !c->lost && !c->track_only) {  // ElasticFusion.cpp:536-585
      timer_begin(c, "indexMap");
      const bool sample_splat = c->ktime_every > 0 && (c->tick % c->ktime_every) == 0 && c->probe_splat.start;
      efm::IndexMaps im_assoc = c->im;   // what the association taps: no colour / time stream (the second predictIndices below writes all four maps)
#ifndef EF_RESOLVE_ALL_MAPS   // (A/B build "resolveall")
      im_assoc.color_time = nullptr;
#endif
      efm::predict_indices(c->cam, c->st->T_cw, c->tick, c->maps[c->cur], &c->st->map_counts[c->cur], c->maxDepthProcessed, c->cfg.time_delta, c->zbuf,
                           im_assoc, s, sample_splat ? &c->probe_splat : nullptr);
      timer_end(c, "indexMap");
      timer_begin(c, "Fuse::Data+Update");
      // (the update pass — k_merge — rides on the splat of the second predictIndices: one launch less; with stage timers on it stays a launch of
      // its own so that the reference's TICK / TOCK stages keep their meaning)
#ifdef EF_SEPARATE_MERGE   // (A/B build "sepmerge")
      const bool defer_merge = false;
#else
      const bool defer_merge = !c->timing;
#endif
      efm::fuse(c->cam, c->st->pose_f, c->tick, c->rgb, c->depth_metric, c->depth_metric_filtered, c->im, c->maxDepthProcessed,
                &c->st->weighting, c->maps[c->cur], &c->st->map_counts[c->cur], c->cand, c->winner, s, defer_merge);
      if (c->reference_download && !defer_merge) efm::copy_map(c->maps[c->cur], &c->st->map_counts[c->cur], c->shadow, s);
      timer_end(c, "Fuse::Data+Update");
      timer_begin(c, "indexMap2");
      efm::predict_indices(c->cam, c->st->T_cw, c->tick, c->maps[c->cur], &c->st->map_counts[c->cur], c->maxDepthProcessed, c->cfg.time_delta, c->zbuf,
                           c->im, s, nullptr, defer_merge ? &c->cand : nullptr, c->winner);
      if (c->reference_download && defer_merge) efm::copy_map(c->maps[c->cur], &c->st->map_counts[c->cur], c->shadow, s);
      timer_end(c, "indexMap2");
      // a pending deformation (ElasticFusion.cpp:558-585): re-predict the depth of the surfels outside the time window, then let
      // clean() move every kept surfel with the graph
      efm::Deformation def{c->graph_dev, c->graph_nodes, c->synth_depth, c->graph_is_fern, c->maxDepthProcessed};
      if (c->graph_nodes > 0 && !c->graph_is_fern)
        efm::synthesize_depth(c->cam, c->st->T_cw, c->maps[c->cur], &c->st->map_counts[c->cur], c->maxDepthProcessed, c->cfg.confidence, c->tick,
                              c->tick - c->cfg.time_delta, 65535, c->zbuf, c->synth_depth, s, c->rays);
      timer_begin(c, "Fuse::Copy");
      c->cs.flip ^= 1;   // (CompactScratch::group_sum: this call's half was cleared by the call before it)
      efm::clean(c->cam, c->st->T_cw, c->tick, c->im, c->cfg.confidence, c->cfg.time_delta, c->maps[c->cur], &c->st->map_counts[c->cur], c->cand,
                 c->winner, c->maps[c->cur ^ 1], &c->st->map_counts[c->cur ^ 1], c->capacity, c->cs, c->overflow, s,
                 c->graph_nodes > 0 ? &def : nullptr);
      c->graph_nodes = 0;
      c->cur ^= 1;
      timer_end(c, "Fuse::Copy");
    } else {
      c->graph_nodes = 0;   // rawGraph is a local of processFrame: a deformation accepted in a frame that does not fuse is never applied
    }
  }
  timer_begin(c, "IndexMap::ACTIVE");
  do_predict(c);  // ElasticFusion.cpp:599
  timer_end(c, "IndexMap::ACTIVE");
  if (c->closure) {   // :588-589, 593, 609-618: pose -> trajectory, graph nodes re-sampled, final fill-in view -> Ferns::addFrame
    timer_begin(c, "ferns");
    int r = flush_end_record(c);           // (normally long done: the frame's closures synchronised)
    if (r == EF_OK) r = enqueue_end_record(c);   // no synchronisation: looked at when the next frame first waits for the stream
    timer_end(c, "ferns");
    if (r != EF_OK) return r;
  }
  if (want_events && c->overlap_mode == 2) EF_HIP(c, hipEventRecord(c->ev_frame_done[c->frame_parity], s));
  c->events_live = want_events;
  if (!c->lost) c->tick++;   // :601-604
  EF_HIP(c, hipGetLastError());
  return EF_OK;
}

int ctx_init(ef_ctx* c) {
  const ef_config& g = c->cfg;
  const int W = g.width, H = g.height;
  const size_t P = (size_t)W * H;
  hipStream_t s = c->stream;
  EF_ALLOC(c, c->rgb, P * 3);
  EF_ALLOC(c, c->depth_raw, P);
  EF_ALLOC(c, c->depth_filtered, P);
  EF_ALLOC(c, c->depth_metric, P);
  EF_ALLOC(c, c->depth_metric_filtered, P);
  EF_ALLOC(c, c->rgb_alt, P * 3);
  EF_ALLOC(c, c->depth_raw_alt, P);
  EF_ALLOC(c, c->depth_filtered_alt, P);
  EF_ALLOC(c, c->depth_metric_alt, P);
  EF_ALLOC(c, c->depth_metric_filtered_alt, P);
  EF_HIP(c, hipStreamCreateWithFlags(&c->in_stream, hipStreamNonBlocking));
  for (auto& e : c->ev_frame_done) EF_HIP(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  EF_HIP(c, hipEventCreateWithFlags(&c->ev_input_done, hipEventDisableTiming));
  EF_HIP(c, hipEventCreateWithFlags(&c->ev_track_done, hipEventDisableTiming));
  EF_HIP(c, hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
  EF_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < ef_ctx::RING; ++i) {
    EF_HIP(c, hipEventCreateWithFlags(&c->ev_h2d[i], hipEventDisableTiming));
    EF_HIP(c, hipHostMalloc((void**)&c->h_rgb_ring[i], P * 3));
    EF_HIP(c, hipHostMalloc((void**)&c->h_depth_ring[i], P * 2));
    EF_ALLOC(c, c->d_rgb_ring[i], P * 3);
    EF_ALLOC(c, c->d_depth_ring[i], P);
  }
  EF_HIP(c, hipHostMalloc((void**)&c->h_consumed, sizeof(unsigned), hipHostMallocMapped));
  *c->h_consumed = 0u;
  EF_HIP(c, hipHostGetDevicePointer((void**)&c->d_consumed, c->h_consumed, 0));
  EF_HIP(c, hipHostMalloc((void**)&c->h_abort, 4 * sizeof(unsigned), hipHostMallocMapped));
  memset(c->h_abort, 0, 4 * sizeof(unsigned));
  EF_HIP(c, hipHostGetDevicePointer((void**)&c->d_abort, c->h_abort, 0));
  // tracker pyramids (zero-filled: the stale y/z planes of quirk Q3 are then deterministic)
  c->pyr.width = W;
  c->pyr.height = H;
  for (int i = 0; i < eft::NUM_PYRS; ++i) {
    const size_t n = (size_t)(W >> i) * (H >> i);
    EF_ALLOC(c, c->pyr.depth_tmp[i], n);
    EF_ALLOC(c, c->pyr.vmap_curr[i], 3 * n);
    EF_ALLOC(c, c->pyr.nmap_curr[i], 3 * n);
    EF_ALLOC(c, c->pyr.vmap_g_prev[i], 3 * n);
    EF_ALLOC(c, c->pyr.nmap_g_prev[i], 3 * n);
    EF_ALLOC(c, c->pyr.lastDepth[i], n);
    c->pyr.nextDepth[i] = c->pyr.lastDepth[i];   // quirk Q1
    EF_ALLOC(c, c->pyr.lastImage[i], n);
    EF_ALLOC(c, c->pyr.nextImage[i], n);
    EF_ALLOC(c, c->pyr.lastNextImage[i], n);
    EF_ALLOC(c, c->pyr.dIdx[i], n);
    EF_ALLOC(c, c->pyr.dIdy[i], n);
    EF_ALLOC(c, c->pyr.corres[i], n);
    EF_ALLOC(c, c->pyr.rgbMask[i], n);
  }
  EF_ALLOC(c, c->pyr.partials, (size_t)eft::PARTIAL_ALLOC_FLOATS);
  EF_ALLOC(c, c->st, 1);
  // prediction images
  EF_ALLOC(c, c->im.index, P);
  EF_ALLOC(c, c->im.vert_conf, P);
  EF_ALLOC(c, c->im.color_time, P);
  EF_ALLOC(c, c->im.norm_rad, P);
  c->im.colmajor = 1;
  EF_ALLOC(c, c->pm.image, P);
  EF_ALLOC(c, c->pm.vertex, P);
  EF_ALLOC(c, c->pm.normal, P);
  EF_ALLOC(c, c->pm.time, P);
  EF_ALLOC(c, c->fm.image, P);
  EF_ALLOC(c, c->fm.vertex, P);
  EF_ALLOC(c, c->fm.normal, P);
  EF_ALLOC(c, c->zbuf, P, 0xFF);
  EF_ALLOC(c, c->graph_dev, 1024 * 16);     // GlobalModel::MAX_NODES x 16 (GlobalModel.cpp:24)
  EF_ALLOC(c, c->synth_depth, P);
  EF_ALLOC(c, c->rays, P * 4);
  EF_ALLOC(c, c->overflow, 1);
  // global model
  c->capacity = g.max_surfels;
  for (int k = 0; k < 2; ++k) {
    EF_ALLOC(c, c->maps[k].pos_conf, c->capacity);
    EF_ALLOC(c, c->maps[k].col_time, c->capacity);
    EF_ALLOC(c, c->maps[k].nrm_rad, c->capacity);
  }
  EF_ALLOC(c, c->winner, c->capacity, 0xFF);
  c->cand.n = (W / 2) * (H / 2);
  EF_ALLOC(c, c->cand.pos_conf, c->cand.n);
  EF_ALLOC(c, c->cand.col_time, c->cand.n);
  EF_ALLOC(c, c->cand.nrm_rad, c->cand.n);
  EF_ALLOC(c, c->cand.best, c->cand.n);
  c->cs.max_chunks = (int)((c->capacity + (size_t)c->cand.n + 2 * P) / efm::CLEAN_ROW + 8);
  EF_ALLOC(c, c->cs.flags, (size_t)c->capacity + c->cand.n + 2 * P);
  EF_ALLOC(c, c->cs.chunk_count, c->cs.max_chunks);
  EF_ALLOC(c, c->cs.chunk_offset, c->cs.max_chunks);
  EF_ALLOC(c, c->cs.totals, 8);
  c->cs.max_groups = c->cs.max_chunks / efm::CLEAN_GROUP + 2;
  EF_ALLOC(c, c->cs.group_sum, 2 * (size_t)c->cs.max_groups * efm::CLEAN_GSTRIDE);   // (zero-filled: what the first clean() expects of its half)
  if (g.close_loops) {
    c->pyr2.width = W;
    c->pyr2.height = H;
    for (int i = 0; i < eft::NUM_PYRS; ++i) {
      const size_t n = (size_t)(W >> i) * (H >> i);
      EF_ALLOC(c, c->pyr2.depth_tmp[i], n);
      EF_ALLOC(c, c->pyr2.vmap_curr[i], 3 * n);
      EF_ALLOC(c, c->pyr2.nmap_curr[i], 3 * n);
      EF_ALLOC(c, c->pyr2.vmap_g_prev[i], 3 * n);
      EF_ALLOC(c, c->pyr2.nmap_g_prev[i], 3 * n);
      EF_ALLOC(c, c->pyr2.lastDepth[i], n);
      EF_ALLOC(c, c->pyr2.nextDepth[i], n);
      EF_ALLOC(c, c->pyr2.lastImage[i], n);
      EF_ALLOC(c, c->pyr2.nextImage[i], n);
      EF_ALLOC(c, c->pyr2.lastNextImage[i], n);
      EF_ALLOC(c, c->pyr2.dIdx[i], n);
      EF_ALLOC(c, c->pyr2.dIdy[i], n);
      EF_ALLOC(c, c->pyr2.corres[i], n);
      EF_ALLOC(c, c->pyr2.rgbMask[i], n);
    }
    EF_ALLOC(c, c->pyr2.partials, (size_t)eft::PARTIAL_ALLOC_FLOATS);
    EF_ALLOC(c, c->st2, 1);
    EF_ALLOC(c, c->old.image, P);
    EF_ALLOC(c, c->old.vertex, P);
    EF_ALLOC(c, c->old.normal, P);
    EF_ALLOC(c, c->old.time, P);
    EF_ALLOC(c, c->cons_dev, (size_t)(W / 20) * (H / 20) * 4 + 4);
    EF_ALLOC(c, c->nodes_dev, (size_t)1024 * 4 + 4);
    EF_HIP(c, hipHostMalloc((void**)&c->h_cons, ((size_t)(W / 20) * (H / 20) * 4 + 4) * sizeof(float)));
    EF_HIP(c, hipHostMalloc((void**)&c->h_states, 3 * sizeof(eft::TrackState)));
    hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, s, c->st2, (W / 20) * (H / 20), W * H);
  }
  c->traj_cap = 1 << 10;   // doubled on demand (grow_trajectory)
  EF_ALLOC(c, c->traj, (size_t)c->traj_cap * 16);
  // T_wc = identity (ElasticFusion.h: T_wc_curr default) -> publish the float matrices
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, s, c->st, (W / 20) * (H / 20), W * H);
#ifndef EF_RESOLVE_TALLY
  c->tally_by_consumer = (W / 20) * (H / 20) <= 1024;
#endif
  efm::build_ray_table(c->cam, c->rays, s);
  const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  eft::pose_injected(c->st, I16, false, 1.0f, false, nullptr, 0, s);
  eft::pose_injected(c->st, I16, true, 1.0f, false, nullptr, 0, s);   // previous pose = identity too
  EF_HIP(c, hipStreamSynchronize(s));
  return EF_OK;
}

void ctx_free(ef_ctx* c) {
  if (c->in_stream) (void)hipStreamSynchronize(c->in_stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream) eft::persistent_chain_forget(c->stream);   // (a borrowed stream may be destroyed by its owner right after this call)
  if (c->in_stream) (void)hipStreamDestroy(c->in_stream);
  for (hipEvent_t e : {c->ev_input_done, c->ev_track_done, c->ev_staged, c->ev_frame_done[0], c->ev_frame_done[1]})
    if (e) (void)hipEventDestroy(e);
  for (void* p : c->allocs) (void)hipFree(p);
  if (c->h_abort) (void)hipHostFree(c->h_abort);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  for (int i = 0; i < ef_ctx::RING; ++i) {
    if (c->ev_h2d[i]) (void)hipEventDestroy(c->ev_h2d[i]);
    if (c->h_rgb_ring[i]) (void)hipHostFree(c->h_rgb_ring[i]);
    if (c->h_depth_ring[i]) (void)hipHostFree(c->h_depth_ring[i]);
  }
  if (c->h_consumed) (void)hipHostFree(c->h_consumed);
  if (c->h_cons) (void)hipHostFree(c->h_cons);
  if (c->h_states) (void)hipHostFree(c->h_states);
  if (c->h_view) (void)hipHostFree(c->h_view);
  if (c->h_view_end) (void)hipHostFree(c->h_view_end);
  if (c->h_codes) (void)hipHostFree(c->h_codes);
  if (c->h_codes_end) (void)hipHostFree(c->h_codes_end);
  if (c->ev_end_record) (void)hipEventDestroy(c->ev_end_record);
  if (c->h_nodes_pinned) (void)hipHostFree(c->h_nodes_pinned);
  if (c->closure) ef_closure_destroy(c->closure);
  for (auto& t : c->timers) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  for (auto e : c->kt_start) (void)hipEventDestroy(e);
  for (auto e : c->kt_stop) (void)hipEventDestroy(e);
  for (auto& g : c->tgraph)
    if (g.exec) (void)hipGraphExecDestroy(g.exec);
  if (c->debug_stream) { (void)hipStreamSynchronize(c->debug_stream); (void)hipStreamDestroy(c->debug_stream); }
  for (auto e : c->ka_start) (void)hipEventDestroy(e);
  for (auto e : c->ka_stop) (void)hipEventDestroy(e);
  for (auto e : c->ks_start) (void)hipEventDestroy(e);
  for (auto e : c->ks_stop) (void)hipEventDestroy(e);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

void ef_default_config(ef_config* cfg) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->width = 640; cfg->height = 480;                          // MainController.cpp:37
  cfg->fx = 528; cfg->fy = 528; cfg->cx = 320; cfg->cy = 240;   // MainController.cpp:42
  cfg->time_delta = 2147483647 / 2;                             // -o, MainController.cpp:179-183
  cfg->confidence = 10.0f;                                      // MainController.cpp:69
  cfg->depth_cut = 3.0f;                                        // MainController.cpp:70
  cfg->icp_weight = 10.0f;                                      // MainController.cpp:71
  cfg->fast_odom = 0; cfg->so3 = 1; cfg->frame_to_frame_rgb = 0; cfg->pyramid = 1; cfg->rgb_only = 0;
  cfg->close_loops = 0;
  cfg->max_surfels = 4u * 1024u * 1024u;
  cfg->device = 0;
  cfg->stream = nullptr;
}

int ef_create(const ef_config* cfg, ef_ctx** out) {
  if (!cfg || !out) { g_create_error = "null argument"; return EF_EINVAL; }
  if (cfg->width <= 0 || cfg->height <= 0 || (cfg->width % 4) || (cfg->height % 4) || cfg->fx <= 0 || cfg->fy <= 0) {
    g_create_error = "width/height must be positive multiples of 4 and focal lengths positive";
    return EF_EINVAL;
  }
  if ((size_t)cfg->max_surfels < (size_t)cfg->width * cfg->height) { g_create_error = "max_surfels must be >= width*height"; return EF_EINVAL; }
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    g_create_error = std::string("no HIP device: ") + hipGetErrorString(e);
    return EF_EHIP;
  }
  if (cfg->device < 0 || cfg->device >= ndev) { g_create_error = "bad device ordinal"; return EF_EINVAL; }
  ef_ctx* c = new ef_ctx();
  c->cfg = *cfg;
  DeviceGuard dg_(c);   // the context's device for the allocations below; the caller's current device is restored on return
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != cfg->device) { g_create_error = "hipSetDevice failed"; delete c; return EF_EHIP; }
  }
  c->cam = efm::Cam{cfg->width, cfg->height, cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  c->intr = eft::Intr{cfg->fx, cfg->fy, cfg->cx, cfg->cy};
  {
    // the persistent tracker launch needs its workgroups resident together, one per CU (fast order: 256, the reference-order launch of
    // the small levels: 128): on a device (or a partition of one: CPX mode exposes 32 CUs) that cannot hold them the per-step script is
    // the default; ef_set_persistent_tracker can still ask for it (the fast order's launch then falls back to one workgroup every frame)
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess) cus = 0;
    if (cus < 256) {
#ifdef EF_FAST_ORDER
      c->persistent = 0;
#else
      c->persistent = cus >= eft::PT_WGS ? 2 : 0;   // round 3's launch of the small levels needs 128 co-resident workgroups
#endif
    }
  }
  if (cfg->stream) {
    c->stream = (hipStream_t)cfg->stream;
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(e); delete c; return EF_EHIP; }
    c->own_stream = true;
  }
  c->bil_table = efm::bilateral_table();
  if (!c->bil_table) {   // the depth filter's weight table of this device (built once per process and device, here at the latest)
    g_create_error = "bilateral weight table: allocation or launch failed";
    if (c->own_stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return EF_EHIP;
  }
  const int r = ctx_init(c);
  if (r != EF_OK) {
    g_create_error = c->err;
    ctx_free(c);
    delete c;
    return r;
  }
  *out = c;
  return EF_OK;
}

void ef_destroy(ef_ctx* c) {
  if (!c) return;
  DeviceGuard dg_(c);
  ctx_free(c);
  delete c;
}
const char* ef_last_error(const ef_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }
void* ef_stream(ef_ctx* c) { return c ? (void*)c->stream : nullptr; }
static int check_capacity(ef_ctx* c);
int ef_synchronize(ef_ctx* c) {
  if (!c) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  {
    const int rf = flush_end_record(c);
    if (rf != EF_OK) return rf;
  }
  for (const eft::Pyramid* p : {&c->pyr, &c->pyr2, &c->pyr3}) {
    const int a = eft::tracker_aborted(*p, c->stream);
    if (a != 0) {
      c->err = a > 0 ? "a persistent tracker launch timed out waiting for another workgroup AFTER its whole grid had reported in (with the fast "
                       "order this is a protocol failure, not a busy chip: a chip partly taken is handled by the one-workgroup fallback, "
                       "ef_get_tracker_fallbacks): results since then are invalid; recreate the context, or run it with "
                       "ef_set_persistent_tracker(ctx, 0)"
                     : "hipMemcpy (tracker status)";
      return EF_EHIP;
    }
  }
  return check_capacity(c);
}

int ef_process_frame(ef_ctx* c, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp, float wm, const double* T) {
  if (!c || !rgb || !depth) return EF_EINVAL;
  DeviceGuard dg_(c);
  const size_t P = (size_t)c->cam.cols * c->cam.rows;
  // The reference uploads inside processFrame (ElasticFusion.cpp:278-280: three texture uploads, blocking).  Here (round 6): the caller's images go
  // into slot r of a ring of pinned staging pairs, are uploaded on copy_stream into the ring's device landing pair — beside the previous
  // frames' kernels — and the frame script reads the landing pair IN PLACE (the bilateral filter reads the depth, the intensity kernel reads the
  // colours and keeps the context's copy: no device-side copy, exactly the device-pointer script).  The compute stream waits for the upload
  // (one barrier packet); nothing waits on the host unless the caller is a whole ring ahead of the GPU: slot r is free once the frame that used
  // it last has been read, which that frame's prediction launch reports through a host-mapped word (no event on the compute stream: an event
  // record there costs a 6 us bubble, profiles/r06f_timeline_single_stream.txt).
  const int r = (int)(c->host_seq % ef_ctx::RING);
  if (c->host_seq >= (unsigned)ef_ctx::RING) {
    const unsigned need = c->host_seq - ef_ctx::RING + 1u;   // the marker the last user of slot r leaves
    volatile unsigned* seen = c->h_consumed;
    int spins = 0;
    while ((int)(*seen - need) < 0) {
      if (++spins > 20000) {   // (~ms: that frame ended early on an error path, or the marker is not visible: join the streams instead)
        EF_HIP(c, hipStreamSynchronize(c->stream));
        break;
      }
      if (spins > 64) std::this_thread::yield();
    }
  }
  memcpy(c->h_rgb_ring[r], rgb, P * 3);
  memcpy(c->h_depth_ring[r], depth, P * 2);
  EF_HIP(c, hipMemcpyAsync(c->d_depth_ring[r], c->h_depth_ring[r], P * 2, hipMemcpyHostToDevice, c->copy_stream));
  EF_HIP(c, hipMemcpyAsync(c->d_rgb_ring[r], c->h_rgb_ring[r], P * 3, hipMemcpyHostToDevice, c->copy_stream));
  EF_HIP(c, hipEventRecord(c->ev_h2d[r], c->copy_stream));
  c->mark_value = c->host_seq + 1u;
  c->host_seq++;
  return process_frame(c, c->d_rgb_ring[r], c->d_depth_ring[r], hipMemcpyDeviceToDevice, timestamp, wm, T, c->ev_h2d[r]);
}
int ef_process_frame_dev(ef_ctx* c, const uint8_t* rgb_dev, const uint16_t* depth_dev, int64_t timestamp, float wm, const double* T) {
  if (!c || !rgb_dev || !depth_dev) return EF_EINVAL;
  DeviceGuard dg_(c);
  return process_frame(c, rgb_dev, depth_dev, hipMemcpyDeviceToDevice, timestamp, wm, T);
}
// The input stream restricted to every n-th CU (hipExtStreamCreateWithCUMask; n <= 1: the whole chip): the next frame's bilateral filter —
// the one ALU-bound kernel of a frame — then runs beside the previous frame's fusion and prediction on a quarter of the chip instead of
// flooding every CU the latency-bound map kernels are trying to run on.
int ef_set_input_cu_mask(ef_ctx* c, int one_in_n) {
  if (!c || one_in_n < 0 || one_in_n > 32) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  if (c->in_stream) { EF_HIP(c, hipStreamSynchronize(c->in_stream)); EF_HIP(c, hipStreamDestroy(c->in_stream)); c->in_stream = nullptr; }
  if (one_in_n <= 1) {
    EF_HIP(c, hipStreamCreateWithFlags(&c->in_stream, hipStreamNonBlocking));
    return EF_OK;
  }
  int cus = 0;
  EF_HIP(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->cfg.device));
  std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
  for (int i = 0; i < cus; i += one_in_n) mask[(size_t)i / 32] |= 1u << (i % 32);
  EF_HIP(c, hipExtStreamCreateWithCUMask(&c->in_stream, (uint32_t)mask.size(), mask.data()));
  return EF_OK;
}
int ef_set_input_overlap(ef_ctx* c, int on) {
  if (!c || on < 0 || on > 2) return EF_EINVAL;
  c->overlap = on != 0;
  c->overlap_mode = on == 2 ? 2 : 1;
  c->events_live = false;   // (the next overlapped frame joins the streams on the host: the events of the other mode were not recorded)
  return EF_OK;
}
int ef_set_deformation(ef_ctx* c, const float* graph, int nodes, int is_fern) {
  if (!c || nodes < 0 || (nodes > 0 && !graph)) return EF_EINVAL;
  DeviceGuard dg_(c);
  if (nodes >= 1024) { c->err = "ef_set_deformation: at most 1023 nodes (GlobalModel::MAX_NODES)"; return EF_EINVAL; }
  if (nodes > 0) EF_HIP(c, hipMemcpyAsync(c->graph_dev, graph, (size_t)nodes * 16 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  if (nodes > 0) EF_HIP(c, hipStreamSynchronize(c->stream));   // the caller's buffer is borrowed for the call only
  c->graph_nodes = nodes;
  c->graph_is_fern = is_fern != 0;
  return EF_OK;
}
int ef_set_loop_thresholds(ef_ctx* c, int icp_count_thresh, float icp_err_thresh, float cov_thresh) {
  if (!c) return EF_EINVAL;
  c->icp_count_thresh = icp_count_thresh; c->icp_err_thresh = icp_err_thresh; c->cov_thresh = cov_thresh;
  return EF_OK;
}
int ef_set_loop_solver(ef_ctx* c, ef_loop_solver fn, void* user) {
  if (!c) return EF_EINVAL;
  if (!c->cfg.close_loops) { c->err = "ef_set_loop_solver: the context was created with close_loops = 0"; return EF_ESTATE; }
  c->solver = fn; c->solver_user = user;
  return EF_OK;
}
int ef_use_builtin_loop_solver(ef_ctx* c, int on) {
  if (!c) return EF_EINVAL;
  if (!c->cfg.close_loops) { c->err = "ef_use_builtin_loop_solver: the context was created with close_loops = 0"; return EF_ESTATE; }
  c->builtin_solver = on != 0;
  return EF_OK;
}
int ef_solve_local_deformation(const float* nodes4, int n_nodes, const double* constraints8, int n_constraints, int64_t src_time,
                               int64_t last_deform_time, float* graph16_out, float* error_out, float* mean_constraint_error_out) {
  if (!nodes4 || !constraints8 || !graph16_out || n_nodes < 0 || n_nodes > 1023 || n_constraints < 0) return EF_EINVAL;
  const efd::Result r = efd::solve_local(nodes4, n_nodes, constraints8, n_constraints, (uint64_t)src_time, (uint64_t)last_deform_time, graph16_out);
  if (error_out) *error_out = r.error;
  if (mean_constraint_error_out) *mean_constraint_error_out = r.meanConsErr;
  return r.ok ? EF_OK : EF_ESTATE;
}
int ef_solve_deformation(const float* nodes4, int n_nodes, const ef_graph_constraint* constraints, int n_constraints, int fern_match,
                         int64_t last_deform_time, double* poses16, const int64_t* pose_times, int n_poses, float* graph16_out, float* error_out,
                         float* mean_constraint_error_out, ef_graph_constraint* new_relative_out, int* n_new_relative_out) {
  return ef_solve_deformation_gated(nodes4, n_nodes, constraints, n_constraints, fern_match, last_deform_time, poses16, pose_times, n_poses, graph16_out,
                                    error_out, mean_constraint_error_out, new_relative_out, n_new_relative_out, nullptr);
}
int ef_solve_deformation_gated(const float* nodes4, int n_nodes, const ef_graph_constraint* constraints, int n_constraints, int fern_match,
                               int64_t last_deform_time, double* poses16, const int64_t* pose_times, int n_poses, float* graph16_out, float* error_out,
                               float* mean_constraint_error_out, ef_graph_constraint* new_relative_out, int* n_new_relative_out, const float* gates3) {
  if (!nodes4 || !constraints || !graph16_out || n_nodes < 0 || n_nodes > 1023 || n_constraints < 0 || n_poses < 0 || (n_poses > 0 && (!poses16 || !pose_times)))
    return EF_EINVAL;
  std::vector<efd::Constraint> cons((size_t)n_constraints);
  for (int i = 0; i < n_constraints; ++i) {
    const ef_graph_constraint& c = constraints[i];
    cons[i] = efd::Constraint{{c.src[0], c.src[1], c.src[2]}, {c.target[0], c.target[1], c.target[2]}, (uint64_t)c.src_time, (uint64_t)c.target_time,
                              c.relative != 0, c.pin != 0};
  }
  efd::Result r{false, 0, 0.f, 0.f};
  std::vector<efd::Constraint> rel;
  const efd::Gates gates = gates3 ? efd::Gates{gates3[0], gates3[1], gates3[2]} : efd::Gates();
  const bool updated = efd::constrain(nodes4, n_nodes, cons.data(), n_constraints, fern_match != 0, (uint64_t)last_deform_time, poses16, pose_times, n_poses,
                                      graph16_out, &r, new_relative_out ? &rel : nullptr, gates);
  if (n_new_relative_out) *n_new_relative_out = (int)rel.size();
  for (size_t i = 0; new_relative_out && i < rel.size(); ++i) {
    ef_graph_constraint& o = new_relative_out[i];
    for (int k = 0; k < 3; ++k) { o.src[k] = rel[i].src[k]; o.target[k] = rel[i].target[k]; }
    o.src_time = (int64_t)rel[i].srcTime; o.target_time = (int64_t)rel[i].targetTime; o.relative = 1; o.pin = 0;
  }
  if (error_out) *error_out = r.error;
  if (mean_constraint_error_out) *mean_constraint_error_out = r.meanConsErr;
  return updated ? EF_OK : EF_ESTATE;
}
int ef_enable_global_closure(ef_ctx* c, int num_ferns, float photo_thresh, float fern_thresh, unsigned seed) {
  if (!c || num_ferns < 50) return EF_EINVAL;
  if (num_ferns > FERN_CODES_PAD) { c->err = "ef_enable_global_closure: at most 512 ferns"; return EF_EINVAL; }   // (every refusal before the first allocation)
  DeviceGuard dg_(c);
  if (!c->cfg.close_loops) { c->err = "ef_enable_global_closure: the context was created with close_loops = 0"; return EF_ESTATE; }
  if (c->closure) { c->err = "ef_enable_global_closure: already enabled"; return EF_ESTATE; }
  const int W = c->cam.cols, H = c->cam.rows;
  if ((W / 8) % 4 || (H / 8) % 4 || W % 8 || H % 8) { c->err = "ef_enable_global_closure: width and height must be multiples of 32"; return EF_EINVAL; }
  c->fern_w = W / 8;
  c->fern_h = H / 8;
  const size_t n = (size_t)c->fern_w * c->fern_h;
  EF_ALLOC(c, c->view_img_dev, n);
  EF_ALLOC(c, c->view_vert_dev, n);
  EF_ALLOC(c, c->view_norm_dev, n);
  EF_ALLOC(c, c->fern_maps_dev, 5 * n);   // 4 maps + a zero image (the 1/8 tracker never reads colour: icpWeight = 100)
  EF_HIP(c, hipHostMalloc((void**)&c->h_view, n * 36));
  EF_HIP(c, hipHostMalloc((void**)&c->h_view_end, n * 36));
  EF_HIP(c, hipHostMalloc((void**)&c->h_codes, FERN_CODES_BYTES));
  EF_HIP(c, hipHostMalloc((void**)&c->h_codes_end, FERN_CODES_BYTES));
  c->fern_num = num_ferns;
  EF_ALLOC(c, c->fern_table_dev, (size_t)num_ferns * 6);
  EF_ALLOC(c, c->fern_codes_dev, (size_t)FERN_CODES_BYTES);
  EF_HIP(c, hipEventCreateWithFlags(&c->ev_end_record, hipEventDisableTiming));
  EF_HIP(c, hipHostMalloc((void**)&c->h_nodes_pinned, ((size_t)1024 * 4 + 4) * sizeof(float)));
  memset(c->h_nodes_pinned, 0, ((size_t)1024 * 4 + 4) * sizeof(float));
  c->intr3 = eft::Intr{c->cfg.fx / 8, c->cfg.fy / 8, c->cfg.cx / 8, c->cfg.cy / 8};   // Ferns.cpp:31-36
  c->pyr3.width = c->fern_w;
  c->pyr3.height = c->fern_h;
  for (int i = 0; i < eft::NUM_PYRS; ++i) {
    const size_t m = (size_t)(c->fern_w >> i) * (c->fern_h >> i);
    EF_ALLOC(c, c->pyr3.depth_tmp[i], m);
    EF_ALLOC(c, c->pyr3.vmap_curr[i], 3 * m);
    EF_ALLOC(c, c->pyr3.nmap_curr[i], 3 * m);
    EF_ALLOC(c, c->pyr3.vmap_g_prev[i], 3 * m);
    EF_ALLOC(c, c->pyr3.nmap_g_prev[i], 3 * m);
    EF_ALLOC(c, c->pyr3.lastDepth[i], m);
    EF_ALLOC(c, c->pyr3.nextDepth[i], m);
    EF_ALLOC(c, c->pyr3.lastImage[i], m);
    EF_ALLOC(c, c->pyr3.nextImage[i], m);
    EF_ALLOC(c, c->pyr3.lastNextImage[i], m);
    EF_ALLOC(c, c->pyr3.dIdx[i], m);
    EF_ALLOC(c, c->pyr3.dIdy[i], m);
    EF_ALLOC(c, c->pyr3.corres[i], m);
    EF_ALLOC(c, c->pyr3.rgbMask[i], m);
  }
  EF_ALLOC(c, c->pyr3.partials, (size_t)eft::PARTIAL_ALLOC_FLOATS);
  EF_ALLOC(c, c->st3, 1);
  hipLaunchKernelGGL(k_init_state, dim3(1), dim3(64), 0, c->stream, c->st3, 1, c->fern_w * c->fern_h);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  memset(&c->gloop, 0, sizeof(c->gloop));
  c->gloop.closest = -1;
  c->closure = ef_closure_create(num_ferns, c->cfg.depth_cut, photo_thresh, fern_thresh, W, H, c->cfg.fx, c->cfg.fy, c->cfg.cx, c->cfg.cy, seed);
  if (!c->closure) { c->err = "ef_closure_create failed"; return EF_ENOMEM; }
  return EF_OK;
}
int ef_set_relocalisation(ef_ctx* c, int on) {
  if (!c) return EF_EINVAL;
  c->reloc = on != 0;
  if (!c->reloc) { c->lost = false; c->last_frame_recovery = false; c->tracking_count = 0; c->tracking_ok = true; }
  return EF_OK;
}
int ef_get_relocalisation(ef_ctx* c, ef_reloc_state* out) {
  if (!c || !out) return EF_EINVAL;
  out->lost = c->lost; out->tracking_ok = c->tracking_ok; out->tracking_count = c->tracking_count; out->last_frame_recovery = c->last_frame_recovery;
  return EF_OK;
}
int ef_get_global_loop(ef_ctx* c, ef_global_loop* info) {
  if (!c || !info) return EF_EINVAL;
  *info = c->gloop;
  return EF_OK;
}
ef_closure* ef_get_closure(ef_ctx* c) {
  if (!c) return nullptr;
  DeviceGuard dg_(c);
  (void)flush_end_record(c);   // the last frame's keyframe decision and trajectory entry are part of what the caller will look at
  return c->closure;
}
int ef_get_local_loop(ef_ctx* c, ef_local_loop* info, double* constraints, int max_constraints, int* n_out) {
  if (!c || !info) return EF_EINVAL;
  *info = c->loop;
  int n = c->loop.n_constraints < max_constraints ? c->loop.n_constraints : max_constraints;
  if (!constraints) n = 0;
  if (n > 0) memcpy(constraints, c->loop_constraints.data(), (size_t)n * 8 * sizeof(double));
  if (n_out) *n_out = n;
  return EF_OK;
}
int ef_sample_graph(ef_ctx* c, float* nodes4, int max_nodes, int* n_out) {
  if (!c || !nodes4 || !n_out || max_nodes <= 0) return EF_EINVAL;
  DeviceGuard dg_(c);
  float* dev = nullptr;
  EF_HIP(c, hipMalloc((void**)&dev, ((size_t)max_nodes * 4 + 4) * sizeof(float)));
  unsigned* n_dev = (unsigned*)(dev + (size_t)max_nodes * 4);
  efm::sample_graph(c->maps[c->cur], &c->st->map_counts[c->cur], 5000, max_nodes, dev, n_dev, c->stream);
  unsigned n = 0;
  hipError_t e = hipMemcpyAsync(&n, n_dev, sizeof(n), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && n > 0) e = hipMemcpy(nodes4, dev, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(dev);
  EF_HIP(c, e);
  *n_out = (int)n;
  return EF_OK;
}
int ef_set_graph_replay(ef_ctx* c, int on) { if (!c) return EF_EINVAL; c->use_graph = on != 0; return EF_OK; }
int ef_set_fused_step(ef_ctx* c, int on) {
  if (!c) return EF_EINVAL;
  c->fused_step = on != 0;
  for (auto& g : c->tgraph)
    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  return EF_OK;
}
int ef_set_track_only(ef_ctx* c, int on) { if (!c) return EF_EINVAL; c->track_only = on != 0; return EF_OK; }
int ef_set_resident_levels(ef_ctx* c, int on) { if (!c) return EF_EINVAL; c->no_resident = on == 0; return EF_OK; }
int ef_set_persistent_tracker(ef_ctx* c, int on) {
  if (!c) return EF_EINVAL;
  c->persistent = on < 0 ? 0 : (on > 2 ? 1 : on);
#ifdef EF_FAST_ORDER
  if (c->persistent == 2) c->persistent = 1;   // (the fast order has no launch of the small levels: 2 means 1 there, also to process_frame's overlap rule)
#endif
  for (auto& g : c->tgraph)   // captured tracker graphs hold the other script
    if (g.exec) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }
  return EF_OK;
}
int ef_predict(ef_ctx* c) {
  if (!c) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipMemsetAsync(&c->st->dense_count, 0, sizeof(unsigned), c->stream));
  return do_predict(c);
}
int ef_get_pose(ef_ctx* c, double* T16) {
  if (!c || !T16) return EF_EINVAL;
  DeviceGuard dg_(c);
  eft::TrackState h;
  EF_HIP(c, hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = h.q[i];
  for (int i = 0; i < 3; ++i) T.t[i] = h.t[i];
  efl::se3_matrix(T, T16);
  return EF_OK;
}
int ef_get_tick(ef_ctx* c, int* tick) { if (!c || !tick) return EF_EINVAL; *tick = c->tick; return EF_OK; }
int ef_set_tick(ef_ctx* c, int tick) { if (!c) return EF_EINVAL; c->tick = tick; return EF_OK; }
int ef_get_tracking_stats(ef_ctx* c, float* out6, double* A36, double* b6) {
  if (!c || !out6) return EF_EINVAL;
  DeviceGuard dg_(c);
  eft::TrackState h;
  EF_HIP(c, hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  out6[0] = h.lastICPError; out6[1] = h.lastICPCount; out6[2] = h.lastRGBError;
  out6[3] = h.lastRGBCount; out6[4] = h.lastSO3Error; out6[5] = h.lastSO3Count;
  if (A36) memcpy(A36, h.lastA, sizeof(h.lastA));
  if (b6) memcpy(b6, h.lastb, sizeof(h.lastb));
  return EF_OK;
}
int ef_get_covariance(ef_ctx* c, double* cov36) {
  if (!c || !cov36) return EF_EINVAL;
  DeviceGuard dg_(c);
  eft::TrackState h;
  EF_HIP(c, hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  efl::lu_inverse<double, 6>(h.lastA, cov36);   // host side, like the reference (Eigen on the CPU)
  return EF_OK;
}
int ef_get_tracker_fallbacks(ef_ctx* c, int* count) {
  if (!c || !count) return EF_EINVAL;
  DeviceGuard dg_(c);
  int total = 0;
  for (const eft::Pyramid* p : {&c->pyr, &c->pyr2, &c->pyr3}) {
    const int n = eft::tracker_fallbacks(*p, c->stream);
    if (n < 0) { c->err = "reading the tracker's fallback counter failed"; return EF_EHIP; }
    total += n;
  }
  *count = total;
  return EF_OK;
}
// developer instrumentation (tests/test_gpu_fallback.py): `workgroups` workgroups of 1024 threads and 128 registers per lane — each fills the
// register file of a whole CU — spin for `microseconds` on a stream of their own: the chip is partly taken, as by another process
__global__ void __launch_bounds__(1024) k_debug_occupy(unsigned long long ticks, unsigned* started) {
  asm volatile("v_mov_b32 v127, 0" ::: "v127");
  if (threadIdx.x == 0) __hip_atomic_fetch_add(started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
int ef_debug_occupy(ef_ctx* c, int workgroups, int microseconds) {
  if (!c || workgroups < 1 || workgroups > 1024 || microseconds < 1 || microseconds > 2000000) return EF_EINVAL;
  DeviceGuard dg_(c);
  // (a stream of ANOTHER priority: the runtime multiplexes the streams of a process onto a few hardware queues, and a spinner that sits in the
  // queue the compute or the copy stream maps to holds up the frame itself — what the spinners stand for, another process, has queues of its own)
  if (!c->debug_stream) {
    int lo = 0, hi = 0;
    EF_HIP(c, hipDeviceGetStreamPriorityRange(&lo, &hi));
    EF_HIP(c, hipStreamCreateWithPriority(&c->debug_stream, hipStreamNonBlocking, hi));
  }
  // The call returns once every spinner IS resident (they report in through a host-mapped word): a spinner needs a whole idle CU, and since round 6
  // the frame script leaves no bubble in which one could slip in — enqueued right before a frame, the spinners would start behind it and the
  // test would exercise nothing.  The compute stream is drained first so that they find the chip idle.
  EF_HIP(c, hipStreamSynchronize(c->stream));
  c->h_abort[3] = 0u;
  hipLaunchKernelGGL(k_debug_occupy, dim3(workgroups), dim3(1024), 0, c->debug_stream, (unsigned long long)microseconds * 100ull, c->d_abort + 3);
  EF_HIP(c, hipGetLastError());
  volatile unsigned* started = c->h_abort + 3;
  for (int spin = 0; spin < 2000000 && *started < (unsigned)workgroups; ++spin) std::this_thread::yield();
  if (*started < (unsigned)workgroups) { c->err = "ef_debug_occupy: the spinners did not become resident"; return EF_EHIP; }
  return EF_OK;
}
// test hook: raises the sticky abort flag of the frame tracker's persistent launches, as a wait that timed out would
int ef_debug_inject_tracker_abort(ef_ctx* c) {
  if (!c) return EF_EINVAL;
  DeviceGuard dg_(c);
  unsigned* w = eft::tracker_abort_word(c->pyr);
  if (!w) return EF_EINVAL;
  const unsigned one = 1u;
  EF_HIP(c, hipMemcpyAsync(w, &one, sizeof(one), hipMemcpyHostToDevice, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  return EF_OK;
}
// developer instrumentation: per-phase clocks of the persistent small-level launch (-DEF_STAGE_CLOCKS builds; tools/small_clocks.py)
int ef_debug_small_clocks(ef_ctx* c, unsigned long long* out32) {
  if (!c || !out32) return EF_EINVAL;
  DeviceGuard dg_(c);
  return eft::tracker_small_clocks(c->pyr, out32, c->stream) == 0 ? EF_OK : EF_EHIP;
}
int ef_debug_clocks(ef_ctx* c, unsigned long long* out16) {
  if (!c || !out16) return EF_EINVAL;
  DeviceGuard dg_(c);
  eft::TrackState h;
  EF_HIP(c, hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(out16, h.dbg_clock, sizeof(h.dbg_clock));
  return EF_OK;
}
int ef_get_trajectory(ef_ctx* c, double* T16s, int64_t* stamps, int max_frames, int* n_frames) {
  if (!c || !n_frames) return EF_EINVAL;
  DeviceGuard dg_(c);
  int n = (int)c->stamps.size();
  if (n > max_frames) n = max_frames;
  if (T16s && n) {
    const int rf = flush_end_record(c);
    if (rf != EF_OK) return rf;
    EF_HIP(c, hipMemcpyAsync(T16s, c->traj, (size_t)n * 16 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    EF_HIP(c, hipStreamSynchronize(c->stream));
    // "Output deformed pose graph" (ElasticFusion.cpp:107-139): every accepted Deformation::constrain moves the poses logged so far
    // (DeformationGraph::applyGraphToPoses), so with loop closures on the log to hand out is the closure object's copy of t_T_wc — the
    // device log holds each frame's pose as it was when the frame ended.  The closure object logs one pose per frame from the frame
    // ef_enable_global_closure was called before: it covers the LAST `m` frames.
    if (c->closure) {
      const int total = (int)c->stamps.size(), m = ef_closure_trajectory(c->closure, nullptr, 0), first = total - m;
      if (m > 0 && first >= 0 && first < n) {
        std::vector<double> P((size_t)m * 16);
        ef_closure_trajectory(c->closure, P.data(), m);
        memcpy(T16s + (size_t)first * 16, P.data(), (size_t)(n - first) * 16 * sizeof(double));
      }
    }
  }
  if (stamps) for (int i = 0; i < n; ++i) stamps[i] = c->stamps[i];
  *n_frames = n;
  return EF_OK;
}
// clean() clamps the new surfel count to the capacity and raises a device flag (the reference's fixed 3072 x 3072 vertex
// buffer simply overflows, GlobalModel.cpp:22-24).  The flag is a WARNING: ef_synchronize reports it once (EF_ECAPACITY) and
// clears it; the clamped map stays readable (ef_map_count / ef_map_download / ef_save_ply proceed with the clamped count).
static int check_capacity(ef_ctx* c) {
  int flag = 0;
  EF_HIP(c, hipMemcpyAsync(&flag, c->overflow, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  if (flag) {
    EF_HIP(c, hipMemsetAsync(c->overflow, 0, sizeof(int), c->stream));
    c->err = "surfel capacity exceeded (ef_config.max_surfels = " + std::to_string(c->capacity) + "): the newest surfels were dropped";
    return EF_ECAPACITY;
  }
  return EF_OK;
}
int ef_map_count(ef_ctx* c, uint32_t* count) {
  if (!c || !count) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipMemcpyAsync(count, &c->st->map_counts[c->cur], sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  return EF_OK;
}
int ef_map_download(ef_ctx* c, float* surfels, uint32_t max_surfels, uint32_t* count) {
  if (!c || !count) return EF_EINVAL;
  DeviceGuard dg_(c);
  uint32_t n = 0;
  int r = ef_map_count(c, &n);
  if (r != EF_OK) return r;
  if (n > max_surfels) n = max_surfels;
  *count = n;
  if (!surfels || !n) return EF_OK;
  // The reference's downloadMap() reads the buffer its update pass wrote — the map BEFORE clean — truncated to the count AFTER
  // clean (quirk Q14); by default this returns model(), the map as it stands; ef_set_reference_download selects the reference's.
  float* tmp = nullptr;
  EF_HIP(c, hipMalloc((void**)&tmp, (size_t)n * 48));
  efm::soa_to_aos(c->reference_download ? c->shadow : c->maps[c->cur], n, tmp, c->stream);
  hipError_t e = hipMemcpyAsync(surfels, tmp, (size_t)n * 48, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  EF_HIP(c, e);
  return EF_OK;
}
int ef_set_reference_download(ef_ctx* c, int on) {
  if (!c) return EF_EINVAL;
  DeviceGuard dg_(c);
  if (on && !c->shadow.pos_conf) {   // zero-filled like the reference's vertex buffers (GlobalModel.cpp:72-75)
    EF_ALLOC(c, c->shadow.pos_conf, c->capacity);
    EF_ALLOC(c, c->shadow.col_time, c->capacity);
    EF_ALLOC(c, c->shadow.nrm_rad, c->capacity);
  }
  c->reference_download = on != 0;
  return EF_OK;
}
int ef_map_upload(ef_ctx* c, const float* surfels, uint32_t count) {
  if (!c || (!surfels && count)) return EF_EINVAL;
  DeviceGuard dg_(c);
  if (count > c->capacity) { c->err = "ef_map_upload: count exceeds max_surfels"; return EF_ECAPACITY; }
  float* tmp = nullptr;
  if (count) {
    EF_HIP(c, hipMalloc((void**)&tmp, (size_t)count * 48));
    hipError_t e = hipMemcpyAsync(tmp, surfels, (size_t)count * 48, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
      efm::aos_to_soa(tmp, count, c->maps[c->cur], c->stream);
      e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(tmp);
    EF_HIP(c, e);
  }
  hipLaunchKernelGGL(k_set_count, dim3(1), dim3(64), 0, c->stream, &c->st->map_counts[c->cur], count);
  return EF_OK;
}
int ef_get_pose_qt(ef_ctx* c, double* q4_t3) {
  if (!c || !q4_t3) return EF_EINVAL;
  DeviceGuard dg_(c);
  eft::TrackState h;
  EF_HIP(c, hipMemcpyAsync(&h, c->st, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  for (int i = 0; i < 4; ++i) q4_t3[i] = h.q[i];
  for (int i = 0; i < 3; ++i) q4_t3[4 + i] = h.t[i];
  return EF_OK;
}
// Resume from a checkpoint (include/ef_hip.h): the end-of-frame state of the frame `rgb_prev` / `depth_prev` was, on the uploaded map
int ef_restore_state(ef_ctx* c, int tick, const double* q4_t3, const uint8_t* rgb_prev, const uint16_t* depth_prev) {
  if (!c || !q4_t3 || !rgb_prev || !depth_prev || tick < 2) return EF_EINVAL;
  DeviceGuard dg_(c);
  hipStream_t s = c->stream;
  const int W = c->cam.cols, H = c->cam.rows;
  if (c->closure) {
    // the fern database and the pose graph are not part of a checkpoint: a record still pending from before the restore belongs to the OLD
    // replay and goes where it was headed before tick and pose change under it
    const int fr = flush_end_record(c);
    if (fr != EF_OK) return fr;
  }
  EF_HIP(c, hipMemcpyAsync(c->rgb, rgb_prev, (size_t)W * H * 3, hipMemcpyHostToDevice, s));
  EF_HIP(c, hipMemcpyAsync(c->depth_raw, depth_prev, (size_t)W * H * 2, hipMemcpyHostToDevice, s));
  EF_HIP(c, hipStreamSynchronize(s));   // the caller's buffers are pageable and only borrowed
  if (!efm::preprocess_depth(c->depth_raw, W, H, c->cfg.depth_cut, c->depth_filtered, c->depth_metric, c->depth_metric_filtered, s, 0u, nullptr,
                             nullptr, nullptr, c->bil_table)) {
    c->err = "bilateral weight table missing on this device";
    return EF_EHIP;
  }
  // the frame's intensity pyramid where the next frame's SO(3) pre-alignment looks for it (lastNextImage: initFirstRGB's target and,
  // after every tracked frame, the swapped-in nextImage, RGBDOdometry.cpp:246-257,284-288)
  eft::init_first_rgb(c->pyr, c->rgb, s);
  efl::SE3 T;
  for (int i = 0; i < 4; ++i) T.q[i] = q4_t3[i];
  for (int i = 0; i < 3; ++i) T.t[i] = q4_t3[4 + i];
  eft::pose_restored(c->st, T.q, T.t, s);
  c->tick = tick;
  c->lost = c->last_frame_recovery = false;
  c->tracking_ok = true;
  c->tracking_count = 0;
  c->graph_nodes = 0;
  const int r = do_predict(c);
  EF_HIP(c, hipGetLastError());
  return r;
}
// Host-only writers (no context, no GPU): the two dumps of the reference, byte for byte.
//   trajectory: ~ElasticFusion, ElasticFusion.cpp:112-139 — "timestamp tx ty tz qx qy qz qw" per pose, the timestamp as microseconds / 1e6
//     with six decimals, the seven numbers as an ostream prints a double by default (six significant digits, %g);
//   map: ElasticFusion::savePly, :684-781 — binary little-endian PLY of the surfels with confidence above the threshold:
//     x y z, r g b unpacked from the colour float, the NEGATED normal (:741-743), the radius.
int ef_write_freiburg(const char* path, const double* T_wc16_array, const int64_t* timestamps, int n) {
  if (!path || (n > 0 && (!T_wc16_array || !timestamps))) return EF_EINVAL;
  FILE* f = fopen(path, "w");
  if (!f) return EF_EINVAL;
  for (int i = 0; i < n; ++i) {
    const efl::SE3 S = efl::se3_from_matrix(T_wc16_array + (size_t)i * 16);
    fprintf(f, "%.6f %g %g %g %g %g %g %g\n", (double)timestamps[i] / 1000000.0, S.t[0], S.t[1], S.t[2], S.q[0], S.q[1], S.q[2], S.q[3]);
  }
  fclose(f);
  return EF_OK;
}
int ef_write_ply(const char* path, const float* surfels, uint32_t count, float confidence_threshold) {
  if (!path || (count > 0 && !surfels)) return EF_EINVAL;
  uint32_t valid = 0;
  for (uint32_t i = 0; i < count; ++i) valid += surfels[(size_t)i * 12 + 3] > confidence_threshold;
  FILE* f = fopen(path, "wb");
  if (!f) return EF_EINVAL;
  fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %u\nproperty float x\nproperty float y\nproperty float z\n"
             "property uchar red\nproperty uchar green\nproperty uchar blue\nproperty float nx\nproperty float ny\nproperty float nz\n"
             "property float radius\nend_header\n", valid);
  for (uint32_t i = 0; i < count; ++i) {
    const float* s = surfels + (size_t)i * 12;
    if (!(s[3] > confidence_threshold)) continue;
    const int col = (int)s[4];
    const unsigned char rgbc[3] = {(unsigned char)((col >> 16) & 0xFF), (unsigned char)((col >> 8) & 0xFF), (unsigned char)(col & 0xFF)};
    const float nr[4] = {s[8] * -1, s[9] * -1, s[10] * -1, s[11]};
    fwrite(s, sizeof(float), 3, f);
    fwrite(rgbc, 1, 3, f);
    fwrite(nr, sizeof(float), 4, f);
  }
  fclose(f);
  return EF_OK;
}
int ef_save_freiburg(ef_ctx* c, const char* path) {
  if (!c || !path) return EF_EINVAL;
  DeviceGuard dg_(c);
  const int n = (int)c->stamps.size();
  std::vector<double> T((size_t)n * 16);
  int got = 0;
  int r = ef_get_trajectory(c, T.data(), nullptr, n, &got);
  if (r != EF_OK) return r;
  r = ef_write_freiburg(path, T.data(), c->stamps.data(), got);
  if (r != EF_OK) c->err = std::string("cannot open ") + path;
  return r;
}
int ef_save_ply(ef_ctx* c, const char* path) {
  if (!c || !path) return EF_EINVAL;
  DeviceGuard dg_(c);
  uint32_t n = 0;
  int r = ef_map_count(c, &n);
  if (r != EF_OK) return r;
  std::vector<float> m((size_t)n * 12);
  r = ef_map_download(c, m.data(), n, &n);
  if (r != EF_OK) return r;
  r = ef_write_ply(path, m.data(), n, c->cfg.confidence);
  if (r != EF_OK) c->err = std::string("cannot open ") + path;
  return r;
}
int ef_set_rgb_only(ef_ctx* c, int v) { if (!c) return EF_EINVAL; c->cfg.rgb_only = v; return EF_OK; }
int ef_set_icp_weight(ef_ctx* c, float v) { if (!c) return EF_EINVAL; c->cfg.icp_weight = v; return EF_OK; }
int ef_set_pyramid(ef_ctx* c, int v) { if (!c) return EF_EINVAL; c->cfg.pyramid = v; return EF_OK; }
int ef_set_fast_odom(ef_ctx* c, int v) { if (!c) return EF_EINVAL; c->cfg.fast_odom = v; return EF_OK; }
int ef_set_so3(ef_ctx* c, int v) { if (!c) return EF_EINVAL; c->cfg.so3 = v; return EF_OK; }
int ef_set_frame_to_frame_rgb(ef_ctx* c, int v) { if (!c) return EF_EINVAL; c->cfg.frame_to_frame_rgb = v; return EF_OK; }
int ef_set_confidence_threshold(ef_ctx* c, float v) { if (!c) return EF_EINVAL; c->cfg.confidence = v; return EF_OK; }
int ef_set_depth_cutoff(ef_ctx* c, float v) { if (!c) return EF_EINVAL; c->cfg.depth_cut = v; return EF_OK; }

int ef_get_image(ef_ctx* c, int which, void* dst, size_t bytes) {
  if (!c || !dst) return EF_EINVAL;
  DeviceGuard dg_(c);
  const size_t P = (size_t)c->cam.cols * c->cam.rows;
  const void* src = nullptr;
  size_t need = 0;
  switch (which) {
    case EF_IMG_DEPTH_FILTERED: src = c->depth_filtered; need = P * 2; break;
    case EF_IMG_DEPTH_METRIC: src = c->depth_metric; need = P * 4; break;
    case EF_IMG_DEPTH_METRIC_FILTERED: src = c->depth_metric_filtered; need = P * 4; break;
    case EF_IMG_PREDICT_IMAGE: src = c->pm.image; need = P * 4; break;
    case EF_IMG_PREDICT_VERTEX: src = c->pm.vertex; need = P * 16; break;
    case EF_IMG_PREDICT_NORMAL: src = c->pm.normal; need = P * 16; break;
    case EF_IMG_PREDICT_TIME: src = c->pm.time; need = P * 2; break;
    case EF_IMG_FILL_IMAGE: src = c->fm.image; need = P * 4; break;
    case EF_IMG_FILL_VERTEX: src = c->fm.vertex; need = P * 16; break;
    case EF_IMG_FILL_NORMAL: src = c->fm.normal; need = P * 16; break;
    case EF_IMG_INDEX: src = c->im.index; need = P * 4; break;
    case EF_IMG_VERT_CONF: src = c->im.vert_conf; need = P * 16; break;
    case EF_IMG_COLOR_TIME: src = c->im.color_time; need = P * 16; break;
    case EF_IMG_NORM_RAD: src = c->im.norm_rad; need = P * 16; break;
    case EF_IMG_OLD_IMAGE: src = c->old.image; need = P * 4; break;
    case EF_IMG_OLD_VERTEX: src = c->old.vertex; need = P * 16; break;
    case EF_IMG_OLD_NORMAL: src = c->old.normal; need = P * 16; break;
    case EF_IMG_OLD_TIME: src = c->old.time; need = P * 2; break;
    default: c->err = "ef_get_image: unknown image"; return EF_EINVAL;
  }
  if (!src) { c->err = "ef_get_image: this image only exists in a close_loops context"; return EF_ESTATE; }
  if (bytes < need) { c->err = "ef_get_image: destination too small"; return EF_EINVAL; }
  void* tmp = nullptr;
  if (c->im.colmajor && which >= EF_IMG_INDEX && which <= EF_IMG_NORM_RAD) {
    EF_HIP(c, hipMalloc(&tmp, need));
    const int W = c->cam.cols, H = c->cam.rows;
    const dim3 g((unsigned)((P + 255) / 256));
    if (which == EF_IMG_INDEX) hipLaunchKernelGGL(k_to_rowmajor<uint32_t>, g, dim3(256), 0, c->stream, (const uint32_t*)src, W, H, (uint32_t*)tmp);
    else hipLaunchKernelGGL(k_to_rowmajor<float4>, g, dim3(256), 0, c->stream, (const float4*)src, W, H, (float4*)tmp);
    src = tmp;
  }
  hipError_t e = hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (tmp) (void)hipFree(tmp);
  EF_HIP(c, e);
  return EF_OK;
}
int ef_get_image_resized(ef_ctx* c, int which, int factor, void* dst, size_t bytes) {
  if (!c || !dst || factor < 1) return EF_EINVAL;
  DeviceGuard dg_(c);
  const int W = c->cam.cols, H = c->cam.rows, dw = W / factor, dh = H / factor;
  const void* src = nullptr;
  int elem = 0;
  switch (which) {
    case EF_IMG_PREDICT_IMAGE: src = c->pm.image; elem = 4; break;
    case EF_IMG_PREDICT_VERTEX: src = c->pm.vertex; elem = 16; break;
    case EF_IMG_PREDICT_NORMAL: src = c->pm.normal; elem = 16; break;
    case EF_IMG_PREDICT_TIME: src = c->pm.time; elem = 2; break;
    case EF_IMG_FILL_IMAGE: src = c->fm.image; elem = 4; break;
    case EF_IMG_FILL_VERTEX: src = c->fm.vertex; elem = 16; break;
    case EF_IMG_FILL_NORMAL: src = c->fm.normal; elem = 16; break;
    case EF_IMG_OLD_IMAGE: src = c->old.image; elem = 4; break;
    case EF_IMG_OLD_VERTEX: src = c->old.vertex; elem = 16; break;
    case EF_IMG_OLD_NORMAL: src = c->old.normal; elem = 16; break;
    case EF_IMG_OLD_TIME: src = c->old.time; elem = 2; break;
    default: c->err = "ef_get_image_resized: a predicted, fill-in or inactive-prediction image"; return EF_EINVAL;
  }
  if (!src) { c->err = "ef_get_image_resized: this image only exists in a close_loops context"; return EF_ESTATE; }
  const size_t need = (size_t)dw * dh * elem;
  if (dw == 0 || dh == 0 || bytes < need) { c->err = "ef_get_image_resized: destination too small"; return EF_EINVAL; }
  void* tmp = nullptr;
  EF_HIP(c, hipMalloc(&tmp, need));
  const dim3 g((unsigned)((dw * dh + 255) / 256));
  if (elem == 16) hipLaunchKernelGGL(k_resize_nearest<float4>, g, dim3(256), 0, c->stream, (const float4*)src, W, dw, dh, factor, (float4*)tmp);
  else if (elem == 4) hipLaunchKernelGGL(k_resize_nearest<uint32_t>, g, dim3(256), 0, c->stream, (const uint32_t*)src, W, dw, dh, factor, (uint32_t*)tmp);
  else hipLaunchKernelGGL(k_resize_nearest<uint16_t>, g, dim3(256), 0, c->stream, (const uint16_t*)src, W, dw, dh, factor, (uint16_t*)tmp);
  hipError_t e = hipMemcpyAsync(dst, tmp, need, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  EF_HIP(c, e);
  return EF_OK;
}
int ef_get_tracker_buffer(ef_ctx* c, int which, int level, void* dst, size_t bytes) {
  if (!c || !dst || level < 0 || level >= eft::NUM_PYRS) return EF_EINVAL;
  DeviceGuard dg_(c);
  const size_t n = (size_t)(c->cam.cols >> level) * (c->cam.rows >> level);
  const void* src = nullptr;
  size_t need = 0;
  const eft::Pyramid& p = c->pyr;
  switch (which) {
    case 0: src = p.vmap_curr[level]; need = n * 12; break;
    case 1: src = p.nmap_curr[level]; need = n * 12; break;
    case 2: src = p.vmap_g_prev[level]; need = n * 12; break;
    case 3: src = p.nmap_g_prev[level]; need = n * 12; break;
    case 4: case 5: src = p.lastDepth[level]; need = n * 4; break;
    case 6: src = p.lastImage[level]; need = n; break;
    case 7: src = p.nextImage[level]; need = n; break;
    case 8: src = p.lastNextImage[level]; need = n; break;
    case 9: src = p.dIdx[level]; need = n * 2; break;
    case 10: src = p.dIdy[level]; need = n * 2; break;
    case 11: src = level == 0 ? c->depth_filtered : p.depth_tmp[level]; need = n * 2; break;
    default: c->err = "ef_get_tracker_buffer: unknown buffer"; return EF_EINVAL;
  }
  if (bytes < need) { c->err = "ef_get_tracker_buffer: destination too small"; return EF_EINVAL; }
  EF_HIP(c, hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, c->stream));
  EF_HIP(c, hipStreamSynchronize(c->stream));
  return EF_OK;
}

int ef_enable_timing(ef_ctx* c, int on) { if (!c) return EF_EINVAL; c->timing = on != 0; return EF_OK; }
int ef_get_timings(ef_ctx* c, ef_timing* out, int max, int* n) {
  if (!c || !n) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  int k = 0;
  for (auto& t : c->timers) {
    if (!t.used || k >= max) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess) ms = -1.f;
    if (out) { out[k].name = t.name; out[k].ms = ms; }
    ++k;
  }
  *n = k;
  return EF_OK;
}

// ---- device helpers ----
int ef_kernel_timing(ef_ctx* c, int every_n_frames) {
  if (!c || every_n_frames < 0) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  c->ktime_every = every_n_frames;
  c->probe.used = 0;
  c->probe_splat.used = 0;
  c->probe_all.used = 0;
  if (every_n_frames > 0 && c->ka_start.empty()) {
    const int cap = 1024;
    c->ka_start.resize(cap);
    c->ka_stop.resize(cap);
    for (int i = 0; i < cap; ++i) {
      EF_HIP(c, hipEventCreate(&c->ka_start[i]));
      EF_HIP(c, hipEventCreate(&c->ka_stop[i]));
    }
    c->probe_all = eft::KernelProbe{c->ka_start.data(), c->ka_stop.data(), cap, 0};
  }
  if (every_n_frames > 0 && c->ks_start.empty()) {
    const int cap = 1024;
    c->ks_start.resize(cap);
    c->ks_stop.resize(cap);
    for (int i = 0; i < cap; ++i) {
      EF_HIP(c, hipEventCreate(&c->ks_start[i]));
      EF_HIP(c, hipEventCreate(&c->ks_stop[i]));
    }
    c->probe_splat = eft::KernelProbe{c->ks_start.data(), c->ks_stop.data(), cap, 0};
  }
  if (every_n_frames > 0 && c->kt_start.empty()) {
    const int cap = 4096;
    c->kt_start.resize(cap);
    c->kt_stop.resize(cap);
    for (int i = 0; i < cap; ++i) {
      EF_HIP(c, hipEventCreate(&c->kt_start[i]));
      EF_HIP(c, hipEventCreate(&c->kt_stop[i]));
    }
    c->probe.start = c->kt_start.data();
    c->probe.stop = c->kt_stop.data();
    c->probe.capacity = cap;
  }
  return EF_OK;
}
int ef_get_kernel_timing(ef_ctx* c, ef_kernel_time* out) {
  if (!c || !out) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  double total_ms = 0;
  for (int i = 0; i < c->probe.used; ++i) {
    float ms = 0;
    EF_HIP(c, hipEventElapsedTime(&ms, c->kt_start[i], c->kt_stop[i]));
    total_ms += ms;
  }
  const bool icp = !c->cfg.rgb_only && c->cfg.icp_weight > 0, rgb = c->cfg.rgb_only || c->cfg.icp_weight < 100;
#ifdef EF_FAST_ORDER
  out->name = "k_se3_accum_fast (level 0 of the launch-per-step script: icpStep + rgbStep Jacobian rows + fast-order sums)";
#else
  out->name = "k_se3_accum (level 0: icpStep + rgbStep Jacobian rows + reference-order sums)";
#endif
  out->launches = c->probe.used;
  out->avg_us = c->probe.used ? (float)(1e3 * total_ms / c->probe.used) : 0.f;
  // algorithmic bytes of ONE launch of this kernel (DESIGN.md "Roofline accounting"): icpStep 48 B per pixel-visit
  // (4 planar float3 maps); rgbStep reads the 4-byte packed correspondence of every pixel — the
  // reference's 16-byte DataTerm + 12-byte cloud are gone, so they are not counted — the ~10 % valid pixels' gathers
  // (depth + 2 gradients) are left out (data dependent): a lower bound, which can only understate `achieved`
  out->bytes_per_launch = (double)c->cam.cols * c->cam.rows * ((icp ? 48.0 : 0.0) + (rgb ? 4.0 : 0.0));
  out->bytes_per_launch_survey = (double)c->cam.cols * c->cam.rows * (icp ? 48.0 : 0.0);   // the ICP reduction alone
  return EF_OK;
}

// the persistent tracker launch (k_track_fast: SO(3) loop + every Gauss-Newton iteration of every level + their update steps)
int ef_get_tracker_timing(ef_ctx* c, ef_kernel_time* out) {
  if (!c || !out) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  double total_ms = 0;
  for (int i = 0; i < c->probe_all.used; ++i) {
    float ms = 0;
    EF_HIP(c, hipEventElapsedTime(&ms, c->ka_start[i], c->ka_stop[i]));
    total_ms += ms;
  }
  const bool icp = !c->cfg.rgb_only && c->cfg.icp_weight > 0, rgb = c->cfg.rgb_only || c->cfg.icp_weight < 100;
#ifdef EF_FAST_ORDER
  out->name = "k_track_fast (the whole tracker as one persistent launch: SO(3) loop + every ICP+RGB iteration of every level + the update steps)";
#else
  out->name = "k_track_ref (the whole tracker as one persistent launch, reference summation order: SO(3) loop + every ICP+RGB iteration of every level + the update steps)";
#endif
  out->launches = c->probe_all.used;
  out->avg_us = c->probe_all.used ? (float)(1e3 * total_ms / c->probe_all.used) : 0.f;
  // algorithmic bytes of one launch: every iteration visits every pixel of its level once (48 B icpStep + 4 B packed correspondence, as
  // ef_get_kernel_timing counts one level-0 launch); the SO(3) loop's two u8 images are left out (a lower bound)
  const int its[3] = {c->cfg.fast_odom ? 3 : 10, c->cfg.pyramid ? 5 : 0, c->cfg.pyramid ? 4 : 0};
  double visits = 0;
  for (int l = 0; l < 3; ++l) visits += (double)its[l] * (double)(c->cam.cols >> l) * (double)(c->cam.rows >> l);
  out->bytes_per_launch = visits * ((icp ? 48.0 : 0.0) + (rgb ? 4.0 : 0.0));
  out->bytes_per_launch_survey = visits * (icp ? 48.0 : 0.0);
  return EF_OK;
}

int ef_get_splat_timing(ef_ctx* c, ef_kernel_time* out) {
  if (!c || !out) return EF_EINVAL;
  DeviceGuard dg_(c);
  EF_HIP(c, hipStreamSynchronize(c->stream));
  double total_ms = 0;
  for (int i = 0; i < c->probe_splat.used; ++i) {
    float ms = 0;
    EF_HIP(c, hipEventElapsedTime(&ms, c->ks_start[i], c->ks_stop[i]));
    total_ms += ms;
  }
  unsigned count = 0;
  EF_HIP(c, hipMemcpy(&count, &c->st->map_counts[c->cur], sizeof(count), hipMemcpyDeviceToHost));
  out->name = "k_index_splat (IndexMap::predictIndices: per-surfel transform + project + 64-bit atomicMin z-buffer)";
  out->launches = c->probe_splat.used;
  out->avg_us = c->probe_splat.used ? (float)(1e3 * total_ms / c->probe_splat.used) : 0.f;
  // algorithmic bytes: the two float4 streams the pass needs (position+confidence, colour+times: 32 B / surfel; the
  // reference's vertex shader fetches all 48) + one 8-byte z-buffer update per surfel (an upper bound: culled surfels issue none)
  out->bytes_per_launch = 40.0 * (double)count;
  out->bytes_per_launch_survey = 48.0 * (double)count;   // 48 B per surfel read by the reference's vertex shader
  return EF_OK;
}

// Box calibration for bench.py (GPU boxes of one pool differ by 10-20 %): an EMPTY kernel and a kernel that streams 16 MB in and 16 MB out
// with 16-byte accesses, 200 back-to-back launches each on `stream`, averaged over the batch with two events (so launch gaps are in).
__global__ void k_calib_empty() {}
__global__ void __launch_bounds__(256) k_calib_stream(const float4* __restrict__ src, float4* __restrict__ dst, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}
int ef_dev_calibrate(void* stream, float* empty_us, float* stream16mb_us) {
  if (!empty_us || !stream16mb_us) return EF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const int n = 1 << 20, reps = 200;   // 1 Mi float4 = 16 MiB
  float4 *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = EF_EHIP;
  float ms = 0;
  if (hipMalloc((void**)&a, (size_t)n * sizeof(float4)) != hipSuccess || hipMalloc((void**)&b, (size_t)n * sizeof(float4)) != hipSuccess) { rc = EF_ENOMEM; goto done; }
  if (hipMemsetAsync(a, 0, (size_t)n * sizeof(float4), s) != hipSuccess) goto done;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) goto done;
  for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
    if (hipEventRecord(e0, s) != hipSuccess) goto done;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_calib_empty, dim3(256), dim3(256), 0, s);
    if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) goto done;
    *empty_us = 1e3f * ms / reps;
    if (hipEventRecord(e0, s) != hipSuccess) goto done;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_calib_stream, dim3(2048), dim3(256), 0, s, (const float4*)a, b, n);
    if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) goto done;
    *stream16mb_us = 1e3f * ms / reps;
  }
  rc = EF_OK;
done:
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  return rc;
}

int ef_dev_alloc(void** dev, size_t bytes) { return hipMalloc(dev, bytes ? bytes : 1) == hipSuccess ? EF_OK : EF_ENOMEM; }
int ef_dev_free(void* dev) { return hipFree(dev) == hipSuccess ? EF_OK : EF_EHIP; }
int ef_dev_upload(void* dev, const void* host, size_t bytes) { return hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess ? EF_OK : EF_EHIP; }
int ef_dev_download(void* host, const void* dev, size_t bytes) { return hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess ? EF_OK : EF_EHIP; }
int ef_dev_memset(void* dev, int value, size_t bytes) { return hipMemset(dev, value, bytes) == hipSuccess ? EF_OK : EF_EHIP; }
int ef_dev_sync(void) { return hipDeviceSynchronize() == hipSuccess ? EF_OK : EF_EHIP; }
int ef_device_count(int* n) { return hipGetDeviceCount(n) == hipSuccess ? EF_OK : EF_EHIP; }
int ef_set_device(int d) { return hipSetDevice(d) == hipSuccess ? EF_OK : EF_EHIP; }

// ---- operator tier: tracking ----
#define OP_TAIL(s)                                                         \
  do {                                                                     \
    hipError_t _e = hipGetLastError();                                     \
    if (_e != hipSuccess) { g_create_error = hipGetErrorString(_e); return EF_EHIP; } \
    return EF_OK;                                                          \
  } while (0)
#define OP_SYNC(s)                                                         \
  do {                                                                     \
    hipError_t _e = hipStreamSynchronize((hipStream_t)(s));                \
    if (_e == hipSuccess) _e = hipGetLastError();                          \
    if (_e != hipSuccess) { g_create_error = hipGetErrorString(_e); return EF_EHIP; } \
  } while (0)

int ef_op_pyr_down(const uint16_t* src, int sc, int sr, uint16_t* dst, void* s) { eft::pyr_down_u16(src, sc, sr, dst, (hipStream_t)s); OP_TAIL(s); }
int ef_op_create_vmap(const ef_intr* k, const uint16_t* depth, int cols, int rows, float cutoff, float* vmap, void* s) {
  eft::create_vmap(depth, cols, rows, eft::Intr{k->fx, k->fy, k->cx, k->cy}, cutoff, vmap, (hipStream_t)s);
  OP_TAIL(s);
}
int ef_op_create_nmap(const float* vmap, int cols, int rows, float* nmap, void* s) { eft::create_nmap(vmap, cols, rows, nmap, (hipStream_t)s); OP_TAIL(s); }
int ef_op_transform_maps(const float* vs, const float* ns, int cols, int rows, const float* R9, const float* t3, float* vd, float* nd, void* s) {
  float* rt = nullptr;
  if (hipMalloc((void**)&rt, 12 * sizeof(float)) != hipSuccess) return EF_ENOMEM;
  float h[12];
  memcpy(h, R9, 36);
  memcpy(h + 9, t3, 12);
  (void)hipMemcpyAsync(rt, h, sizeof(h), hipMemcpyHostToDevice, (hipStream_t)s);
  eft::transform_maps(vs, ns, cols, rows, rt, rt + 9, vd, nd, (hipStream_t)s);
  (void)hipStreamSynchronize((hipStream_t)s);
  (void)hipFree(rt);
  OP_TAIL(s);
}
int ef_op_copy_maps(const float* v4, const float* n4, int cols, int rows, float* tmp, float* vd, float* nd, void* s) {
  eft::copy_maps(v4, n4, cols, rows, tmp, vd, nd, (hipStream_t)s);
  OP_TAIL(s);
}
int ef_op_resize_vmap(const float* in, int sc, int sr, float* out, void* s) { eft::resize_map(in, sc, sr, out, false, (hipStream_t)s); OP_TAIL(s); }
int ef_op_resize_nmap(const float* in, int sc, int sr, float* out, void* s) { eft::resize_map(in, sc, sr, out, true, (hipStream_t)s); OP_TAIL(s); }
int ef_op_pyr_down_gauss_f(const float* src, int sc, int sr, float* dst, void* s) { eft::pyr_down_gauss_f(src, sc, sr, dst, (hipStream_t)s); OP_TAIL(s); }
int ef_op_pyr_down_uchar_gauss(const uint8_t* src, int sc, int sr, uint8_t* dst, void* s) { eft::pyr_down_uchar_gauss(src, sc, sr, dst, (hipStream_t)s); OP_TAIL(s); }
int ef_op_vertices_to_depth(const float* tmp, int cols, int rows, float cutoff, float* dst, void* s) { eft::vertices_to_depth(tmp, cols, rows, cutoff, dst, (hipStream_t)s); OP_TAIL(s); }
int ef_op_image_bgr_to_intensity(const uint8_t* rgba, int cols, int rows, uint8_t* dst, void* s) { eft::bgr_to_intensity(rgba, 4, cols, rows, dst, (hipStream_t)s); OP_TAIL(s); }
int ef_op_compute_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy, void* s) {
  eft::derivative_images(src, cols, rows, dx, dy, (hipStream_t)s);
  OP_TAIL(s);
}
int ef_op_project_to_point_cloud(const float* depth, int cols, int rows, const ef_intr* k0, int level, float* cloud, void* s) {
  eft::project_to_point_cloud(depth, cols, rows, eft::intr_level(eft::Intr{k0->fx, k0->fy, k0->cx, k0->cy}, level), cloud, (hipStream_t)s);
  OP_TAIL(s);
}

static int op_scratch(float** partials, float** out, int nfloats_out) {
  if (hipMalloc((void**)partials, (size_t)eft::OP_SCRATCH_FLOATS * sizeof(float)) != hipSuccess) return EF_ENOMEM;
  if (hipMalloc((void**)out, nfloats_out * sizeof(float)) != hipSuccess) { (void)hipFree(*partials); return EF_ENOMEM; }
  return EF_OK;
}
static void unpack29_host(const float* h, float* A, float* b) {
  int shift = 0;
  for (int i = 0; i < 6; ++i)
    for (int j = i; j < 7; ++j) {
      const float v = h[shift++];
      if (j == 6) b[i] = v;
      else A[j * 6 + i] = A[i * 6 + j] = v;
    }
}
int ef_op_icp_step(const float* Rc, const float* tc, const float* vc, const float* nc, const float* Rpi, const float* tp, const ef_intr* k,
                   const float* vg, const float* ng, float dist, float ang, int cols, int rows, float* A, float* b, float* res, void* s) {
  if (cols <= 0 || rows <= 0 || cols > 2048 || rows > 2048) return EF_EINVAL;
  eft::IcpArgs a;
  memcpy(a.Rcurr, Rc, 36); memcpy(a.tcurr, tc, 12); memcpy(a.Rprev_inv, Rpi, 36); memcpy(a.tprev, tp, 12);
  a.k = eft::Intr{k->fx, k->fy, k->cx, k->cy};
  a.distThres = dist; a.angleThres = ang;
  float *partials, *out;
  int r = op_scratch(&partials, &out, 32);
  if (r != EF_OK) return r;
  eft::icp_step_op(a, vc, nc, vg, ng, cols, rows, partials, out, (hipStream_t)s);
  float h[32];
  (void)hipMemcpyAsync(h, out, 29 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)s);
  hipError_t e = hipStreamSynchronize((hipStream_t)s);
  (void)hipFree(partials); (void)hipFree(out);
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return EF_EHIP; }
  unpack29_host(h, A, b);
  res[0] = h[27]; res[1] = h[28];
  return EF_OK;
}
int ef_op_compute_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth, const float* nextDepth,
                               const uint8_t* lastImage, const uint8_t* nextImage, void* corres, float maxDepthDelta, const float* kt,
                               const float* krkinv, int cols, int rows, int* sigma, int* count, void* s) {
  eft::RgbResidualArgs a;
  a.minScale = minScale; a.maxDepthDelta = maxDepthDelta;
  memcpy(a.kt, kt, 12); memcpy(a.krkinv, krkinv, 36);
  int* out;
  if (hipMalloc((void**)&out, 2 * sizeof(int)) != hipSuccess) return EF_ENOMEM;
  eft::rgb_residual_op(a, dIdx, dIdy, lastDepth, nextDepth, lastImage, nextImage, corres, cols, rows, out, (hipStream_t)s);
  int h[2] = {0, 0};
  hipError_t e = hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(out);
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return EF_EHIP; }
  *count = h[0];
  *sigma = h[1];
  return EF_OK;
}
int ef_op_rgb_step(const void* corres, float sigma, const float* cloud, float fx, float fy, const int16_t* dIdx, const int16_t* dIdy,
                   float sobelScale, int cols, int rows, float* A, float* b, void* s) {
  float *partials, *out;
  int r = op_scratch(&partials, &out, 32);
  if (r != EF_OK) return r;
  eft::rgb_step_op(corres, sigma, cloud, fx, fy, dIdx, dIdy, sobelScale, cols, rows, partials, out, (hipStream_t)s);
  float h[32];
  (void)hipMemcpyAsync(h, out, 29 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)s);
  hipError_t e = hipStreamSynchronize((hipStream_t)s);
  (void)hipFree(partials); (void)hipFree(out);
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return EF_EHIP; }
  unpack29_host(h, A, b);
  return EF_OK;
}
int ef_op_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* ib, const float* kinv, const float* krlr, int cols,
                   int rows, float* A, float* b, float* res, void* s) {
  eft::So3Args a;
  memcpy(a.imageBasis, ib, 36); memcpy(a.kinv, kinv, 36); memcpy(a.krlr, krlr, 36);
  float *partials, *out;
  int r = op_scratch(&partials, &out, 16);
  if (r != EF_OK) return r;
  eft::so3_step_op(a, lastImage, nextImage, cols, rows, partials, out, (hipStream_t)s);
  float h[11];
  (void)hipMemcpyAsync(h, out, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)s);
  hipError_t e = hipStreamSynchronize((hipStream_t)s);
  (void)hipFree(partials); (void)hipFree(out);
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return EF_EHIP; }
  int shift = 0;
  for (int i = 0; i < 3; ++i)
    for (int j = i; j < 4; ++j) {
      const float v = h[shift++];
      if (j == 3) b[i] = v;
      else A[j * 3 + i] = A[i * 3 + j] = v;
    }
  res[0] = h[9]; res[1] = h[10];
  return EF_OK;
}

// ---- operator tier: the driver's small linear algebra, evaluated on the device ----
}  // extern "C"
namespace {
__global__ void k_linalg_probe(int which, const double* __restrict__ in, double* __restrict__ out) {
  if (which == EF_LINALG_LDLT6_WAVE) {  // the one-element-per-lane factorisation the tracker uses (ef_solve_dev.hpp)
    __shared__ efs::SolveScratch S;
    const int lane = threadIdx.x;
    if (lane < 6) S.b[lane] = in[36 + lane];
    efs::wave_sync();
    efs::ldlt6_wave(in[lane < 36 ? lane : 0], S);
    if (lane < 6) out[lane] = S.x[lane];
    return;
  }
  if (threadIdx.x != 0) return;
  switch (which) {
    case EF_LINALG_LDLT6: efl::ldlt_solve<double, 6>(in, in + 36, out); break;
    case EF_LINALG_LDLT3F: {
      float A[9], b[3], x[3];
      for (int i = 0; i < 9; ++i) A[i] = (float)in[i];
      for (int i = 0; i < 3; ++i) b[i] = (float)in[9 + i];
      efl::ldlt_solve<float, 3>(A, b, x);
      for (int i = 0; i < 3; ++i) out[i] = (double)x[i];
      break;
    }
    case EF_LINALG_POLAR3: efl::polar3(in, out); break;
    case EF_LINALG_RODRIGUES: efl::rodrigues(in, out); break;
    case EF_LINALG_SE3_INVERSE: efl::se3_matrix(efl::se3_inverse(efl::se3_from_matrix(in)), out); break;
    case EF_LINALG_SE3_LOG_NORM: out[0] = efl::se3_log_norm(efl::se3_from_matrix(in)); break;
    case EF_LINALG_SCALAR:
      out[0] = sqrt(in[0]); out[1] = in[0] / in[1]; out[2] = sin(in[0]); out[3] = cos(in[0]); out[4] = atan2(in[0], in[1]);
      break;
    default: break;
  }
}
}  // namespace
extern "C" {
int ef_op_linalg(int which, const double* in, int n_in, double* out, int n_out) {
  if (!in || !out || n_in <= 0 || n_out <= 0 || n_in > 64 || n_out > 64 || which < 0 || which > EF_LINALG_LDLT6_WAVE) return EF_EINVAL;
  double* d;
  if (hipMalloc((void**)&d, 128 * sizeof(double)) != hipSuccess) return EF_ENOMEM;
  (void)hipMemset(d, 0, 128 * sizeof(double));
  (void)hipMemcpy(d, in, n_in * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_linalg_probe, dim3(1), dim3(64), 0, 0, which, (const double*)d, d + 64);
  hipError_t e = hipMemcpy(out, d + 64, n_out * sizeof(double), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) { g_create_error = hipGetErrorString(e); return EF_EHIP; }
  return EF_OK;
}

// ---- operator tier: pre-processing + map ----
int ef_op_filter_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered, void* s) {
  if (!efm::filter_depth(raw, cols, rows, maxD, filtered, (hipStream_t)s)) return EF_EHIP;   // (no weight table: device ordinal >= 64, allocation or launch failure)
  OP_TAIL(s);
}
int ef_op_metricise_depth(const uint16_t* in, int cols, int rows, float maxD, float* out, void* s) {
  efm::metricise_depth(in, cols, rows, maxD, out, (hipStream_t)s);
  OP_TAIL(s);
}

}  // extern "C"
namespace {
struct OpMap {  // temporary SoA mirror of an AoS surfel list + scratch, for the operator tier
  std::vector<void*> allocs;
  template <typename T>
  T* alloc(size_t n, int fill = 0) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemset(p, fill, (n ? n : 1) * sizeof(T));
    allocs.push_back(p);
    return (T*)p;
  }
  efm::SurfelSoA soa(size_t n) { return efm::SurfelSoA{alloc<float4>(n), alloc<float4>(n), alloc<float4>(n)}; }
  ~OpMap() { for (void* p : allocs) (void)hipFree(p); }
};
efm::Cam to_cam(const ef_cam* c) { return efm::Cam{c->cols, c->rows, c->fx, c->fy, c->cx, c->cy}; }
// device copies of the two float pose matrices derived from a double T_wc
void pose_mats(const double* T16, float* Tcw_host, float* pose_host) {
  const efl::SE3 T = efl::se3_from_matrix(T16);
  efl::se3_inverse_matrix_f(T, Tcw_host);
  efl::se3_castf_matrix(T, pose_host);
}
}  // namespace
extern "C" {

int ef_op_seed_map(const ef_cam* cam, const uint8_t* rgb, const float* dm, const float* dmf, int time, float maxDepth, float* surfels,
                   uint32_t* count_host, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  const size_t P = (size_t)cam->cols * cam->rows;
  efm::SurfelSoA soa = m.soa(P);
  efm::CompactScratch cs;
  cs.max_chunks = (int)(2 * P / efm::CHUNK + 8);
  cs.flags = m.alloc<uint8_t>(2 * P);
  cs.chunk_count = m.alloc<uint32_t>(cs.max_chunks);
  cs.chunk_offset = m.alloc<uint32_t>(cs.max_chunks);
  cs.totals = m.alloc<uint32_t>(8);
  unsigned* cnt = m.alloc<unsigned>(1);
  efm::seed_map(to_cam(cam), rgb, dm, dmf, time, maxDepth, soa, cnt, cs, s);
  unsigned h = 0;
  (void)hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, s);
  OP_SYNC(s);
  efm::soa_to_aos(soa, h, surfels, s);
  OP_SYNC(s);
  *count_host = h;
  return EF_OK;
}

int ef_op_predict_indices(const ef_cam* cam, const double* T16, int time, const float* surfels, uint32_t count, float maxDepth, int timeDelta,
                          uint32_t* index, float* vc, float* ct, float* nr, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  const size_t P = (size_t)cam->cols * cam->rows;
  efm::SurfelSoA soa = m.soa(count);
  efm::aos_to_soa(surfels, count, soa, s);
  float h[32];
  pose_mats(T16, h, h + 16);
  float* mats = m.alloc<float>(32);
  (void)hipMemcpyAsync(mats, h, sizeof(h), hipMemcpyHostToDevice, s);
  unsigned* cnt = m.alloc<unsigned>(1);
  (void)hipMemcpyAsync(cnt, &count, sizeof(unsigned), hipMemcpyHostToDevice, s);
  unsigned long long* zbuf = m.alloc<unsigned long long>(P, 0xFF);
  efm::IndexMaps im{index, (float4*)vc, (float4*)ct, (float4*)nr};
  efm::predict_indices(to_cam(cam), mats, time, soa, cnt, maxDepth, timeDelta, zbuf, im, s);
  OP_SYNC(s);
  return EF_OK;
}

int ef_op_combined_predict(const ef_cam* cam, const double* T16, const float* surfels, uint32_t count, float maxDepth, float confThreshold,
                           int time, int maxTime, int timeDelta, uint8_t* image, float* vertex, float* normal, uint16_t* timeMap, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  const size_t P = (size_t)cam->cols * cam->rows;
  efm::SurfelSoA soa = m.soa(count);
  efm::aos_to_soa(surfels, count, soa, s);
  float h[32];
  pose_mats(T16, h, h + 16);
  float* mats = m.alloc<float>(32);
  (void)hipMemcpyAsync(mats, h, sizeof(h), hipMemcpyHostToDevice, s);
  unsigned* cnt = m.alloc<unsigned>(1);
  (void)hipMemcpyAsync(cnt, &count, sizeof(unsigned), hipMemcpyHostToDevice, s);
  unsigned long long* zbuf = m.alloc<unsigned long long>(P, 0xFF);
  efm::PredictMaps pm{(uchar4*)image, (float4*)vertex, (float4*)normal, timeMap};
  efm::FillMaps none{nullptr, nullptr, nullptr};
  efm::combined_predict(to_cam(cam), mats, soa, cnt, maxDepth, confThreshold, time, maxTime, timeDelta, zbuf, pm, none, nullptr, nullptr, false,
                        nullptr, s);
  OP_SYNC(s);
  return EF_OK;
}

int ef_op_synthesize_depth(const ef_cam* cam, const double* T16, const float* surfels, uint32_t count, float maxDepth, float confThreshold,
                           int time, int maxTime, int timeDelta, float* depth, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  const size_t P = (size_t)cam->cols * cam->rows;
  efm::SurfelSoA soa = m.soa(count);
  efm::aos_to_soa(surfels, count, soa, s);
  float h[32];
  pose_mats(T16, h, h + 16);
  float* mats = m.alloc<float>(32);
  (void)hipMemcpyAsync(mats, h, sizeof(h), hipMemcpyHostToDevice, s);
  unsigned* cnt = m.alloc<unsigned>(1);
  (void)hipMemcpyAsync(cnt, &count, sizeof(unsigned), hipMemcpyHostToDevice, s);
  unsigned long long* zbuf = m.alloc<unsigned long long>(P, 0xFF);
  efm::synthesize_depth(to_cam(cam), mats, soa, cnt, maxDepth, confThreshold, time, maxTime, timeDelta, zbuf, depth, s);
  OP_SYNC(s);
  return EF_OK;
}

int ef_op_fill_in(const ef_cam* cam, const uint8_t* image, const float* vertex, const float* normal, const uint16_t* depthFiltered,
                  const uint8_t* rgb, int passthrough, int passthroughImage, uint8_t* fimage, float* fvertex, float* fnormal, void* s_) {
  efm::PredictMaps pm{(uchar4*)image, (float4*)vertex, (float4*)normal, nullptr};
  efm::FillMaps fm{(uchar4*)fimage, (float4*)fvertex, (float4*)fnormal};
  efm::fill_in(to_cam(cam), pm, depthFiltered, rgb, passthrough != 0, passthroughImage != 0, fm, (hipStream_t)s_);
  OP_TAIL(s_);
}

int ef_op_dense_enough(const ef_cam* cam, const uint8_t* image, int* dense_host, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  unsigned* cnt = m.alloc<unsigned>(1);
  efm::dense_count(to_cam(cam), (const uchar4*)image, cnt, s);
  unsigned h = 0;
  (void)hipMemcpyAsync(&h, cnt, sizeof(h), hipMemcpyDeviceToHost, s);
  OP_SYNC(s);
  *dense_host = ((float)h / (float)((cam->cols / 20) * (cam->rows / 20)) > 0.75f) ? 1 : 0;
  return EF_OK;
}

int ef_op_fuse(const ef_cam* cam, const double* T16, int time, const uint8_t* rgb, const float* dm, const float* dmf, const uint32_t* index,
               const float* vc, const float* ct, const float* nr, float maxDepth, float weighting, float* surfels, uint32_t count,
               float* newUnstable, uint32_t* newCount, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  efm::SurfelSoA soa = m.soa(count);
  efm::aos_to_soa(surfels, count, soa, s);
  float h[33];
  pose_mats(T16, h, h + 16);
  h[32] = weighting;
  float* mats = m.alloc<float>(33);
  (void)hipMemcpyAsync(mats, h, sizeof(h), hipMemcpyHostToDevice, s);
  unsigned* cnt = m.alloc<unsigned>(2);
  (void)hipMemcpyAsync(cnt, &count, sizeof(unsigned), hipMemcpyHostToDevice, s);
  efm::Candidates cand;
  cand.n = (cam->cols / 2) * (cam->rows / 2);
  cand.pos_conf = m.alloc<float4>(cand.n);
  cand.col_time = m.alloc<float4>(cand.n);
  cand.nrm_rad = m.alloc<float4>(cand.n);
  cand.best = m.alloc<uint32_t>(cand.n);
  uint32_t* winner = m.alloc<uint32_t>(count, 0xFF);
  efm::IndexMaps im{(uint32_t*)index, (float4*)vc, (float4*)ct, (float4*)nr};
  efm::fuse(to_cam(cam), mats + 16, time, rgb, dm, dmf, im, maxDepth, mats + 32, soa, cnt, cand, winner, s);
  efm::soa_to_aos(soa, count, surfels, s);
  efm::CompactScratch cs;
  cs.max_chunks = cand.n / efm::CHUNK + 8;
  cs.flags = m.alloc<uint8_t>(cand.n);
  cs.chunk_count = m.alloc<uint32_t>(cs.max_chunks);
  cs.chunk_offset = m.alloc<uint32_t>(cs.max_chunks);
  cs.totals = m.alloc<uint32_t>(8);
  efm::candidates_to_aos(cand, newUnstable, cnt + 1, cs, s);
  unsigned hn = 0;
  (void)hipMemcpyAsync(&hn, cnt + 1, sizeof(hn), hipMemcpyDeviceToHost, s);
  OP_SYNC(s);
  *newCount = hn;
  return EF_OK;
}

}  // extern "C"
namespace {
// scatter an AoS "newUnstable" list (draw order) back into candidate slots 0..n-1: the clean kernels only
// need the relative order, which consecutive slots preserve
__global__ void k_aos_to_cand(const float4* __restrict__ aos, uint32_t n, efm::Candidates cand) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (uint32_t)cand.n) return;
  if (i < n) {
    cand.pos_conf[i] = aos[(size_t)i * 3];
    cand.col_time[i] = aos[(size_t)i * 3 + 1];
    cand.nrm_rad[i] = aos[(size_t)i * 3 + 2];
  } else {
    cand.col_time[i] = make_float4(0, 0, 0, 0);
  }
}
}  // namespace
extern "C" {

int ef_op_clean_deform(const ef_cam* cam, const double* T16, int time, const uint32_t* index, const float* vc, const float* ct, const float* nr,
                       float confThreshold, int timeDelta, float maxDepth, const float* surfels, uint32_t count, const float* newUnstable,
                       uint32_t newCount, const float* graph, int nodes, const float* depth, int isFern, float* surfels_out,
                       uint32_t* outCount, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  OpMap m;
  const uint32_t cap = count + newCount;
  efm::SurfelSoA soa = m.soa(count), out = m.soa(cap);
  efm::aos_to_soa(surfels, count, soa, s);
  float h[32];
  pose_mats(T16, h, h + 16);
  float* mats = m.alloc<float>(32);
  (void)hipMemcpyAsync(mats, h, sizeof(h), hipMemcpyHostToDevice, s);
  unsigned* cnt = m.alloc<unsigned>(1);
  (void)hipMemcpyAsync(cnt, &count, sizeof(unsigned), hipMemcpyHostToDevice, s);
  efm::Candidates cand;
  cand.n = (int)(newCount ? newCount : 1);
  cand.pos_conf = m.alloc<float4>(cand.n);
  cand.col_time = m.alloc<float4>(cand.n);
  cand.nrm_rad = m.alloc<float4>(cand.n);
  cand.best = m.alloc<uint32_t>(cand.n);
  hipLaunchKernelGGL(k_aos_to_cand, dim3((cand.n + 255) / 256), dim3(256), 0, s, (const float4*)newUnstable, newCount, cand);
  uint32_t* winner = m.alloc<uint32_t>(count, 0xFF);
  efm::CompactScratch cs;
  cs.max_chunks = (int)((cap + 1) / efm::CLEAN_ROW + 8);
  cs.flags = m.alloc<uint8_t>((size_t)cap + 1);
  cs.chunk_count = m.alloc<uint32_t>(cs.max_chunks);
  cs.chunk_offset = m.alloc<uint32_t>(cs.max_chunks);
  cs.totals = m.alloc<uint32_t>(8);
  efm::IndexMaps im{(uint32_t*)index, (float4*)vc, (float4*)ct, (float4*)nr};
  unsigned* cnt_out = m.alloc<unsigned>(1);
  const efm::Deformation def{graph, nodes, depth, isFern, maxDepth};
  efm::clean(to_cam(cam), mats, time, im, confThreshold, timeDelta, soa, cnt, cand, winner, out, cnt_out, cap, cs, nullptr, s,
             nodes > 0 ? &def : nullptr);
  unsigned hn = 0;
  (void)hipMemcpyAsync(&hn, cnt_out, sizeof(hn), hipMemcpyDeviceToHost, s);
  OP_SYNC(s);
  efm::soa_to_aos(out, hn, surfels_out, s);
  OP_SYNC(s);
  *outCount = hn;
  return EF_OK;
}
int ef_op_clean(const ef_cam* cam, const double* T16, int time, const uint32_t* index, const float* vc, const float* ct, const float* nr,
                float confThreshold, int timeDelta, float maxDepth, const float* surfels, uint32_t count, const float* newUnstable,
                uint32_t newCount, float* surfels_out, uint32_t* outCount, void* s_) {
  return ef_op_clean_deform(cam, T16, time, index, vc, ct, nr, confThreshold, timeDelta, maxDepth, surfels, count, newUnstable, newCount,
                            nullptr, 0, nullptr, 0, surfels_out, outCount, s_);
}

}  // extern "C"
