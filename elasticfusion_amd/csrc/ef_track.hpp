// Tracking side of libefusion_hip: device-resident Gauss-Newton state + launchers.
// Replaces Core/Utils/RGBDOdometry.{h,cpp} and the CUDA operators of Core/Cuda/{cudafuncs,reduce}.cu.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ef_build.hpp"
// Summation order of the fp32 normal-equation sums (ef_build.hpp decides).  The shipped default keeps the REFERENCE's order
// (reduce.cu:57-140,313-317; kernels of DESIGN_reference_order.md, pinned against the compiled reduce.cu); -DEF_FAST_BUILD
// (libefusion_hip_fast.so, opt-in) uses THE FAST ORDER (ef_track_fast.inc: per-lane register accumulation, adjacent-pair trees; specified in
// DESIGN.md 5.1, which the test suite's CPU checker restates bit for bit).

namespace eft {

constexpr int NUM_PYRS = 3;          // RGBDOdometry.h:114
constexpr int REDUCE_BLOCK = 256;    // threads per workgroup of the integer (order-free) residual reduction
// The fp32 normal-equation sums reproduce the reference's summation ORDER exactly (reduce.cu:313-317, :97-140):
// a <<<64,256>>> grid-stride launch = 16384 "virtual threads", each summing pixels g, g+16384, ... in order, then a
// warp32 shuffle tree, an 8-warp tree per block and a tree over the 64 block partials.  The kernels here run one
// workgroup per virtual WARP (512 of them) and leave one partial vector per virtual warp, acc-major:
// partials[acc * VWARPS + warp].
constexpr int VTHREADS = 16384;      // 64 x 256, types.cuh:62-63
constexpr int VWARPS = VTHREADS / 32;
constexpr int RGB_SLOTS = 64;
constexpr int SE3_ACCS = 29;         // JtJJtrSE3, types.cuh:98-143
constexpr int SO3_ACCS = 11;         // JtJJtrSO3, types.cuh:145-168
// SE(3) normal equations: k_se3_accum leaves one partial per accumulator and PAIR of virtual warps (warps w and w + 4 of a reference
// block: blockReduceSum's first tree level is done inside the workgroup), layout [term][acc][block 0..63][pair 0..3];
// the SO(3) kernel leaves one partial per accumulator and virtual warp, layout [acc][warp]
constexpr int SE3_PAIRS = VWARPS / 2;
constexpr int PARTIAL_FLOATS = 2 * SE3_ACCS * SE3_PAIRS > SO3_ACCS * VWARPS ? 2 * SE3_ACCS * SE3_PAIRS : SO3_ACCS * VWARPS;
// The persistent small-level tracker (k_track_small, round 3): 128 co-resident workgroups of 512 threads run k_track_begin, the SO(3)
// loop and every Gauss-Newton iteration of the levels with <= PT_MAX_PIXELS pixels in ONE launch; hand-overs between workgroups
// alternate between two partial regions, and a small synchronisation record (PtSync) sits behind them.
constexpr int PT_WGS = 128, PT_BLOCK = 512, PT_MAX_ITER = 16, PT_MAX_PIXELS = 8 * VTHREADS, PT_SYNC_FLOATS = 1024;
// (the fast order's persistent tracker lays its exchange areas over the same allocation: FT_*_OFF in ef_track_fast.inc, 45.8 K floats)
// (round 6: + the copies of the reference-order launch's all-to-all areas behind float 49152: FT_G2R_OFF / FT_GAR_OFF, 512 + 64 KB)
constexpr int PARTIAL_ALLOC_FLOATS = (2 * PARTIAL_FLOATS + PT_SYNC_FLOATS) > 196608 ? (2 * PARTIAL_FLOATS + PT_SYNC_FLOATS) : 196608;
constexpr int FT_EPOCHS = 64;        // exchange epochs one launch of the persistent tracker may use (<= 10 SO(3) + 2 x 24 iterations)

struct Intr { float fx, fy, cx, cy; };
__host__ __device__ inline Intr intr_level(const Intr& k, int level) {  // CameraModel::operator()(level), types.cuh:92-95
  const int div = 1 << level;
  return Intr{k.fx / div, k.fy / div, k.cx / div, k.cy / div};
}

// What one Gauss-Newton update hands to the next.  Double-buffered (TrackState::gn): the update of iteration i is evaluated at the
// head of iteration i + 1's first kernel by EVERY workgroup redundantly (reading gn[cur], the pair partials and the residual sums);
// workgroup 0 alone writes the result into gn[cur ^ 1], so no workgroup ever reads a word another one is writing.
struct GNState {
  float Rcurr[9], tcurr[3];
  double resultRt[16];
  float krkinv[9], kt[3];     // K R K^-1 and K t of the coming iteration (RGBDOdometry.cpp:407-417)
  float lastRGBErrorLevel;    // rgbOnly early-exit bookkeeping (RGBDOdometry.cpp:445-450)
  int rgb_broken;
};

// Everything the 19-iteration loop mutates lives here, in HBM, so that no iteration needs the host
// (the reference round-trips 3x per iteration: RGBDOdometry.cpp:424-512).
struct TrackState {
  // camera pose, Sophus-style (unit quaternion xyzw + translation), double
  double q[4];
  double t[3];
  double q_prev[4];   // pose before this frame's tracking (velocity weighting, ElasticFusion.cpp:369-383)
  double t_prev[3];
  // constants of one getIncrementalTransformation call
  float Rprev[9], tprev[3], Rprev_inv[9];
  // Gauss-Newton variables, double-buffered (see GNState); track() starts in gn[0]
  GNState gn[2];
  // {count, sum diff^2} of the residual pass, integer => order independent (reduce.cu:687-709).  Spread over
  // RGB_SLOTS cache lines (workgroup b adds to slot b % RGB_SLOTS) so the device-scope atomics do not serialise
  // on one address; consumers add the slots up.  Two sets: iteration i adds into set i & 1 while the update of
  // iteration i - 1 (same kernel, head) still reads the other one; k_se3_accum of iteration i re-zeroes set (i + 1) & 1.
  int rgb_slots[2][RGB_SLOTS][16];
  // outputs (RGBDOdometry.h:74-82)
  float lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count;
  double lastA[36], lastb[6];
  int so3_iterations;
  // SO(3) pre-alignment loop state (RGBDOdometry.cpp:284-369); the loop runs as <= 10 multi-workgroup launches
  double so3_resultR[9], so3_lastResultR[9];
  float so3_R_lr[9];
  float so3_mats[27];         // imageBasis (K R K^-1), K^-1, K R of the coming iteration, float (RGBDOdometry.cpp:309-316)
  float so3_lastError, so3_lastCount;
  int so3_done;
  unsigned so3_ticket;        // last-workgroup-done counter
  // per-frame scalars produced on the device
  float weighting;            // fusion weight (ElasticFusion.cpp:371-383)
  // denseEnough() tally of the last predict() (Resize::image samples with r,g,b > 0; ElasticFusion.cpp:256-268):
  // surface_resolve adds to it, the next frame's tracker reads it (fill-in maps are used when
  // !(dense_count / dense_samples > 0.75), :304-305) and k_track_begin / k_pose_injected re-arm it.
  unsigned dense_count;
  int dense_samples;
  int tick;
  // Level-0 iterations with the update step INSIDE the correspondence-search launch (ef_set_fused_step): workgroup 0 evaluates the update
  // and hands K R K^-1 / K t / the rgbOnly flag to the other workgroups of the same launch as 13 tagged 8-byte granules
  // {value, (call_seq << 6) | iteration + 1}; call_seq counts getIncrementalTransformation calls (first kernel of the call)
  unsigned long long step_rec[16];
  unsigned call_seq;
  unsigned step_timeout;      // sticky: a workgroup gave up waiting for the record (bounded spin)
  unsigned long long dbg_clock[16];   // developer instrumentation (EF_STAGE_CLOCKS builds only)
  unsigned map_counts[2];     // live surfels of the two ping-pong map buffers (clean reads one, writes the other)
  unsigned model_view_stamp;  // model-to-model tracker (local loop closure): the INACTIVE prediction stamps this with the frame's value when it shows a surfel
  // float matrices consumed by the map kernels
  float T_cw[16];             // T_wc.inverse().matrix().cast<float>()  (IndexMap.cpp:208)
  float pose_f[16];           // T_wc.cast<float>().matrix()            (GlobalModel.cpp:403)
  float R_wc_f[9], t_wc_f[3]; // T_wc.rotationMatrix().cast<float>()    (RGBDOdometry.cpp:192-193)
};

struct Pyramid {               // one RGBDOdometry instance's device buffers (RGBDOdometry.h:87-140)
  int width, height;
  uint16_t* depth_tmp[NUM_PYRS];
  float* vmap_curr[NUM_PYRS];
  float* nmap_curr[NUM_PYRS];
  float* vmap_g_prev[NUM_PYRS];
  float* nmap_g_prev[NUM_PYRS];
  float* lastDepth[NUM_PYRS];
  float* nextDepth[NUM_PYRS];      // frame-to-model tracking: the SAME buffers as lastDepth (quirk Q1); model-to-model: its own
  uint8_t* lastImage[NUM_PYRS];
  uint8_t* nextImage[NUM_PYRS];
  uint8_t* lastNextImage[NUM_PYRS];
  int16_t* dIdx[NUM_PYRS];
  int16_t* dIdy[NUM_PYRS];
  // photometric correspondences of the current iteration, 4 bytes per pixel instead of the reference's 16-byte
  // DataTerm (types.cuh:81-86): bit31 valid | (diff+255) << 22 | v0 << 11 | u0 ("one" is the pixel itself)
  uint32_t* corres[NUM_PYRS];
  uint8_t* rgbMask[NUM_PYRS];      // iteration-invariant part of residualKernel's gates, built once per frame
  float* partials;                 // PARTIAL_ALLOC_FLOATS: region 0 (what the per-step kernels use), region 1, PtSync (zero-filled at allocation)
  unsigned epoch = 1;              // next unused exchange epoch of this instance's persistent launches (host side; 0 = "never written")
  int last_mode = 0;               // host side: 1 = the exchange areas hold a 256-workgroup persistent launch's granules (k_track_fast / k_track_ref)
  // host side: what the sticky words of the exchange areas held when a context switched scripts (the areas are cleared at a switch: ADVICE r5) —
  // tracker_aborted / tracker_fallbacks add them to what the device holds now
  unsigned sticky_abort = 0, fallbacks_base = 0;
  int W(int l) const { return width >> l; }
  int H(int l) const { return height >> l; }
};

// Optional HIP-event sampling of the level-0 normal-equation kernel (bench.py roofline leg): events are
// recorded on the stream the kernel runs on, immediately before and after the launch.
struct KernelProbe {
  hipEvent_t* start;
  hipEvent_t* stop;
  int capacity;
  int used;
};

struct TrackParams {           // host-side knobs of getIncrementalTransformation.  Compared and copied with memcmp / memcpy (the hipGraph cache key,
                               // ef_context.hip): NO padding anywhere — every byte is a member (static_assert below)
  bool rgbOnly, pyramid, fastOdom, so3;
  float icpWeight;
  float distThres, angleThres; // RGBDOdometry.h:41-42
  int fused_step = 0;          // level-0 iterations: update step inside the correspondence-search launch (two launches per iteration)
  int persistent = 1;          // 1: the whole call as ONE persistent launch of 256 workgroups (k_track_fast / k_track_ref); 0: one launch per step (round 2);
                               // 2 (reference-order builds): round 3's launch of the small levels (k_track_small, 128 workgroups) + one launch per level-0 step
  // persistent launch only: when *empty_model_flag != empty_model_value the MODEL side of this call is known to be empty (the
  // inactive prediction of the local loop closure showed no surfel: ElasticFusion.cpp:451-471 still runs the tracker on it) — no pixel can
  // find a correspondence in any iteration, every sum is zero, and the launch leaves what nineteen zero updates leave, at once
  const unsigned* empty_model_flag = nullptr;
  unsigned empty_model_value = 0;
  int no_resident = 0;         // development (A/B): 1 = the persistent launch streams every level's pixel data as in round 5 (ef_set_resident_levels(ctx, 0));
                               // (this member also makes the tail explicit: no padding)
};
static_assert(sizeof(TrackParams) == 4 + 4 + 8 + 4 + 4 + 8 + 4 + 4, "TrackParams has no padding bytes (it is compared with memcmp)");

// ---- operator-tier launchers (raw device pointers) ----
void pyr_down_u16(const uint16_t* src, int scols, int srows, uint16_t* dst, hipStream_t s);
void create_vmap(const uint16_t* depth, int cols, int rows, Intr k, float cutoff, float* vmap, hipStream_t s);
void create_nmap(const float* vmap, int cols, int rows, float* nmap, hipStream_t s);
void transform_maps(const float* vsrc, const float* nsrc, int cols, int rows, const float* R9_dev, const float* t3_dev,
                    float* vdst, float* ndst, hipStream_t s);
void copy_maps(const float* vtex, const float* ntex, int cols, int rows, float* vmaps_tmp, float* vmap, float* nmap, hipStream_t s);
void resize_map(const float* in, int scols, int srows, float* out, bool normalize, hipStream_t s);
void pyr_down_gauss_f(const float* src, int scols, int srows, float* dst, hipStream_t s);
void pyr_down_uchar_gauss(const uint8_t* src, int scols, int srows, uint8_t* dst, hipStream_t s);
void vertices_to_depth(const float* vmaps_tmp, int cols, int rows, float cutoff, float* dst, hipStream_t s);
void bgr_to_intensity(const uint8_t* src, int channels, int cols, int rows, uint8_t* dst, hipStream_t s);
void derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy, hipStream_t s);
void project_to_point_cloud(const float* depth, int cols, int rows, Intr k, float* cloud, hipStream_t s);

// reductions with explicit parameters (operator tier); results left in out_dev (29 / 29 / 11 floats, 2 ints)
struct IcpArgs {
  float Rcurr[9], tcurr[3], Rprev_inv[9], tprev[3];
  Intr k;
  float distThres, angleThres;
};
// scratch: >= OP_SCRATCH_FLOATS device floats; out*_dev receive the final sums in the reference's member order
constexpr int OP_SCRATCH_FLOATS = SE3_ACCS * VWARPS + 64;
void icp_step_op(const IcpArgs& a, const float* vmap_curr, const float* nmap_curr, const float* vmap_g_prev,
                 const float* nmap_g_prev, int cols, int rows, float* scratch, float* out29_dev, hipStream_t s);
struct RgbResidualArgs {
  float minScale, maxDepthDelta;
  float kt[3], krkinv[9];
};
void rgb_residual_op(const RgbResidualArgs& a, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                     const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, void* corres, int cols,
                     int rows, int* out2_dev, hipStream_t s);
void rgb_step_op(const void* corres, float sigma, const float* cloud, float fx, float fy, const int16_t* dIdx,
                 const int16_t* dIdy, float sobelScale, int cols, int rows, float* scratch, float* out29_dev, hipStream_t s);
struct So3Args { float imageBasis[9], kinv[9], krlr[9]; };
void so3_step_op(const So3Args& a, const uint8_t* lastImage, const uint8_t* nextImage, int cols, int rows, float* scratch,
                 float* out11_dev, hipStream_t s);

// ---- frame-tier launchers (device-resident state; nothing here synchronises) ----
// RGBDOdometry::initICP(filteredDepth, cutoff): u16 pyramid + vertex/normal maps, RGBDOdometry.cpp:121-147
void init_icp(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, hipStream_t s);
// initICPModel + initRGBModel's depth half: predicted (or fill-in, chosen by the dense_count tally) float4 maps ->
// world-frame planar pyramids + model depth L0.  RGBDOdometry.cpp:171-210, :217
// (pred_image_rgba / fill_image_rgba given: the model's level-0 intensity image — populateRGBDData(model) — is written by the same launch)
// `with` given: the depth pre-processing of the new frame (efm::preprocess_depth's work, with the frame's level-0 intensity) is part of the same
// launch — the two read nothing of each other's (round 6: k_frame_inputs)
struct FramePreprocess {
  const uint16_t* raw;          // the frame's raw depth
  float maxD;                   // depth cut-off in metres
  const float* table;           // efm::bilateral_table()
  uint16_t* filtered;
  float* metric;
  float* metric_filtered;
  const uint8_t* rgb3;          // the frame's colours -> Pyramid::nextImage[0]
  uint8_t* rgb_keep;            // copy of the colours, or null
};
void init_icp_model(const Pyramid& p, const float* pred_vertex, const float* pred_normal, const float* fill_vertex,
                    const float* fill_normal, const TrackState* st, float maxDepthRGB, hipStream_t s, const uint8_t* pred_image_rgba = nullptr,
                    const uint8_t* fill_image_rgba = nullptr, bool frameToFrameRGB = false, const FramePreprocess* with = nullptr,
                    bool tally = false /* take denseEnough()'s tally from pred_image_rgba first (ModelMapsArgs::tally_image) */);
// the rest of populateRGBDData (RGBDOdometry.cpp:212-244, :275-279) in three independently enqueueable parts, so that
// the part that only needs the new frame can run on the input stream while the previous frame is still being fused:
//   model ("last"): Gaussian depth pyramid + intensity pyramid of the predicted (or fill-in) image
//   frame ("next"): intensity pyramid of the camera image
//   sobel: derivative images of the "next" pyramid + the iteration-invariant photometric gates (needs both)
void init_rgb_model(const Pyramid& p, const uint8_t* pred_image_rgba, const uint8_t* fill_image_rgba, bool frameToFrameRGB,
                    const TrackState* st, hipStream_t s);
void init_rgb_frame(const Pyramid& p, const uint8_t* rgb3, hipStream_t s);
// initICP(predictedVertices, predictedNormals) + initRGB(predictedImage), RGBDOdometry.cpp:149-169,241-244: the "current" side of
// model-to-model tracking from a model prediction (camera-frame maps, nextDepth from the same vertices, intensity of the image)
void init_icp_maps(const Pyramid& p, const float* vertex4, const float* normal4, const uint8_t* image_rgba, const TrackState* st,
                   float maxDepthRGB, hipStream_t s);
void init_rgb_sobel(const Pyramid& p, hipStream_t s);
// build_pyramids for the two-stream frame script: the half that reads nothing but the new frame / the half that reads the model prediction
void build_pyramids_frame_side(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, const uint8_t* rgb3, hipStream_t s,
                               uint8_t* rgb_keep);
void build_pyramids_model_side(const Pyramid& p, const uint8_t* pred_image_rgba, const uint8_t* fill_image_rgba, bool frameToFrameRGB,
                               const TrackState* st, hipStream_t s);
// init_icp_model + init_rgb_model (model view, its own image, no fill-in) + init_icp_maps (current view) of a model-to-model tracker in five launches
void init_model_pair(const Pyramid& p, const float* model_vertex4, const float* model_normal4, const uint8_t* model_image_rgba,
                     const float* cur_vertex4, const float* cur_normal4, const uint8_t* cur_image_rgba, const TrackState* st, float maxDepthRGB,
                     hipStream_t s);
// init_icp + init_rgb_model + init_rgb_frame in three launches (single-stream frame script; needs init_icp_model first)
void build_pyramids(const Pyramid& p, const uint16_t* depth_filtered, Intr k, float cutoff, const uint8_t* pred_image_rgba,
                    const uint8_t* fill_image_rgba, bool frameToFrameRGB, const uint8_t* rgb3, const TrackState* st, hipStream_t s,
                    uint8_t* rgb_keep = nullptr,    // rgb_keep: also store the frame's RGB there (the caller's buffer is only borrowed)
                    bool with_sobel = false);       // with_sobel: init_rgb_sobel's work in the same launch as the vertex / normal maps
// initFirstRGB, RGBDOdometry.cpp:246-257
void init_first_rgb(const Pyramid& p, const uint8_t* rgb3, hipStream_t s);
// getIncrementalTransformation, RGBDOdometry.cpp:259-571, entirely enqueued.  The update step of an iteration is evaluated at the head of the
// NEXT iteration's first kernel, the last one at the head of track_end's: TrackTail says where that chain stands (which GNState buffer,
// which residual-sum set, the step's parameters).  Pass it to the track_end that follows on the same stream.
struct TrackTail {
  int cur, slots;
  bool has_head, icp, rgb, rgbOnly;
  float icpWeight;
  Intr k0;
  const float* pairs;
  int ng = 0;                  // fast order: group partials per accumulator the head's tree looks at (1: the persistent launch left totals)
  // reference-order builds, after the persistent launch: track_end is ONE launch that also stands where k_track_serial stood (k_track_ref_end)
  bool merged_end = false;
  unsigned epoch = 0;          // the persistent launch's first epoch (its admission verdict is tagged with it)
  float* partials = nullptr;
};
// probe: samples the level-0 normal-equation launches of the launch-per-step script; probe_all: samples the persistent launch (fast order)
TrackTail track(Pyramid& p, TrackState* st, Intr k, const TrackParams& tp, hipStream_t s, KernelProbe* probe = nullptr,
                KernelProbe* probe_all = nullptr);
// 1 when a persistent launch of this tracker instance gave up waiting in a grid barrier (its workgroups were not co-resident), 0
// otherwise, < 0 on a HIP error; synchronises the stream
int tracker_aborted(const Pyramid& p, hipStream_t s);
// the script track() will run with these parameters on a stream that is NOT capturing; switches the instance over to it (clearing the exchange
// areas, carrying their sticky words over) — for a caller that is about to capture track() and must not have the clear inside its graph
void track_prepare(Pyramid& p, const TrackParams& tp, hipStream_t s);
// fast order: persistent launches of this tracker instance that found the chip partly taken (admission failed) and ran on ONE workgroup
// instead — same results, ~25x the time; < 0 on a HIP error; synchronises the stream
int tracker_fallbacks(const Pyramid& p, hipStream_t s);
// a stream that launched persistent trackers is about to be destroyed (or synchronised for good): the per-device chain must not record on it
void persistent_chain_forget(hipStream_t s);
int tracker_small_clocks(const Pyramid& p, unsigned long long* out32, hipStream_t s);   // developer instrumentation (-DEF_STAGE_CLOCKS)
void track_swap(Pyramid& p, const TrackParams& tp);   // the pointer swap track() ends with (for hipGraph replay)
// tail of getIncrementalTransformation (0.3 m guard, SVD re-orthonormalisation, RGBDOdometry.cpp:555-570) +
// velocity weighting (ElasticFusion.cpp:369-383) + the float matrices of the map passes
// traj / slot: device trajectory log (16 doubles per frame; t_T_wc of ElasticFusion.cpp:588) or null
// abort_word / abort_report: the instance's sticky abort flag (tracker_abort_word) and a word of host-mapped pinned memory it is copied to
// when set (both null: not reported)
void track_end(TrackState* st, const TrackTail& tail, bool rgb, float weightMultiplier, double* traj, int slot, hipStream_t s,
               const unsigned* abort_word = nullptr, unsigned* abort_report = nullptr);
unsigned* tracker_abort_word(const Pyramid& p);
// caller-supplied pose (in_T_wc, ElasticFusion.cpp:367-369): sets q/t from the row-major 4x4, optionally keeping the old
// pose as "previous" for the velocity weighting, publishes the float matrices, re-arms the denseEnough() tally
void pose_injected(TrackState* st, const double* T_wc16, bool save_prev, float weightMultiplier, bool with_weighting, double* traj,
                   int slot, hipStream_t s);
void log_pose(const TrackState* st, double* traj, int slot, hipStream_t s);
// checkpoint restore (ef_restore_state): the pose exactly as a context held it (quaternion xyzw + translation), float matrices published,
// the denseEnough() tally re-armed for the predict() that follows
void pose_restored(TrackState* st, const double* q4, const double* t3, hipStream_t s);
// local loop closure plumbing (ElasticFusion.cpp:469-527): T_wc_est := T_wc_curr before the model-to-model tracker runs;
// T_wc_curr := T_wc_est after an accepted deformation; the (W/20)x(H/20) constraint samples {x, y, z, inactive time}
void copy_pose(TrackState* dst, const TrackState* src, hipStream_t s);
void adopt_pose(TrackState* st, const TrackState* est, double* traj, int slot, hipStream_t s);
void sample_constraints(const float* vertex4, const uint16_t* old_time, int cols, int rows, int step, float* out4, hipStream_t s);

}  // namespace eft
