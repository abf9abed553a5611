/* libefusion_hip.so — C ABI of the MI355X-native ElasticFusion per-frame engine.
 *
 * Plain C, opaque context, raw pointers and sizes only; every call returns 0 on success or a negative
 * EF_E* code (ef_last_error() gives the message).  Nothing exits the process, nothing is global: the
 * reference's process-wide Resolution/Intrinsics singletons (Core/Utils/Resolution.h:25-58,
 * Intrinsics.h:25-51) become fields of ef_config.  A context is single-owner and externally
 * synchronised, the same contract as the reference's ElasticFusion object.
 *
 * Three tiers, mirroring the reference's own layering:
 *   1. frame tier      ef_create / ef_process_frame / getters      <-> class ElasticFusion
 *                                                                       (Core/ElasticFusion.h:40-255)
 *   2. subsystem tier  ef_preprocess / ef_track / ef_predict / ...  <-> RGBDOdometry, IndexMap,
 *                                                                       GlobalModel, FillIn, ComputePack
 *   3. operator tier   ef_op_*  on raw DEVICE pointers               <-> the 17 free functions of
 *                                                                       Core/Cuda/cudafuncs.cuh:61-169
 *                                                                       and the GLSL passes (one each)
 * All image layouts are the reference's (planar float[3*rows][cols] maps, 16-byte DataTerm,
 * 48-byte surfels = 3 x vec4).  "dev" pointers are HIP device pointers.
 */
#ifndef EF_HIP_H_
#define EF_HIP_H_
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EF_OK 0
#define EF_EINVAL (-1)   /* bad argument */
#define EF_EHIP (-2)     /* HIP runtime error (message has the hipError string) */
#define EF_ENOMEM (-3)
#define EF_ESTATE (-4)   /* call not valid in the current state */
#define EF_ECAPACITY (-5) /* surfel capacity exceeded: the map was clamped to max_surfels (a warning: ef_synchronize reports it once and clears it; the clamped map stays readable) */

typedef struct ef_ctx ef_ctx;

/* ---- configuration: ElasticFusion ctor arguments (Core/ElasticFusion.h:42-58) + the two singletons */
typedef struct ef_config {
  int width, height;          /* Resolution::getInstance(w,h)                       */
  float fx, fy, cx, cy;       /* Intrinsics::getInstance(fx,fy,cx,cy)               */
  int time_delta;             /* timeDelta       (200; INT_MAX/2 in open loop)      */
  float confidence;           /* confidence      (10)                               */
  float depth_cut;            /* depthCut        (3 m)                              */
  float icp_weight;           /* icpThresh       (10)                               */
  int fast_odom;              /* fastOdom        (0)                                */
  int so3;                    /* so3             (1)                                */
  int frame_to_frame_rgb;     /* frameToFrameRGB (0)                                */
  int pyramid;                /* setPyramid      (1)                                */
  int rgb_only;               /* setRgbOnly      (0)                                */
  int close_loops;            /* closeLoops (0 = -o): 1 runs the LOCAL loop closure every frame (below) and, once
                                 ef_enable_global_closure was called, the fern-based GLOBAL one before it */
  uint32_t max_surfels;       /* surfel capacity; reference: 3072*3072 (GlobalModel.cpp:22-24) */
  int device;                 /* HIP device ordinal                                 */
  void* stream;               /* hipStream_t to run on, or NULL to create a private one */
} ef_config;

void ef_default_config(ef_config* cfg);   /* front-end defaults, MainController.cpp:37-43,69-104, -o */

/* ---- lifecycle ---- */
int ef_create(const ef_config* cfg, ef_ctx** out);
void ef_destroy(ef_ctx* ctx);
const char* ef_last_error(const ef_ctx* ctx);   /* ctx may be NULL: last error of a failed ef_create */
void* ef_stream(ef_ctx* ctx);                   /* the hipStream_t all work is enqueued on */
int ef_synchronize(ef_ctx* ctx);

/* ---- frame tier ----
 * ef_process_frame == ElasticFusion::processFrame(rgb, depth, timestamp, weightMultiplier, in_T_wc)
 * (Core/ElasticFusion.h:70-75).  rgb: W*H*3 bytes row-major; depth: W*H uint16 millimetres, 0 invalid.
 * Host pointers are borrowed for the duration of the call (staged synchronously, like the reference's
 * glTexture upload, ElasticFusion.cpp:278-280); all device work is only ENQUEUED: the call does not
 * wait for the GPU.  in_T_wc: 16 doubles row-major or NULL.
 * ef_process_frame_dev takes DEVICE pointers (frames already resident in HBM; the bench path). */
int ef_process_frame(ef_ctx* ctx, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp,
                     float weight_multiplier, const double* in_T_wc16);
int ef_process_frame_dev(ef_ctx* ctx, const uint8_t* rgb_dev, const uint16_t* depth_dev, int64_t timestamp,
                         float weight_multiplier, const double* in_T_wc16);
/* Input-stage overlap (default off; on = 1, or 2 to start the copy-in and the bilateral filter already during the previous tracker): the part of a frame that
 * needs only the new images (copy-in, bilateral filter + metric depth, frame-side pyramids) is enqueued on a second
 * internal stream and runs while the previous frame is still being fused.  Results are identical either way.  A context whose tracker is the
 * persistent launch (ef_set_persistent_tracker 1, the default on a 256-CU chip) treats 2 as 1: that launch needs every CU to itself. */
int ef_set_input_overlap(ef_ctx* ctx, int on);
/* ... and that second stream restricted to every n-th CU of the chip (n <= 1: no restriction, the default): the bilateral filter of frame k + 1,
 * the one ALU-bound kernel of a frame, then shares the chip with the latency-bound fusion / prediction kernels of frame k instead of displacing
 * them.  (With the persistent tracker, which needs the whole chip to itself, use overlap mode 1: the input stage waits for the tracker.) */
int ef_set_input_cu_mask(ef_ctx* ctx, int one_in_n);
/* hipGraph replay of the tracker (default off): the ~70 kernel launches of one getIncrementalTransformation
 * (RGBDOdometry.cpp:259-571) are captured once per pyramid parity and replayed with one hipGraphLaunch per frame.
 * Identical results; it only trims host-side launch work (BASELINE.json configs[4]). */
int ef_set_graph_replay(ef_ctx* ctx, int on);
/* The whole of RGBDOdometry::getIncrementalTransformation (RGBDOdometry.cpp:259-553: the SO(3) pre-alignment loop and every Gauss-Newton
 * iteration of every pyramid level, with their update steps) as ONE persistent launch of 256 co-resident workgroups, one per CU (default, on)
 * instead of one launch per step (off: 68 launches per call).  The same sums in the same order: results are bit-identical either way
 * (tests/test_gpu_frame.py runs both).  The launch needs the whole chip at once; it checks that at its start, and when other work holds
 * part of the chip for milliseconds (another process, a long kernel on another stream) the call runs on one workgroup instead — slower, the
 * same results, nothing for the caller to do (ef_get_tracker_fallbacks counts these).  Persistent launches of one process on one device are
 * chained in enqueue order, so several contexts never starve one another.  With rgbOnly and under ef_set_graph_replay the launch-per-step
 * script runs.  on = 2 (reference-order builds: the default library; what a device that reports 128 .. 255 CUs gets by default, fewer: 0):
 * round 3's form — the levels of at most 131072 pixels + the SO(3) loop as one launch of 128 workgroups, three launches per level-0 iteration —
 * whose time-out makes ef_synchronize return EF_EHIP.  A protocol failure of either persistent form (a wait that timed out after the whole
 * grid had reported in) is sticky and is reported by the NEXT ef_process_frame[_dev] and by ef_synchronize (EF_EHIP). */
int ef_set_persistent_tracker(ef_ctx* ctx, int on);
/* Level-0 Gauss-Newton iterations as TWO launches instead of three: the update step (one workgroup's worth of work) is evaluated by
 * workgroup 0 of the correspondence-search launch and handed to that launch's other workgroups as tagged granules (they poll with their
 * pixel loads in flight) instead of being a launch of its own.  Same arithmetic, bit-identical results.  Off by default (measured:
 * DESIGN.md 6). */
int ef_set_fused_step(ef_ctx* ctx, int on);
/* The persistent tracker launch keeps the pose-INDEPENDENT inputs of a pyramid level's pixel visits (current vertex / normal maps: the operands
 * of icpStep's search that the pixel itself addresses, Core/Cuda/reduce.cu:228-262; the frame's depth, intensity, gradients and the photometric
 * gate: residualKernel's, reduce.cu:631-667) in LDS for all iterations of the level (round 6; default on).  0 = stream them from memory in every
 * iteration as round 5's launch did (A/B).  Same arithmetic on the same values: bit-identical results. */
int ef_set_resident_levels(ef_ctx* ctx, int on);
/* Odometry only (BASELINE.json configs[4], "open-loop odometry-only ... throughput ceiling"): frames are pre-processed, tracked against
 * the model prediction and the prediction is renewed at the new pose, but nothing is fused (the map stays as it is: indexMap, fuse and
 * clean of ElasticFusion.cpp:536-585 are skipped, like a frame whose tracking failed under relocalisation).  Off by default. */
int ef_set_track_only(ef_ctx* ctx, int on);
/* Device half of a loop closure: hands a deformation graph (HOST pointer, nodes x 16 floats sorted by time, layout of
 * GlobalModel::clean's rawGraph, GlobalModel.cpp:536-546) to the NEXT ef_process_frame, whose clean pass applies it to the
 * whole map exactly as ElasticFusion.cpp:558-585 does (synthesizeDepth first unless is_fern).  For a caller that finds loop closures
 * and optimises the graph itself — or with ef_closure_* / ef_solve_deformation below. */
int ef_set_deformation(ef_ctx* ctx, const float* graph_host, int nodes, int is_fern);
/* ---- local loop closure, front half (ElasticFusion.cpp:447-527; contexts created with close_loops = 1) ----
 * After tracking, every frame: the INACTIVE part of the model (surfels not seen for time_delta frames) is predicted into the
 * camera (IndexMap::combinedPredict(..., INACTIVE), IndexMap.cpp:293-393), a second tracker (RGBDOdometry modelToModel) registers
 * it against the ACTIVE prediction, and if the covariance / ICP-count / ICP-error gates hold (:473-484) the surface constraints
 * of :485-509 are sampled every 20 pixels (Resize::vertex / Resize::time).  What the reference does next — Deformation::constrain,
 * a sparse non-linear solve with CHOLMOD — is the registered solver's job: it receives the constraints and may return a
 * deformation graph (same layout as ef_set_deformation); the engine then does what :514-527 and :558-585 do: T_wc := T_wc_est,
 * synthesizeDepth, and this frame's clean pass applies the graph to the whole map.  Costs one stream synchronisation per frame,
 * where the reference reads the constraint buffers back. */
typedef struct ef_local_loop {
  int attempted;            /* the front half ran in the last ef_process_frame (tick > 1) */
  int cov_ok;               /* no diagonal entry of modelToModel's covariance above covThresh */
  int gates_ok;             /* cov_ok && lastICPCount > icpCountThresh && lastICPError < icpErrThresh */
  int n_constraints;        /* surface constraints sampled (0 unless gates_ok) */
  int applied;              /* the solver accepted: pose replaced, graph (if any) applied by this frame's clean */
  int graph_nodes;
  int graph_capacity;       /* nodes the solver's graph_out has room for (1023 = GlobalModel::MAX_NODES - 1); set before the solver is called */
  int reserved_;
  float stats[6];           /* modelToModel: lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count */
  double cov_diag[6];       /* diagonal of getCovariance() up to and including the first entry above covThresh */
  double T_wc_curr[16];     /* pose after frame-to-model tracking, row-major */
  double T_wc_est[16];      /* pose proposed by the model-to-model registration */
} ef_local_loop;
/* constraints: n rows of 8 doubles {vert_w_curr xyz (source), vert_w_est xyz (target), time the inactive surface was last seen,
 * pin (1 while no deformation has been applied yet, ElasticFusion.cpp:507-508)}.  graph_out has room for info->graph_capacity x 16 floats
 * (1023): a solver must not write more; a *nodes_out beyond it makes the frame fail with EF_EINVAL.
 * Return non-zero to accept (Deformation::constrain returning true); called on the thread inside ef_process_frame. */
typedef int (*ef_loop_solver)(void* user, const ef_local_loop* info, const double* constraints, int n, float* graph_out, int* nodes_out);
int ef_set_loop_solver(ef_ctx* ctx, ef_loop_solver fn, void* user);   /* NULL: gates and constraints are still evaluated */
/* The built-in solver in that place: Deformation::constrain(..., fernMatch = false) re-implemented as a host-side banded Gauss-Newton
 * optimiser of the embedded deformation graph (Deformation.cpp:88-215, DeformationGraph.cpp; no CHOLMOD), run on the graph nodes
 * ef_sample_graph yields for the current map.  With it a closeLoops context closes LOCAL loops end to end.  A registered
 * ef_loop_solver takes precedence. */
int ef_use_builtin_loop_solver(ef_ctx* ctx, int on);
/* the same optimiser on explicit inputs (host arrays, no context, no GPU): nodes4 = n_nodes x {x, y, z, time} ascending in time,
 * constraints8 as ef_get_local_loop returns them (all sourced at src_time), nodes no younger than last_deform_time stay fixed;
 * graph16_out = n_nodes x 16 floats in the layout ef_set_deformation takes.  EF_ESTATE when there are not more than 4 nodes. */
int ef_solve_local_deformation(const float* nodes4, int n_nodes, const double* constraints8, int n_constraints, int64_t src_time,
                               int64_t last_deform_time, float* graph16_out, float* error_out, float* mean_constraint_error_out);
/* Deformation::constrain in its general form (Deformation.cpp:88-215), the GLOBAL deformation included: constraints are the entries of
 * Deformation::constraints — `relative`: the source has to land wherever the graph carries the target (the constraints a local closure
 * leaves behind, Deformation.cpp:160-173); `pin`: a target held in place (src == target) —, fern_match selects the global closure's
 * rules (nothing to do below 0.06 m mean constraint error; accepted only when the optimised mean error is below 3e-4 and the energy
 * below 0.12; pass last_deform_time = 0 as Deformation.cpp:143 does).  poses16 (n_poses x 16, in/out) with pose_times: the keyframe
 * poses (and with fern_match the trajectory) carried along by DeformationGraph::applyGraphToPoses — translation only, as observed in
 * the reference.  For the global graph nodes4 is every 5th node of ef_sample_graph (Deformation::sampleGraphFrom, :217-237).
 * new_relative_out (room for n_constraints entries, optional): after an accepted LOCAL solve, the relative constraints to hand to
 * later global ones (newRelativeCons: the deformed source of every plain constraint against its target).
 * EF_OK: accepted, graph16_out / poses16 written; EF_ESTATE: rejected (or not more than 4 nodes). */
typedef struct ef_graph_constraint {
  double src[3], target[3];
  int64_t src_time, target_time;
  int relative, pin;
} ef_graph_constraint;
int ef_solve_deformation(const float* nodes4, int n_nodes, const ef_graph_constraint* constraints, int n_constraints, int fern_match,
                         int64_t last_deform_time, double* poses16_inout, const int64_t* pose_times, int n_poses, float* graph16_out,
                         float* error_out, float* mean_constraint_error_out, ef_graph_constraint* new_relative_out_or_null,
                         int* n_new_relative_out_or_null);
/* the same with the three gates of a global closure spelled out (NULL = the reference's constants {0.06, 3e-4, 0.12}: entry below which
 * there is nothing to close, acceptance bounds on the optimised mean constraint error and on the energy) */
int ef_solve_deformation_gated(const float* nodes4, int n_nodes, const ef_graph_constraint* constraints, int n_constraints, int fern_match,
                               int64_t last_deform_time, double* poses16_inout, const int64_t* pose_times, int n_poses, float* graph16_out,
                               float* error_out, float* mean_constraint_error_out, ef_graph_constraint* new_relative_out_or_null,
                               int* n_new_relative_out_or_null, const float* gates3_or_null);
/* icpCountThresh, icpErrThresh, covThresh of the constructor (ElasticFusion.h:44-46; defaults 35000, 5e-05, 1e-05) */
int ef_set_loop_thresholds(ef_ctx* ctx, int icp_count_thresh, float icp_err_thresh, float cov_thresh);
int ef_get_local_loop(ef_ctx* ctx, ef_local_loop* info, double* constraints_or_null, int max_constraints, int* n_out_or_null);
/* Deformation::sampleGraphModel (Deformation.cpp:232-306, sample.vert + sample.geom): the deformation graph's nodes, every 5000th
 * surfel of the current model in map order as {x, y, z, initTime} (times ascend because the map keeps creation order); what the
 * reference hands to DeformationGraph::initialiseGraph.  nodes4_host: max_nodes x 4 floats.  Synchronises. */
int ef_sample_graph(ef_ctx* ctx, float* nodes4_host, int max_nodes, int* n_out);
/* ---- fern database (Core/Ferns.h:35-184, Core/Ferns.cpp:22-393): the keyframe store of the GLOBAL loop closure and of
 * relocalisation.  Host-side object with no GPU state of its own: it encodes the 1/8-resolution predicted views
 * (ef_get_image_resized(EF_IMG_FILL_*, 8, ...)) with `num` random ferns (4 binary tests each: r, g, b
 * against 0..255, depth in mm against 400..max_depth_mm), keeps a frame when it differs enough from all stored ones, and proposes
 * the most similar stored frame for a new view.  The fern-to-view registration (an 80x60 ICP, Ferns.cpp:243-258) is the caller's:
 * ef_fern_tracker gets both vertex / normal images and the stored pose, refines T_inout16 and reports the ICP statistics.
 * Images: rgb = rows of `rgb_channels` (3 or 4) bytes per pixel, verts4 / norms4 = f32x4 per pixel, (W/8) x (H/8) pixels. */
typedef struct ef_ferns ef_ferns;
/* Ferns::Ferns(n, maxDepth, photoThresh) with Resolution / Intrinsics spelled out; the table is drawn from std::mt19937(seed) in the
 * reference's order (Ferns.cpp:62-77; the reference seeds with time(0)). */
ef_ferns* ef_ferns_create(int num, int max_depth_mm, float photo_thresh, int width, int height, float fx, float fy, float cx, float cy,
                          unsigned seed);
void ef_ferns_destroy(ef_ferns* f);
/* the fern table, num rows of {x, y, r, g, b, d}; setting it is only allowed while no frame is stored */
int ef_ferns_get_table(const ef_ferns* f, int* table6);
int ef_ferns_set_table(ef_ferns* f, const int* table6);
/* Ferns::addFrame (Ferns.cpp:78-160): returns 1 when the frame was stored, 0 when it was too similar (or had no valid code),
 * a negative EF_E* on bad arguments */
int ef_ferns_add_frame(ef_ferns* f, const uint8_t* rgb, int rgb_channels, const float* verts4, const float* norms4, const double* T_wc16,
                       int src_time, float threshold);
typedef void (*ef_fern_tracker)(void* user, const float* fern_verts4, const float* fern_norms4, const double* T_wc_fern16,
                                const float* cur_verts4, const float* cur_norms4, double* T_inout16, float* icp_error, float* icp_count);
/* Ferns::findFrame (Ferns.cpp:162-298).  T_est16_out: the recovered pose (identity when no candidate passed the code gates);
 * constraints6_out: up to max_constraints rows {T_wc * p (source), T_est * p (target)} (Ferns.cpp:268-293).  Returns lastClosest
 * (the matched frame's id) or -1. */
int ef_ferns_find_frame(ef_ferns* f, const uint8_t* rgb, int rgb_channels, const float* verts4, const float* norms4, const double* T_wc16,
                        int time, int lost, ef_fern_tracker tracker, void* user, double* T_est16_out, double* constraints6_out,
                        int max_constraints, int* n_constraints_out);
/* The same two with the view's fern codes computed by the CALLER — on the device: ef_process_frame runs one 512-thread kernel over
 * the full-resolution fill-in maps (texel (8 x + 4, 8 y + 4) under each fern = what Resize::image / Resize::vertex would hand to
 * Ferns.cpp:97-118) and reads back `num` code bytes (255 = no valid depth) + their count instead of three 1/8-resolution images.  The
 * view itself is asked for through `fetch` only when it is needed: a keyframe passed the code gates (findFrame) or the frame is kept
 * (addFrame).  fetch returns EF_OK and the three images (rgb with 3 or 4 channels); at most one call per call. */
typedef int (*ef_view_fetch)(void* user, const uint8_t** rgb_out, int* rgb_channels_out, const float** verts4_out, const float** norms4_out);
int ef_ferns_add_frame_coded(ef_ferns* f, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int src_time,
                             float threshold);
int ef_ferns_find_frame_coded(ef_ferns* f, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int time,
                              int lost, ef_fern_tracker tracker, void* user, double* T_est16_out, double* constraints6_out, int max_constraints,
                              int* n_constraints_out);
/* 1 when some stored frame is more than 300 ticks older than `time` (Ferns.cpp:218), i.e. findFrame CAN match; 0: it returns -1 whatever the view */
int ef_ferns_candidate_possible(ef_ferns* f, int time);   /* answering 0 it also resets lastClosest to -1, as the findFrame it stands for would */
int ef_ferns_table_version(const ef_ferns* f);            /* bumped by every ef_ferns_set_table */
int ef_ferns_count(const ef_ferns* f);          /* frames.size() */
int ef_ferns_last_closest(const ef_ferns* f);   /* lastClosest */
/* one stored frame: any output may be NULL.  codes_out: num bytes (255 = no valid depth under that fern). */
int ef_ferns_get_frame(const ef_ferns* f, int id, uint8_t* codes_out, int* good_codes_out, int* src_time_out, double* T_wc16_out,
                       uint8_t* rgb3_out, float* verts4_out, float* norms4_out);
/* Deformation::constrain's applyGraphToPoses (Deformation.cpp:196-203) hands the deformed poses back to the stored frames */
int ef_ferns_set_frame_pose(ef_ferns* f, int id, const double* T_wc16);
/* the two private measures, for tests: Ferns::blockHDAware of two stored frames; Ferns::photometricCheck of a view against frame id */
float ef_ferns_block_hd_aware(const ef_ferns* f, int id_a, int id_b);
float ef_ferns_photometric_check(const ef_ferns* f, const uint8_t* rgb, int rgb_channels, const float* verts4, const double* T_est16, int id);
/* ---- the host side of the loop closures around that database (ElasticFusion.cpp:392-445, 511-526, 588-589, 609-618): one object holds
 * the fern database, the relative constraints local closures leave behind, the trajectory (t_T_wc) and the two deformation counters, and
 * takes the decisions — which constraints go to which graph, what an accepted closure changes (keyframe and trajectory poses deformed
 * along).  No device work: the caller brings the 1/8-resolution fill-in views, the pose, the sampled graph (ef_sample_graph; the global
 * graph is every 5th node of it, Deformation::sampleGraphFrom) and the fern-to-view registration (ef_fern_tracker).  ef_process_frame
 * drives one of these itself once ef_enable_global_closure was called (below). */
typedef struct ef_closure ef_closure;
ef_closure* ef_closure_create(int num_ferns, float depth_cut, float photo_thresh, float fern_thresh, int width, int height, float fx, float fy, float cx,
                              float cy, unsigned seed);   /* Ferns(500, depthCut * 1000, photoThresh); fernThresh 0.3095 */
void ef_closure_destroy(ef_closure* c);
ef_ferns* ef_closure_ferns(ef_closure* c);                /* owned by the closure object */
/* mid-frame, after predict() (:392-445): fern match -> fern constraints with their pins + the kept relative constraints -> global
 * deformation.  1: accepted — T_recovery16_out is the new pose, graph16_out / nodes_out go to the clean pass with is_fern = 1 and the
 * local closure is skipped; 0: no match or rejected (T_recovery16_out still holds the registration, identity without a candidate). */
int ef_closure_global(ef_closure* c, const uint8_t* rgb, int rgb_channels, const float* verts4, const float* norms4, const double* T_wc16, int tick,
                      ef_fern_tracker tracker, void* user, const float* nodes4, int n_nodes, double* T_recovery16_out, float* graph16_out, int* nodes_out);
/* the local closure's far half (:511-526), gates already open: constraints8 as ef_get_local_loop returns them.  1: accepted (graph16_out
 * over all n_nodes; the keyframe poses followed; a third of the new relative constraints kept), 0: rejected */
int ef_closure_local(ef_closure* c, const double* constraints8, int n, int tick, const float* nodes4, int n_nodes, float* graph16_out, int* nodes_out);
/* end of the frame (:588-589, 609-618): pose -> trajectory, final fill-in view -> Ferns::addFrame; returns 1 when it became a keyframe */
int ef_closure_end_frame(ef_closure* c, const uint8_t* rgb, int rgb_channels, const float* verts4, const float* norms4, const double* T_wc16, int tick);
int ef_closure_counts(const ef_closure* c, int* deforms, int* fern_deforms, int* relative_constraints, int* trajectory_poses);
/* the gates of the GLOBAL deformation (defaults = the reference's hard-coded 0.06 m entry, 3e-4 m / 0.12 acceptance: DeformationGraph.cpp:425,
 * Deformation.cpp:154; tuned on room-scale trajectories) */
int ef_closure_set_gates(ef_closure* c, float entry_mean_error, float accept_mean_error, float accept_energy);
int ef_closure_set_fern_thresh(ef_closure* c, float fern_thresh);   /* ElasticFusion::setFernThresh: Ferns::addFrame's threshold from now on */
/* introspection for tests: the rows the last closure handed to the optimiser (returns their number) and its two error figures */
int ef_closure_last_rows(const ef_closure* c, ef_graph_constraint* rows_or_null, int max_rows, float* error_or_null, float* mean_constraint_error_or_null);
int ef_closure_relative(const ef_closure* c, ef_graph_constraint* rows_or_null, int max_rows);
int ef_closure_trajectory(const ef_closure* c, double* poses16_or_null, int max_poses);
/* a LOST camera's frame (ElasticFusion.cpp:395-413 with lost = true): Ferns::findFrame only — 1 and the registered pose when a keyframe
 * passes the gates (ICP count > 1400), 0 otherwise; and its end (:588-589 without :601-604): the pose joins the trajectory, no keyframe */
int ef_closure_relocalise(ef_closure* c, const uint8_t* rgb, int channels, const float* verts4, const float* norms4, const double* T_wc16, int tick,
                          ef_fern_tracker tracker, void* user, double* T_recovery16_out);
int ef_closure_log_pose(ef_closure* c, const double* T_wc16, int tick);
/* ef_closure_global / ef_closure_end_frame / ef_closure_relocalise on device-computed fern codes (ef_ferns_*_coded above) */
int ef_closure_global_coded(ef_closure* c, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int tick,
                            ef_fern_tracker tracker, void* user, const float* nodes4, int n_nodes, double* T_recovery16_out, float* graph16_out,
                            int* nodes_out);
int ef_closure_candidate_possible(ef_closure* c, int tick);   /* 0: this frame's findFrame cannot match (no keyframe older than 300 ticks); the object is left as after such a call */
int ef_closure_end_frame_coded(ef_closure* c, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int tick);
int ef_closure_relocalise_coded(ef_closure* c, const uint8_t* codes, int good_codes, ef_view_fetch fetch, void* fetch_user, const double* T_wc16, int tick,
                                ef_fern_tracker tracker, void* user, double* T_recovery16_out);
/* ---- the GLOBAL loop closure inside ef_process_frame (ElasticFusion.cpp:392-445, 588-589, 609-618; contexts created with
 * close_loops = 1).  Creates the context's closure object (ef_closure_* above: Ferns(num_ferns, depth_cut * 1000, photo_thresh), the
 * relative constraints, the trajectory) and a third tracker instance at 1/8 resolution.  From then on every frame (tick > 1):
 *   after tracking    predict() (ACTIVE + fill-in) at the new pose; the fill-in view, NEAREST-resized by 8, goes to Ferns::findFrame;
 *                     a candidate keyframe is registered against the view by the 1/8-resolution tracker on the device (ICP only,
 *                     10 iterations at one level); its surface constraints with their pins + the kept relative constraints go to the
 *                     global deformation (every 5th graph node); accepted => T_wc := the recovered pose, keyframe and trajectory poses
 *                     deformed along, the graph applied by this frame's clean pass as a fern match, the local closure skipped;
 *                     otherwise the local closure runs — through the closure object when the built-in solver is on
 *                     (ef_use_builtin_loop_solver), so keyframe poses follow and a third of the new relative constraints is kept;
 *   end of the frame  the final fill-in view goes to Ferns::addFrame, the pose joins the trajectory.
 * Two stream synchronisations per frame, where the reference reads the views back (Resize.cpp:50-159).  The reference seeds its fern
 * table from time(0); here the seed is an argument (equal seeds => equal runs). */
int ef_enable_global_closure(ef_ctx* ctx, int num_ferns, float photo_thresh, float fern_thresh, unsigned seed);
/* ---- relocalisation (the reference constructor's `reloc`; ElasticFusion.cpp:326-366, 402-413, 536, 601-604, 624-649).  With it on,
 * every TRACKED frame (no injected pose) is judged by its own statistics, read back right after the tracker (one more synchronisation
 * per frame): trackingOk = lastICPError < 1e-4 and no diagonal entry of getCovariance() above 1e-4.  A frame that is not ok is not
 * fused; more than ten of them in a row and the camera is LOST: the fill-in passes the raw frame through, the tick stands still, no
 * keyframe is stored, neither closure runs — but Ferns::findFrame keeps looking (ICP count gate 1400 instead of 2400) and a match
 * becomes the pose (ef_global_loop.closest >= 0 with accepted = 0).  The frame after such a recovery is predicted from the whole model
 * (time = 0) and, if its tracking is ok, the camera is found again.  Recovery needs the global closure (ef_enable_global_closure);
 * without it a lost camera stays lost, as in the reference with closeLoops = false. */
int ef_set_relocalisation(ef_ctx* ctx, int on);
typedef struct ef_reloc_state {
  int lost;                 /* ElasticFusion::getLost() */
  int tracking_ok;          /* of the last frame (1 for frames with an injected pose) */
  int tracking_count;       /* consecutive frames not ok, towards "lost" at > 10 */
  int last_frame_recovery;  /* the last frame's pose came from a fern match while lost */
} ef_reloc_state;
int ef_get_relocalisation(ef_ctx* ctx, ef_reloc_state* out);
typedef struct ef_global_loop {
  int attempted;            /* Ferns::findFrame ran in the last ef_process_frame */
  int closest;              /* matched keyframe (Ferns::lastClosest), -1: none passed the gates */
  int n_constraints;        /* surface constraints of the match */
  int accepted;             /* the global deformation was accepted: pose replaced, graph applied as a fern match */
  int graph_nodes;
  float icp_error, icp_count;   /* of the 1/8-resolution registration, when it ran */
  double T_wc_recovery[16];     /* the registered pose (identity without a candidate) */
} ef_global_loop;
int ef_get_global_loop(ef_ctx* ctx, ef_global_loop* info);
ef_closure* ef_get_closure(ef_ctx* ctx);   /* the context's closure object (NULL before ef_enable_global_closure); owned by the context */
int ef_predict(ef_ctx* ctx);                                  /* ElasticFusion::predict() */
int ef_get_pose(ef_ctx* ctx, double* T_wc16);                 /* get_T_wc(); synchronises */
int ef_get_tick(ef_ctx* ctx, int* tick);                      /* getTick() */
int ef_set_tick(ef_ctx* ctx, int tick);                       /* setTick() */
/* lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count (RGBDOdometry.h:74-79) */
int ef_get_tracking_stats(ef_ctx* ctx, float* out6, double* lastA36_or_null, double* lastb6_or_null);
/* RGBDOdometry::getCovariance (RGBDOdometry.cpp:573-575): lastA.lu().inverse(), 36 doubles row-major; synchronises */
int ef_get_covariance(ef_ctx* ctx, double* cov36);
int ef_get_trajectory(ef_ctx* ctx, double* T_wc16_array, int64_t* timestamps, int max_frames, int* n_frames);
int ef_map_count(ef_ctx* ctx, uint32_t* count);               /* GlobalModel::lastCount(); synchronises */
int ef_map_download(ef_ctx* ctx, float* surfels, uint32_t max_surfels, uint32_t* count); /* downloadMap(), 12 floats each */
int ef_map_upload(ef_ctx* ctx, const float* surfels, uint32_t count);  /* test/bench seeding */
/* Checkpoint / resume of a replay.  What ElasticFusion carries from one processFrame to the next is the map, the tick,
 * T_wc and the previous frame (its intensity pyramid is the SO(3) reference of RGBDOdometry.cpp:284-288, its filtered depth and colour
 * feed FillIn, ElasticFusion.cpp:621-653); everything else is re-derived.  ef_get_pose_qt returns T_wc exactly as the engine holds it
 * (unit quaternion x y z w + translation, doubles: no round trip through a matrix).  ef_restore_state, called on a context whose map
 * was brought in with ef_map_upload, sets tick and pose, takes the frame that was processed LAST before the checkpoint (host pointers,
 * W*H*3 bytes and W*H uint16), rebuilds its pre-processing and SO(3) reference and runs predict(): the next ef_process_frame then tracks
 * and fuses exactly as the checkpointed context's next frame does (tests/test_gpu_one_frame.py).  The velocity weighting needs no extra
 * state: it compares the pose before and after the tracker of the same frame. */
int ef_get_pose_qt(ef_ctx* ctx, double* q4_t3);
int ef_restore_state(ef_ctx* ctx, int tick, const double* q4_t3, const uint8_t* rgb_prev, const uint16_t* depth_prev);
/* Which buffer ef_map_download / ef_save_ply read.  0 (default): model(), the map as it stands after the frame's clean pass.
 * 1: what GlobalModel::downloadMap really reads (GlobalModel.cpp:673-706, quirk Q14): vbos[renderSource], i.e. the buffer the frame's
 * UPDATE pass wrote (the map before clean) truncated to the count AFTER clean — entries beyond the pre-clean count are whatever older
 * update passes left there (zeros at first).  Byte-for-byte the reference's downloadMap / savePly output for a run; costs one map copy
 * per frame.  Switch it on before the first frame (class ElasticFusion of libefusion.so does). */
int ef_set_reference_download(ef_ctx* ctx, int on);
int ef_save_freiburg(ef_ctx* ctx, const char* path);          /* trajectory dump of ~ElasticFusion, :112-139 */
int ef_save_ply(ef_ctx* ctx, const char* path);               /* ElasticFusion::savePly, :684-781 */
/* the same two writers on HOST arrays (no context, no GPU): byte for byte the reference's files — the trajectory with six
 * significant digits per number as its ostream prints them, the PLY with the normals negated as savePly does (:741-743) */
int ef_write_freiburg(const char* path, const double* T_wc16_array, const int64_t* timestamps, int n);
int ef_write_ply(const char* path, const float* surfels12, uint32_t count, float confidence_threshold);
/* setters (Core/ElasticFusion.h:135-183) */
int ef_set_rgb_only(ef_ctx*, int v);
int ef_set_icp_weight(ef_ctx*, float v);
int ef_set_pyramid(ef_ctx*, int v);
int ef_set_fast_odom(ef_ctx*, int v);
int ef_set_so3(ef_ctx*, int v);
int ef_set_frame_to_frame_rgb(ef_ctx*, int v);
int ef_set_confidence_threshold(ef_ctx*, float v);
int ef_set_depth_cutoff(ef_ctx*, float v);

/* named internal images, copied to HOST (synchronises); for tests and for a front-end's drawing code */
enum ef_image {
  EF_IMG_DEPTH_FILTERED = 0,      /* u16  */
  EF_IMG_DEPTH_METRIC,            /* f32  */
  EF_IMG_DEPTH_METRIC_FILTERED,   /* f32  */
  EF_IMG_PREDICT_IMAGE,           /* u8x4 IndexMap::imageTex   */
  EF_IMG_PREDICT_VERTEX,          /* f32x4 IndexMap::vertexTex */
  EF_IMG_PREDICT_NORMAL,          /* f32x4 IndexMap::normalTex */
  EF_IMG_PREDICT_TIME,            /* u16  IndexMap::timeTex    */
  EF_IMG_FILL_IMAGE,              /* u8x4 FillIn::imageTexture */
  EF_IMG_FILL_VERTEX,             /* f32x4 */
  EF_IMG_FILL_NORMAL,             /* f32x4 */
  EF_IMG_INDEX,                   /* u32  IndexMap::indexTex   */
  EF_IMG_VERT_CONF,               /* f32x4 */
  EF_IMG_COLOR_TIME,              /* f32x4 */
  EF_IMG_NORM_RAD,                /* f32x4 */
  EF_IMG_OLD_IMAGE,               /* u8x4 IndexMap::oldImageTex  (INACTIVE prediction; close_loops contexts only) */
  EF_IMG_OLD_VERTEX,              /* f32x4 IndexMap::oldVertexTex */
  EF_IMG_OLD_NORMAL,              /* f32x4 IndexMap::oldNormalTex */
  EF_IMG_OLD_TIME                 /* u16  IndexMap::oldTimeTex   */
};
int ef_get_image(ef_ctx* ctx, int which, void* host_dst, size_t bytes);
/* Resize::image / vertex / time (Core/Shaders/Resize.cpp:50-159): the predicted, fill-in or inactive-prediction image `which`
 * downsampled NEAREST by an integer factor on the device ((W/factor) x (H/factor) elements copied to the host): what the fern
 * database encodes (factor 8, Ferns.cpp:31-36,78-116) and what the constraint sampling reads (factor 20).  Synchronises. */
int ef_get_image_resized(ef_ctx* ctx, int which, int factor, void* host_dst, size_t bytes);
/* tracker pyramids (RGBDOdometry private state) for kernel-level parity tests:
 * which: 0 vmap_curr 1 nmap_curr 2 vmap_g_prev 3 nmap_g_prev 4 lastDepth 5 nextDepth 6 lastImage
 *        7 nextImage 8 lastNextImage 9 dIdx 10 dIdy 11 depth_tmp */
int ef_get_tracker_buffer(ef_ctx* ctx, int which, int level, void* host_dst, size_t bytes);

/* per-stage GPU time of the last ef_process_frame (hipEvent pairs; names follow the reference's
 * TICK/TOCK sites, Core/Utils/Stopwatch.h): fills up to max entries, returns count in *n */
typedef struct ef_timing { const char* name; float ms; } ef_timing;
int ef_enable_timing(ef_ctx* ctx, int on);
int ef_get_timings(ef_ctx* ctx, ef_timing* out, int max, int* n);

/* Sampling of the dominant kernel (the level-0 icpStep+rgbStep normal-equation kernel) every `every_n_frames`-th frame (0 = off;
 * resets the samples): the sampled launches go through hipExtLaunchKernelGGL with a start and a stop event, which receive the
 * DISPATCH's own begin / end timestamps -- the duration rocprofv3 --kernel-trace reports for the same launch, no marker packets
 * in between.  ef_get_kernel_timing synchronises and returns the average launch duration, the number of sampled launches and the
 * algorithmic bytes one launch must move (bytes_per_launch_survey: SURVEY.md 8d's narrower numerator) --
 * bench.py's roofline leg. */
typedef struct ef_kernel_time {
  const char* name; float avg_us; int launches; double bytes_per_launch; double bytes_per_launch_survey;
} ef_kernel_time;
int ef_kernel_timing(ef_ctx* ctx, int every_n_frames);
int ef_get_kernel_timing(ef_ctx* ctx, ef_kernel_time* out);
/* same sampling (switched on by ef_kernel_timing) of the IndexMap point splat: k_index_splat of the frame's first predictIndices */
int ef_get_splat_timing(ef_ctx* ctx, ef_kernel_time* out);
/* same sampling of the persistent tracker launch (k_track_fast: the whole of RGBDOdometry::getIncrementalTransformation, RGBDOdometry.cpp:259-553,
 * as one launch); launches = 0 while the launch-per-step script runs (ef_set_persistent_tracker(ctx, 0), rgbOnly, graph replay) */
int ef_get_tracker_timing(ef_ctx* ctx, ef_kernel_time* out);

/* Persistent tracker launches (the default: RGBDOdometry::getIncrementalTransformation, RGBDOdometry.cpp:259-553, as one launch of 256 co-resident
 * workgroups) that found part of the chip taken by other work and ran on one workgroup instead: same results, ~25x the tracking time, nothing for
 * the caller to do — a server that sees the count grow is sharing the GPU with something that holds CUs for milliseconds.  Synchronises. */
int ef_get_tracker_fallbacks(ef_ctx* ctx, int* count);
/* developer instrumentation for the test of that path: `workgroups` workgroups that each fill one CU spin for `microseconds` on a stream of their own */
int ef_debug_occupy(ef_ctx* ctx, int workgroups, int microseconds);
/* test hook: raises the sticky "a persistent tracker launch gave up waiting" flag of the frame tracker, as a wait that timed out after admission
 * would.  From then on every persistent launch of the context returns at once; the frame whose tracker saw the flag hands it to the host, the
 * NEXT ef_process_frame[_dev] (class ElasticFusion::processFrame: throws) returns EF_EHIP without having enqueued anything, and so does
 * ef_synchronize.  Results since the flag was raised are invalid. */
int ef_debug_inject_tracker_abort(ef_ctx* ctx);

/* developer instrumentation: the 16 wall_clock64() (100 MHz) stamps the last tracking solve left in the device state;
 * all zero unless the library was built with -DEF_STAGE_CLOCKS (EF_HIPCC_FLAGS=-DEF_STAGE_CLOCKS python -m elasticfusion_amd.build) */
int ef_debug_clocks(ef_ctx* ctx, unsigned long long* out16);
/* developer instrumentation of the persistent tracker launch (zeros unless the library was built with -DEF_STAGE_CLOCKS): 32 sums
 * of 10 ns ticks per phase since the last call; tools/fast_clocks.py (reference-order builds: tools/small_clocks.py) names them */
int ef_debug_small_clocks(ef_ctx* ctx, unsigned long long* out32);

/* ---- device memory helpers (so that a non-HIP host can drive the operator tier) ---- */
int ef_dev_alloc(void** dev, size_t bytes);
int ef_dev_free(void* dev);
int ef_dev_upload(void* dev, const void* host, size_t bytes);
int ef_dev_download(void* host, const void* dev, size_t bytes);
int ef_dev_memset(void* dev, int value, size_t bytes);
int ef_dev_sync(void);
/* box calibration for benchmarks (GPU boxes of one pool differ by 10-20 %): average time per launch of 200 back-to-back launches of an EMPTY
 * kernel and of a kernel that copies 16 MiB with 16-byte accesses (16 MiB read + 16 MiB written), on `stream` (a hipStream_t, 0 = null stream) */
int ef_dev_calibrate(void* stream, float* empty_us, float* stream16mb_us);
int ef_device_count(int* n);
int ef_set_device(int device);

/* ---- operator tier: tracking (Core/Cuda/cudafuncs.cuh:61-169). All pointers are DEVICE pointers,
 * work runs on `stream` (hipStream_t, NULL = default stream) and is synchronous only where the
 * reference returns host results (icp/rgb/so3 steps, rgb residual). ---- */
typedef struct ef_intr { float fx, fy, cx, cy; } ef_intr;   /* CameraModel, types.cuh:88-96 */

int ef_op_pyr_down(const uint16_t* src, int src_cols, int src_rows, uint16_t* dst, void* stream);             /* pyrDown */
int ef_op_create_vmap(const ef_intr* intr, const uint16_t* depth, int cols, int rows, float depth_cutoff,
                      float* vmap, void* stream);                                                              /* createVMap */
int ef_op_create_nmap(const float* vmap, int cols, int rows, float* nmap, void* stream);                       /* createNMap */
int ef_op_transform_maps(const float* vmap_src, const float* nmap_src, int cols, int rows, const float* R9,
                         const float* t3, float* vmap_dst, float* nmap_dst, void* stream);                      /* tranformMaps */
int ef_op_copy_maps(const float* vmap_src_f4, const float* nmap_src_f4, int cols, int rows, float* vmaps_tmp,
                    float* vmap_dst, float* nmap_dst, void* stream);                                            /* copyMaps */
int ef_op_resize_vmap(const float* in, int src_cols, int src_rows, float* out, void* stream);                  /* resizeVMap */
int ef_op_resize_nmap(const float* in, int src_cols, int src_rows, float* out, void* stream);                  /* resizeNMap */
int ef_op_pyr_down_gauss_f(const float* src, int src_cols, int src_rows, float* dst, void* stream);            /* pyrDownGaussF */
int ef_op_pyr_down_uchar_gauss(const uint8_t* src, int src_cols, int src_rows, uint8_t* dst, void* stream);    /* pyrDownUcharGauss */
int ef_op_vertices_to_depth(const float* vmaps_tmp, int cols, int rows, float cutoff, float* dst, void* stream); /* verticesToDepth */
int ef_op_image_bgr_to_intensity(const uint8_t* rgba, int cols, int rows, uint8_t* dst, void* stream);         /* imageBGRToIntensity */
int ef_op_compute_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy, void* stream); /* computeDerivativeImages */
int ef_op_project_to_point_cloud(const float* depth, int cols, int rows, const ef_intr* intr_level0, int level,
                                 float* cloud_f3, void* stream);                                                /* projectToPointCloud */
/* icpStep: host outputs A[36] row-major symmetric, b[6], residual[2] = {sum r^2, inliers} */
int ef_op_icp_step(const float* Rcurr9, const float* tcurr3, const float* vmap_curr, const float* nmap_curr,
                   const float* Rprev_inv9, const float* tprev3, const ef_intr* intr, const float* vmap_g_prev,
                   const float* nmap_g_prev, float dist_thres, float angle_thres, int cols, int rows,
                   float* A_host36, float* b_host6, float* residual_host2, void* stream);
/* computeRgbResidual: corres_img is W*H 16-byte DataTerm records (types.cuh:81-86) */
int ef_op_compute_rgb_residual(float min_scale, const int16_t* dIdx, const int16_t* dIdy, const float* last_depth,
                               const float* next_depth, const uint8_t* last_image, const uint8_t* next_image,
                               void* corres_img, float max_depth_delta, const float* kt3, const float* krkinv9,
                               int cols, int rows, int* sigma_sum_host, int* count_host, void* stream);
int ef_op_rgb_step(const void* corres_img, float sigma, const float* cloud_f3, float fx, float fy,
                   const int16_t* dIdx, const int16_t* dIdy, float sobel_scale, int cols, int rows,
                   float* A_host36, float* b_host6, void* stream);
int ef_op_so3_step(const uint8_t* last_image, const uint8_t* next_image, const float* image_basis9,
                   const float* kinv9, const float* krlr9, int cols, int rows, float* A_host9, float* b_host3,
                   float* residual_host2, void* stream);

/* The tracking driver's small linear algebra (Eigen / Sophus arithmetic of RGBDOdometry.cpp:356,526-534,566-570,
 * OdometryProvider.h:34-96, ElasticFusion.cpp:371-374) as the DEVICE evaluates it, host vectors in and out
 * (<= 64 doubles each) -- for parity tests against the oracle's efo_ldlt6 / efo_polar3 / ... */
enum ef_linalg_op {
  EF_LINALG_LDLT6 = 0,        /* in A[36] b[6]              -> x[6]   Eigen::LDLT solve, double            */
  EF_LINALG_LDLT3F,           /* in A[9] b[3] (as doubles)  -> x[3]   Eigen::LDLT solve, float             */
  EF_LINALG_POLAR3,           /* in A[9]                    -> R[9]   JacobiSVD U V^T                      */
  EF_LINALG_RODRIGUES,        /* in v[3]                    -> R[9]   OdometryProvider::rodrigues          */
  EF_LINALG_SE3_INVERSE,      /* in T[16]                   -> T^-1[16] Sophus::SE3d::inverse().matrix()    */
  EF_LINALG_SE3_LOG_NORM,     /* in T[16]                   -> |log(T)| Sophus::SE3d::log().norm()          */
  EF_LINALG_SCALAR,           /* in a, b -> sqrt(a), a/b, sin(a), cos(a), atan2(a,b) (fp64 device math)     */
  EF_LINALG_LDLT6_WAVE        /* as EF_LINALG_LDLT6, through the one-element-per-lane wavefront version      */
};
int ef_op_linalg(int which, const double* in_host, int n_in, double* out_host, int n_out);

/* ---- operator tier: pre-processing and surfel map (the reference's GLSL passes) ---- */
typedef struct ef_cam { int cols, rows; float fx, fy, cx, cy; } ef_cam;

int ef_op_filter_depth(const uint16_t* raw, int cols, int rows, float max_d, uint16_t* filtered, void* stream);   /* depth_bilateral.frag */
int ef_op_metricise_depth(const uint16_t* in, int cols, int rows, float max_d, float* out, void* stream);         /* depth_metric.frag */
/* vertex_feedback x2 + init_unstable: returns the number of seeded surfels in *count_host */
int ef_op_seed_map(const ef_cam* cam, const uint8_t* rgb, const float* depth_metric, const float* depth_metric_filtered,
                   int time, float max_depth, float* surfels_aos, uint32_t* count_host, void* stream);
/* IndexMap::predictIndices: surfels_aos = count x 12 floats */
int ef_op_predict_indices(const ef_cam* cam, const double* T_wc16, int time, const float* surfels_aos, uint32_t count,
                          float max_depth, int time_delta, uint32_t* index_map, float* vert_conf, float* color_time,
                          float* norm_rad, void* stream);
/* IndexMap::combinedPredict (ACTIVE) */
int ef_op_combined_predict(const ef_cam* cam, const double* T_wc16, const float* surfels_aos, uint32_t count,
                           float max_depth, float conf_threshold, int time, int max_time, int time_delta,
                           uint8_t* image_rgba, float* vertex, float* normal, uint16_t* time_map, void* stream);
/* IndexMap::synthesizeDepth (splat.vert + depth_splat.frag, IndexMap.cpp:395-476): float depth, 0 = nothing drawn */
int ef_op_synthesize_depth(const ef_cam* cam, const double* T_wc16, const float* surfels_aos, uint32_t count,
                           float max_depth, float conf_threshold, int time, int max_time, int time_delta,
                           float* depth, void* stream);
/* FillIn::{vertex,normal,image} */
int ef_op_fill_in(const ef_cam* cam, const uint8_t* image_rgba, const float* vertex, const float* normal,
                  const uint16_t* depth_filtered, const uint8_t* rgb, int passthrough, int passthrough_image,
                  uint8_t* fill_image, float* fill_vertex, float* fill_normal, void* stream);
/* Resize::image + denseEnough -> *dense_host in {0,1} */
int ef_op_dense_enough(const ef_cam* cam, const uint8_t* image_rgba, int* dense_host, void* stream);
/* GlobalModel::fuse: surfels updated in place; new_unstable gets the tagged candidates in draw order */
int ef_op_fuse(const ef_cam* cam, const double* T_wc16, int time, const uint8_t* rgb, const float* depth_metric,
               const float* depth_metric_filtered, const uint32_t* index_map, const float* vert_conf,
               const float* color_time, const float* norm_rad, float max_depth, float weighting, float* surfels_aos,
               uint32_t count, float* new_unstable_aos, uint32_t* new_count_host, void* stream);
/* GlobalModel::clean (no deformation graph) */
int ef_op_clean(const ef_cam* cam, const double* T_wc16, int time, const uint32_t* index_map, const float* vert_conf,
                const float* color_time, const float* norm_rad, float conf_threshold, int time_delta, float max_depth,
                const float* surfels_aos, uint32_t count, const float* new_unstable_aos, uint32_t new_count,
                float* surfels_out_aos, uint32_t* out_count_host, void* stream);
/* GlobalModel::clean with the deformation graph applied to every kept surfel (copy_unstable.vert:128-322):
 * graph = nodes x 16 floats sorted by time {position 3, rotation 9 column-major, translation 3, time} (device pointer), the
 * content of the reference's node texture (GlobalModel.cpp:540-546); depth = ef_op_synthesize_depth image (device, read
 * unless is_fern).  nodes == 0 is ef_op_clean. */
int ef_op_clean_deform(const ef_cam* cam, const double* T_wc16, int time, const uint32_t* index_map, const float* vert_conf,
                       const float* color_time, const float* norm_rad, float conf_threshold, int time_delta, float max_depth,
                       const float* surfels_aos, uint32_t count, const float* new_unstable_aos, uint32_t new_count,
                       const float* graph, int nodes, const float* depth, int is_fern, float* surfels_out_aos,
                       uint32_t* out_count_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EF_HIP_H_ */
