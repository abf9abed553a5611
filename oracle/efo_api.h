/* TEST INFRASTRUCTURE — CPU oracle C API (host pointers only). See oracle/README.md.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. */
#ifndef EFO_API_H_
#define EFO_API_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- tracking operators (Core/Cuda/cudafuncs.cu, reduce.cu) ---- */
void efo_pyr_down_u16(const uint16_t* src, int scols, int srows, uint16_t* dst);
void efo_create_vmap(const uint16_t* depth, int cols, int rows, float fx, float fy, float cx, float cy,
                     float depthCutoff, float* vmap);
void efo_create_nmap(const float* vmap, int cols, int rows, float* nmap);
void efo_transform_maps(float* vmap, float* nmap, int cols, int rows, const float* R9, const float* t3);
void efo_copy_maps(const float* vtex, const float* ntex, int cols, int rows, float* vmaps_tmp, float* vmap, float* nmap);
void efo_resize_map(const float* in, int scols, int srows, float* out, int normalize);
void efo_pyr_down_gauss_f(const float* src, int scols, int srows, float* dst);
void efo_pyr_down_uchar_gauss(const uint8_t* src, int scols, int srows, uint8_t* dst);
void efo_vertices_to_depth(const float* vmaps_tmp, int cols, int rows, float cutOff, float* dst);
void efo_bgr_to_intensity(const uint8_t* rgba, int cols, int rows, uint8_t* dst);
void efo_derivative_images(const uint8_t* src, int cols, int rows, int16_t* dx, int16_t* dy);
void efo_project_to_point_cloud(const float* depth, int cols, int rows, float fx, float fy, float cx, float cy, float* cloud);
void efo_icp_step(const float* Rcurr, const float* tcurr, const float* vmap_curr, const float* nmap_curr,
                  const float* Rprev_inv, const float* tprev, float fx, float fy, float cx, float cy,
                  const float* vmap_g_prev, const float* nmap_g_prev, float distThres, float angleThres, int cols,
                  int rows, float* A36, float* b6, float* residual2);
void efo_rgb_residual(float minScale, const int16_t* dIdx, const int16_t* dIdy, const float* lastDepth,
                      const float* nextDepth, const uint8_t* lastImage, const uint8_t* nextImage, void* corres_out,
                      float maxDepthDelta, const float* kt3, const float* krkinv9, int cols, int rows, int* sigmaSum,
                      int* count);
void efo_rgb_step(const void* corres_in, float sigma, const float* cloud, float fx, float fy, const int16_t* dIdx,
                  const int16_t* dIdy, float sobelScale, int cols, int rows, float* A36, float* b6);
void efo_so3_step(const uint8_t* lastImage, const uint8_t* nextImage, const float* imageBasis9, const float* kinv9,
                  const float* krlr9, int cols, int rows, float* A9, float* b3, float* residual2);

/* ---- tracking driver (Core/Utils/RGBDOdometry.cpp) ---- */
typedef struct efo_odometry efo_odometry;
efo_odometry* efo_odom_create(int w, int h, float cx, float cy, float fx, float fy);
void efo_odom_destroy(efo_odometry*);
void efo_odom_init_icp(efo_odometry*, const uint16_t* filteredDepth, float depthCutoff);
void efo_odom_init_icp_model(efo_odometry*, const float* vtex, const float* ntex, const double* T_wc16);
/* initICP(predictedVertices, predictedNormals), RGBDOdometry.cpp:149-169 */
void efo_odom_init_icp_maps(efo_odometry*, const float* vtex, const float* ntex);
void efo_odom_init_rgb_model(efo_odometry*, const uint8_t* rgba);
void efo_odom_init_rgb(efo_odometry*, const uint8_t* rgba);
void efo_odom_init_first_rgb(efo_odometry*, const uint8_t* rgba);
void efo_odom_track(efo_odometry*, double* T_wc16, int rgbOnly, float icpWeight, int pyramid, int fastOdom, int so3);
void efo_odom_stats(const efo_odometry*, float* out6, double* lastA36, double* lastb6);
const void* efo_odom_buffer(const efo_odometry*, int which, int level);

/* ---- small linear algebra (Eigen / Sophus restatement) for known-answer tests ---- */
void efo_ldlt6(const double* A36, const double* b6, double* x6);
void efo_ldlt3f(const float* A9, const float* b3, float* x3);
void efo_polar3(const double* A9, double* R9);
void efo_rodrigues(const double* v3, double* R9);
void efo_se3_inverse(const double* T16, double* out16);
double efo_se3_log_norm(const double* T16, double* out6);
float efo_expf_spec(float x);
void efo_covariance(const double* lastA36, double* cov36);   /* getCovariance: PartialPivLU inverse of lastA */

/* ---- pre-processing + surfel map (Core/Shaders GLSL passes, IndexMap.cpp, GlobalModel.cpp) ---- */
void efo_filter_depth(const uint16_t* raw, int cols, int rows, float maxD, uint16_t* filtered);
void efo_metricise_depth(const uint16_t* in, int cols, int rows, float maxD, float* out);

typedef struct efo_cam { int cols, rows; float fx, fy, cx, cy; } efo_cam;

/* surfels are the reference's 3 x vec4 = 12 floats: {x,y,z,conf} {colour,0,initTime,lastTime} {nx,ny,nz,radius} */
int efo_seed_map(const efo_cam* cam, const uint8_t* rgb, const float* depthMetric, const float* depthMetricFiltered,
                 int time, float maxDepth, float* surfels_out /* cols*rows*12 */);
void efo_predict_indices(const efo_cam* cam, const double* T_wc16, int time, const float* surfels, int count,
                         float maxDepth, int timeDelta, uint32_t* indexMap, float* vertConf, float* colorTime,
                         float* normRad);
void efo_combined_predict(const efo_cam* cam, const double* T_wc16, const float* surfels, int count, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, uint8_t* image_rgba,
                          float* vertex, float* normal, uint16_t* timeMap);
/* IndexMap::synthesizeDepth (G6): float depth image, 0 = nothing drawn */
void efo_synthesize_depth(const efo_cam* cam, const double* T_wc16, const float* surfels, int count, float maxDepth,
                          float confThreshold, int time, int maxTime, int timeDelta, float* depth);
void efo_fill_in(const efo_cam* cam, const uint8_t* image_rgba, const float* vertex, const float* normal,
                 const uint16_t* depthFiltered, const uint8_t* rgb, int passthrough, int passthroughImage,
                 uint8_t* fill_image_rgba, float* fill_vertex, float* fill_normal);
int efo_dense_enough(const efo_cam* cam, const uint8_t* image_rgba);
/* fuse data pass + update pass; surfels updated in place; new unstable candidates (tags -1/-2 in colour.w)
 * written to newUnstable (<= cols*rows*12 floats) in draw (column-major) order; returns their number */
int efo_fuse(const efo_cam* cam, const double* T_wc16, int time, const uint8_t* rgb, const float* depthMetric,
             const float* depthMetricFiltered, const uint32_t* indexMap, const float* vertConf,
             const float* colorTime, const float* normRad, float maxDepth, float weighting, float* surfels,
             int count, float* newUnstable);
/* clean: old surfels then newUnstable through copy_unstable.{vert,geom}; returns new count */
int efo_clean(const efo_cam* cam, const double* T_wc16, int time, const uint32_t* indexMap, const float* vertConf,
              const float* colorTime, const float* normRad, float confThreshold, int timeDelta, float maxDepth,
              const float* surfels, int count, const float* newUnstable, int newCount, float* surfels_out);

/* clean with the deformation graph applied (copy_unstable.vert:128-322): graph = nodes x 16 floats sorted by time
 * {position 3, rotation 9 column-major, translation 3, time}; depth = synthesizeDepth image; nodes == 0 => efo_clean */
int efo_clean_deform(const efo_cam* cam, const double* T_wc16, int time, const uint32_t* indexMap, const float* vertConf,
                     const float* colorTime, const float* normRad, float confThreshold, int timeDelta, float maxDepth,
                     const float* surfels, int count, const float* newUnstable, int newCount, const float* graph, int nodes,
                     const float* depth, int isFern, float* surfels_out);

/* ---- whole-frame orchestration (Core/ElasticFusion.cpp:270-653, open loop) ---- */
typedef struct efo_fusion efo_fusion;
typedef struct efo_fusion_params {
  int width, height;
  float fx, fy, cx, cy;
  int timeDelta;
  float confidence, depthCut, icpWeight;
  int fastOdom, so3, frameToFrameRGB, pyramid, rgbOnly;
  int maxSurfels;
} efo_fusion_params;
void efo_fusion_default_params(efo_fusion_params*);
efo_fusion* efo_fusion_create(const efo_fusion_params*);
void efo_fusion_destroy(efo_fusion*);
void efo_fusion_process_frame(efo_fusion*, const uint8_t* rgb, const uint16_t* depth, int64_t timestamp,
                              float weightMultiplier, const double* in_T_wc16_or_null);
void efo_fusion_get_pose(const efo_fusion*, double* T_wc16);
int efo_fusion_map_count(const efo_fusion*);
void efo_fusion_map_download(const efo_fusion*, float* surfels /* count*12 */);
void efo_fusion_map_download_reference(const efo_fusion*, float* surfels /* count*12: the buffer GlobalModel::downloadMap reads (Q14) */);
int efo_fusion_tick(const efo_fusion*);
/* host threads used by the parallel loops of the restatement (bilateral rows, reduction blocks); results do not depend on it */
void efo_set_threads(int n);
/* deformation graph (nodes x 16, sorted by time) applied by the next frame's clean, as after a loop closure */
void efo_fusion_set_deformation(efo_fusion*, const float* graph, int nodes, int isFern);
void efo_fusion_stats(const efo_fusion*, float* out6);
/* trace of the frame loop: one line per step with its parameters (tests/test_oracle_vs_reference_frame.py) */
void efo_fusion_trace(efo_fusion*, int on);
const char* efo_fusion_take_trace(efo_fusion*);
/* ---- local loop closure, front half (ElasticFusion.cpp:447-511): INACTIVE prediction, model-to-model odometry, covariance and
 * error gates, surface constraints sampled every consSample = 20 pixels.  The deformation-graph optimisation on their far side
 * (Deformation::constrain) is the caller's: a solver callback receives the constraints and may return a graph, which is then
 * applied by this frame's clean pass, with T_wc := T_wc_est, exactly as :514-527 / :558-585 do. */
typedef struct efo_local_loop {
  int attempted, cov_ok, gates_ok, n_constraints, applied, graph_nodes;
  float stats[6];        /* modelToModel: lastICPError, lastICPCount, lastRGBError, lastRGBCount, lastSO3Error, lastSO3Count */
  double cov_diag[6];
  double T_wc_curr[16], T_wc_est[16];
} efo_local_loop;
/* constraints: n rows of 8 doubles {vert_w_curr xyz, vert_w_est xyz, time of the inactive surface, pin}; return non-zero to accept */
typedef int (*efo_loop_solver)(void* user, const efo_local_loop* info, const double* constraints, int n, float* graph_out /* 1024 x 16 */,
                               int* nodes_out);
/* the constraint sampling + arithmetic of ElasticFusion.cpp:485-509 on explicit inputs (what efo_fusion's local loop closure runs) */
int efo_loop_constraints(const float* vertex4, const uint16_t* oldTime, int width, int height, int consSample, const double* T_wc_curr16,
                         const double* T_wc_est16, float maxDepth, int pin, double* rows8);
void efo_fusion_set_close_loops(efo_fusion*, int on, int icpCountThresh, float icpErrThresh, float covThresh);
void efo_fusion_set_loop_solver(efo_fusion*, efo_loop_solver fn, void* user);
int efo_fusion_local_loop(const efo_fusion*, efo_local_loop* info, double* constraints, int max_constraints);
/* T_cw = float(T_wc^-1) and pose = T_wc cast to float, row-major 4x4: the matrices the map passes hand to their shaders */
void efo_pose_matrices(const double* T_wc16, float* T_cw16, float* pose16);
/* Resize::{image,vertex,time} (Resize.cpp:50-159): NEAREST downsample by an integer factor, any element size */
void efo_resize_nearest(const void* src, int cols, int rows, int elemBytes, int factor, void* dst);
/* Deformation::sampleGraphModel: every 5000th surfel -> {x, y, z, initTime}; returns the node count */
int efo_sample_graph(const float* surfels, int count, float* out4);
/* which: 0 image 1 vertex 2 normal 3 time of the INACTIVE prediction */
const void* efo_fusion_old_buffer(const efo_fusion*, int which);
/* which: 0 image_rgba(u8x4) 1 vertex(f4) 2 normal(f4) 3 time(u16) 4 fill_image 5 fill_vertex 6 fill_normal
 * 7 indexMap(u32) 8 vertConf 9 colorTime 10 normRad 11 depthFiltered(u16) 12 depthMetric 13 depthMetricFiltered */
const void* efo_fusion_buffer(const efo_fusion*, int which);
efo_odometry* efo_fusion_odometry(efo_fusion*);
/* tracking-only timing hook for bench.py cpu_baseline: runs initICP/initRGB/track on the current state */

/* ---- fern keyframe database (Core/Ferns.cpp), efo_ferns.cpp; signatures as include/ef_hip.h ef_ferns_* ---- */
typedef struct efo_ferns efo_ferns;
typedef void (*efo_fern_tracker)(void* user, const float* fern_verts4, const float* fern_norms4, const double* T_wc_fern16, const float* cur_verts4,
                                 const float* cur_norms4, double* T_inout16, float* icp_error, float* icp_count);
efo_ferns* efo_ferns_create(int num, int max_depth_mm, float photo_thresh, int width, int height, float fx, float fy, float cx, float cy, unsigned seed);
void efo_ferns_destroy(efo_ferns*);
int efo_ferns_get_table(const efo_ferns*, int* table6);
int efo_ferns_set_table(efo_ferns*, const int* table6);
int efo_ferns_add_frame(efo_ferns*, const uint8_t* rgb, int rgb_channels, const float* verts4, const float* norms4, const double* T_wc16, int src_time,
                        float threshold);
int efo_ferns_find_frame(efo_ferns*, const uint8_t* rgb, int rgb_channels, const float* verts4, const float* norms4, const double* T_wc16, int time, int lost,
                         efo_fern_tracker tracker, void* user, double* T_est16_out, double* constraints6_out, int max_constraints, int* n_constraints_out);
int efo_ferns_count(const efo_ferns*);
int efo_ferns_last_closest(const efo_ferns*);
int efo_ferns_get_frame(const efo_ferns*, int id, uint8_t* codes, int* good_codes, int* src_time, double* T_wc16, uint8_t* rgb3, float* verts4, float* norms4);
int efo_ferns_set_frame_pose(efo_ferns*, int id, const double* T_wc16);
float efo_ferns_block_hd_aware(const efo_ferns*, int a, int b);
float efo_ferns_photometric_check(const efo_ferns*, const uint8_t* rgb, int rgb_channels, const float* verts4, const double* T_est16, int id);

/* ---- global loop closure (ElasticFusion.cpp:392-445,609-618) in the frame loop: efo_fusion_enable_ferns gives the instance its fern
 * database (Ferns(num, depthCut * 1000, photoThresh), seed instead of time(0)) and 1/8-resolution tracker; a general solver callback
 * stands where Deformation::constrain does, for the global AND (when registered) the local closure.
 * rows10: {src xyz, target xyz, srcTime, targetTime, relative, pin}; poses16 in/out: keyframe poses, then (fernMatch) the trajectory;
 * new_relative_rows10 / n_new_relative: newRelativeCons of a local closure (NULL for a global one).  Non-zero return = accepted. */
typedef int (*efo_deform_solver)(void* user, int fernMatch, const double* rows10, int n_rows, double* poses16_inout, const int64_t* pose_times, int n_poses,
                                 float* graph_out /* 1024 x 16 */, int* nodes_out, double* new_relative_rows10, int* n_new_relative);
typedef struct efo_global_loop {
  int attempted, closest, n_constraints, accepted, graph_nodes;
  float icp_error, icp_count;          /* of the fern tracker, when it ran */
  double T_wc_recovery[16];
} efo_global_loop;
void efo_fusion_get_pose_qt(const efo_fusion*, double* q4_t3);
void efo_fusion_set_tick(efo_fusion*, int tick);
/* test hook: the oracle's side of ef_map_upload + ef_restore_state (map, tick, pose as held: quaternion xyzw + translation, the frame processed last) */
void efo_fusion_restore(efo_fusion*, const float* surfels12, int count, int tick, const double* q4_t3, const uint8_t* rgb_prev, const uint16_t* depth_prev);
/* relocalisation (the reference constructor's `reloc`, ElasticFusion.cpp:326-366,411-413,536,601-604,624-649) */
void efo_fusion_set_reloc(efo_fusion*, int on);
void efo_fusion_reloc_state(const efo_fusion*, int* out4 /* lost, trackingOk, trackingCount, lastFrameRecovery */);
void efo_fusion_enable_ferns(efo_fusion*, int num, float photoThresh, float fernThresh, unsigned seed);
efo_ferns* efo_fusion_ferns(efo_fusion*);
/* the 1/8-resolution fill-in views of the last frame: which = 0 what Ferns::findFrame saw mid-frame, 1 what Ferns::addFrame saw at its end */
int efo_fusion_fern_view(const efo_fusion*, int which, uint8_t* rgba, float* verts4, float* norms4);
void efo_fusion_set_deform_solver(efo_fusion*, efo_deform_solver fn, void* user);
void efo_fusion_global_loop(const efo_fusion*, efo_global_loop* info);
int efo_fusion_relative_constraints(const efo_fusion*, double* rows10, int max_rows);
int efo_fusion_trajectory(const efo_fusion*, double* poses16, int max_poses);

#ifdef __cplusplus
}
#endif
#endif
