/* =====================================================================================
 * lama_host.h -- C surface of liblama_host.so: the host-side mirror of the reference interface
 * (lama::PFSlam2D of include/lama/pf_slam2d.h) flattened for FFI users (Python ctypes in this repo:
 * tests, bench.py, iris_lama_amd/distributed.py), plus the synthetic workload generator.
 *
 * C++ consumers (iris_lama_ros) use the classes in include/lama/ directly; this header exists because the
 * measuring/test harness is Python.  All functions catch C++ exceptions; a negative return is an error and
 * lama_pf_last_error() holds the message.
 * ===================================================================================== */
#ifndef LAMA_HOST_H
#define LAMA_HOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Seeded synthetic corridor log.  pts: (steps+1) x beams x 3 doubles (sensor frame),
 * odom_xyr / truth_xyr: (steps+1) x 3 doubles (x, y, yaw).  truth_xyr may be NULL.  Returns 0. */
int lama_corridor_generate(int steps, int beams, double* pts, double* odom_xyr, double* truth_xyr);

/* PFSlam2D::Options (include/lama/pf_slam2d.h) as a POD. */
typedef struct lama_pf_options {
    uint32_t particles;
    double srr, str, stt, srt;
    double meas_sigma, meas_sigma_gain;
    double trans_thresh, rot_thresh;
    double l2_max, truncated_ray, truncated_range, resolution;
    uint32_t patch_size, max_iter;
    uint32_t seed;
    int32_t create_summary;
    int32_t gpu_device;
    uint32_t shard_rank, shard_world;
    int32_t profile;
    uint32_t brushfire_mode;
    uint32_t window_patches, dm_patch_capacity, occ_patch_capacity, queue_capacity;   /* device map storage, 0 = defaults */
    int32_t gpus;                 /* PFSlam2D::Options::gpus: > 1 = one object drives that many shards / devices from C++ (0 / 1: one) */
} lama_pf_options;

typedef struct lama_pf lama_pf;

void lama_pf_default_options(lama_pf_options* o);

#ifdef LAMA_TESTING
/* ONLY in a host library compiled with -DLAMA_TESTING (the test-suite builds its own copy, tests/cpu_engine/Makefile); the
 * shipped liblama_host.so neither declares nor exports it (tests/test_cabi.py asserts that) and always binds liblama_hip.so
 * next to itself.  Device library used by the objects created afterwards (NULL = liblama_hip.so): the test-suite points it at
 * an oracle-backed test double to exercise the host / multi-rank logic on machines without a GPU. */
int lama_host_set_engine_library(const char* path);
#endif
/* Path of the device library bound to `pf` (so callers can assert that the HIP library is the one in use). */
const char* lama_pf_engine_origin(const lama_pf* pf);

/* err (optional, errcap bytes) receives the failure message when NULL is returned. */
lama_pf* lama_pf_create(const lama_pf_options* o, char* err, int errcap);
void lama_pf_destroy(lama_pf* pf);
const char* lama_pf_last_error(const lama_pf* pf);

void lama_pf_set_prior(lama_pf* pf, double x, double y, double yaw);

/* PFSlam2D::update: 1 = update done, 0 = motion gate closed, <0 = error. */
int lama_pf_update(lama_pf* pf, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz,
                   const double* odom_xyr, double timestamp);

/* step-wise API (sharded operation), see include/lama/pf_slam2d.h */
int lama_pf_update_begin(lama_pf* pf, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz,
                         const double* odom_xyr, double timestamp);   /* 0 no update, 1 first scan, 2 matched */
int lama_pf_local_range(const lama_pf* pf, uint32_t* lo, uint32_t* hi);
int lama_pf_local_loglik(const lama_pf* pf, double* out /* hi-lo */);
int lama_pf_plan_resample(lama_pf* pf, const double* all_loglik /* P */, int32_t* sample_idx_out /* P */); /* 1 = resample */
int lama_pf_apply_resample(lama_pf* pf, const int32_t* sample_idx /* P */);
int lama_pf_update_maps(lama_pf* pf);
void* lama_pf_device_context(const lama_pf* pf);   /* lama_hip_ctx* of the local shard */

/* state queries; poses are {c,s,tx,ty} for all P particles (valid for locally owned ones) */
int lama_pf_get_poses(const lama_pf* pf, double* poses_P4);
int lama_pf_set_pose(lama_pf* pf, uint32_t i, const double* pose4);
int lama_pf_get_weights(const lama_pf* pf, double* weight, double* normalized_weight, double* weight_sum);
int lama_pf_set_weights(lama_pf* pf, const double* weight, const double* weight_sum);
double lama_pf_neff(const lama_pf* pf);
int lama_pf_best(const lama_pf* pf);
int lama_pf_best_pose_xyr(const lama_pf* pf, double* xyr);
uint32_t lama_pf_num_resamples(const lama_pf* pf);
/* Options::gpus > 1: seconds the last update spent gathering log-likelihoods / shipping (export + peer copy) / importing particles,
 * particles and bytes shipped between shards by it.  out8 = {gather_s, ship_s, import_s, particles, bytes, local_copies_s (resample copies
 * inside a shard), phase_begin_s (motion + scan match on all shards), phase_maps_s (map-update launch + state mirror)}.  Returns the number of shards. */
int lama_pf_exchange_times(const lama_pf* pf, double* out8);
/* lama_hip_ctx* of shard r of a gpus > 1 object (NULL when out of range) */
void* lama_pf_shard_context(const lama_pf* pf, uint32_t r);
uint64_t lama_pf_memory_usage(const lama_pf* pf);
/* Summary::report() into buf; returns the length needed */
int lama_pf_summary(const lama_pf* pf, char* buf, int cap);
/* Summary buckets of the last `update` (seconds): total, solving, normalizing, resampling, mapping */
int lama_pf_last_times(const lama_pf* pf, double* out5);

/* host-logic hooks (RNG replay tests): same formulas as src/pf_slam2d.cpp:365-391, 511-556 */
int lama_pf_draw_from_motion(lama_pf* pf, const double* delta4, double* pose4_inout);
double lama_pf_normalize(lama_pf* pf);
int lama_pf_resample_indices(const lama_pf* pf, double u01, int32_t* out_P);
/* pose algebra: out = a^-1 * b (Pose2D::operator-), {c,s,tx,ty} */
void lama_pose_minus(const double* a4, const double* b4, double* out4);
void lama_pose_from_xyr(double x, double y, double yaw, double* out4);

/* ---- lama::Slam2D (include/lama/slam2d.h), flattened ---- */
typedef struct lama_slam lama_slam;
typedef struct lama_slam_options {
    double trans_thresh, rot_thresh, l2_max, truncated_ray, truncated_range, resolution;
    uint32_t patch_size, max_iter;
    int32_t gpu_device;
    int32_t transient_map;      /* Slam2D::Options::transient_map (src/slam2d.cpp:322-379) */
    int32_t lm;                 /* Slam2D::Options::strategy == "lm" (Levenberg-Marquardt instead of Gauss-Newton) */
} lama_slam_options;
void lama_slam_default_options(lama_slam_options* o);
lama_slam* lama_slam_create(const lama_slam_options* o, char* err, int errcap);
void lama_slam_destroy(lama_slam* s);
const char* lama_slam_last_error(const lama_slam* s);
void lama_slam_set_pose(lama_slam* s, double x, double y, double yaw);
int lama_slam_get_pose(const lama_slam* s, double* pose4);
/* Slam2D::update: 1 = update done, 0 = not enough motion, <0 = error */
int lama_slam_update(lama_slam* s, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz,
                     const double* odom_xyr, double timestamp);
int lama_slam_enough_motion(lama_slam* s, const double* odom_xyr);
uint32_t lama_slam_processed_cells(const lama_slam* s);
uint32_t lama_slam_iterations(const lama_slam* s);
uint32_t lama_slam_deleted_patches(const lama_slam* s);
void* lama_slam_device_context(const lama_slam* s);
/* Queries on the snapshots returned by Slam2D::getOccupancyMap() / getDistanceMap() (include/lama/sdm_maps.h: the const
 * query API of the reference's map classes).  which: 0 = occupancy, 1 = distance.  All return < 0 when no map is available. */
int lama_slam_view_bounds(lama_slam* s, int which, uint32_t* min3, uint32_t* max3, double* wmin3, double* wmax3);
int64_t lama_slam_view_cells(lama_slam* s, int which, uint32_t* xy_out, uint64_t cap);      /* visit_all_cells; returns the count */
int lama_slam_view_occupancy(lama_slam* s, uint64_t n, const uint32_t* xy, uint8_t* is_free, uint8_t* is_occupied,
                             uint8_t* is_unknown, double* probability);
int lama_slam_view_distance_cells(lama_slam* s, uint64_t n, const uint32_t* xy, double* distance);
int lama_slam_view_distance_points(lama_slam* s, uint64_t n, const double* xy, double* dist_gx_gy);   /* n x 3 out */
/* lama::MatchSurface2D / lama::Solve (include/lama/match_surface_2d.h, include/lama/nlls/solver.h) on the distance map
 * Slam2D::getDistanceMap() returns -- the scan-matching problem a caller of the reference builds by hand:
 *   match_eval : MatchSurface2D::eval at pose4 {c,s,tx,ty}: residuals[n], jacobian n x 3 column-major (may be NULL), rmse_out =
 *                MatchSurface2D::error() (may be NULL);
 *   match_solve: Solve(options, problem, &cov): strategy "gn" | "lm", weight "cauchy" | "unit" | "tukey" | "huber" with its
 *                parameter; pose4 in/out, cov9 row-major (may be NULL).  Returns 0, or -3 when the configuration has no device
 *                kernel (message through lama_slam_last_error). */
int lama_slam_match_eval(lama_slam* s, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz,
                         const double* pose4, double* residuals, double* jacobian, double* rmse_out);
int lama_slam_match_solve(lama_slam* s, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz,
                          double* pose4, const char* strategy, const char* weight, double weight_param, uint32_t max_iterations,
                          double* cov9, uint32_t* iterations);
const char* lama_slam_engine_origin(const lama_slam* s);

/* ---- lama::Loc2D (include/lama/loc2d.h), flattened ---- */
typedef struct lama_loc lama_loc;
lama_loc* lama_loc_create(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t max_iter,
                          int32_t gpu_device, char* err, int errcap);
void lama_loc_destroy(lama_loc* l);
const char* lama_loc_last_error(const lama_loc* l);
const char* lama_loc_engine_origin(const lama_loc* l);
/* distance_map->addObstacle(w2m(x, y)) for every world point, then distance_map->update() */
int lama_loc_set_obstacles_world(lama_loc* l, const double* xy, uint32_t n);
/* distance_map->write(file) / distance_map->read(file): the reference's .sdm format (Map::write / Map::read, src/sdm/map.cpp:489-575).
 * read REPLACES the device map by the file's patches (lama_hip_pf_upload_map): a static map is loaded instead of rebuilt. */
int lama_loc_write_distance_map(lama_loc* l, const char* filename);
int lama_loc_read_distance_map(lama_loc* l, const char* filename);
void* lama_loc_device_context(const lama_loc* l);      /* the lama_hip_ctx behind the object (NULL before the first map / update) */
void lama_loc_set_pose(lama_loc* l, double x, double y, double yaw);
int lama_loc_get_pose(const lama_loc* l, double* pose4);
int lama_loc_update(lama_loc* l, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz,
                    const double* odom_xyr, double timestamp, int force_update);   /* 1 / 0 / <0 */
int lama_loc_covar(const lama_loc* l, double* out9);
double lama_loc_rmse(const lama_loc* l);
uint32_t lama_loc_iterations(const lama_loc* l);
/* global localisation / sampling covariance (src/loc2d.cpp:194-286) */
lama_loc* lama_loc_create2(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t max_iter,
                           uint32_t gloc_particles, uint32_t gloc_iters, double gloc_thresh, double cov_blend,
                           int32_t gpu_device, char* err, int errcap);
lama_loc* lama_loc_create3(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t max_iter,
                           uint32_t gloc_particles, uint32_t gloc_iters, double gloc_thresh, double cov_blend,
                           const char* strategy /* "gn" | "lm" */, int32_t gpu_device, char* err, int errcap);
/* occupancy_map->setFree / setUnknown / setOccupied on map cells (x, y pairs); state -1 / 0 / 1 */
int lama_loc_occ_set_cells(lama_loc* l, const uint32_t* cells_xy, uint32_t n, int state);
int lama_loc_occ_bounds(const lama_loc* l, double* out6);
void lama_loc_trigger_global_localization(lama_loc* l);
int lama_loc_global_localization_active(const lama_loc* l);
uint32_t lama_loc_gloc_candidates(const lama_loc* l, double* poses4, double* errors, uint32_t cap);
uint32_t lama_loc_sampling_likelihoods(const lama_loc* l, double* out, uint32_t cap);
/* ---- lama::LidarOdometry2D (include/lama/lidar_odometry_2d.h), flattened ---- */
typedef struct lama_lo lama_lo;
lama_lo* lama_lo_create(double resolution, uint32_t max_iter, int32_t gpu_device, char* err, int errcap);
void lama_lo_destroy(lama_lo* l);
const char* lama_lo_last_error(const lama_lo* l);
const char* lama_lo_engine_origin(const lama_lo* l);
int lama_lo_update(lama_lo* l, const double* pts_xyz, uint32_t n, const double* origin3, const double* quat_wxyz, double timestamp);   /* 1 / <0 */
int lama_lo_get_odom(const lama_lo* l, double* pose4);
uint32_t lama_lo_iterations(const lama_lo* l);
uint32_t lama_lo_deleted_patches(const lama_lo* l);
void* lama_lo_device_context(const lama_lo* l);

/* ---- lama::sdm map formats (include/lama/sdm_io.h): the reference's `.sdm` file (Map::write/read, src/sdm/map.cpp:489-575)
 * and the export images of src/sdm/export.cpp:46-110 for maps downloaded from the device.
 * kind: 0 DynamicDistanceMap (10 B cells), 1 FrequencyOccupancyMap (4 B), 2 SimpleOccupancyMap (1 B). */
int lama_sdm_write(const char* file, int kind, double resolution, uint32_t max_sqdist, uint32_t n,
                   const uint64_t* ids, const uint8_t* cells, const uint64_t* masks);
int lama_sdm_read(const char* file, int* kind, double* resolution, uint32_t* max_sqdist, uint32_t cap,
                  uint64_t* ids, uint8_t* cells, uint64_t* masks, uint32_t* n);       /* cap = 0: query n */
int lama_sdm_image(int kind, double resolution, uint32_t max_sqdist, uint32_t n, const uint64_t* ids, const uint8_t* cells,
                   const uint64_t* masks, uint32_t* width, uint32_t* height, uint8_t* out, uint64_t cap);
int lama_sdm_export_png(int kind, double resolution, uint32_t max_sqdist, uint32_t n, const uint64_t* ids, const uint8_t* cells,
                        const uint64_t* masks, const char* file);
/* The FIRST build of lama::Loc2D's distance map on the host (iris_lama_amd/host/dm_builder.hpp: addObstacle for n cells of an empty
 * map in the given order + one update(), /root/reference/src/loc2d.cpp:61-108, src/sdm/dynamic_distance_map.cpp:160-226,281-330) --
 * what lama::DynamicDistanceMap::update() of a Loc2D runs before it uploads the result to the device.  Exposed so that the harness can
 * compare it with the checker and time it.  lama_dm_build returns the number of patches (kept until the next call on this thread;
 * -1: not buildable on the host, e.g. max_sqdist beyond the device's 14 bits) and update()'s return value in *processed;
 * lama_dm_build_fetch copies the records out (ids n, cells n x 10240 B, masks n x 16 words). */
int64_t lama_dm_build(const uint32_t* cells_xy, uint64_t n, uint32_t max_sqdist, uint32_t* processed);
int lama_dm_build_fetch(uint64_t* ids, uint8_t* cells, uint64_t* masks);
/* lama::random (include/lama/random.h) */
void lama_random_set_seed(uint32_t seed);
double lama_random_uniform(void);

#ifdef __cplusplus
}
#endif
#endif
