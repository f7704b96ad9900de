/* =====================================================================================
 * lama_hip.h -- C-ABI of the MI355X (gfx950) particle-filter scan-matching path.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its seam
 * is the C++ class API of include/lama/ (static library `iris_lama`).  The entry points below
 * are what a host-side `lama::PFSlam2D` calls INSTEAD of the bodies of the reference's two
 * per-particle parallel regions and of resample():
 *
 *   reference (file:line, relative to /root/reference)            entry point here
 *   -----------------------------------------------------------   --------------------------
 *   PFSlam2D::update first-scan block   src/pf_slam2d.cpp:185-228  lama_hip_pf_init
 *   region 1: scanMatch per particle    src/pf_slam2d.cpp:254-266  lama_hip_pf_scan_match
 *     = MatchSurface2D::eval            src/match_surface_2d.cpp:42-90
 *     + Solver::solve / GaussNewton     src/nlls/solver.cpp:53-107, src/nlls/gauss_newton.cpp:53-86
 *     + CauchyWeight                    src/nlls/robust_cost.cpp:66-73
 *     + calculateLikelihood             src/pf_slam2d.cpp:393-414
 *   resample(): particle-set copy       src/pf_slam2d.cpp:558-574  lama_hip_pf_resample
 *   region 2: updateParticleMaps        src/pf_slam2d.cpp:292-302  lama_hip_pf_update_maps
 *     = Map::computeRay                 src/sdm/map.cpp:198-227
 *     + FrequencyOccupancyMap::set*     src/sdm/frequency_occupancy_map.cpp:65-91
 *     + DynamicDistanceMap add/remove/update  src/sdm/dynamic_distance_map.cpp:160-330
 *   getOccupancyMap()/getDistanceMap()  include/lama/pf_slam2d.h:211-225  lama_hip_pf_download_map
 *   Particle::pose read/write           include/lama/pf_slam2d.h:77      lama_hip_pf_set_poses / _get_poses
 *
 * Host-side pieces that stay on the host (RNG order is part of the result): drawFromMotion
 * (:365-391), normalize (:511-535), systematic-resampling index generation (:537-556).
 *
 * Conventions
 *   - plain C, POD arguments only; no exceptions, STL, Eigen or torch types cross this boundary;
 *   - every function returns int32_t status: LAMA_HIP_OK (0) or a negative LAMA_HIP_E_* code;
 *     lama_hip_last_error() gives a human-readable message for the last failure on that context;
 *   - host buffers are borrowed for the duration of the call; outputs go to caller-provided arrays;
 *   - calls on one context must be serialised by the caller (same contract as PFSlam2D::update);
 *   - a pose is the reference's SE2 state: 4 doubles {c, s, tx, ty} (unit complex + translation,
 *     include/lama/sophus/se2.hpp); the scan is PointCloudXYZ (include/lama/types.h:111-120):
 *     n x 3 doubles + sensor_origin_[3] + sensor_orientation_ quaternion {w,x,y,z}.
 * ===================================================================================== */
#ifndef LAMA_HIP_H
#define LAMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LAMA_HIP_OK 0
#define LAMA_HIP_E_INVALID (-1)    /* bad argument / unsupported option                         */
#define LAMA_HIP_E_HIP (-2)        /* a HIP runtime call failed                                 */
#define LAMA_HIP_E_WINDOW (-3)     /* a cell fell outside the device map window                 */
#define LAMA_HIP_E_CAPACITY (-4)   /* patch arena / brushfire queue capacity exceeded           */
#define LAMA_HIP_E_STATE (-5)      /* call sequence error (e.g. scan_match before init)         */
#define LAMA_HIP_E_NUMERIC (-6)    /* zero-norm unit complex (the reference throws SophusException) */

typedef struct lama_hip_ctx lama_hip_ctx;

/* Subset of PFSlam2D::Options (include/lama/pf_slam2d.h:132-185) the device path needs, plus
 * device-side capacities (0 = default). */
typedef struct lama_hip_cfg {
    uint32_t particles;          /* particles owned by THIS context (one shard of the pool)       */
    double resolution;           /* Options::resolution (0.05)                                    */
    uint32_t patch_size;         /* Options::patch_size; only 32 is supported on the device       */
    double l2_max;               /* Options::l2_max -> DynamicDistanceMap::setMaxDistance.  ceil(l2_max / resolution) <= 127 cells in
                                    liblama_hip.so; liblama_hip_wide.so -- the same sources and the same C-ABI, built with a 4-byte
                                    distance plane -- takes up to 255 cells, which is all the reference's own record can hold
                                    (distance_t::sqdist is a uint16_t).  The host classes and iris_lama_amd.ffi bind the library
                                    that the reach of the map asks for; a C caller links the one it needs.               */
    double meas_sigma;           /* Options::meas_sigma (likelihood divisor, pf_slam2d.cpp:411)   */
    uint32_t max_iter;           /* Options::max_iter                                             */
    double truncated_ray;        /* Options::truncated_ray                                        */
    double truncated_range;      /* Options::truncated_range                                      */
    int32_t device;              /* HIP device ordinal                                            */
    uint32_t window_patches;     /* INITIAL side of the square map window in patches (default 128 = 204.8 m; a multiple of 8).  The window
                                    follows the robot and GROWS (re-allocated at least half as large again) when the mapped area
                                    outgrows it, up to 1016 patches = 1.6 km at 0.05 m; only a wider map is LAMA_HIP_E_WINDOW */
    uint32_t dm_patch_capacity;  /* smallest DM region a particle gets, in patches (default 128: 10 m x 10 m of map; regions grow on demand).  The reference's maps are unbounded; here
                                    every particle owns one contiguous region per map kind inside pooled planes: the region grows with
                                    ITS particle's map (moved to a larger one, that particle alone, when it runs short), the pools grow
                                    chunk by chunk.  A map update that needs more patches than are free reports that BEFORE it modifies
                                    any cell (its allocation phase comes first), upon which the particles that ran short get what they
                                    asked for and the update is run again.  LAMA_HIP_E_CAPACITY remains for the hard limit of 32767
                                    patches per particle and map, and for a device that has no memory left for another pool chunk */
    uint32_t occ_patch_capacity; /* smallest occupancy region a particle gets (default 128)      */
    uint32_t queue_capacity;     /* brushfire queue entries per particle (default 32768)          */
    uint32_t profile;            /* !=0: bracket every kernel with hipEvents (lama_hip_get_counters) */
    uint32_t active_capacity;    /* parallel ray-cast: max. order-sensitive cell visits per particle and scan (default 8192) */
    uint32_t sequential_raycast; /* ray-cast kernels: 0 / 2 = parallel form (default), 1 = beam-sequential form; bit-identical */
    uint32_t brushfire_mode;     /* 0 = exact (default): bit-identical to the reference incl. libstdc++'s tie order;
                                    1 = level-synchronous with a canonical tie rule (parallel; identical sqdist/valid/masks on
                                        the measured logs, obstacle offsets of tie cells -- and, once in millions of cells, a
                                        distance -- may differ).  NOT bit-identical: only a caller that sets
                                        this field gets it (no environment variable selects it) and lama_hip_get_counters
                                        reports the mode that ran.  Other values are rejected. */
    uint32_t brushfire_waves;    /* exact brushfire: 0 / 2 = a helper wave per particle owns the heap (default), 1 = one wave per
                                    particle; bit-identical */
    uint32_t occupancy_policy;   /* cell policy of the occupancy map: 0 = FrequencyOccupancyMap {uint16 occupied, uint16 visited}
                                    (PFSlam2D, Slam2D); 1 = ProbabilisticOccupancyMap {float log-odds}
                                    (src/sdm/probabilistic_occupancy_map.cpp:53-107; LidarOdometry2D) -- beam-sequential ray-cast */
    uint32_t ray_rule;           /* where a ray starts: 0 = PFSlam2D/Slam2D (truncated_ray / truncated_range options),
                                    1 = LidarOdometry2D::updateMaps (src/lidar_odometry_2d.cpp:108-114: the last metre before the hit) */
    uint32_t solver_strategy;    /* nlls strategy of the scan matcher: 0 = GaussNewton (src/nlls/gauss_newton.cpp; PFSlam2D always),
                                    1 = LevenbergMarquard (src/nlls/levenberg_marquardt.cpp; Slam2D / Loc2D with strategy "lm") */
} lama_hip_cfg;

void lama_hip_default_cfg(lama_hip_cfg* cfg);

int32_t lama_hip_device_count(int32_t* count);

int32_t lama_hip_ctx_create(const lama_hip_cfg* cfg, lama_hip_ctx** out);
void lama_hip_ctx_destroy(lama_hip_ctx* ctx);
const char* lama_hip_last_error(const lama_hip_ctx* ctx);

/* First scan (src/pf_slam2d.cpp:185-228): every particle's pose = pose0; particle 0 builds its maps
 * from the scan (updateParticleMaps), all others become copies of particle 0. */
int32_t lama_hip_pf_init(lama_hip_ctx* ctx, const double* pts_xyz, uint32_t n,
                         const double* sensor_origin3, const double* sensor_quat_wxyz,
                         const double* pose0);

/* Particle poses, P x 4 doubles {c,s,tx,ty}. */
int32_t lama_hip_pf_set_poses(lama_hip_ctx* ctx, const double* poses);
int32_t lama_hip_pf_get_poses(lama_hip_ctx* ctx, double* poses);

/* Region 1 (scanMatch): per particle, Gauss-Newton + Cauchy(0.15) on the particle's own distance map
 * starting from its current pose; the refined pose replaces the particle's pose.
 * Outputs (any may be NULL): poses_out P x 4, loglik_out P (calculateLikelihood),
 * iters_out P (Solver iterations = applied + reverted steps). */
int32_t lama_hip_pf_scan_match(lama_hip_ctx* ctx, const double* pts_xyz, uint32_t n,
                               const double* sensor_origin3, const double* sensor_quat_wxyz,
                               double* poses_out, double* loglik_out, int32_t* iters_out);

/* resample(): new particle i := copy of old particle sample_idx[i] (pose + both maps). */
int32_t lama_hip_pf_resample(lama_hip_ctx* ctx, const int32_t* sample_idx);

/* Region 2 (updateParticleMaps): per particle ray-cast of the scan from the particle's pose into its
 * FrequencyOccupancyMap, add/remove obstacle events into its DynamicDistanceMap, then dm->update().
 * pts_xyz may be NULL to reuse the scan (same n) uploaded by the preceding lama_hip_pf_scan_match call. */
int32_t lama_hip_pf_update_maps(lama_hip_ctx* ctx, const double* pts_xyz, uint32_t n,
                                const double* sensor_origin3, const double* sensor_quat_wxyz);

/* Map download in the REFERENCE's record formats, so a host lama::Map can be rebuilt byte for byte:
 *   kind LAMA_HIP_MAP_DISTANCE : cells are DynamicDistanceMap::distance_t (10 B:
 *        int16 obstacle[3], uint16 sqdist, bool valid_obstacle, bool is_queued), 10240 B / patch
 *   kind LAMA_HIP_MAP_OCCUPANCY: cells are FrequencyOccupancyMap::frequency (4 B: uint16 occupied,
 *        uint16 visited), 4096 B / patch
 *   patch_ids: the reference's 64-bit patch index (Map::m2p, include/lama/sdm/map.h:153-161), ascending
 *   masks    : 16 x uint64 per patch (Container::mask, include/lama/sdm/container.h:57-65)
 * lama_hip_pf_map_patches returns the number of patches; download fills up to `cap` patches. */
#define LAMA_HIP_MAP_DISTANCE 0
#define LAMA_HIP_MAP_OCCUPANCY 1
int32_t lama_hip_pf_map_patches(lama_hip_ctx* ctx, uint32_t particle, int32_t kind, uint32_t* num_patches);
int32_t lama_hip_pf_download_map(lama_hip_ctx* ctx, uint32_t particle, int32_t kind, uint32_t cap,
                                 uint64_t* patch_ids, uint8_t* cells, uint64_t* masks, uint32_t* num_patches);

/* The inverse of lama_hip_pf_download_map -- Map::read (src/sdm/map.cpp:533-575) for a map that lives on the device: particle
 * `particle`'s map of `kind` is REPLACED by the given patches (the reference's record formats, as download returns them and as a
 * `.sdm` file stores them: lama::sdm::read).  A distance map built elsewhere -- by the reference, by an earlier run, by another
 * context -- is put in place without replaying addObstacle() + update(): Loc2D's static map of a building is one serial brushfire
 * chain of millions of pops when it is built on the device (bench.py: next_rows.loc2d_map_load), a file read when it is loaded.
 * patch_ids need not be sorted; a patch outside the largest window is LAMA_HIP_E_WINDOW.  On a context that has not seen
 * lama_hip_pf_init it also places the map window (like lama_hip_map_add_obstacles). */
int32_t lama_hip_pf_upload_map(lama_hip_ctx* ctx, uint32_t particle, int32_t kind, uint32_t num_patches,
                               const uint64_t* patch_ids, const uint8_t* cells, const uint64_t* masks);

/* The same map update split in two, so that the host work of the NEXT scan (odometry prediction, motion sampling) overlaps
 * with the kernels: _begin queues the work on the context's stream and returns; its status (window / capacity errors) and
 * counters are collected by lama_hip_sync, or implicitly at the start of the next call on the context, whichever comes
 * first -- a deferred error is then returned by THAT call (message prefixed "deferred from ...").
 * lama_hip_pf_update_maps == _begin followed by lama_hip_sync. */
int32_t lama_hip_pf_update_maps_begin(lama_hip_ctx* ctx, const double* pts_xyz, uint32_t n,
                                      const double* sensor_origin3, const double* sensor_quat_wxyz);
int32_t lama_hip_sync(lama_hip_ctx* ctx);
/* HIP device ordinal the context lives on (cfg.device); < 0 for a null context.  Lets a caller of a multi-GPU object check
 * that its shards really sit on distinct devices. */
int32_t lama_hip_ctx_device(const lama_hip_ctx* ctx);

/* Patch bookkeeping for transient maps (LidarOdometry2D::updateMaps, src/lidar_odometry_2d.cpp:128-199):
 *   lama_hip_pf_patch_ids     : Map::visit_all_patches -- the reference patch indices (Map::m2p) of the allocated patches;
 *   lama_hip_pf_delete_patches: Map::deletePatchAt (src/sdm/map.cpp:465-488) on BOTH maps of the particle for every listed patch
 *                               index (absent patches are skipped); the arenas stay dense.  *deleted = distance-map patches removed. */
int32_t lama_hip_pf_patch_ids(lama_hip_ctx* ctx, uint32_t particle, int32_t kind, uint32_t cap, uint64_t* patch_ids, uint32_t* num_patches);
/* One 64-bit checksum per particle of its distance / occupancy map, computed on the device: order-independent sum over the
 * allocated patches (reference patch indices, Map::m2p) and their cells of a hash of what the reference stores per cell
 * (distance_t / frequency, include/lama/sdm/dynamic_distance_map.h:48-53, frequency_occupancy_map.h:43-46) and of the
 * Container mask bit (include/lama/sdm/container.h:167-183).  Two maps with equal patch sets, cells and masks have equal
 * checksums; used to compare ALL particles of a large filter with a checker without downloading the maps.  out: P values. */
int32_t lama_hip_pf_map_checksums(lama_hip_ctx* ctx, int32_t kind, uint64_t* out);
int32_t lama_hip_pf_delete_patches(lama_hip_ctx* ctx, uint32_t particle, const uint64_t* patch_ids, uint32_t n, uint32_t* deleted);

/* Batched evaluation on ONE particle's distance map (Loc2D::globalLocalization-style):
 * B poses -> B log-likelihoods (calculateLikelihood) without changing any state. */
int32_t lama_hip_match_batch(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                             const double* sensor_origin3, const double* sensor_quat_wxyz,
                             const double* poses, uint32_t num_poses, double* loglik_out);

/* Loc2D::globalLocalization's candidate evaluation (src/loc2d.cpp:249-286, the loop body :271-284): for each of the
 * B poses {c, s, tx, ty} the squared norm of MatchSurface2D's residuals (bilinear distance per point, no robust
 * weight; src/match_surface_2d.cpp:58-101) on `particle`'s distance map, and/or the particle filter's log-likelihood
 * (either output may be NULL).  No state changes.  The candidates themselves are drawn on the host (random::uniform). */
int32_t lama_hip_eval_batch(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                            const double* sensor_origin3, const double* sensor_quat_wxyz,
                            const double* poses, uint32_t num_poses, double* sqnorm_out, double* loglik_out);

/* Loc2D::addSamplingCovariance's likelihood samples (src/loc2d.cpp:199-234): for each of the K world positions
 * xy[k] the scan is placed there with heading `yaw`; every `point_step`-th point (0, step, 2 step, ...; at most 128
 * of them) looks up the NON-interpolated distance d and l_out[k] = sum exp(-d^2 / 0.01)^3, summed in point order. */
int32_t lama_hip_map_sample_likelihood(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                                       const double* sensor_origin3, const double* sensor_quat_wxyz, double yaw,
                                       const double* xy, uint32_t num_samples, uint32_t point_step, double* l_out);

/* Static-map localisation support (Loc2D, src/loc2d.cpp:61-192) on a one-particle context:
 *  lama_hip_map_add_obstacles: DynamicDistanceMap::addObstacle for every cell of the list (map coordinates as
 *      returned by Map::w2m, x then y, in the given order) followed by dm->update().  On a context that has not seen
 *      lama_hip_pf_init it also places the map window (centred on the first cell).
 *  lama_hip_match_solve: Solve(GaussNewton + Cauchy(0.15), MatchSurface2D(dm, scan, pose), &cov): pose_inout {c,s,tx,ty};
 *      out7 = lower triangle of J^T J (weighted J at the solution: 00,10,11,20,21,22) and the sum of squared unweighted
 *      residuals; iters_out = solver iterations.  do_solve == 0 only evaluates at the given pose. */
int32_t lama_hip_map_add_obstacles(lama_hip_ctx* ctx, uint32_t particle, const uint32_t* cells_xy, uint32_t n);
int32_t lama_hip_match_solve(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                             const double* sensor_origin3, const double* sensor_quat_wxyz, double* pose_inout,
                             double* out7, int32_t* iters_out, int32_t do_solve);
/* The same with the strategy chosen per call (0 = GaussNewton, 1 = LevenbergMarquard) instead of cfg.solver_strategy:
 * and its own iteration limit: Solve(options, MatchSurface2D&, &cov) of a caller that builds its own Solver::Options
 * (include/lama/nlls/solver.h:52-67). */
int32_t lama_hip_match_solve_with(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                                  const double* sensor_origin3, const double* sensor_quat_wxyz, double* pose_inout,
                                  double* out7, int32_t* iters_out, int32_t strategy, uint32_t max_iterations /* 0: cfg.max_iter */);
/* MatchSurface2D::eval (src/match_surface_2d.cpp:42-90; the reference's nlls::Problem interface, include/lama/nlls/problem.h:
 * 40-57) against particle `particle`'s distance map at pose {c,s,tx,ty}: residuals[n] = interpolated distance at every beam's
 * end point (no robust weight), jacobian (may be NULL) = n x 3 COLUMN-major [gx | gy | gy*hx - gx*hy]. */
int32_t lama_hip_match_eval(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                            const double* sensor_origin3, const double* sensor_quat_wxyz, const double* pose,
                            double* residuals, double* jacobian);
/* The per-beam terms of MatchSurface2D::error (src/match_surface_2d.cpp:92-116): distances[i] = DynamicDistanceMap::distance
 * of the CELL w2m(tf * p_i) (no interpolation); the caller takes sqrt(sum of squares / n). */
int32_t lama_hip_match_cell_distances(lama_hip_ctx* ctx, uint32_t particle, const double* pts_xyz, uint32_t n,
                                      const double* sensor_origin3, const double* sensor_quat_wxyz, const double* pose,
                                      double* distances);

/* Particle shipping for multi-GPU resampling (one context per GPU): serialise one particle (pose + both
 * maps, used patches only) into a DEVICE buffer / restore it into slot `particle` of this context.
 * export returns the number of bytes needed in *bytes when buf == NULL. */
int32_t lama_hip_pf_export_particle(lama_hip_ctx* ctx, uint32_t particle, void* device_buf, uint64_t cap, uint64_t* bytes);
int32_t lama_hip_pf_import_particle(lama_hip_ctx* ctx, uint32_t particle, const void* device_buf, uint64_t bytes);
/* The same for ALL outgoing / incoming particles of a resample at once: one kernel launch and one synchronisation per batch instead
 * of a dozen copies per particle (a 3000-particle filter on 8 GPUs ships hundreds of particles per resample).  export: device_bufs ==
 * NULL only reports the sizes in bytes_out[n]; import: a slot may appear once per batch. */
int32_t lama_hip_pf_export_particles(lama_hip_ctx* ctx, uint32_t n, const uint32_t* particles, void* const* device_bufs, const uint64_t* caps,
                                     uint64_t* bytes_out);
int32_t lama_hip_pf_import_particles(lama_hip_ctx* ctx, uint32_t n, const uint32_t* particles, const void* const* device_bufs, const uint64_t* bytes);

/* (Peer access: the first lama_hip_blob_copy between two devices queries hipDeviceCanAccessPeer and enables access in both
 * directions, once per pair and process; when the platform refuses it the copy still works -- the runtime stages it through host
 * memory -- and lama_hip_counters::peer_access says 0.)
 * Device staging buffers for particle shipping when ONE process drives several contexts (lama::PFSlam2D with Options::gpus > 1: a
 * host thread per GPU, src/pf_slam2d.cpp:254-302's two parallel regions become G device streams): allocate / free a buffer on the
 * context's device, and copy between buffers of two contexts -- hipMemcpyPeerAsync when they live on different GPUs (xGMI), a
 * plain device copy otherwise.  The copy is complete when the call returns. */
int32_t lama_hip_blob_alloc(lama_hip_ctx* ctx, uint64_t bytes, void** device_buf);
int32_t lama_hip_blob_free(lama_hip_ctx* ctx, void* device_buf);
int32_t lama_hip_blob_copy(lama_hip_ctx* dst_ctx, void* dst_device_buf, lama_hip_ctx* src_ctx, const void* src_device_buf, uint64_t bytes);

/* SE2 pose-graph linearisation: the per-factor body and the accumulation of minisam's
 * linearzationLowerHessian (vendor/minisam/minisam/nonlinear/linearization.cpp:150-272,290-341) for PriorFactor<SE2d> /
 * BetweenFactor<SE2d> with DiagonalLoss, as built by SimplePGO::optimize (src/simple_pgo.cpp:48-105) and
 * GraphSlam2D::optimizePoseGraph (src/graph_slam2d.cpp:394-430).  The graph structure and measurements are uploaded once
 * (create), every Levenberg-Marquardt iteration calls linearize with the current poses {c, s, tx, ty}.  Outputs, dense 3x3
 * blocks row-major (the caller scatters them into its sparse lower Hessian and runs the Cholesky on the host):
 *   err   [F][3]  whitened error           Hoff [F][9]  J_i^T J_j of factor f (zero for a prior, fj = -1)
 *   Hdiag [N][9]  sum of J_v^T J_v over the factors of v, in factor order     b [N][3]  Atb = -sum J_v^T err
 * Any output may be NULL.  kernel_ms: device time of the two kernels (hipEvents on the graph's stream). */
typedef struct lama_hip_pgo lama_hip_pgo;
int32_t lama_hip_pgo_create(int32_t device, uint32_t num_poses, const int32_t* fi, const int32_t* fj, const double* meas4,
                            const double* sqrt_info3, uint32_t num_factors, lama_hip_pgo** out);
void lama_hip_pgo_destroy(lama_hip_pgo* g);
const char* lama_hip_pgo_last_error(const lama_hip_pgo* g);
int32_t lama_hip_pgo_linearize(lama_hip_pgo* g, const double* poses4, double* err, double* Hdiag, double* Hoff, double* b,
                               double* chi2, double* kernel_ms);

/* Accumulated per-kernel device time (hipEvent elapsed, on the stream the kernels run on) and work
 * counters since the last reset; valid when cfg.profile != 0. */
typedef struct lama_hip_counters {
    double ms_scan_match;  uint64_t launches_scan_match;
    double ms_update_maps; uint64_t launches_update_maps;   /* = raycast + brushfire */
    double ms_raycast;     uint64_t launches_raycast;
    double ms_brushfire;   uint64_t launches_brushfire;
    double ms_resample;    uint64_t launches_resample;
    uint64_t gn_iterations;     /* sum over particles and calls                                    */
    uint64_t gn_evals;          /* residual evaluations (with or without Jacobian) + likelihood     */
    uint64_t ray_cells;         /* Bresenham cells visited                                          */
    uint64_t bf_cells;          /* brushfire cells processed (DynamicDistanceMap::update return)     */
    uint64_t dm_patches;        /* sum of allocated DM patches over particles (current)             */
    uint64_t occ_patches;       /* sum of allocated occupancy patches over particles (current)      */
    double ms_eval_batch;  uint64_t launches_eval_batch;     /* lama_hip_eval_batch                   */
    uint64_t arena_growths;     /* batches of particles moved to larger regions (maps grow on demand)  */
    uint64_t window_shifts;     /* times the map window was re-centred (the window follows the robot)  */
    uint64_t wrap_guard_scans;  /* scans ray-cast beam by beam because a uint16 `visited` counter could wrap inside them */
    /* what actually ran (set by every map update; not cleared by lama_hip_reset_counters' zeroing of the sums above): */
    uint32_t brushfire_mode;    /* cfg.brushfire_mode of the last map update: 0 = exact, 1 = canonical tie rule (not bit-identical) */
    uint32_t brushfire_waves;   /* exact brushfire of the last map update: 2 = wave pair per particle, 1 = one wave per particle */
    uint64_t sequential_raycast_scans;  /* map updates whose ray-cast ran beam by beam (k_raycast)                           */
    uint64_t parallel_raycast_scans;    /* map updates whose ray-cast ran in the parallel, patch-centric form                */
    uint32_t brushfire_handovers;       /* particle updates whose brushfire queue outgrew the first stage's LDS window (1020 entries) and
                                           were finished by the resume stage (sum since the last reset).  Such an update is one long
                                           serial chain: at 3000 particles one of them (13.9 ms) is what made the resume kernel's MEAN
                                           1.2 ms in the round-3 profile while its median is 5 us (profiles/r04_timeline_3000_before.txt) */
    uint32_t replay_handovers;          /* the same for the ordered replay of the parallel ray-cast (more than 2048 order-sensitive visits) */
    uint32_t window_patches;            /* current side of the map window in patches (it grows with the mapped area, up to 1016)            */
    uint32_t window_growths;            /* times the window directories were re-allocated with a larger side                                */
    uint64_t bf_longest_chain_sum;      /* sum over map updates of the LARGEST brushfire cell count of a single particle: a particle's exact
                                           brushfire is one serial chain, so a map update lasts as long as its longest chain --
                                           bf_longest_chain_sum / map updates against bf_cells / (particles x map updates) is the spread   */
    uint64_t bf_longest_chain_last;     /* the largest per-particle count of the last map update                                           */
    uint64_t brushfire_early;           /* ... of which: particles the PREVIOUS update had routed and that therefore ran their ray-cast and their
                                           brushfire in the early lane, ahead of everybody else's ray-cast (counted in brushfire_routed too)   */
    uint64_t brushfire_routed;          /* particle updates whose brushfire ran in the big-queue stage from the start, beside the first stage:
                                           their obstacle-event count (known before the brushfire starts) marked them as the long chains   */
    /* round 5: memory that follows the maps (one particle set, in-place resampling, per-particle regions in pooled planes) */
    uint64_t pool_growths;              /* times a patch pool was re-allocated larger (geometric)                                          */
    uint64_t resample_clones;           /* particles a resample / the first scan really COPIED: a particle that survives keeps its home and
                                           regions, only the second and further copies of a multiply drawn particle are written           */
    uint64_t resample_bytes;            /* bytes those copies read (= wrote): directories + used slots of every plane                       */
    uint64_t hbm_bytes_allocated;       /* device memory the context holds for the maps now: patch pools + directories                     */
    uint64_t hbm_bytes_used;            /* ... of which used: the allocated patches of all particles (all planes) + directories            */
    uint64_t hbm_bytes_total;           /* everything the context has allocated on the device (maps, queues, scratch)                      */
    uint32_t peer_access;               /* lama_hip_blob_copy between two devices: 1 = the last such copy went GPU to GPU directly (peer access
                                           enabled both ways), 0 = it was staged by the runtime (peer access refused) or none was made      */
    uint32_t struct_bytes;              /* sizeof(lama_hip_counters) of the library that filled the struct (see lama_hip_get_counters_sized) */
    double peer_copy_ms;                /* device time (hipEvents on the destination's stream) of the cross-device lama_hip_blob_copy calls ... */
    uint64_t peer_copy_bytes;           /* ... and the bytes they moved INTO this context: bytes / ms = the achieved xGMI rate               */
} lama_hip_counters;
int32_t lama_hip_get_counters(lama_hip_ctx* ctx, lama_hip_counters* out);
/* The same for a caller compiled against another version of this header: at most `bytes` bytes are written (the struct
 * grows at its END only, a consumer built against an older header passes its own sizeof and is never written past). */
int32_t lama_hip_get_counters_sized(lama_hip_ctx* ctx, void* out, uint32_t bytes);
uint32_t lama_hip_counters_bytes(void);     /* sizeof(lama_hip_counters) in the library (a binding checks its own mirror against it) */
int32_t lama_hip_reset_counters(lama_hip_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* LAMA_HIP_H */
