// lama/loc2d.h -- host-side lama::Loc2D (localisation on a fixed map) on the MI355X path.
//
// Same class name, Options fields and public methods as the reference's include/lama/loc2d.h:47-165; update()
// follows src/loc2d.cpp:126-192.  The public members `occupancy_map` / `distance_map` have the reference's types
// (include/lama/loc2d.h:103-104): consumers (iris_lama_ros' loc2d_ros) fill them cell by cell before the first update.
// The SimpleOccupancyMap is a plain host map; the DynamicDistanceMap is LIVE: addObstacle() buffers cells on the host,
// update() runs DynamicDistanceMap::addObstacle + update() on the device, queries read a snapshot downloaded on first use
// (lama/sdm_maps.h).  On the device: scan matching with covariance
// (Solve(..., &cov)) and the RMSE (lama_hip_match_solve), the candidate evaluation of globalLocalization (:249-286,
// lama_hip_eval_batch; the candidates are drawn on the host from lama::random like the reference) and the likelihood
// samples of addSamplingCovariance (:199-247, lama_hip_map_sample_likelihood); strategy "lm" = Levenberg-Marquardt in the
// same kernel (cfg.solver_strategy).
#pragma once

#include <cstdint>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "pose2d.h"
#include "sdm/dynamic_distance_map.h"
#include "sdm/simple_occupancy_map.h"

struct lama_hip_ctx;

namespace lama {

struct HipEngine;

class Loc2D {
public:
    struct Options {
        Options();
        double trans_thresh, rot_thresh, l2_max, resolution;
        uint32_t patch_size, max_iter;
        std::string strategy;
        uint32_t gloc_particles, gloc_iters;
        double gloc_thresh, cov_blend;
        int32_t gpu_device = 0;      // addition
    };

    SimpleOccupancyMap* occupancy_map = nullptr;      // include/lama/loc2d.h:103-104
    DynamicDistanceMap* distance_map = nullptr;

    Loc2D() = default;
    void Init(const Options& options = Options());
    virtual ~Loc2D();

    bool enoughMotion(const Pose2D& odometry);
    bool update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp, bool force_update = false);
    void triggerGlobalLocalization();
    void setPose(const Pose2D& pose) { pose_ = pose; has_first_scan = false; }
    const Pose2D& getPose() const { return pose_; }
    const Matrix3d& getCovar() const { return cov_; }     // Eigen::Matrix3d when Eigen is available (lama/types.h)
    double getRMSE() const { return rmse_; }
    bool globalLocalizationIsActive() const { return do_global_localization_; }
    // candidates and errors of the last globalLocalization call (instrumentation for the parity tests)
    const std::vector<double>& lastGlocPoses() const { return gloc_poses_; }
    const std::vector<double>& lastGlocErrors() const { return gloc_errors_; }
    const std::vector<double>& lastSamplingLikelihoods() const { return sampling_l_; }
    uint32_t getLastIterations() const { return last_iterations_; }
    // true when the first build of the distance map was replayed on the host and uploaded (iris_lama_amd/host/dm_builder.hpp)
    bool distanceMapBuiltOnHost() const { return host_built_; }
    lama_hip_ctx* deviceContext() const { return ctx_; }
    const HipEngine* engine() const { return eng_.get(); }

private:
    void ensureContext();
    void solve(const PointCloudXYZ& surface, bool do_solve);
    void globalLocalization(const PointCloudXYZ& surface);
    void addSamplingCovariance(const PointCloudXYZ& surface);
    void fail(int32_t rc, const char* what) const;

    Options opt_;
    std::shared_ptr<HipEngine> eng_;
    lama_hip_ctx* ctx_ = nullptr;
    Pose2D odom_, pose_;
    Matrix3d cov_;
    double rmse_ = 0.0;
    bool has_first_scan = false;
    uint32_t last_iterations_ = 0;
    bool host_built_ = false;
    bool do_global_localization_ = false;
    uint32_t gloc_cur_iter_ = 0;
    double cov_blend_ = 0.0;
    std::vector<Vector2d> sampling_steps_;
    std::vector<double> gloc_poses_, gloc_errors_, sampling_l_;
};

} // namespace lama
