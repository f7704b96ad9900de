// lama/lidar_odometry_2d.h -- host-side lama::LidarOdometry2D on the MI355X path.
//
// Same struct name, Options and methods as the reference's include/lama/lidar_odometry_2d.h:45-80; update() /
// updateMaps() follow src/lidar_odometry_2d.cpp:60-200: scan-to-map Gauss-Newton (lama_hip_match_solve), then -- after
// 0.1 m / 0.5 rad of motion -- the map update on the device with the ProbabilisticOccupancyMap cell policy
// (cfg.occupancy_policy = 1: float log-odds, src/sdm/probabilistic_occupancy_map.cpp:82-107) and LidarOdometry2D's ray rule
// (the last metre before the hit), the dynamic distance map (max distance 1 m), and the transient-map step that deletes
// every patch whose box does not meet the expanded box of the scan (lama_hip_pf_patch_ids / _delete_patches).
// Differences: `distance_map` / `occupancy_map` are not host objects -- use downloadDistanceMap / downloadOccupancyMap
// (sdm::HostMap, occupancy cells are 4-byte float log-odds); Options gains gpu_device.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

#include "pose2d.h"
#include "sdm_io.h"

struct lama_hip_ctx;

namespace lama {

struct HipEngine;

struct LidarOdometry2D {
    bool has_first_scan = false;
    Pose2D odom;
    Pose2D map_update_odom;

    struct Options {
        Options() {}
        double resolution = 0.05;       // resolution of the maps
        uint32_t max_iter = 100;        // maximum number of optimiser iterations
        int32_t gpu_device = 0;         // addition
    };

    explicit LidarOdometry2D(const Options& options = Options());
    virtual ~LidarOdometry2D();
    LidarOdometry2D(const LidarOdometry2D&) = delete;
    LidarOdometry2D& operator=(const LidarOdometry2D&) = delete;

    bool update(const PointCloudXYZ::Ptr& surface, double timestamp);
    void updateMaps(const PointCloudXYZ::Ptr& surface);

    bool downloadDistanceMap(sdm::HostMap& m) const;
    bool downloadOccupancyMap(sdm::HostMap& m) const;      // kind = kProbabilisticOccupancyMap: 4-byte float log-odds cells
    uint32_t getLastIterations() const { return last_iterations_; }
    uint32_t getLastDeletedPatches() const { return last_deleted_; }
    lama_hip_ctx* deviceContext() const { return ctx_; }
    const HipEngine* engine() const { return eng_.get(); }

private:
    void fail(int32_t rc, const char* what) const;
    Options opt_;
    std::shared_ptr<HipEngine> eng_;
    lama_hip_ctx* ctx_ = nullptr;
    bool device_initialised_ = false;
    uint32_t last_iterations_ = 0, last_deleted_ = 0;
};

} // namespace lama
