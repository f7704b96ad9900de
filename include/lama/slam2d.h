// lama/slam2d.h -- host-side lama::Slam2D (online SLAM: one pose, one map pair) on the MI355X path.
//
// Same class name, Options fields and public methods as the reference's include/lama/slam2d.h:57-188; the body of
// update() follows src/slam2d.cpp:143-198.  Scan matching (MatchSurface2D + Solve, :172-175) and updateMaps()
// (:247-321) run on the device through the C-ABI of include/lama_hip.h with a one-particle context -- the same
// kernels as PFSlam2D.  Differences: getOccupancyMap()/getDistanceMap() become downloadOccupancyMap()/
// downloadDistanceMap(); strategy "lm" runs Levenberg-Marquardt on the device (cfg.solver_strategy);
// there is no CPU fallback.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "pose2d.h"
#include "sdm_io.h"
#include "sdm_maps.h"

struct lama_hip_ctx;

namespace lama {

struct HipEngine;

class Slam2D {
public:
    // include/lama/slam2d.h:59-88: per-update timing / memory record, filled when Options::create_summary is set
    struct Summary {
        DynamicArray<double> timestamp, time, time_solving, time_mapping, memory;
        std::string report() const;
    };
    Summary* summary = nullptr;

    struct Options {
        Options() {}
        double trans_thresh = 0.5;
        double rot_thresh = 0.5;
        double l2_max = 0.5;
        double truncated_ray = 0.0;
        double truncated_range = 0.0;
        double resolution = 0.05;
        uint32_t patch_size = 32;
        uint32_t max_iter = 100;
        std::string strategy = "gn";
        bool use_compression = false;
        uint32_t cache_size = 100;
        std::string calgorithm = "lz4";
        bool transient_map = false;
        bool create_summary = false;
        // ---- additions ----
        int32_t gpu_device = 0;
        // device map storage, 0 = defaults (see PFSlam2D::Options: arenas grow on demand, the window is fixed at creation)
        uint32_t window_patches = 0, dm_patch_capacity = 0, occ_patch_capacity = 0, queue_capacity = 0;
    };

    explicit Slam2D(const Options& options = Options());
    virtual ~Slam2D();

    bool enoughMotion(const Pose2D& odometry);
    bool update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp);

    uint64_t getMemoryUsage() const;
    uint32_t getNumberOfProcessedCells() const { return number_of_proccessed_cells_; }
    uint32_t getLastIterations() const { return last_iterations_; }
    uint32_t getLastDeletedPatches() const { return last_deleted_; }   // transient_map: patches removed by the last update

    void setPose(const Pose2D& pose) { pose_ = pose; }
    Pose2D getPose() const { return pose_; }

    bool downloadDistanceMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const;
    bool downloadOccupancyMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const;
    // the same as lama::sdm::HostMap, ready for sdm::write (the reference's .sdm file) / sdm::export_to_png
    bool downloadDistanceMap(sdm::HostMap& m) const
    {
        m.kind = sdm::kDistanceMap; m.resolution = resolution_;
        const uint32_t r = (uint32_t)std::ceil(l2_max_ * (1.0 / resolution_));      // DynamicDistanceMap::setMaxDistance :149-153
        m.max_sqdist = r * r;
        return downloadDistanceMap(m.ids, m.cells, m.masks);
    }
    bool downloadOccupancyMap(sdm::HostMap& m) const
    {
        m.kind = sdm::kFrequencyOccupancyMap; m.resolution = resolution_;
        return downloadOccupancyMap(m.ids, m.cells, m.masks);
    }
    // The reference's accessors (include/lama/slam2d.h:150-156) as host snapshots with the const query API (lama/sdm_maps.h);
    // downloaded on first use after an update, kept until the next update() call; nullptr before the first scan.
    const FrequencyOccupancyMap* getOccupancyMap() const
    {
        if (!occ_view_) {
            sdm::HostMap m;
            if (!downloadOccupancyMap(m)) return nullptr;
            occ_view_.reset(new FrequencyOccupancyMap(std::move(m)));
        }
        return occ_view_.get();
    }
    const DynamicDistanceMap* getDistanceMap() const
    {
        if (!dm_view_) {
            sdm::HostMap m;
            if (!downloadDistanceMap(m)) return nullptr;
            dm_view_.reset(new DynamicDistanceMap(std::move(m)));
            dm_view_->bindDevice(eng_, ctx_, 0u);
        }
        return dm_view_.get();
    }
    // include/lama/slam2d.h:157-161
    void saveOccImage(const std::string& name) const { const FrequencyOccupancyMap* m = getOccupancyMap(); if (m) sdm::export_to_png(m->snapshot(), name); }
    void saveDistImage(const std::string& name) const { const DynamicDistanceMap* m = getDistanceMap(); if (m) sdm::export_to_png(m->snapshot(), name); }
    // src/slam2d.cpp getMemoryUsage(occmem, dmmem): patch payloads in the reference's record sizes
    uint64_t getMemoryUsage(uint64_t& occmem, uint64_t& dmmem) const
    {
        const FrequencyOccupancyMap* o = getOccupancyMap();
        const DynamicDistanceMap* d = getDistanceMap();
        occmem = o ? (uint64_t)o->patches() * 4096ull : 0; dmmem = d ? (uint64_t)d->patches() * 10240ull : 0;
        return occmem + dmmem;
    }
    lama_hip_ctx* deviceContext() const { return ctx_; }
    const HipEngine* engine() const { return eng_.get(); }

private:
    void fail(int32_t rc, const char* what) const;
    mutable std::unique_ptr<FrequencyOccupancyMap> occ_view_;
    mutable std::unique_ptr<DynamicDistanceMap> dm_view_;
    std::shared_ptr<HipEngine> eng_;
    lama_hip_ctx* ctx_ = nullptr;
    Pose2D odom_, pose_;
    double trans_thresh_, rot_thresh_;
    double resolution_ = 0.05, l2_max_ = 0.5, truncated_range_ = 0.0;
    bool transient_map_ = false;
    uint32_t last_deleted_ = 0;
    bool has_first_scan = false;
    uint32_t number_of_proccessed_cells_ = 0;
    uint32_t last_iterations_ = 0;
};

} // namespace lama
