// lama/pf_slam2d.h -- host-side lama::PFSlam2D whose per-particle work runs on the MI355X.
//
// Same class name, Options fields, and public methods as the reference's include/lama/pf_slam2d.h:49-277
// (update / getPose / getBestParticleIdx / getParticles / getNeff / setPrior / getTimestamps / summary ...), so
// that a consumer such as iris_lama_ros keeps compiling.  What changed underneath:
//   * the two per-particle regions of update() (src/pf_slam2d.cpp:254-266, 292-302), the first-scan block
//     (:185-228) and the particle copies of resample() (:558-574) are calls into the C-ABI of
//     include/lama_hip.h (resolved at run time from liblama_hip.so; there is NO CPU fallback -- the constructor
//     throws std::runtime_error when the device library or a GPU is missing);
//   * what must replay the host RNG stays on the host, statement for statement: drawFromMotion (:365-391),
//     normalize (:511-535), systematic-resampling indices (:537-556), std::mt19937 seeded like random.cpp;
//   * the particles' maps live in HBM: Particle::dm / Particle::occ are not host members any more.  getOccupancyMap() /
//     getDistanceMap() (the best particle's, as in the reference) and getParticleOccupancyMap(i) / getParticleDistanceMap(i)
//     (any particle's: what Particle::occ / Particle::dm gave) return host SNAPSHOTS with the const map API of lama/sdm_maps.h;
//     download*Map() return the same patches as raw arrays in the reference's record formats;
//   * Options gains `gpu_device`, `shard_rank`, `shard_world`, ..., `gpus` at the END (aggregate/default use is unchanged).
//     With gpus > 1 ONE object drives several GPUs from C++ (a host thread per device, peer copies for cross-shard clones):
//     update() stays the whole step, nothing else changes for the consumer.
//     With shard_world > 1 one process per GPU owns a contiguous block of the particle pool and the step-wise
//     API (updateBegin / planResample / applyResample / updateMaps) is driven by the caller with an
//     all-gather of the per-particle log-likelihoods in between (iris_lama_amd/distributed.py).
#pragma once

#include <cmath>
#include <cstdint>
#include <deque>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "pose2d.h"
#include "sdm_io.h"
#include "sdm_maps.h"

struct lama_hip_ctx;

namespace lama {

struct HipEngine;   // function table of include/lama_hip.h resolved with dlopen

class PFSlam2D {
public:
    struct Particle {
        double weight = 0.0;
        double normalized_weight = 0.0;
        double weight_sum = 0.0;
        Pose2D pose;
        DynamicArray<Pose2D> poses;   // history (kept for locally owned particles, single-shard runs)
    };

    struct Summary {
        DynamicArray<double> timestamp, time, time_solving, time_normalizing, time_resampling, time_mapping, memory;
        std::string report() const;
    };
    Summary* summary = nullptr;

    struct Options {
        Options() {}
        uint32_t particles = 30;          // (no default in the reference)
        double srr = 0.1;
        double str = 0.2;
        double stt = 0.1;
        double srt = 0.2;
        double meas_sigma = 0.05;
        double meas_sigma_gain = 3;
        double trans_thresh = 0.5;
        double rot_thresh = 0.5;
        double l2_max = 0.5;
        double truncated_ray = 0.0;
        double truncated_range = 0.0;
        double resolution = 0.05;
        uint32_t patch_size = 32;
        uint32_t max_iter = 100;
        std::string strategy = "gn";       // ignored by scanMatch in the reference too (always GN, :423-427)
        int32_t threads = -1;              // unused: the particle loop runs on the GPU
        uint32_t seed = 0;
        bool use_compression = false;      // unsupported on the device (288 GB HBM); must stay false
        uint32_t cache_size = 100;
        std::string calgorithm = "lz4";
        bool create_summary = false;
        // ---- additions (at the end) ----
        int32_t gpu_device = 0;
        uint32_t shard_rank = 0;
        uint32_t shard_world = 1;
        bool profile = false;              // bracket kernels with hipEvents (see lama_hip_get_counters)
        uint32_t brushfire_mode = 0;       // 0 exact (default, bit-identical to the reference), 1 level-synchronous canonical tie rule (opt-in, NOT bit-identical; lama_hip.h)
        // Device map storage (0 = the device library's defaults).  The reference's maps are unbounded (src/sdm/map.cpp:400-411);
        // here every particle's patch regions GROW with its own maps (one particle at a time, inside pooled planes that grow chunk by
        // chunk; the two capacities below are the smallest region a particle gets) and the map window (window_patches x 1.6 m at
        // 0.05 m: starts at 128 = 204.8 m) FOLLOWS the robot and GROWS with the mapped area up to 1016 patches = 1.6 km; only a map
        // wider than that -- or one whose window directories no longer fit the device memory -- is an error.
        uint32_t window_patches = 0;       // side of the square map window in patches (multiple of 8)
        uint32_t dm_patch_capacity = 0;    // initial distance-map patches per particle
        uint32_t occ_patch_capacity = 0;   // initial occupancy patches per particle
        uint32_t queue_capacity = 0;       // brushfire queue entries per particle
        // Several GPUs behind ONE object (shard_world must stay 1): the pool is split in `gpus` contiguous blocks (particle i on
        // shard floor(i * gpus / P)), shard r runs on HIP device (gpu_device + r) mod (visible devices), each driven by
        // its own host thread.  update() is then the whole sharded step -- the reference's two parallel regions
        // (src/pf_slam2d.cpp:254-266, 292-302) run on all devices at once, the log-likelihoods are gathered on the host (they are
        // host results of the scan match already), normalisation / Neff / resampling indices are computed once per shard from the
        // same random stream, and a clone whose source lives on another shard is exported, copied GPU to GPU
        // (hipMemcpyPeerAsync: xGMI) and imported.  Results are bit-identical for every value of `gpus`.
        int32_t gpus = 1;
    };

    explicit PFSlam2D(const Options& options = Options());
    virtual ~PFSlam2D();

    const Options& getOptions() const { return options_; }

    // Whole update, single shard (shard_world == 1): same contract as the reference (returns false when the
    // motion gate did not open).  Throws std::runtime_error on a device error.
    bool update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp);

    size_t getBestParticleIdx() const;
    Pose2D getPose() const;
    const std::deque<double>& getTimestamps() const { return timestamps_; }
    // all P particles (weights are global; poses are valid for locally owned particles, see ownsParticle)
    const std::vector<Particle>& getParticles() const { return particles_; }
    double getNeff() const { return neff_; }
    void setPrior(const Pose2D& prior) { pose_ = prior; }
    uint64_t getMemoryUsage() const;
    // src/pf_slam2d.cpp:164-176: P times the memory of particle 0's maps (the reference indexes particle 0 in its loop); patch
    // payloads in the reference's record sizes.  Sharded: the first locally owned particle stands in when particle 0 lives elsewhere.
    uint64_t getMemoryUsage(uint64_t& occmem, uint64_t& dmmem) const;
    // include/lama/pf_slam2d.h:227: PNG of the best particle's occupancy map (sdm::export_to_png on its snapshot)
    void saveOccImage(const std::string& name) const
    {
        const FrequencyOccupancyMap* m = getOccupancyMap();
        if (m) sdm::export_to_png(m->snapshot(), name);
    }

    // Best particle's maps in the reference's on-host record formats; patch ids are Map::m2p indices.
    // cells: 10240 B (distance_t) or 4096 B (frequency) per patch, masks: 16 x uint64 per patch.
    bool downloadDistanceMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const;
    bool downloadOccupancyMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const;
    // Any particle's maps (the reference's public Particle::dm / Particle::occ, include/lama/pf_slam2d.h:83-84): the same arrays
    // for particle i, and host snapshots with the const map API (valid until they are released; they do not follow later
    // updates).  false / nullptr before the first scan, for i >= particles, or when particle i lives on another PROCESS' shard
    // (a gpus > 1 object reaches all of its shards).
    bool downloadParticleDistanceMap(size_t i, std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const;
    bool downloadParticleOccupancyMap(size_t i, std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const;
    std::shared_ptr<const FrequencyOccupancyMap> getParticleOccupancyMap(size_t i) const;
    std::shared_ptr<const DynamicDistanceMap> getParticleDistanceMap(size_t i) const;
    // the same as lama::sdm::HostMap, ready for sdm::write (the reference's .sdm file) / sdm::export_to_png
    bool downloadDistanceMap(sdm::HostMap& m) const
    {
        m.kind = sdm::kDistanceMap; m.resolution = options_.resolution;
        const uint32_t r = (uint32_t)std::ceil(options_.l2_max * (1.0 / options_.resolution));    // DynamicDistanceMap::setMaxDistance :149-153
        m.max_sqdist = r * r;
        return downloadDistanceMap(m.ids, m.cells, m.masks);
    }
    bool downloadOccupancyMap(sdm::HostMap& m) const
    {
        m.kind = sdm::kFrequencyOccupancyMap; m.resolution = options_.resolution;
        return downloadOccupancyMap(m.ids, m.cells, m.masks);
    }

    // The reference's accessors (include/lama/pf_slam2d.h:211-225): the best particle's occupancy / distance map, with the
    // const query API consumers use (lama/sdm_maps.h: bounds, visit_all_cells, isFree / isOccupied / getProbability,
    // distance, ...).  The maps live in HBM: what is returned is a host SNAPSHOT, downloaded on first use after an update and
    // kept until the next update() call.  nullptr before the first scan or when the best particle lives on another shard.
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
            const uint32_t b = (uint32_t)getBestParticleIdx();
            const PFSlam2D* o = ownerOf(b);
            dm_view_->bindDevice(o->eng_, o->ctx_, b - o->lo_);
        }
        return dm_view_.get();
    }

    // ------------------------------------------------------------------ step-wise API (sharded operation)
    enum Phase { kNoUpdate = 0, kFirstScan = 1, kMatched = 2 };
    uint32_t localBegin() const { return lo_; }
    uint32_t localEnd() const { return hi_; }
    bool ownsParticle(uint32_t i) const { return i >= lo_ && i < hi_; }
    // first scan, or predict + motion gate + scan match of the local shard; fills local log-likelihoods
    Phase updateBegin(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp);
    const std::vector<double>& localLogLik() const { return local_loglik_; }
    // add the (all-gathered) per-particle log-likelihoods of ALL P particles, normalise, compute Neff;
    // returns true and fills sample_idx (size P) when resampling is due (Neff < P/2)
    bool planResample(const double* all_loglik, std::vector<int32_t>& sample_idx);
    // apply sample_idx to the replicated weights and to the locally owned particles.  Slots whose source is
    // owned by another shard must afterwards be filled with importParticle().
    void applyResample(const std::vector<int32_t>& sample_idx);
    void updateMaps();                                      // region 2 on the local shard
    lama_hip_ctx* deviceContext() const;                    // for export/import of particles (multi-process sharding); gpus > 1: shard 0's
    const HipEngine* engine() const;
    uint32_t numResamples() const { return num_resamples_; }
    // gpus > 1: the shard objects (each a PFSlam2D with shard_rank = r, shard_world = gpus on its own device); empty otherwise
    size_t numShards() const;
    const PFSlam2D* shard(size_t r) const;
    // seconds the last update() spent in the exchange steps of a multi-GPU object: gathering the log-likelihoods, shipping
    // particles between devices (export + peer copy), importing them
    struct ExchangeTimes { double gather = 0, ship = 0, import_ = 0, local_copies = 0, phase_begin = 0, phase_maps = 0; uint64_t shipped_particles = 0, shipped_bytes = 0; };
    const ExchangeTimes& exchangeTimes() const { return xt_; }

    // Exposed for tests (host logic, identical formulas to src/pf_slam2d.cpp:365-391,511-556)
    void drawFromMotion(const Pose2D& delta, Pose2D& pose);
    void normalize();
    std::vector<int32_t> resampleIndices(double u01) const;

private:
    struct Group;                           // shards + their worker threads (Options::gpus > 1)
    std::unique_ptr<Group> group_;
    ExchangeTimes xt_;
    bool updateGroup(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp);
    void mirrorShards(bool poses_changed);
    const PFSlam2D* ownerOf(uint32_t i) const;
    void syncDevice();
    double normal(double stddev);
    void fail(int32_t rc, const char* what) const;
    void uploadLocalPoses();
    void scanToArrays(const PointCloudXYZ& s);

    Options options_;
    std::shared_ptr<HipEngine> eng_;
    lama_hip_ctx* ctx_ = nullptr;
    std::mt19937 gen_;
    std::vector<Particle> particles_;       // all P (weights replicated on every shard)
    uint32_t lo_ = 0, hi_ = 0;              // locally owned block [lo, hi)
    Pose2D odom_, pose_;
    double acc_trans_ = 0.0, acc_rot_ = 0.0;
    bool has_first_scan = false;
    double neff_ = 0.0;
    uint32_t num_resamples_ = 0;
    std::deque<double> timestamps_;
    double last_timestamp_ = 0.0;      // of the update() call in progress
    PointCloudXYZ::Ptr current_surface_;
    std::vector<double> pts_;               // n x 3
    double origin_[3] = {0, 0, 0};
    double quat_[4] = {1, 0, 0, 0};
    std::vector<double> local_loglik_;
    // summary bookkeeping
    double t_begin_ = 0, t_solve_ = 0;
    bool scan_resident_ = false;
    mutable std::unique_ptr<FrequencyOccupancyMap> occ_view_;     // snapshots handed out by getOccupancyMap / getDistanceMap
    mutable std::unique_ptr<DynamicDistanceMap> dm_view_;
    void dropMapViews() { occ_view_.reset(); dm_view_.reset(); }
};

} // namespace lama
