// lama/sdm_maps.h -- read-only host views of maps downloaded from the device, with the const query API of the
// reference's map classes, so that consumer code written against `const lama::FrequencyOccupancyMap*` /
// `const lama::DynamicDistanceMap*` (iris_lama_ros builds its nav_msgs/OccupancyGrid and distance images from
// getOccupancyMap() / getDistanceMap() this way) keeps compiling:
//   lama::Map                     include/lama/sdm/map.h:109-296   w2m / m2w / bounds / visit_all_cells / visit_all_patches /
//                                                                  resolution / patch_length / patches / memory / write
//   lama::FrequencyOccupancyMap   src/sdm/frequency_occupancy_map.cpp:38-45,113-172   isFree / isOccupied / isUnknown / getProbability
//   lama::ProbabilisticOccupancyMap src/sdm/probabilistic_occupancy_map.cpp:38-46,126-175   (LidarOdometry2D's log-odds map)
//   lama::DynamicDistanceMap      src/sdm/dynamic_distance_map.cpp:66-91,140-158   distance(cell) / distance(point, gradient) / maxDistance
// The views own a snapshot (lama::sdm::HostMap: the reference's record formats + Container masks); they are not the
// live maps -- those stay in HBM.  Two classes can also be WRITTEN on the host, because consumers of lama::Loc2D fill them cell by
// cell (iris_lama_ros' loc2d_ros): lama::SimpleOccupancyMap (src/sdm/simple_occupancy_map.cpp, a plain host map) and a
// lama::DynamicDistanceMap created from (resolution, patch_size), whose addObstacle() / update() are forwarded to the device
// map of the Loc2D that owns it.  Header only.  The reference's include paths lama/sdm/<class>.h forward here.
#pragma once

#include <mutex>
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <vector>
#include <string>
#include <unordered_map>
#include <utility>

#include "sdm_io.h"
#include "types.h"

struct lama_hip_ctx;

namespace lama {

struct HipEngine;

class Map {
public:
    // an empty map (the reference's Map(resolution, cell_size, patch_size, is3d), src/sdm/map.cpp:42-59; 2-D only here)
    Map(double res, sdm::MapKind kind, uint32_t patch_size = 32) : Map(emptyHost(res, kind, patch_size)) {}
    explicit Map(sdm::HostMap m) : host_(std::move(m))
    {
        resolution = host_.resolution;
        scale = 1.0 / resolution;
        patch_length = host_.patch_length;
        patch_volume = patch_length * patch_length;
        log2dim_ = 0;
        while ((1u << log2dim_) < patch_length) ++log2dim_;
        off_ = double(UNIVERSAL_CONSTANT >> 1) * double(patch_length);                 // src/sdm/map.cpp:42-59
        index_.reserve(host_.ids.size() * 2);
        for (size_t i = 0; i < host_.ids.size(); ++i) index_[host_.ids[i]] = i;
    }
    virtual ~Map() {}

    static constexpr uint64_t UNIVERSAL_CONSTANT = 2642244;                            // include/lama/sdm/map.h:88
    double resolution = 0.05, scale = 20.0;
    uint32_t patch_length = 32, patch_volume = 1024;

    // include/lama/sdm/map.h:125-126, 137-138, 147-148
    Vector3ui w2m(const Vector3d& p) const
    {
        return Vector3ui((uint32_t)(scale * p[0] + off_ + 0.5), (uint32_t)(scale * p[1] + off_ + 0.5), (uint32_t)(scale * p[2] + off_ + 0.5));
    }
    Vector3d w2m_nocast(const Vector3d& p) const { return Vector3d(scale * p[0] + off_, scale * p[1] + off_, scale * p[2] + off_); }
    Vector3d m2w(const Vector3ui& c) const
    {
        // tf_inv_ = (Translation(off) * Scaling(scale)).inverse(): linear = (s*s) * (1 / ((s*s)*s)), translation = -(linear * off)
        const double l = (scale * scale) * (1.0 / ((scale * scale) * scale)), t = -(l * off_);
        return Vector3d(l * (double)c(0) + t, l * (double)c(1) + t, l * (double)c(2) + t);
    }
    // src/sdm/map.cpp:139-157 (cells) and include/lama/sdm/map.h:221-225 (world): extent of the allocated patches
    void bounds(Vector3ui& min, Vector3ui& max) const
    {
        sync();
        min = Vector3ui(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        max = Vector3ui(0, 0, 0);
        for (uint64_t id : host_.ids) {
            const Vector3ui a = p2m(id);
            for (int k = 0; k < 3; ++k) { min[k] = std::min(min(k), a(k)); max[k] = std::max(max(k), a(k)); }
        }
        for (int k = 0; k < 3; ++k) max[k] += patch_length;
    }
    void bounds(Vector3d& min, Vector3d& max) const
    {
        Vector3ui a, b;
        bounds(a, b);
        min = m2w(a); max = m2w(b);
    }
    // src/sdm/map.cpp:352-367: every cell whose Container mask bit is on / the anchor of every patch
    template <typename F>
    void visit_all_cells(F&& walker) const
    {
        sync();
        for (size_t i = 0; i < host_.ids.size(); ++i) {
            const Vector3ui anchor = p2m(host_.ids[i]);
            const uint64_t* mask = &host_.masks[i * (patch_volume / 64)];
            for (uint32_t c = 0; c < patch_volume; ++c)
                if ((mask[c >> 6] >> (c & 63)) & 1ull)
                    walker(Vector3ui(anchor(0) + (c & (patch_length - 1)), anchor(1) + (c >> log2dim_), anchor(2)));
        }
    }
    template <typename F>
    void visit_all_patches(F&& walker) const { sync(); for (uint64_t id : host_.ids) walker(p2m(id)); }

    size_t patches() const { sync(); return host_.ids.size(); }                          // number of allocated patches
    size_t memory() const { sync(); return host_.cells.size() + host_.masks.size() * sizeof(uint64_t); }
    bool write(const std::string& filename) const { sync(); return sdm::write(host_, filename); }   // the reference's .sdm format
    const sdm::HostMap& snapshot() const { sync(); return host_; }

protected:
    // Every accessor of the host copy goes through sync() first: a map that mirrors a live device map (DynamicDistanceMap bound to
    // a Loc2D) refreshes its copy there when the device map has changed since; plain snapshots have nothing to do.
    virtual void sync() const {}
    static sdm::HostMap emptyHost(double res, sdm::MapKind kind, uint32_t patch_size)
    {
        sdm::HostMap m;
        m.kind = kind; m.resolution = res;
        m.patch_length = 1;
        while (m.patch_length * 2 <= patch_size) m.patch_length *= 2;          // 1 << int(log2(patch_size)), src/sdm/map.cpp:45
        return m;
    }
    void resetHost(sdm::HostMap m) const        // replace the snapshot (a live map downloads a fresh one)
    {
        host_ = std::move(m);
        index_.clear();
        for (size_t i = 0; i < host_.ids.size(); ++i) index_[host_.ids[i]] = i;
    }
    // non-const Map::get (src/sdm/map.cpp:371-412): allocates the (zeroed) patch, turns the cell's mask bit on
    uint8_t* getOrCreate(const Vector3ui& c)
    {
        const uint64_t id = uint64_t(c(0) >> log2dim_) * UNIVERSAL_CONSTANT + uint64_t(c(1) >> log2dim_);
        auto it = index_.find(id);
        if (it == index_.end()) {
            it = index_.emplace(id, host_.ids.size()).first;
            host_.ids.push_back(id);
            host_.cells.resize(host_.cells.size() + (size_t)patch_volume * host_.cellSize(), 0);
            host_.masks.resize(host_.masks.size() + patch_volume / 64, 0);
        }
        const uint32_t m = patch_length - 1;
        const uint32_t cell = (c(0) & m) | ((c(1) & m) << log2dim_);
        host_.masks[it->second * (patch_volume / 64) + (cell >> 6)] |= 1ull << (cell & 63);
        return &host_.cells[(it->second * (size_t)patch_volume + cell) * host_.cellSize()];
    }
    // const Map::get (src/sdm/map.cpp:414-455): nullptr for an absent patch or a cell whose mask bit is off
    const uint8_t* get(const Vector3ui& c) const
    {
        sync();
        const uint64_t id = uint64_t(c(0) >> log2dim_) * UNIVERSAL_CONSTANT + uint64_t(c(1) >> log2dim_);   // m2p, map.h:153-161
        const auto it = index_.find(id);
        if (it == index_.end()) return nullptr;
        const uint32_t m = patch_length - 1;
        const uint32_t cell = (c(0) & m) | ((c(1) & m) << log2dim_);                                            // m2c, map.h:182-189
        if (!((host_.masks[it->second * (patch_volume / 64) + (cell >> 6)] >> (cell & 63)) & 1ull)) return nullptr;
        return &host_.cells[(it->second * (size_t)patch_volume + cell) * host_.cellSize()];
    }
    Vector3ui p2m(uint64_t id) const                                                                            // map.h:166-177
    {
        return Vector3ui(uint32_t((id / UNIVERSAL_CONSTANT) << log2dim_), uint32_t((id % UNIVERSAL_CONSTANT) << log2dim_), 0);
    }

    mutable sdm::HostMap host_;                              // mutable: sync() of a live map replaces the copy behind const accessors
    mutable std::unordered_map<uint64_t, size_t> index_;
    uint32_t log2dim_ = 5;
    double off_ = 0.0;
};

class OccupancyMap : public Map {                     // include/lama/sdm/occupancy_map.h:66-76 (the const part)
public:
    using Map::Map;
    virtual bool isFree(const Vector3ui& coordinates) const = 0;
    virtual bool isOccupied(const Vector3ui& coordinates) const = 0;
    virtual bool isUnknown(const Vector3ui& coordinates) const = 0;
    virtual double getProbability(const Vector3ui& coordinates) const = 0;
    bool isFree(const Vector3d& p) const { return isFree(w2m(p)); }
    bool isOccupied(const Vector3d& p) const { return isOccupied(w2m(p)); }
    bool isUnknown(const Vector3d& p) const { return isUnknown(w2m(p)); }
    double getProbability(const Vector3d& p) const { return getProbability(w2m(p)); }
};

class FrequencyOccupancyMap : public OccupancyMap {
public:
    struct frequency { uint16_t occupied, visited; };                                   // frequency_occupancy_map.h:43-46
    explicit FrequencyOccupancyMap(sdm::HostMap m) : OccupancyMap(std::move(m)) {}
    using OccupancyMap::isFree; using OccupancyMap::isOccupied; using OccupancyMap::isUnknown; using OccupancyMap::getProbability;
    bool isFree(const Vector3ui& c) const override { frequency f; return cell(c, f) && prob(f) < 0.25; }          // :119-125
    bool isOccupied(const Vector3ui& c) const override { frequency f; return cell(c, f) && prob(f) > 0.25; }      // :132-138
    bool isUnknown(const Vector3ui& c) const override { frequency f; return !cell(c, f) || f.visited == 0; }      // :145-151
    double getProbability(const Vector3ui& c) const override { frequency f; return cell(c, f) ? prob(f) : 0.25; } // :166-172
    bool counters(const Vector3ui& c, frequency& f) const { return cell(c, f); }

private:
    static double prob(const frequency& f) { return f.visited == 0 ? 0.25 : ((double)f.occupied) / ((double)f.visited); }   // :38-45
    bool cell(const Vector3ui& c, frequency& f) const
    {
        const uint8_t* p = get(c);
        if (!p) return false;
        std::memcpy(&f, p, sizeof(f));
        return true;
    }
};

class ProbabilisticOccupancyMap : public OccupancyMap {
public:
    explicit ProbabilisticOccupancyMap(sdm::HostMap m) : OccupancyMap(std::move(m)) {}
    using OccupancyMap::isFree; using OccupancyMap::isOccupied; using OccupancyMap::isUnknown; using OccupancyMap::getProbability;
    bool isFree(const Vector3ui& c) const override { float l; return cell(c, l) && l < 0.0f; }                     // :131-137 (occ_thresh_ = 0)
    bool isOccupied(const Vector3ui& c) const override { float l; return cell(c, l) && l > 0.0f; }                 // :144-150
    bool isUnknown(const Vector3ui& c) const override { float l; return !cell(c, l) || l == 0.0f; }                // :157-163
    double getProbability(const Vector3ui& c) const override { float l; return cell(c, l) ? prob(l) : prob(0.0f); }   // :169-175
    bool logOdds(const Vector3ui& c, float& l) const { return cell(c, l); }

private:
    static float prob(const float& logods) { return 1.0 - 1.0 / (1.0 + std::exp(logods)); }                      // :38-41
    bool cell(const Vector3ui& c, float& l) const
    {
        const uint8_t* p = get(c);
        if (!p) return false;
        std::memcpy(&l, p, sizeof(l));
        return true;
    }
};

// include/lama/sdm/simple_occupancy_map.h + src/sdm/simple_occupancy_map.cpp: int8 tri-state (-1 free, 1 occupied, 0 unknown),
// a plain host map that can be written (Loc2D::occupancy_map is one)
class SimpleOccupancyMap : public OccupancyMap {
public:
    explicit SimpleOccupancyMap(double resolution, uint32_t patch_size = 32, bool /*is3d*/ = false)
        : OccupancyMap(resolution, sdm::kSimpleOccupancyMap, patch_size) {}
    explicit SimpleOccupancyMap(sdm::HostMap m) : OccupancyMap(std::move(m)) {}
    using OccupancyMap::isFree; using OccupancyMap::isOccupied; using OccupancyMap::isUnknown; using OccupancyMap::getProbability;
    bool setFree(const Vector3d& p) { return setFree(w2m(p)); }
    bool setOccupied(const Vector3d& p) { return setOccupied(w2m(p)); }
    bool setUnknown(const Vector3d& p) { return setUnknown(w2m(p)); }
    bool setFree(const Vector3ui& c) { return set(c, -1); }                             // :50-58 ("changed")
    bool setOccupied(const Vector3ui& c) { return set(c, 1); }                          // :65-73
    bool setUnknown(const Vector3ui& c) { return set(c, 0); }                           // :80-88
    bool isFree(const Vector3ui& c) const override { const uint8_t* p = get(c); return p && (int8_t)*p == -1; }       // :95-102
    bool isOccupied(const Vector3ui& c) const override { const uint8_t* p = get(c); return p && (int8_t)*p == 1; }    // :109-116
    bool isUnknown(const Vector3ui& c) const override { const uint8_t* p = get(c); return !p || (int8_t)*p == 0; }    // :123-129
    double getProbability(const Vector3ui& c) const override { return isFree(c) ? 0.0 : (isOccupied(c) ? 1.0 : 0.5); }   // :136-145
    bool empty() const { return host_.ids.empty(); }

private:
    bool set(const Vector3ui& c, int8_t v) { int8_t* cell = (int8_t*)getOrCreate(c); if (*cell == v) return false; *cell = v; return true; }
};

class DynamicDistanceMap : public Map {
public:
#pragma pack(push, 1)
    struct distance_t { int16_t obstacle[3]; uint16_t sqdist; bool valid_obstacle; bool is_queued; };   // dynamic_distance_map.h:48-53 (10 bytes)
#pragma pack(pop)
    explicit DynamicDistanceMap(sdm::HostMap m) : Map(std::move(m)) {}
    // The reference's constructor (src/sdm/dynamic_distance_map.cpp:36-47).  A map made this way is LIVE: it belongs to a
    // lama::Loc2D, which binds addObstacle() / update() to its device map (bindWriter); queries after an update() read a
    // snapshot that is downloaded on first use.  Unbound, the mutating members throw.
    explicit DynamicDistanceMap(double resolution, uint32_t patch_size = 32, bool /*is3d*/ = false)
        : Map(resolution, sdm::kDistanceMap, patch_size) {}
    struct Writer {
        std::function<uint32_t(std::vector<uint32_t>& /*x,y pairs*/, double /*max distance*/)> apply;   // addObstacle list + update()
        std::function<bool(sdm::HostMap&)> download;
        std::function<void(const sdm::HostMap&)> upload;           // Map::read for a device map: replace it by these patches
    };
    void bindWriter(Writer w) { writer_ = std::move(w); }
    void setMaxDistance(double distance)                                                                    // :149-153
    {
        const uint32_t r = (uint32_t)std::ceil(distance * scale);
        host_.max_sqdist = r * r;
        max_distance_ = distance;
    }
    // Map::read (src/sdm/map.cpp:533-575) for the live device map: the patches of `m` / of the `.sdm` file REPLACE the map on the
    // device (lama_hip_pf_upload_map) -- a distance map built once (by the reference, by an earlier run through write()) is loaded
    // instead of rebuilt obstacle by obstacle.  Resolution, patch size and the squared radius must be this map's.
    void load(const sdm::HostMap& m)
    {
        if (!writer_.upload) throw std::logic_error("lama::DynamicDistanceMap::load: this map is a host snapshot of a device map");
        if (m.kind != sdm::kDistanceMap || m.patch_length != patch_length || std::abs(m.resolution - resolution) > 1e-6 * resolution  /* the file stores a float */)
            throw std::invalid_argument("lama::DynamicDistanceMap::load: not a distance map of this resolution / patch size");
        if (m.max_sqdist != host_.max_sqdist) throw std::invalid_argument("lama::DynamicDistanceMap::load: the map was built with another l2_max");
        pending_.clear();
        writer_.upload(m);
        stale_.store(true, std::memory_order_release);
    }
    bool read(const std::string& filename)
    {
        sdm::HostMap m;
        if (!sdm::read(m, filename)) return false;
        load(m);
        return true;
    }
    void addObstacle(const Vector3ui& c) { pending_.push_back(c(0)); pending_.push_back(c(1)); }             // :212-226 (runs in update())
    void addObstacle(const Vector3d& p) { addObstacle(w2m(p)); }
    bool hasPending() const { return !pending_.empty(); }
    uint32_t update()                                                                                        // :160-197, on the device
    {
        if (pending_.empty()) return 0;
        if (!writer_.apply) throw std::logic_error("lama::DynamicDistanceMap::update: this map is a host snapshot of a device map");
        const uint32_t n = writer_.apply(pending_, max_distance_);      // cells the brushfire processed in THIS update (:196)
        pending_.clear();
        stale_.store(true, std::memory_order_release);
        return n;
    }

    // Which live device map this snapshot was taken from (set by Slam2D / PFSlam2D::getDistanceMap()): lama::MatchSurface2D
    // evaluates against THAT map on the GPU, not against the host copy.
    struct DeviceBinding {
        std::shared_ptr<HipEngine> engine;
        ::lama_hip_ctx* ctx = nullptr;
        uint32_t particle = 0;
    };
    void bindDevice(std::shared_ptr<HipEngine> e, ::lama_hip_ctx* ctx, uint32_t particle) { dev_.engine = std::move(e); dev_.ctx = ctx; dev_.particle = particle; }
    const DeviceBinding& device() const { return dev_; }

    double maxDistance() const { return std::sqrt((double)host_.max_sqdist) * resolution; }                      // :155-158
    double maxDistanceOption() const { return max_distance_; }
    uint32_t maxSqDist() const { return host_.max_sqdist; }       // the configured squared radius in cells (no refresh)
    // :140-147
    double distance(const Vector3ui& coordinates) const
    {
        distance_t d;
        if (!cell(coordinates, d) || !d.valid_obstacle) return maxDistance();
        return std::sqrt((double)d.sqdist) * resolution;
    }
    // :66-91 (2-D branch): bilinear value and gradient
    double distance(const Vector3d& coordinates, Vector3d* gradient = nullptr) const
    {
        const Vector3d m = w2m_nocast(coordinates);
        const uint32_t dx = (uint32_t)m[0], dy = (uint32_t)m[1];
        const double mu0 = m[0] - (double)dx, mu1 = m[1] - (double)dy, muinv0 = 1.0 - mu0, muinv1 = 1.0 - mu1;
        const double v0 = distance(Vector3ui(dx, dy, 0)), v1 = distance(Vector3ui(dx + 1, dy, 0));
        const double v2 = distance(Vector3ui(dx, dy + 1, 0)), v3 = distance(Vector3ui(dx + 1, dy + 1, 0));
        if (gradient) {
            (*gradient)[0] = -((v0 - v1) * muinv1 + (v2 - v3) * mu1) * scale;
            (*gradient)[1] = -((v0 - v2) * muinv0 + (v1 - v3) * mu0) * scale;
            (*gradient)[2] = 0.0;
        }
        return v0 * muinv0 * muinv1 + v1 * muinv1 * mu0 + v2 * muinv0 * mu1 + v3 * mu0 * mu1;
    }
    bool cell(const Vector3ui& c, distance_t& d) const
    {
        const uint8_t* p = get(c);
        if (!p) return false;
        std::memcpy(&d, p, sizeof(d));
        return true;
    }

protected:
    // live map: fetch the device map once after an update(), whichever accessor comes first (distance(), write(), snapshot(),
    // bounds(), the visitors ...); concurrent readers are serialised on the refresh
    void sync() const override
    {
        if (!stale_.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lock(sync_mutex_);
        if (!stale_.load(std::memory_order_relaxed)) return;
        if (writer_.download) {
            sdm::HostMap m;
            if (writer_.download(m)) { m.resolution = resolution; resetHost(std::move(m)); }
        }
        stale_.store(false, std::memory_order_release);
    }

private:
    DeviceBinding dev_;
    Writer writer_;
    std::vector<uint32_t> pending_;
    double max_distance_ = 0.5;
    mutable std::atomic<bool> stale_{false};
    mutable std::mutex sync_mutex_;
};

} // namespace lama
