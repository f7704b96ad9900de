// slam2d.cpp -- host-side lama::Slam2D (include/lama/slam2d.h); orchestration of src/slam2d.cpp:143-198.
#include "lama/slam2d.h"

#include <chrono>

#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "hip_engine.hpp"
#include "transient_map.hpp"

namespace lama {

namespace {
void scan_arrays(const PointCloudXYZ& s, std::vector<double>& pts, double o[3], double q[4])
{
    pts.resize(s.points.size() * 3);
    for (size_t i = 0; i < s.points.size(); ++i) { pts[3 * i] = s.points[i].x(); pts[3 * i + 1] = s.points[i].y(); pts[3 * i + 2] = s.points[i].z(); }
    o[0] = s.sensor_origin_.x(); o[1] = s.sensor_origin_.y(); o[2] = s.sensor_origin_.z();
    q[0] = s.sensor_orientation_.w(); q[1] = s.sensor_orientation_.x(); q[2] = s.sensor_orientation_.y(); q[3] = s.sensor_orientation_.z();
}
} // namespace

Slam2D::Slam2D(const Options& o) : trans_thresh_(o.trans_thresh), rot_thresh_(o.rot_thresh), resolution_(o.resolution), l2_max_(o.l2_max)
{
    if (o.use_compression) throw std::runtime_error("lama::Slam2D: use_compression is not supported on the device path");
    transient_map_ = o.transient_map; truncated_range_ = o.truncated_range;
    if (o.create_summary) summary = new Summary;                 // src/slam2d.cpp:117-118
    eng_ = defaultEngine(o.l2_max, o.resolution);
    lama_hip_cfg cfg;
    eng_->default_cfg(&cfg);
    cfg.particles = 1;
    cfg.resolution = o.resolution; cfg.patch_size = o.patch_size; cfg.l2_max = o.l2_max; cfg.max_iter = o.max_iter;
    cfg.truncated_ray = o.truncated_ray; cfg.truncated_range = o.truncated_range; cfg.device = o.gpu_device;
    cfg.meas_sigma = 0.05;                       // unused by Slam2D (no likelihood)
    cfg.solver_strategy = o.strategy == "lm" ? 1u : 0u;           // makeStrategy, src/slam2d.cpp:226-233
    if (o.window_patches) cfg.window_patches = o.window_patches;
    if (o.dm_patch_capacity) cfg.dm_patch_capacity = o.dm_patch_capacity;
    if (o.occ_patch_capacity) cfg.occ_patch_capacity = o.occ_patch_capacity;
    if (o.queue_capacity) cfg.queue_capacity = o.queue_capacity;
    const int32_t rc = eng_->ctx_create(&cfg, &ctx_);
    if (rc != 0 || !ctx_) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "lama::Slam2D: lama_hip_ctx_create failed (status %d): no usable MI355X / HIP device; there is no CPU fallback", rc);
        throw std::runtime_error(msg);
    }
}

Slam2D::~Slam2D() { if (ctx_) eng_->ctx_destroy(ctx_); delete summary; }

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// src/slam2d.cpp:46-90 (same lines and buckets; plain loops instead of Eigen::Map)
std::string Slam2D::Summary::report() const
{
    auto stats = [](const DynamicArray<double>& v, double out[4]) {
        out[0] = out[1] = out[2] = out[3] = 0;
        if (v.empty()) return;
        double s = 0, mn = v[0], mx = v[0];
        for (double x : v) { s += x; mn = std::min(mn, x); mx = std::max(mx, x); }
        const double mean = s / v.size();
        double ss = 0;
        for (double x : v) ss += (x - mean) * (x - mean);
        out[0] = mean * 1e3; out[1] = (v.size() > 1 ? std::sqrt(ss / (v.size() - 1)) : 0.0) * 1e3; out[2] = mn * 1e3; out[3] = mx * 1e3;
    };
    double t[4], ts[4], tm[4];
    stats(time, t); stats(time_solving, ts); stats(time_mapping, tm);
    double span = 0, maxmem = 0;
    for (double x : time) span += x;
    for (double m : memory) maxmem = std::max(maxmem, m);
    const double stampdiff = timestamp.empty() ? 0.0 : timestamp.back() - timestamp.front();
    char buf[1536];
    std::snprintf(buf, sizeof(buf),
                  "\n LaMa Slam2D - Report\n ====================\n"
                  " Number of updates     %zu\n Max memory usage      %.2f MiB\n"
                  " Problem time span     %d minute(s) and %d second(s)\n Execution time span   %d minute(s) and %d second(s)\n"
                  " Execution frequency   %.2f Hz\n Realtime factor       %.2fx\n"
                  "\n Execution time (mean +- std [min, max]) in milliseconds\n"
                  " --------------------------------------------------------\n"
                  " Update          %f +- %f [%f, %f]\n   Optimization  %f +- %f [%f, %f]\n   Mapping       %f +- %f [%f, %f]\n",
                  time.size(), maxmem / 1024.0 / 1024.0, ((uint32_t)stampdiff) / 60, ((uint32_t)stampdiff) % 60,
                  ((uint32_t)span) / 60, ((uint32_t)span) % 60, time.empty() || span <= 0 ? 0.0 : 1.0 / (span / time.size()),
                  span > 0 ? stampdiff / span : 0.0, t[0], t[1], t[2], t[3], ts[0], ts[1], ts[2], ts[3], tm[0], tm[1], tm[2], tm[3]);
    return std::string(buf);
}

void Slam2D::fail(int32_t rc, const char* what) const
{
    char msg[512];
    std::snprintf(msg, sizeof(msg), "lama::Slam2D: %s failed (status %d): %s", what, rc, eng_->last_error(ctx_));
    throw std::runtime_error(msg);
}

bool Slam2D::enoughMotion(const Pose2D& odometry)                 // src/slam2d.cpp:129-141
{
    if (!has_first_scan) return true;
    Pose2D odelta = odom_ - odometry;
    if (odelta.xy().norm() <= trans_thresh_ && std::abs(odelta.rotation()) <= rot_thresh_) return false;
    return true;
}

bool Slam2D::update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp)
{
    if (!surface || surface->points.empty()) throw std::runtime_error("lama::Slam2D::update: empty scan");
    const double t_begin = now_s();
    occ_view_.reset(); dm_view_.reset();
    std::vector<double> pts;
    double o[3], q[4], p[4];
    scan_arrays(*surface, pts, o, q);
    const uint32_t n = (uint32_t)surface->points.size();
    lama_hip_counters c0, c1;

    if (!has_first_scan) {                                        // :147-160
        odom_ = odometry;
        pose_.state.toArray(p);
        int32_t rc = eng_->pf_init(ctx_, pts.data(), n, o, q, p); // updateMaps(surface) at pose_
        if (rc) fail(rc, "lama_hip_pf_init");
        if (eng_->get_counters(ctx_, &c1) == 0) number_of_proccessed_cells_ = (uint32_t)c1.bf_cells;
        if (transient_map_) {                                      // :322-379
            rc = transient::prune(eng_.get(), ctx_, *surface, pose_, resolution_, l2_max_, truncated_range_, 2.0, &last_deleted_);
            if (rc) fail(rc, "transient map");
        }
        if (summary) {                                             // :151-157
            const double el = now_s() - t_begin;
            summary->timestamp.push_back(timestamp); summary->time.push_back(el); summary->time_mapping.push_back(el);
            summary->memory.push_back((double)getMemoryUsage());
        }
        has_first_scan = true;
        return true;
    }
    // 1. predict from odometry                                    :163-173
    Pose2D odelta = odom_ - odometry;
    Pose2D ppose = pose_ + odelta;
    if (odelta.xy().norm() <= trans_thresh_ && std::abs(odelta.rotation()) <= rot_thresh_) return false;
    pose_ = ppose;
    odom_ = odometry;
    // 2. optimise                                                 :175-181
    const double t_solve = now_s();
    pose_.state.toArray(p);
    int32_t rc = eng_->pf_set_poses(ctx_, p);
    if (rc) fail(rc, "lama_hip_pf_set_poses");
    int32_t iters = 0;
    rc = eng_->pf_scan_match(ctx_, pts.data(), n, o, q, p, nullptr, &iters);
    if (rc) fail(rc, "lama_hip_pf_scan_match");
    pose_.state = SE2d::fromArray(p);
    last_iterations_ = (uint32_t)iters;
    if (summary) summary->time_solving.push_back(now_s() - t_solve);
    // 3. update maps                                              :184-186
    const double t_map = now_s();
    (void)eng_->get_counters(ctx_, &c0);
    rc = eng_->pf_update_maps(ctx_, pts.data(), n, o, q);
    if (rc) fail(rc, "lama_hip_pf_update_maps");
    if (eng_->get_counters(ctx_, &c1) == 0) number_of_proccessed_cells_ = (uint32_t)(c1.bf_cells - c0.bf_cells);
    if (transient_map_) {                                          // :322-379
        rc = transient::prune(eng_.get(), ctx_, *surface, pose_, resolution_, l2_max_, truncated_range_, 2.0, &last_deleted_);
        if (rc) fail(rc, "transient map");
    }
    if (summary) {                                                 // :188-194
        summary->time_mapping.push_back(now_s() - t_map);
        summary->time.push_back(now_s() - t_begin);
        summary->timestamp.push_back(timestamp);
        summary->memory.push_back((double)getMemoryUsage());
    }
    return true;
}

uint64_t Slam2D::getMemoryUsage() const
{
    lama_hip_counters c;
    if (eng_->get_counters(ctx_, &c) != 0) return 0;
    return c.dm_patches * 10240ull + c.occ_patches * 4096ull;
}

static bool dl(const HipEngine* e, lama_hip_ctx* ctx, int kind, size_t cell_bytes, std::vector<uint64_t>& ids,
               std::vector<uint8_t>& cells, std::vector<uint64_t>& masks)
{
    uint32_t n = 0;
    if (e->pf_map_patches(ctx, 0, kind, &n) != 0) return false;
    ids.assign(n, 0); cells.assign((size_t)n * cell_bytes * 1024, 0); masks.assign((size_t)n * 16, 0);
    uint32_t got = 0;
    return e->pf_download_map(ctx, 0, kind, n, ids.data(), cells.data(), masks.data(), &got) == 0 && got == n;
}

bool Slam2D::downloadDistanceMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const
{ return has_first_scan && dl(eng_.get(), ctx_, LAMA_HIP_MAP_DISTANCE, 10, ids, cells, masks); }

bool Slam2D::downloadOccupancyMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const
{ return has_first_scan && dl(eng_.get(), ctx_, LAMA_HIP_MAP_OCCUPANCY, 4, ids, cells, masks); }

} // namespace lama
