// lidar_odometry_2d.cpp -- host-side lama::LidarOdometry2D (include/lama/lidar_odometry_2d.h); orchestration of
// src/lidar_odometry_2d.cpp:42-200 over a one-particle device context.
#include "lama/lidar_odometry_2d.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
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

LidarOdometry2D::LidarOdometry2D(const Options& o) : opt_(o)              // src/lidar_odometry_2d.cpp:42-52
{
    eng_ = defaultEngine();
    lama_hip_cfg cfg;
    eng_->default_cfg(&cfg);
    cfg.particles = 1;
    cfg.resolution = o.resolution; cfg.patch_size = 32; cfg.l2_max = 1.0; cfg.max_iter = o.max_iter;
    cfg.device = o.gpu_device;
    cfg.occupancy_policy = 1;                  // ProbabilisticOccupancyMap
    cfg.ray_rule = 1;                          // the last metre before the hit
    cfg.dm_patch_capacity = 1024; cfg.occ_patch_capacity = 1024;
    cfg.queue_capacity = 1u << 18;
    const int32_t rc = eng_->ctx_create(&cfg, &ctx_);
    if (rc != 0 || !ctx_) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "lama::LidarOdometry2D: lama_hip_ctx_create failed (status %d): no usable MI355X / HIP device; there is no CPU fallback", rc);
        throw std::runtime_error(msg);
    }
}

LidarOdometry2D::~LidarOdometry2D() { if (ctx_) eng_->ctx_destroy(ctx_); }

void LidarOdometry2D::fail(int32_t rc, const char* what) const
{
    char msg[512];
    std::snprintf(msg, sizeof(msg), "lama::LidarOdometry2D: %s failed (status %d): %s", what, rc, eng_->last_error(ctx_));
    throw std::runtime_error(msg);
}

bool LidarOdometry2D::update(const PointCloudXYZ::Ptr& surface, double)   // :60-83
{
    if (!surface || surface->points.empty()) throw std::runtime_error("lama::LidarOdometry2D::update: empty scan");
    if (!has_first_scan) {
        updateMaps(surface);
        has_first_scan = true;
        return true;
    }
    std::vector<double> pts; double o[3], q[4];
    scan_arrays(*surface, pts, o, q);
    double p[4], out7[7];
    int32_t iters = 0;
    odom.state.toArray(p);
    const int32_t rc = eng_->match_solve(ctx_, 0, pts.data(), (uint32_t)surface->points.size(), o, q, p, out7, &iters, 1);
    if (rc) fail(rc, "lama_hip_match_solve");
    odom.state = SE2d::fromArray(p);                                       // :72
    last_iterations_ = (uint32_t)iters;
    Pose2D odelta = map_update_odom - odom;                                // :75-80
    if (odelta.xy().norm() > 0.1 || std::abs(odelta.rotation()) > 0.5) {
        updateMaps(surface);
        map_update_odom = odom;
    }
    return true;
}

void LidarOdometry2D::updateMaps(const PointCloudXYZ::Ptr& surface)        // :85-200
{
    const PointCloudXYZ& s = *surface;
    std::vector<double> pts; double o[3], q[4];
    scan_arrays(s, pts, o, q);
    double p[4];
    odom.state.toArray(p);
    const uint32_t n = (uint32_t)s.points.size();
    int32_t rc;
    if (!device_initialised_) {
        rc = eng_->pf_init(ctx_, pts.data(), n, o, q, p);
        if (rc) fail(rc, "lama_hip_pf_init");
        device_initialised_ = true;
    } else {
        rc = eng_->pf_set_poses(ctx_, p);
        if (rc) fail(rc, "lama_hip_pf_set_poses");
        rc = eng_->pf_update_maps(ctx_, pts.data(), n, o, q);
        if (rc) fail(rc, "lama_hip_pf_update_maps");
    }
    // transient map (:128-199)
    rc = transient::prune(eng_.get(), ctx_, s, odom, opt_.resolution, 1.0, 0.0, 1.0, &last_deleted_);
    if (rc) fail(rc, "lama_hip_pf_patch_ids / lama_hip_pf_delete_patches");
}

static bool download(const HipEngine* e, lama_hip_ctx* ctx, int kind, size_t cell_bytes, sdm::HostMap& m)
{
    uint32_t n = 0;
    if (e->pf_map_patches(ctx, 0, kind, &n) != 0) return false;
    m.ids.assign(n, 0); m.cells.assign((size_t)n * 1024 * cell_bytes, 0); m.masks.assign((size_t)n * 16, 0);
    if (n == 0) return true;
    uint32_t got = 0;
    return e->pf_download_map(ctx, 0, kind, n, m.ids.data(), m.cells.data(), m.masks.data(), &got) == 0 && got == n;
}

bool LidarOdometry2D::downloadDistanceMap(sdm::HostMap& m) const
{
    m.kind = sdm::kDistanceMap; m.resolution = opt_.resolution;
    const uint32_t r = (uint32_t)std::ceil(1.0 * (1.0 / opt_.resolution));
    m.max_sqdist = r * r;
    return download(eng_.get(), ctx_, LAMA_HIP_MAP_DISTANCE, 10, m);
}

bool LidarOdometry2D::downloadOccupancyMap(sdm::HostMap& m) const
{
    m.kind = sdm::kProbabilisticOccupancyMap;   // 4-byte cells: ProbabilisticOccupancyMap::prob_tag {float}
    m.resolution = opt_.resolution;
    return download(eng_.get(), ctx_, LAMA_HIP_MAP_OCCUPANCY, 4, m);
}

} // namespace lama
