// lidar_odometry_2d.cpp -- host-side lama::LidarOdometry2D (include/lama/lidar_odometry_2d.h); orchestration of
// src/lidar_odometry_2d.cpp:42-200 over a one-particle device context.
#include "lama/lidar_odometry_2d.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <limits>
#include <stdexcept>

#include "hip_engine.hpp"

namespace lama {

namespace {
struct Aff { double R[3][3]; double t[3]; };
// Translation3d(origin) * quaternion (Eigen Quaternion::toRotationMatrix restated), src/lidar_odometry_2d.cpp:88
Aff moving_tf(const PointCloudXYZ& s)
{
    Aff a;
    const double w = s.sensor_orientation_.w(), x = s.sensor_orientation_.x(), y = s.sensor_orientation_.y(), z = s.sensor_orientation_.z();
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    a.R[0][0] = 1.0 - (tyy + tzz); a.R[0][1] = txy - twz;         a.R[0][2] = txz + twy;
    a.R[1][0] = txy + twz;         a.R[1][1] = 1.0 - (txx + tzz); a.R[1][2] = tyz - twx;
    a.R[2][0] = txz - twy;         a.R[2][1] = tyz + twx;         a.R[2][2] = 1.0 - (txx + tyy);
    a.t[0] = s.sensor_origin_.x(); a.t[1] = s.sensor_origin_.y(); a.t[2] = s.sensor_origin_.z();
    return a;
}
// Translation3d(x, y, 0) * AngleAxisd(rotation, UnitZ), :89
Aff fixed_tf(const Pose2D& p)
{
    Aff a;
    const double th = p.rotation();
    const double sn = std::sin(th), cs = std::cos(th);
    const double F[3][3] = {{cs, 0.0 - sn, 0.0}, {sn, cs, 0.0}, {0.0, 0.0, (1.0 - cs) + cs}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a.R[i][j] = F[i][j];
    a.t[0] = p.x(); a.t[1] = p.y(); a.t[2] = 0.0;
    return a;
}
Aff mul(const Aff& A, const Aff& B)
{
    Aff r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) r.R[i][j] = (A.R[i][0] * B.R[0][j] + A.R[i][1] * B.R[1][j]) + A.R[i][2] * B.R[2][j];
        r.t[i] = ((A.R[i][0] * B.t[0] + A.R[i][1] * B.t[1]) + A.R[i][2] * B.t[2]) + A.t[i];
    }
    return r;
}
struct Box {                                                            // include/lama/aabb.h:41-74
    double center[3], hwidth[3];
    Box(const double mn[3], const double mx[3]) { for (int k = 0; k < 3; ++k) { const double l = mx[k] - mn[k]; hwidth[k] = l * 0.5; center[k] = mn[k] + hwidth[k]; } }
    bool meets(const Box& o) const
    {
        bool r = true;
        for (int k = 0; k < 3; ++k) r = r && (std::abs(center[k] - o.center[k]) <= (hwidth[k] + o.hwidth[k]));
        return r;
    }
};
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
    eng_ = engineOverride() ? engineOverride() : loadHipEngine();
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
    // transient map (:128-199): box of the hits, made symmetric about the pose, expanded by twice the distance map's range
    const Aff tf = mul(fixed_tf(odom), moving_tf(s));
    double mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<double>::max(); mx[k] = -std::numeric_limits<double>::max(); }
    for (size_t i = 0; i < s.points.size(); ++i) {
        const double px = s.points[i].x(), py = s.points[i].y(), pz = s.points[i].z();
        const double h[3] = {((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0],
                             ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1],
                             ((tf.R[2][0] * px + tf.R[2][1] * py) + tf.R[2][2] * pz) + tf.t[2]};
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], h[k]); mx[k] = std::max(mx[k], h[k]); }
    }
    mn[2] = mx[2] = 0;
    const double xdist = std::max(odom.x() - mn[0], mx[0] - odom.x());
    const double ydist = std::max(odom.y() - mn[1], mx[1] - odom.y());
    mn[0] = odom.x() - xdist; mn[1] = odom.y() - ydist;
    mx[0] = odom.x() + xdist; mx[1] = odom.y() + ydist;
    Box a(mn, mx);
    const double scale = 1.0 / opt_.resolution;
    const double max_dist = std::sqrt((double)((uint32_t)std::ceil(1.0 * scale) * (uint32_t)std::ceil(1.0 * scale))) * opt_.resolution;   // maxDistance()
    for (int k = 0; k < 3; ++k) a.hwidth[k] += 2.0 * max_dist;
    uint32_t np = 0;
    rc = eng_->pf_patch_ids(ctx_, 0, LAMA_HIP_MAP_DISTANCE, 0, nullptr, &np);
    if (rc) fail(rc, "lama_hip_pf_patch_ids");
    std::vector<uint64_t> ids(np), to_remove;
    if (np) { rc = eng_->pf_patch_ids(ctx_, 0, LAMA_HIP_MAP_DISTANCE, np, ids.data(), &np); if (rc) fail(rc, "lama_hip_pf_patch_ids"); }
    const double off = double(2642244ull >> 1) * 32.0;
    const double l = (scale * scale) * (1.0 / ((scale * scale) * scale)), t = -(l * off);      // Map::m2w = tf_inv_ * m (see loc2d.cpp)
    for (uint64_t id : ids) {
        const uint32_t ox = (uint32_t)((id / 2642244ull) << 5), oy = (uint32_t)((id % 2642244ull) << 5);   // Map::p2m
        const double ws[3] = {l * (double)ox + t, l * (double)oy + t, 0.0};
        const double we[3] = {l * (double)(ox + 32u) + t, l * (double)(oy + 32u) + t, 0.0};
        Box b(ws, we);
        if (a.meets(b)) continue;
        to_remove.push_back(id);
    }
    last_deleted_ = 0;
    if (!to_remove.empty()) {
        rc = eng_->pf_delete_patches(ctx_, 0, to_remove.data(), (uint32_t)to_remove.size(), &last_deleted_);
        if (rc) fail(rc, "lama_hip_pf_delete_patches");
    }
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
