// transient_map.hpp -- the "transient map" step shared by Slam2D::updateMaps (src/slam2d.cpp:322-379) and
// LidarOdometry2D::updateMaps (src/lidar_odometry_2d.cpp:128-199): the box of the scan's hit points is made symmetric about
// the pose, expanded by twice the distance map's range, and every distance-map patch whose box does not meet it is deleted
// from both maps (Map::deletePatchAt) -- on the device through lama_hip_pf_patch_ids / lama_hip_pf_delete_patches.
#pragma once

#include <algorithm>
#include <cmath>
#include <limits>
#include <vector>

#include "hip_engine.hpp"
#include "lama/pose2d.h"
#include "lama/types.h"

namespace lama {
namespace transient {

struct Aff { double R[3][3]; double t[3]; };

// Translation3d(origin) * quaternion (Eigen Quaternion::toRotationMatrix restated)
inline Aff moving_tf(const PointCloudXYZ& s)
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
// Translation3d(x, y, 0) * AngleAxisd(rotation, UnitZ)
inline Aff fixed_tf(const Pose2D& p)
{
    Aff a;
    const double th = p.rotation();
    const double sn = std::sin(th), cs = std::cos(th);
    const double F[3][3] = {{cs, 0.0 - sn, 0.0}, {sn, cs, 0.0}, {0.0, 0.0, (1.0 - cs) + cs}};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a.R[i][j] = F[i][j];
    a.t[0] = p.x(); a.t[1] = p.y(); a.t[2] = 0.0;
    return a;
}
inline Aff mul(const Aff& A, const Aff& B)
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

// Returns the engine status (0 = ok); *deleted = distance-map patches removed.
//   truncated_range: Slam2D's option (a hit beyond it is pulled in before it enters the box, src/slam2d.cpp:279-288); 0 = off
//   dist_factor    : 2.0 for Slam2D (:334-335), 1.0 for LidarOdometry2D (:137-138)
inline int32_t prune(const HipEngine* eng, lama_hip_ctx* ctx, const PointCloudXYZ& s, const Pose2D& pose, double resolution, double l2_max,
                     double truncated_range, double dist_factor, uint32_t* deleted)
{
    if (deleted) *deleted = 0;
    const Aff tf = mul(fixed_tf(pose), moving_tf(s));
    double mn[3], mx[3];
    for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<double>::max(); mx[k] = -std::numeric_limits<double>::max(); }
    for (size_t i = 0; i < s.points.size(); ++i) {
        const double px = s.points[i].x(), py = s.points[i].y(), pz = s.points[i].z();
        double h[3] = {((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0],
                       ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1],
                       ((tf.R[2][0] * px + tf.R[2][1] * py) + tf.R[2][2] * pz) + tf.t[2]};
        if (truncated_range > 0.0) {
            const double ab[3] = {h[0] - tf.t[0], h[1] - tf.t[1], h[2] - tf.t[2]};
            const double len = std::sqrt((ab[0] * ab[0] + ab[1] * ab[1]) + ab[2] * ab[2]);
            if (truncated_range < len) for (int k = 0; k < 3; ++k) h[k] = tf.t[k] + ab[k] / len * truncated_range;
        }
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], h[k]); mx[k] = std::max(mx[k], h[k]); }
    }
    mn[2] = mx[2] = 0;
    const double xdist = std::max(pose.x() - mn[0], mx[0] - pose.x()) * dist_factor;
    const double ydist = std::max(pose.y() - mn[1], mx[1] - pose.y()) * dist_factor;
    mn[0] = pose.x() - xdist; mn[1] = pose.y() - ydist;
    mx[0] = pose.x() + xdist; mx[1] = pose.y() + ydist;
    Box a(mn, mx);
    const double scale = 1.0 / resolution;
    const uint32_t r = (uint32_t)std::ceil(l2_max * scale);              // DynamicDistanceMap::setMaxDistance / maxDistance
    const double max_dist = std::sqrt((double)(r * r)) * resolution;
    for (int k = 0; k < 3; ++k) a.hwidth[k] += 2.0 * max_dist;
    uint32_t np = 0;
    int32_t rc = eng->pf_patch_ids(ctx, 0, LAMA_HIP_MAP_DISTANCE, 0, nullptr, &np);
    if (rc) return rc;
    std::vector<uint64_t> ids(np), to_remove;
    if (np) { rc = eng->pf_patch_ids(ctx, 0, LAMA_HIP_MAP_DISTANCE, np, ids.data(), &np); if (rc) return rc; }
    const double off = double(2642244ull >> 1) * 32.0;
    const double l = (scale * scale) * (1.0 / ((scale * scale) * scale)), t = -(l * off);      // Map::m2w = tf_inv_ * m
    for (uint64_t id : ids) {
        const uint32_t ox = (uint32_t)((id / 2642244ull) << 5), oy = (uint32_t)((id % 2642244ull) << 5);   // Map::p2m
        const double ws[3] = {l * (double)ox + t, l * (double)oy + t, 0.0};
        const double we[3] = {l * (double)(ox + 32u) + t, l * (double)(oy + 32u) + t, 0.0};
        if (a.meets(Box(ws, we))) continue;
        to_remove.push_back(id);
    }
    if (to_remove.empty()) return 0;
    return eng->pf_delete_patches(ctx, 0, to_remove.data(), (uint32_t)to_remove.size(), deleted);
}

} // namespace transient
} // namespace lama
