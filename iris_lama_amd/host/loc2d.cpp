// loc2d.cpp -- host-side lama::Loc2D (include/lama/loc2d.h); orchestration of src/loc2d.cpp:61-192.
#include "lama/loc2d.h"

#include <cmath>
#include <cstdio>
#include <stdexcept>

#include "hip_engine.hpp"

namespace lama {

Loc2D::Options::Options()                      // src/loc2d.cpp:46-58
{
    trans_thresh = 0.5; rot_thresh = 0.5; l2_max = 1.0; resolution = 0.05; patch_size = 32;
    gloc_particles = 3000; gloc_iters = 10; gloc_thresh = 0.15; max_iter = 100; cov_blend = 0.0;
}

Vector3ui Loc2D::MapProxy::w2m(const Vector3d& p) const
{
    const double off = double(2642244ull >> 1) * 32.0;        // src/sdm/map.cpp:55-58
    return Vector3ui((uint32_t)(scale * p.x() + off + 0.5), (uint32_t)(scale * p.y() + off + 0.5), (uint32_t)(scale * p.z() + off + 0.5));
}

static uint64_t key(const Vector3ui& c) { return ((uint64_t)c(0) << 32) | c(1); }
// SimpleOccupancyMap semantics (src/sdm/simple_occupancy_map.cpp): cell = -1 free, 1 occupied, 0 unknown; returns "changed"
bool Loc2D::OccupancyMapProxy::setFree(const Vector3ui& c) { int8_t& v = cells[key(c)]; if (v == -1) return false; v = -1; return true; }
bool Loc2D::OccupancyMapProxy::setOccupied(const Vector3ui& c) { int8_t& v = cells[key(c)]; if (v == 1) return false; v = 1; return true; }
bool Loc2D::OccupancyMapProxy::setUnknown(const Vector3ui& c) { int8_t& v = cells[key(c)]; if (v == 0) return false; v = 0; return true; }
bool Loc2D::OccupancyMapProxy::isFree(const Vector3ui& c) const { auto it = cells.find(key(c)); return it != cells.end() && it->second == -1; }
bool Loc2D::OccupancyMapProxy::isOccupied(const Vector3ui& c) const { auto it = cells.find(key(c)); return it != cells.end() && it->second == 1; }

void Loc2D::Init(const Options& o)
{
    if (o.strategy == "lm") throw std::runtime_error("lama::Loc2D: strategy \"lm\" is not available on the device path");
    if (o.cov_blend > 0.0) throw std::runtime_error("lama::Loc2D: cov_blend > 0 (sampling covariance) is not available on the device path");
    opt_ = o;
    delete occupancy_map; delete distance_map;
    occupancy_map = new OccupancyMapProxy;
    distance_map = new DistanceMapProxy;
    occupancy_map->resolution = distance_map->resolution = o.resolution;
    occupancy_map->scale = distance_map->scale = 1.0 / o.resolution;
    distance_map->l2_max = o.l2_max;
    distance_map->owner = this;
    rmse_ = 0.0;
    cov_ = Matrix3d_();
    has_first_scan = false;
}

Loc2D::~Loc2D()
{
    if (ctx_) eng_->ctx_destroy(ctx_);
    delete occupancy_map;
    delete distance_map;
}

void Loc2D::fail(int32_t rc, const char* what) const
{
    char msg[512];
    std::snprintf(msg, sizeof(msg), "lama::Loc2D: %s failed (status %d): %s", what, rc, eng_->last_error(ctx_));
    throw std::runtime_error(msg);
}

void Loc2D::ensureContext()
{
    if (ctx_) return;
    if (!distance_map) throw std::runtime_error("lama::Loc2D: Init() must be called first");
    eng_ = engineOverride() ? engineOverride() : loadHipEngine();
    lama_hip_cfg cfg;
    eng_->default_cfg(&cfg);
    cfg.particles = 1;
    cfg.resolution = opt_.resolution; cfg.patch_size = opt_.patch_size; cfg.l2_max = distance_map->l2_max; cfg.max_iter = opt_.max_iter;
    cfg.device = opt_.gpu_device;
    cfg.dm_patch_capacity = 4096;            // a static building-scale map; occupancy is not kept on the device
    cfg.occ_patch_capacity = 8;
    cfg.queue_capacity = 1u << 20;
    const int32_t rc = eng_->ctx_create(&cfg, &ctx_);
    if (rc != 0 || !ctx_) {
        char msg[200];
        std::snprintf(msg, sizeof(msg), "lama::Loc2D: lama_hip_ctx_create failed (status %d): no usable MI355X / HIP device; there is no CPU fallback", rc);
        throw std::runtime_error(msg);
    }
}

uint32_t Loc2D::DistanceMapProxy::update()
{
    if (pending.empty()) return 0;
    owner->ensureContext();
    const int32_t rc = owner->eng_->map_add_obstacles(owner->ctx_, 0, pending.data(), (uint32_t)(pending.size() / 2));
    if (rc) owner->fail(rc, "lama_hip_map_add_obstacles");
    pending.clear();
    lama_hip_counters c;
    return owner->eng_->get_counters(owner->ctx_, &c) == 0 ? (uint32_t)c.bf_cells : 0;
}

bool Loc2D::enoughMotion(const Pose2D& odometry)               // src/loc2d.cpp:113-124
{
    if (!has_first_scan) return true;
    Pose2D odelta = odom_ - odometry;
    if (odelta.xy().norm() <= opt_.trans_thresh && std::abs(odelta.rotation()) <= opt_.rot_thresh) return false;
    return true;
}

// symmetric 3x3 inverse (the full-rank branch of Solver::calculateCovariance, src/nlls/solver.cpp:141-142)
static bool inverse_sym3(const double L[6] /*00,10,11,20,21,22*/, double out[9])
{
    const double a = L[0], b = L[1], c = L[3], e = L[2], f = L[4], i = L[5];       // [a b c; b e f; c f i]
    const double det = a * (e * i - f * f) - b * (b * i - f * c) + c * (b * f - e * c);
    if (!(std::fabs(det) > 0)) return false;
    out[0] = (e * i - f * f) / det; out[1] = (c * f - b * i) / det; out[2] = (b * f - c * e) / det;
    out[3] = out[1];                out[4] = (a * i - c * c) / det; out[5] = (c * b - a * f) / det;
    out[6] = out[2];                out[7] = out[5];                out[8] = (a * e - b * b) / det;
    return true;
}

void Loc2D::solve(const PointCloudXYZ& s, bool do_solve)
{
    ensureContext();
    std::vector<double> pts(s.points.size() * 3);
    for (size_t i = 0; i < s.points.size(); ++i) { pts[3 * i] = s.points[i].x(); pts[3 * i + 1] = s.points[i].y(); pts[3 * i + 2] = s.points[i].z(); }
    const double o[3] = {s.sensor_origin_.x(), s.sensor_origin_.y(), s.sensor_origin_.z()};
    const double q[4] = {s.sensor_orientation_.w(), s.sensor_orientation_.x(), s.sensor_orientation_.y(), s.sensor_orientation_.z()};
    double p[4], out7[7];
    int32_t iters = 0;
    pose_.state.toArray(p);
    const int32_t rc = eng_->match_solve(ctx_, 0, pts.data(), (uint32_t)s.points.size(), o, q, p, out7, &iters, do_solve ? 1 : 0);
    if (rc) fail(rc, "lama_hip_match_solve");
    if (do_solve) {
        pose_.state = SE2d::fromArray(p);
        last_iterations_ = (uint32_t)iters;
        double c9[9];
        if (inverse_sym3(out7, c9)) for (int k = 0; k < 9; ++k) cov_.m[k] = c9[k];      // rank-deficient J: covariance left unchanged
    }
    rmse_ = std::sqrt(out7[6] / ((double)(s.points.size() - 1)));                       // :178-180
}

bool Loc2D::update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double, bool force_update)
{
    if (!surface || surface->points.size() < 2) throw std::runtime_error("lama::Loc2D::update: empty scan");
    if (distance_map && !distance_map->pending.empty()) distance_map->update();
    if (!has_first_scan) {                                      // :128-143
        odom_ = odometry;
        has_first_scan = true;
        if (!force_update) return true;
        solve(*surface, false);
    }
    Pose2D odelta = odom_ - odometry;                           // :146-147
    Pose2D ppose = pose_ + odelta;
    if (!force_update && !enoughMotion(odometry)) return false; // :151-152
    pose_ = ppose;
    odom_ = odometry;
    solve(*surface, true);                                      // :168-180
    return true;
}

void Loc2D::triggerGlobalLocalization()
{
    throw std::runtime_error("lama::Loc2D::triggerGlobalLocalization is not available on the device path yet");
}

} // namespace lama
