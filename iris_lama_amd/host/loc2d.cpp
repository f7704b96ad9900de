// loc2d.cpp -- host-side lama::Loc2D (include/lama/loc2d.h); orchestration of src/loc2d.cpp:61-192.
#include "lama/loc2d.h"

#include <cmath>
#include <algorithm>
#include <cstdio>
#include <limits>
#include <stdexcept>

#include "covariance3.hpp"
#include "dm_builder.hpp"
#include "hip_engine.hpp"
#include "lama/random.h"

namespace lama {

Loc2D::Options::Options()                      // src/loc2d.cpp:46-58
{
    trans_thresh = 0.5; rot_thresh = 0.5; l2_max = 1.0; resolution = 0.05; patch_size = 32;
    gloc_particles = 3000; gloc_iters = 10; gloc_thresh = 0.15; max_iter = 100; cov_blend = 0.0;
}

void Loc2D::Init(const Options& o)
{
    opt_ = o;
    // a second Init() (new map, resolution, l2_max or strategy) starts from fresh maps like the reference's (src/loc2d.cpp:61-78):
    // the device context is rebuilt from the new options by the next ensureContext()
    if (ctx_) { eng_->ctx_destroy(ctx_); ctx_ = nullptr; }
    delete occupancy_map; delete distance_map;
    occupancy_map = new SimpleOccupancyMap(o.resolution, o.patch_size);                // src/loc2d.cpp:63-66
    distance_map = new DynamicDistanceMap(o.resolution, o.patch_size);
    distance_map->setMaxDistance(o.l2_max);
    DynamicDistanceMap::Writer w;
    w.apply = [this](std::vector<uint32_t>& cells_xy, double max_distance) -> uint32_t {
        (void)max_distance;                                 // part of the context's configuration (ensureContext)
        ensureContext();
        {   // The FIRST build -- every occupied cell of a static map added to an empty distance map, one update(): src/loc2d.cpp:61-108
            // with the caller's loop -- is one serial chain of pops with nothing to parallelise over; it is replayed on the host
            // (dm_builder.hpp: from-scratch code, std::priority_queue for the reference's tie order) and uploaded like a map read
            // from a file.  Later updates of the map that now exists run on the device.
            uint32_t have = 0;
            if (eng_->pf_map_patches(ctx_, 0, 0 /* distance map */, &have) == 0 && have == 0) {
                sdm::HostMap m;
                m.kind = sdm::kDistanceMap; m.resolution = distance_map->resolution; m.patch_length = distance_map->patch_length;
                m.max_sqdist = distance_map->maxSqDist();
                uint32_t processed = 0;
                if (detail::build_distance_map(cells_xy.data(), cells_xy.size() / 2, m.max_sqdist, m, processed)) {
                    const int32_t ru = eng_->pf_upload_map(ctx_, 0, 0, (uint32_t)m.ids.size(), m.ids.data(), m.cells.data(), m.masks.data());
                    if (ru) fail(ru, "lama_hip_pf_upload_map (first build of the distance map)");
                    host_built_ = true;
                    return processed;
                }
            }
        }
        lama_hip_counters c0, c1;
        const bool have0 = eng_->get_counters(ctx_, &c0) == 0;
        const int32_t rc = eng_->map_add_obstacles(ctx_, 0, cells_xy.data(), (uint32_t)(cells_xy.size() / 2));
        if (rc) fail(rc, "lama_hip_map_add_obstacles");
        // DynamicDistanceMap::update returns the cells processed by THIS call (src/sdm/dynamic_distance_map.cpp:196)
        return (have0 && eng_->get_counters(ctx_, &c1) == 0) ? (uint32_t)(c1.bf_cells - c0.bf_cells) : 0;
    };
    w.download = [this](sdm::HostMap& m) -> bool {
        if (!ctx_) return false;
        uint32_t n = 0, got = 0;
        if (eng_->pf_map_patches(ctx_, 0, 0 /* distance map */, &n) != 0) return false;
        m.kind = sdm::kDistanceMap; m.max_sqdist = distance_map->maxSqDist();
        m.ids.assign(n, 0); m.cells.assign((size_t)n * 10 * 1024, 0); m.masks.assign((size_t)n * 16, 0);
        return eng_->pf_download_map(ctx_, 0, 0, n, m.ids.data(), m.cells.data(), m.masks.data(), &got) == 0 && got == n;
    };
    w.upload = [this](const sdm::HostMap& m) {
        ensureContext();
        const int32_t rc = eng_->pf_upload_map(ctx_, 0, 0 /* distance map */, (uint32_t)m.ids.size(), m.ids.data(), m.cells.data(), m.masks.data());
        if (rc) fail(rc, "lama_hip_pf_upload_map");
    };
    distance_map->bindWriter(std::move(w));
    rmse_ = 0.0;
    host_built_ = false;
    cov_ = Matrix3d::Identity();
    has_first_scan = false;
    do_global_localization_ = false;                                   // :80-90
    gloc_cur_iter_ = 0;
    cov_blend_ = std::max(std::min(o.cov_blend, 1.0), 0.0);
    if (!sampling_steps_.empty()) return;                              // :93-107
    const double sstep = distance_map->resolution;
    sampling_steps_.push_back(Vector2d(0.0, 0.0));
    for (int i = 1; i <= 20; ++i) {
        sampling_steps_.push_back(Vector2d(i * sstep, 0.0));   sampling_steps_.push_back(Vector2d(0.0, i * sstep));
        sampling_steps_.push_back(Vector2d(-i * sstep, 0.0));  sampling_steps_.push_back(Vector2d(0.0, -i * sstep));
        sampling_steps_.push_back(Vector2d(i * sstep, i * sstep));   sampling_steps_.push_back(Vector2d(-i * sstep, i * sstep));
        sampling_steps_.push_back(Vector2d(i * sstep, -i * sstep));  sampling_steps_.push_back(Vector2d(-i * sstep, -i * sstep));
    }
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
    eng_ = defaultEngine(distance_map->maxDistanceOption(), opt_.resolution);
    lama_hip_cfg cfg;
    eng_->default_cfg(&cfg);
    cfg.particles = 1;
    cfg.resolution = opt_.resolution; cfg.patch_size = opt_.patch_size; cfg.l2_max = distance_map->maxDistanceOption(); cfg.max_iter = opt_.max_iter;
    cfg.device = opt_.gpu_device;
    cfg.solver_strategy = opt_.strategy == "lm" ? 1u : 0u;       // makeStrategy, src/loc2d.cpp:288-294
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

bool Loc2D::enoughMotion(const Pose2D& odometry)               // src/loc2d.cpp:113-124
{
    if (!has_first_scan) return true;
    Pose2D odelta = odom_ - odometry;
    if (odelta.xy().norm() <= opt_.trans_thresh && std::abs(odelta.rotation()) <= opt_.rot_thresh) return false;
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
        detail::covariance_from_normal3(out7, c9);
        for (int k = 0; k < 9; ++k) cov_(k / 3, k % 3) = c9[k];
        if (cov_blend_ > 0.0) addSamplingCovariance(s);                                 // :175-176
    }
    rmse_ = std::sqrt(out7[6] / ((double)(s.points.size() - 1)));                       // :178-180
}

bool Loc2D::update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double, bool force_update)
{
    if (!surface || surface->points.size() < 2) throw std::runtime_error("lama::Loc2D::update: empty scan");
    if (distance_map && distance_map->hasPending()) distance_map->update();
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
    if (do_global_localization_) {                              // :157-168
        if (gloc_cur_iter_ < opt_.gloc_iters) {
            ++gloc_cur_iter_;
            globalLocalization(*surface);
        } else {
            do_global_localization_ = false;
            gloc_cur_iter_ = 0;
        }
    }
    solve(*surface, true);                                      // :170-180 (cov, optional sampling covariance, rmse)
    if (do_global_localization_ && rmse_ < opt_.gloc_thresh) {  // :182-189
        do_global_localization_ = false;
        gloc_cur_iter_ = 0;
    }
    return true;
}

void Loc2D::triggerGlobalLocalization() { do_global_localization_ = true; }      // :194-197

static void scan_arrays(const PointCloudXYZ& s, std::vector<double>& pts, double o[3], double q[4])
{
    pts.resize(s.points.size() * 3);
    for (size_t i = 0; i < s.points.size(); ++i) { pts[3 * i] = s.points[i].x(); pts[3 * i + 1] = s.points[i].y(); pts[3 * i + 2] = s.points[i].z(); }
    o[0] = s.sensor_origin_.x(); o[1] = s.sensor_origin_.y(); o[2] = s.sensor_origin_.z();
    q[0] = s.sensor_orientation_.w(); q[1] = s.sensor_orientation_.x(); q[2] = s.sensor_orientation_.y(); q[3] = s.sensor_orientation_.z();
}

// :249-286.  The candidates are drawn exactly like the reference (x, y until the cell is free in the occupancy map,
// then the heading; lama::random's stream); their squared residual norms come from the device in one batch and the
// FIRST smallest one wins (the reference's strict `<`).
void Loc2D::globalLocalization(const PointCloudXYZ& surface)
{
    ensureContext();
    Vector3d mn, mx;
    occupancy_map->bounds(mn, mx);
    const double diff0 = mx[0] - mn[0], diff1 = mx[1] - mn[1];
    const uint32_t B = opt_.gloc_particles;
    if (B == 0) return;
    bool any_free = false;
    occupancy_map->visit_all_cells([&](const Vector3ui& c) { if (!any_free && occupancy_map->isFree(c)) any_free = true; });
    if (!any_free) throw std::runtime_error("lama::Loc2D::globalLocalization: the occupancy map has no free cell");
    gloc_poses_.assign((size_t)4 * B, 0.0);
    gloc_errors_.assign(B, 0.0);
    for (uint32_t i = 0; i < B; ++i) {
        double x, y, a;
        for (;;) {
            x = mn[0] + random::uniform() * diff0;
            y = mn[1] + random::uniform() * diff1;
            if (!occupancy_map->isFree(Vector3d(x, y, 0.0))) continue;
            a = random::uniform() * 2 * M_PI - M_PI;
            break;
        }
        Pose2D(x, y, a).state.toArray(&gloc_poses_[4 * i]);
    }
    std::vector<double> pts; double o[3], q[4];
    scan_arrays(surface, pts, o, q);
    const int32_t rc = eng_->eval_batch(ctx_, 0, pts.data(), (uint32_t)surface.points.size(), o, q, gloc_poses_.data(), B, gloc_errors_.data(), nullptr);
    if (rc) fail(rc, "lama_hip_eval_batch");
    double best_error = std::numeric_limits<double>::max();
    for (uint32_t i = 0; i < B; ++i)
        if (gloc_errors_[i] < best_error) { best_error = gloc_errors_[i]; pose_.state = SE2d::fromArray(&gloc_poses_[4 * i]); }
}

// :199-247 (Olson 2009).  The 161 likelihood samples come from the device, the 2x2 statistics are accumulated here in
// the reference's order.
void Loc2D::addSamplingCovariance(const PointCloudXYZ& surface)
{
    const size_t num_points = surface.points.size();
    const size_t step = std::max(num_points / 100, size_t(1));
    const size_t K = sampling_steps_.size();
    std::vector<double> xy(2 * K);
    for (size_t i = 0; i < K; ++i) { xy[2 * i] = pose_.x() + sampling_steps_[i].x(); xy[2 * i + 1] = pose_.y() + sampling_steps_[i].y(); }
    sampling_l_.assign(K, 0.0);
    std::vector<double> pts; double o[3], q[4];
    scan_arrays(surface, pts, o, q);
    const int32_t rc = eng_->map_sample_likelihood(ctx_, 0, pts.data(), (uint32_t)num_points, o, q, pose_.rotation(), xy.data(), (uint32_t)K,
                                                   (uint32_t)step, sampling_l_.data());
    if (rc) fail(rc, "lama_hip_map_sample_likelihood");
    double Km[2][2] = {{0, 0}, {0, 0}}, u[2] = {0, 0}, s = 0;
    for (size_t i = 0; i < K; ++i) {
        const double x = xy[2 * i], y = xy[2 * i + 1], l = sampling_l_[i];
        Km[0][0] = Km[0][0] + x * x * l; Km[0][1] = Km[0][1] + x * y * l;
        Km[1][0] = Km[1][0] + y * x * l; Km[1][1] = Km[1][1] + y * y * l;
        u[0] = u[0] + x * l; u[1] = u[1] + y * l;
        s = s + l;
    }
    const double a1 = 1.0 / s, a2 = 1.0 / (s * s);
    const double sc[2][2] = {{a1 * Km[0][0] - a2 * u[0] * u[0], a1 * Km[0][1] - a2 * u[0] * u[1]},
                             {a1 * Km[1][0] - a2 * u[1] * u[0], a1 * Km[1][1] - a2 * u[1] * u[1]}};
    const double alpha = cov_blend_;
    for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) cov_(r, c) = alpha * sc[r][c] + (1.0 - alpha) * cov_(r, c);
}

} // namespace lama
