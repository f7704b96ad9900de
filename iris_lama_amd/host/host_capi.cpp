// host_capi.cpp -- extern "C" flattening of lama::PFSlam2D (include/lama_host.h).
#include "lama_host.h"
#include "lama/match_surface_2d.h"
#include "lama/nlls/solver.h"

#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>

#include "hip_engine.hpp"
#include "lama/pf_slam2d.h"
#include "lama/lidar_odometry_2d.h"
#include "lama/loc2d.h"
#include "lama/random.h"
#include "lama/sdm_io.h"
#include "dm_builder.hpp"
#include "lama/slam2d.h"

using namespace lama;

struct lama_pf {
    std::unique_ptr<PFSlam2D> pf;
    std::string error;
    std::string origin;
    double times[5] = {0, 0, 0, 0, 0};
};

namespace {

PointCloudXYZ::Ptr make_cloud(const double* pts, uint32_t n, const double* origin3, const double* quat)
{
    auto c = std::make_shared<PointCloudXYZ>();
    c->points.resize(n);
    for (uint32_t i = 0; i < n; ++i) c->points[i] = Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
    if (origin3) c->sensor_origin_ = Vector3d(origin3[0], origin3[1], origin3[2]);
    if (quat) c->sensor_orientation_ = Quaterniond(quat[0], quat[1], quat[2], quat[3]);
    return c;
}

template <class F>
int guarded(lama_pf* pf, F&& f)
{
    try {
        return f();
    } catch (const std::exception& e) {
        pf->error = e.what();
        return -1;
    } catch (...) {
        pf->error = "unknown exception";
        return -1;
    }
}

void collect_times(lama_pf* h, size_t before[5])
{
    const PFSlam2D::Summary* s = h->pf->summary;
    if (!s) return;
    const DynamicArray<double>* v[5] = {&s->time, &s->time_solving, &s->time_normalizing, &s->time_resampling, &s->time_mapping};
    for (int k = 0; k < 5; ++k) h->times[k] = v[k]->size() > before[k] ? v[k]->back() : 0.0;
}

void sizes(const lama_pf* h, size_t out[5])
{
    const PFSlam2D::Summary* s = h->pf->summary;
    for (int k = 0; k < 5; ++k) out[k] = 0;
    if (!s) return;
    out[0] = s->time.size(); out[1] = s->time_solving.size(); out[2] = s->time_normalizing.size();
    out[3] = s->time_resampling.size(); out[4] = s->time_mapping.size();
}

} // namespace

extern "C" {

void lama_pf_default_options(lama_pf_options* o)
{
    PFSlam2D::Options d;
    std::memset(o, 0, sizeof(*o));
    o->particles = d.particles; o->srr = d.srr; o->str = d.str; o->stt = d.stt; o->srt = d.srt;
    o->meas_sigma = d.meas_sigma; o->meas_sigma_gain = d.meas_sigma_gain;
    o->trans_thresh = d.trans_thresh; o->rot_thresh = d.rot_thresh; o->l2_max = d.l2_max;
    o->truncated_ray = d.truncated_ray; o->truncated_range = d.truncated_range; o->resolution = d.resolution;
    o->patch_size = d.patch_size; o->max_iter = d.max_iter; o->seed = d.seed;
    o->create_summary = 1; o->gpu_device = 0; o->shard_rank = 0; o->shard_world = 1; o->profile = 0;
}

#ifdef LAMA_TESTING      // test builds of the host library only; the shipped liblama_host.so does not export this
int lama_host_set_engine_library(const char* path)
{
    try {
        if (!path || !*path) { setEngineOverride(nullptr); return 0; }
        setEngineOverride(loadHipEngine(path));
        return 0;
    } catch (...) {
        return -1;
    }
}
#endif

lama_pf* lama_pf_create(const lama_pf_options* o, char* err, int errcap)
{
    auto* h = new lama_pf;
    try {
        PFSlam2D::Options p;
        p.particles = o->particles; p.srr = o->srr; p.str = o->str; p.stt = o->stt; p.srt = o->srt;
        p.meas_sigma = o->meas_sigma; p.meas_sigma_gain = o->meas_sigma_gain;
        p.trans_thresh = o->trans_thresh; p.rot_thresh = o->rot_thresh; p.l2_max = o->l2_max;
        p.truncated_ray = o->truncated_ray; p.truncated_range = o->truncated_range; p.resolution = o->resolution;
        p.patch_size = o->patch_size; p.max_iter = o->max_iter; p.seed = o->seed;
        p.create_summary = o->create_summary != 0; p.gpu_device = o->gpu_device;
        p.shard_rank = o->shard_rank; p.shard_world = o->shard_world; p.profile = o->profile != 0;
        p.brushfire_mode = o->brushfire_mode;
        p.window_patches = o->window_patches; p.dm_patch_capacity = o->dm_patch_capacity;
        p.occ_patch_capacity = o->occ_patch_capacity; p.queue_capacity = o->queue_capacity;
        p.gpus = o->gpus > 1 ? o->gpus : 1;
        h->pf.reset(new PFSlam2D(p));
        h->origin = h->pf->engine()->origin;
        return h;
    } catch (const std::exception& e) {
        if (err && errcap > 0) { std::strncpy(err, e.what(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
}

void lama_pf_destroy(lama_pf* pf) { delete pf; }
int lama_pf_exchange_times(const lama_pf* h, double* out8)
{
    const PFSlam2D::ExchangeTimes& x = h->pf->exchangeTimes();
    out8[0] = x.gather; out8[1] = x.ship; out8[2] = x.import_; out8[3] = (double)x.shipped_particles; out8[4] = (double)x.shipped_bytes;
    out8[5] = x.local_copies; out8[6] = x.phase_begin; out8[7] = x.phase_maps;
    return (int)h->pf->numShards();
}
void* lama_pf_shard_context(const lama_pf* h, uint32_t r)
{
    const PFSlam2D* s = h->pf->shard(r);
    return s ? (void*)s->deviceContext() : nullptr;
}
const char* lama_pf_last_error(const lama_pf* pf) { return pf ? pf->error.c_str() : "null handle"; }
const char* lama_pf_engine_origin(const lama_pf* pf) { return pf ? pf->origin.c_str() : ""; }

void lama_pf_set_prior(lama_pf* pf, double x, double y, double yaw) { pf->pf->setPrior(Pose2D(x, y, yaw)); }

int lama_pf_update(lama_pf* h, const double* pts, uint32_t n, const double* origin3, const double* quat,
                   const double* odom_xyr, double ts)
{
    return guarded(h, [&] {
        size_t before[5]; sizes(h, before);
        bool r = h->pf->update(make_cloud(pts, n, origin3, quat), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts);
        collect_times(h, before);
        return r ? 1 : 0;
    });
}

int lama_pf_update_begin(lama_pf* h, const double* pts, uint32_t n, const double* origin3, const double* quat,
                         const double* odom_xyr, double ts)
{
    return guarded(h, [&] { return (int)h->pf->updateBegin(make_cloud(pts, n, origin3, quat), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts); });
}

int lama_pf_local_range(const lama_pf* h, uint32_t* lo, uint32_t* hi)
{
    *lo = h->pf->localBegin(); *hi = h->pf->localEnd();
    return 0;
}

int lama_pf_local_loglik(const lama_pf* h, double* out)
{
    const auto& v = h->pf->localLogLik();
    std::memcpy(out, v.data(), sizeof(double) * v.size());
    return (int)v.size();
}

int lama_pf_plan_resample(lama_pf* h, const double* all_loglik, int32_t* idx_out)
{
    return guarded(h, [&] {
        std::vector<int32_t> idx;
        const bool r = h->pf->planResample(all_loglik, idx);
        if (r) std::memcpy(idx_out, idx.data(), sizeof(int32_t) * idx.size());
        return r ? 1 : 0;
    });
}

int lama_pf_apply_resample(lama_pf* h, const int32_t* idx)
{
    return guarded(h, [&] {
        std::vector<int32_t> v(idx, idx + h->pf->getOptions().particles);
        h->pf->applyResample(v);
        return 0;
    });
}

int lama_pf_update_maps(lama_pf* h) { return guarded(h, [&] { h->pf->updateMaps(); return 0; }); }
void* lama_pf_device_context(const lama_pf* h) { return (void*)h->pf->deviceContext(); }

int lama_pf_get_poses(const lama_pf* h, double* out)
{
    const auto& ps = h->pf->getParticles();
    for (size_t i = 0; i < ps.size(); ++i) ps[i].pose.state.toArray(out + 4 * i);
    return (int)ps.size();
}

int lama_pf_set_pose(lama_pf* h, uint32_t i, const double* pose4)
{
    auto& ps = const_cast<std::vector<PFSlam2D::Particle>&>(h->pf->getParticles());
    if (i >= ps.size()) return -1;
    ps[i].pose.state = SE2d::fromArray(pose4);
    return 0;
}

int lama_pf_get_weights(const lama_pf* h, double* w, double* nw, double* ws)
{
    const auto& ps = h->pf->getParticles();
    for (size_t i = 0; i < ps.size(); ++i) {
        if (w) w[i] = ps[i].weight;
        if (nw) nw[i] = ps[i].normalized_weight;
        if (ws) ws[i] = ps[i].weight_sum;
    }
    return (int)ps.size();
}

int lama_pf_set_weights(lama_pf* h, const double* w, const double* ws)
{
    auto& ps = const_cast<std::vector<PFSlam2D::Particle>&>(h->pf->getParticles());
    for (size_t i = 0; i < ps.size(); ++i) {
        if (w) ps[i].weight = w[i];
        if (ws) ps[i].weight_sum = ws[i];
    }
    return 0;
}

double lama_pf_neff(const lama_pf* h) { return h->pf->getNeff(); }
int lama_pf_best(const lama_pf* h) { return (int)h->pf->getBestParticleIdx(); }
int lama_pf_best_pose_xyr(const lama_pf* h, double* xyr)
{
    const Pose2D p = h->pf->getPose();
    xyr[0] = p.x(); xyr[1] = p.y(); xyr[2] = p.rotation();
    return 0;
}
uint32_t lama_pf_num_resamples(const lama_pf* h) { return h->pf->numResamples(); }
uint64_t lama_pf_memory_usage(const lama_pf* h) { return h->pf->getMemoryUsage(); }

int lama_pf_summary(const lama_pf* h, char* buf, int cap)
{
    if (!h->pf->summary) return 0;
    const std::string r = h->pf->summary->report();
    if (buf && cap > 0) { std::strncpy(buf, r.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
    return (int)r.size() + 1;
}

int lama_pf_last_times(const lama_pf* h, double* out5)
{
    std::memcpy(out5, h->times, sizeof(h->times));
    return 0;
}

int lama_pf_draw_from_motion(lama_pf* h, const double* delta4, double* pose4)
{
    return guarded(h, [&] {
        Pose2D pose(SE2d::fromArray(pose4));
        h->pf->drawFromMotion(Pose2D(SE2d::fromArray(delta4)), pose);
        pose.state.toArray(pose4);
        return 0;
    });
}

double lama_pf_normalize(lama_pf* h) { h->pf->normalize(); return h->pf->getNeff(); }

int lama_pf_resample_indices(const lama_pf* h, double u01, int32_t* out)
{
    const std::vector<int32_t> v = h->pf->resampleIndices(u01);
    std::memcpy(out, v.data(), sizeof(int32_t) * v.size());
    return (int)v.size();
}

void lama_pose_minus(const double* a4, const double* b4, double* out4)
{
    (Pose2D(SE2d::fromArray(a4)) - Pose2D(SE2d::fromArray(b4))).state.toArray(out4);
}

void lama_pose_from_xyr(double x, double y, double yaw, double* out4) { Pose2D(x, y, yaw).state.toArray(out4); }

// ------------------------------------------------------------------ Slam2D
struct lama_slam {
    std::unique_ptr<Slam2D> s;
    std::string error, origin;
};

void lama_slam_default_options(lama_slam_options* o)
{
    Slam2D::Options d;
    o->trans_thresh = d.trans_thresh; o->rot_thresh = d.rot_thresh; o->l2_max = d.l2_max; o->truncated_ray = d.truncated_ray;
    o->truncated_range = d.truncated_range; o->resolution = d.resolution; o->patch_size = d.patch_size; o->max_iter = d.max_iter;
    o->gpu_device = 0;
    o->transient_map = d.transient_map ? 1 : 0;
    o->lm = 0;
}

lama_slam* lama_slam_create(const lama_slam_options* o, char* err, int errcap)
{
    auto* h = new lama_slam;
    try {
        Slam2D::Options p;
        p.trans_thresh = o->trans_thresh; p.rot_thresh = o->rot_thresh; p.l2_max = o->l2_max; p.truncated_ray = o->truncated_ray;
        p.truncated_range = o->truncated_range; p.resolution = o->resolution; p.patch_size = o->patch_size; p.max_iter = o->max_iter;
        p.gpu_device = o->gpu_device;
        p.transient_map = o->transient_map != 0;
        if (o->lm) p.strategy = "lm";
        h->s.reset(new Slam2D(p));
        h->origin = h->s->engine()->origin;
        return h;
    } catch (const std::exception& e) {
        if (err && errcap > 0) { std::strncpy(err, e.what(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
}
void lama_slam_destroy(lama_slam* s) { delete s; }
const char* lama_slam_last_error(const lama_slam* s) { return s ? s->error.c_str() : "null handle"; }
const char* lama_slam_engine_origin(const lama_slam* s) { return s ? s->origin.c_str() : ""; }
void lama_slam_set_pose(lama_slam* s, double x, double y, double yaw) { s->s->setPose(Pose2D(x, y, yaw)); }
int lama_slam_get_pose(const lama_slam* s, double* pose4) { s->s->getPose().state.toArray(pose4); return 0; }
int lama_slam_update(lama_slam* h, const double* pts, uint32_t n, const double* origin3, const double* quat, const double* odom_xyr, double ts)
{
    try {
        return h->s->update(make_cloud(pts, n, origin3, quat), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts) ? 1 : 0;
    } catch (const std::exception& e) { h->error = e.what(); return -1; }
}
int lama_slam_enough_motion(lama_slam* h, const double* odom_xyr) { return h->s->enoughMotion(Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2])) ? 1 : 0; }
uint32_t lama_slam_processed_cells(const lama_slam* h) { return h->s->getNumberOfProcessedCells(); }
uint32_t lama_slam_deleted_patches(const lama_slam* h) { return h->s->getLastDeletedPatches(); }
uint32_t lama_slam_iterations(const lama_slam* h) { return h->s->getLastIterations(); }
void* lama_slam_device_context(const lama_slam* h) { return (void*)h->s->deviceContext(); }

}   // extern "C"
namespace {
template <class M>
int view_bounds(const M* m, uint32_t* min3, uint32_t* max3, double* wmin3, double* wmax3)
{
    if (!m) return -1;
    Vector3ui a, b;
    Vector3d wa, wb;
    m->bounds(a, b);
    m->bounds(wa, wb);
    for (int k = 0; k < 3; ++k) { min3[k] = a(k); max3[k] = b(k); wmin3[k] = wa[k]; wmax3[k] = wb[k]; }
    return 0;
}
template <class M>
int64_t view_cells(const M* m, uint32_t* xy_out, uint64_t cap)
{
    if (!m) return -1;
    uint64_t n = 0;
    m->visit_all_cells([&](const Vector3ui& c) {
        if (n < cap) { xy_out[2 * n] = c(0); xy_out[2 * n + 1] = c(1); }
        ++n;
    });
    return (int64_t)n;
}
}
extern "C" {

int lama_slam_view_bounds(lama_slam* h, int which, uint32_t* min3, uint32_t* max3, double* wmin3, double* wmax3)
{
    try {
        return which == 0 ? view_bounds(h->s->getOccupancyMap(), min3, max3, wmin3, wmax3) : view_bounds(h->s->getDistanceMap(), min3, max3, wmin3, wmax3);
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}
int64_t lama_slam_view_cells(lama_slam* h, int which, uint32_t* xy_out, uint64_t cap)
{
    try {
        return which == 0 ? view_cells(h->s->getOccupancyMap(), xy_out, cap) : view_cells(h->s->getDistanceMap(), xy_out, cap);
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}
int lama_slam_view_occupancy(lama_slam* h, uint64_t n, const uint32_t* xy, uint8_t* is_free, uint8_t* is_occupied, uint8_t* is_unknown, double* probability)
{
    try {
        const FrequencyOccupancyMap* m = h->s->getOccupancyMap();
        if (!m) return -1;
        for (uint64_t i = 0; i < n; ++i) {
            const Vector3ui c(xy[2 * i], xy[2 * i + 1], 0);
            is_free[i] = m->isFree(c); is_occupied[i] = m->isOccupied(c); is_unknown[i] = m->isUnknown(c); probability[i] = m->getProbability(c);
        }
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}
int lama_slam_view_distance_cells(lama_slam* h, uint64_t n, const uint32_t* xy, double* distance)
{
    try {
        const DynamicDistanceMap* m = h->s->getDistanceMap();
        if (!m) return -1;
        for (uint64_t i = 0; i < n; ++i) distance[i] = m->distance(Vector3ui(xy[2 * i], xy[2 * i + 1], 0));
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}
static PointCloudXYZ::Ptr cloud_of(const double* pts, uint32_t n, const double* o, const double* q)
{
    PointCloudXYZ::Ptr c(new PointCloudXYZ);
    c->points.reserve(n);
    for (uint32_t i = 0; i < n; ++i) c->points.push_back(Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    if (o) c->sensor_origin_ = Vector3d(o[0], o[1], o[2]);
    if (q) c->sensor_orientation_ = Quaterniond(q[0], q[1], q[2], q[3]);
    return c;
}

int lama_slam_match_eval(lama_slam* h, const double* pts, uint32_t n, const double* o, const double* q, const double* pose4,
                         double* residuals, double* jacobian, double* rmse_out)
{
    try {
        const DynamicDistanceMap* dm = h->s->getDistanceMap();
        if (!dm) return -1;
        MatchSurface2D problem(dm, cloud_of(pts, n, o, q), SE2d::fromArray(pose4));
        VectorXd r;
        MatrixXd J;
        problem.eval(r, jacobian ? &J : nullptr);
        for (uint32_t i = 0; i < n; ++i) residuals[i] = r[i];
        if (jacobian) for (uint32_t c = 0; c < 3; ++c) for (uint32_t i = 0; i < n; ++i) jacobian[(size_t)c * n + i] = J(i, c);
        if (rmse_out) *rmse_out = problem.error();
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}

int lama_slam_match_solve(lama_slam* h, const double* pts, uint32_t n, const double* o, const double* q, double* pose4, const char* strategy,
                          const char* weight, double weight_param, uint32_t max_iterations, double* cov9, uint32_t* iterations)
{
    try {
        const DynamicDistanceMap* dm = h->s->getDistanceMap();
        if (!dm) return -1;
        MatchSurface2D problem(dm, cloud_of(pts, n, o, q), SE2d::fromArray(pose4));
        Solver::Options so;
        so.max_iterations = max_iterations;
        const std::string st(strategy ? strategy : "gn"), w(weight ? weight : "cauchy");
        if (st == "lm") so.strategy.reset(new LevenbergMarquard); else so.strategy.reset(new GaussNewton);
        if (w == "cauchy") so.robust_cost.reset(new CauchyWeight(weight_param));
        else if (w == "tukey") so.robust_cost.reset(new TukeyWeight(weight_param));
        else if (w == "huber") so.robust_cost.reset(new HuberWeight(weight_param));
        else so.robust_cost.reset(new UnitWeight);
        Solver solver(so);
        MatrixXd cov;
        try {
            solver.solve(problem, cov9 ? &cov : nullptr);
        } catch (const std::invalid_argument& e) { h->error = e.what(); return -3; }
        problem.getState().toArray(pose4);
        if (cov9) for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov9[3 * r + c] = cov(r, c);
        if (iterations) *iterations = solver.lastIterations();
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}

int lama_slam_view_distance_points(lama_slam* h, uint64_t n, const double* xy, double* out)
{
    try {
        const DynamicDistanceMap* m = h->s->getDistanceMap();
        if (!m) return -1;
        for (uint64_t i = 0; i < n; ++i) {
            Vector3d g;
            out[3 * i] = m->distance(Vector3d(xy[2 * i], xy[2 * i + 1], 0.0), &g);
            out[3 * i + 1] = g[0]; out[3 * i + 2] = g[1];
        }
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -2; }
}


// ------------------------------------------------------------------ Loc2D
struct lama_loc {
    Loc2D l;
    std::string error;
};

lama_loc* lama_loc_create(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t max_iter,
                          int32_t gpu_device, char* err, int errcap)
{
    auto* h = new lama_loc;
    try {
        Loc2D::Options o;
        o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.max_iter = max_iter;
        o.gpu_device = gpu_device;
        h->l.Init(o);
        return h;
    } catch (const std::exception& e) {
        if (err && errcap > 0) { std::strncpy(err, e.what(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
}
void lama_loc_destroy(lama_loc* l) { delete l; }
const char* lama_loc_last_error(const lama_loc* l) { return l ? l->error.c_str() : "null handle"; }
const char* lama_loc_engine_origin(const lama_loc* l) { return (l && l->l.engine()) ? l->l.engine()->origin.c_str() : ""; }
int lama_loc_set_obstacles_world(lama_loc* h, const double* xy, uint32_t n)
{
    try {
        for (uint32_t i = 0; i < n; ++i) {
            const Vector3ui c = h->l.distance_map->w2m(Vector3d(xy[2 * i], xy[2 * i + 1], 0.0));
            h->l.occupancy_map->setOccupied(c);
            h->l.distance_map->addObstacle(c);
        }
        h->l.distance_map->update();
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -1; }
}
void* lama_loc_device_context(const lama_loc* h) { return h ? (void*)h->l.deviceContext() : nullptr; }
int lama_loc_write_distance_map(lama_loc* h, const char* filename)
{
    try { return h->l.distance_map->write(filename) ? 0 : -1; } catch (const std::exception& e) { h->error = e.what(); return -1; }
}
int lama_loc_read_distance_map(lama_loc* h, const char* filename)
{
    try {
        if (!h->l.distance_map->read(filename)) { h->error = std::string("cannot read ") + filename; return -1; }
        return 0;
    } catch (const std::exception& e) { h->error = e.what(); return -1; }
}
void lama_loc_set_pose(lama_loc* h, double x, double y, double yaw) { h->l.setPose(Pose2D(x, y, yaw)); }
int lama_loc_get_pose(const lama_loc* h, double* pose4) { h->l.getPose().state.toArray(pose4); return 0; }
int lama_loc_update(lama_loc* h, const double* pts, uint32_t n, const double* origin3, const double* quat, const double* odom_xyr, double ts, int force)
{
    try {
        return h->l.update(make_cloud(pts, n, origin3, quat), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts, force != 0) ? 1 : 0;
    } catch (const std::exception& e) { h->error = e.what(); return -1; }
}
int lama_loc_covar(const lama_loc* h, double* out9)      // row-major, whatever lama::Matrix3d is (Eigen's is column-major)
{
    const lama::Matrix3d& c = h->l.getCovar();
    for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) out9[3 * r + k] = c(r, k);
    return 0;
}
double lama_loc_rmse(const lama_loc* h) { return h->l.getRMSE(); }
uint32_t lama_loc_iterations(const lama_loc* h) { return h->l.getLastIterations(); }

lama_loc* lama_loc_create2(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t max_iter,
                           uint32_t gloc_particles, uint32_t gloc_iters, double gloc_thresh, double cov_blend,
                           int32_t gpu_device, char* err, int errcap)
{
    auto* h = new lama_loc;
    try {
        Loc2D::Options o;
        o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.max_iter = max_iter;
        o.gloc_particles = gloc_particles; o.gloc_iters = gloc_iters; o.gloc_thresh = gloc_thresh; o.cov_blend = cov_blend;
        o.gpu_device = gpu_device;
        h->l.Init(o);
        return h;
    } catch (const std::exception& e) {
        if (err && errcap > 0) { std::strncpy(err, e.what(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
}
lama_loc* lama_loc_create3(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t max_iter,
                           uint32_t gloc_particles, uint32_t gloc_iters, double gloc_thresh, double cov_blend,
                           const char* strategy, int32_t gpu_device, char* err, int errcap)
{
    auto* h = new lama_loc;
    try {
        Loc2D::Options o;
        o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.max_iter = max_iter;
        o.gloc_particles = gloc_particles; o.gloc_iters = gloc_iters; o.gloc_thresh = gloc_thresh; o.cov_blend = cov_blend;
        o.strategy = strategy ? strategy : "gn";
        o.gpu_device = gpu_device;
        h->l.Init(o);
        return h;
    } catch (const std::exception& e) {
        if (err && errcap > 0) { std::strncpy(err, e.what(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
}
int lama_loc_occ_set_cells(lama_loc* h, const uint32_t* cells_xy, uint32_t n, int state)
{
    for (uint32_t i = 0; i < n; ++i) {
        const Vector3ui c(cells_xy[2 * i], cells_xy[2 * i + 1], 0);
        if (state < 0) h->l.occupancy_map->setFree(c); else if (state > 0) h->l.occupancy_map->setOccupied(c); else h->l.occupancy_map->setUnknown(c);
    }
    return 0;
}
int lama_loc_occ_bounds(const lama_loc* h, double* out6)
{
    Vector3d a, b;
    h->l.occupancy_map->bounds(a, b);
    for (int i = 0; i < 3; ++i) { out6[i] = a[i]; out6[3 + i] = b[i]; }
    return 0;
}
void lama_loc_trigger_global_localization(lama_loc* h) { h->l.triggerGlobalLocalization(); }
int lama_loc_global_localization_active(const lama_loc* h) { return h->l.globalLocalizationIsActive() ? 1 : 0; }
uint32_t lama_loc_gloc_candidates(const lama_loc* h, double* poses4, double* errors, uint32_t cap)
{
    const uint32_t n = (uint32_t)h->l.lastGlocErrors().size();
    for (uint32_t i = 0; i < n && i < cap; ++i) {
        for (int k = 0; k < 4; ++k) poses4[4 * i + k] = h->l.lastGlocPoses()[4 * i + k];
        errors[i] = h->l.lastGlocErrors()[i];
    }
    return n;
}
uint32_t lama_loc_sampling_likelihoods(const lama_loc* h, double* out, uint32_t cap)
{
    const uint32_t n = (uint32_t)h->l.lastSamplingLikelihoods().size();
    for (uint32_t i = 0; i < n && i < cap; ++i) out[i] = h->l.lastSamplingLikelihoods()[i];
    return n;
}
// ------------------------------------------------------------------ LidarOdometry2D
struct lama_lo {
    std::unique_ptr<LidarOdometry2D> l;
    std::string error;
};
lama_lo* lama_lo_create(double resolution, uint32_t max_iter, int32_t gpu_device, char* err, int errcap)
{
    auto* h = new lama_lo;
    try {
        LidarOdometry2D::Options o;
        o.resolution = resolution; o.max_iter = max_iter; o.gpu_device = gpu_device;
        h->l.reset(new LidarOdometry2D(o));
        return h;
    } catch (const std::exception& e) {
        if (err && errcap > 0) { std::strncpy(err, e.what(), (size_t)errcap - 1); err[errcap - 1] = 0; }
        delete h;
        return nullptr;
    }
}
void lama_lo_destroy(lama_lo* h) { delete h; }
const char* lama_lo_last_error(const lama_lo* h) { return h ? h->error.c_str() : "null handle"; }
const char* lama_lo_engine_origin(const lama_lo* h) { return (h && h->l->engine()) ? h->l->engine()->origin.c_str() : ""; }
int lama_lo_update(lama_lo* h, const double* pts, uint32_t n, const double* origin3, const double* quat, double ts)
{
    try {
        return h->l->update(make_cloud(pts, n, origin3, quat), ts) ? 1 : 0;
    } catch (const std::exception& e) { h->error = e.what(); return -1; }
}
int lama_lo_get_odom(const lama_lo* h, double* pose4) { h->l->odom.state.toArray(pose4); return 0; }
uint32_t lama_lo_iterations(const lama_lo* h) { return h->l->getLastIterations(); }
uint32_t lama_lo_deleted_patches(const lama_lo* h) { return h->l->getLastDeletedPatches(); }
void* lama_lo_device_context(const lama_lo* h) { return h->l->deviceContext(); }

static lama::sdm::HostMap host_map(int kind, double resolution, uint32_t max_sqdist, uint32_t n, const uint64_t* ids, const uint8_t* cells,
                                    const uint64_t* masks)
{
    lama::sdm::HostMap m;
    m.kind = (lama::sdm::MapKind)kind; m.resolution = resolution; m.max_sqdist = max_sqdist;
    const size_t pb = (size_t)1024 * m.cellSize();
    m.ids.assign(ids, ids + n);
    m.cells.assign(cells, cells + (size_t)n * pb);
    m.masks.assign(masks, masks + (size_t)n * 16);
    return m;
}
int lama_sdm_write(const char* file, int kind, double resolution, uint32_t max_sqdist, uint32_t n, const uint64_t* ids, const uint8_t* cells,
                   const uint64_t* masks)
{
    return lama::sdm::write(host_map(kind, resolution, max_sqdist, n, ids, cells, masks), file) ? 0 : -1;
}
int lama_sdm_read(const char* file, int* kind, double* resolution, uint32_t* max_sqdist, uint32_t cap, uint64_t* ids, uint8_t* cells,
                  uint64_t* masks, uint32_t* n)
{
    lama::sdm::HostMap m;
    if (!lama::sdm::read(m, file)) return -1;
    if (kind) *kind = (int)m.kind;
    if (resolution) *resolution = m.resolution;
    if (max_sqdist) *max_sqdist = m.max_sqdist;
    if (n) *n = (uint32_t)m.numPatches();
    if (cap >= m.numPatches() && ids && cells && masks) {
        std::memcpy(ids, m.ids.data(), m.ids.size() * 8);
        std::memcpy(cells, m.cells.data(), m.cells.size());
        std::memcpy(masks, m.masks.data(), m.masks.size() * 8);
    }
    return 0;
}
int lama_sdm_image(int kind, double resolution, uint32_t max_sqdist, uint32_t n, const uint64_t* ids, const uint8_t* cells,
                   const uint64_t* masks, uint32_t* width, uint32_t* height, uint8_t* out, uint64_t cap)
{
    lama::sdm::Image im;
    lama::sdm::build_image(host_map(kind, resolution, max_sqdist, n, ids, cells, masks), im);
    if (width) *width = im.width;
    if (height) *height = im.height;
    if (out && cap >= im.data.size()) std::memcpy(out, im.data.data(), im.data.size());
    return 0;
}
int lama_sdm_export_png(int kind, double resolution, uint32_t max_sqdist, uint32_t n, const uint64_t* ids, const uint8_t* cells,
                        const uint64_t* masks, const char* file)
{
    return lama::sdm::export_to_png(host_map(kind, resolution, max_sqdist, n, ids, cells, masks), file) ? 0 : -1;
}
static thread_local lama::sdm::HostMap g_dm_build;
int64_t lama_dm_build(const uint32_t* cells_xy, uint64_t n, uint32_t max_sqdist, uint32_t* processed)
{
    uint32_t done = 0;
    g_dm_build = lama::sdm::HostMap();
    g_dm_build.kind = lama::sdm::kDistanceMap; g_dm_build.max_sqdist = max_sqdist;
    try {
        if (!cells_xy || !lama::detail::build_distance_map(cells_xy, (size_t)n, max_sqdist, g_dm_build, done)) return -1;
    } catch (...) { return -1; }
    if (processed) *processed = done;
    return (int64_t)g_dm_build.numPatches();
}
int lama_dm_build_fetch(uint64_t* ids, uint8_t* cells, uint64_t* masks)
{
    if (!ids || !cells || !masks) return -1;
    std::memcpy(ids, g_dm_build.ids.data(), g_dm_build.ids.size() * 8);
    std::memcpy(cells, g_dm_build.cells.data(), g_dm_build.cells.size());
    std::memcpy(masks, g_dm_build.masks.data(), g_dm_build.masks.size() * 8);
    return 0;
}
void lama_random_set_seed(uint32_t seed) { lama::random::setSeed(seed); }
double lama_random_uniform(void) { return lama::random::uniform(); }

} // extern "C"
