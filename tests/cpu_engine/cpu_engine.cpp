// cpu_engine.cpp -- TEST DOUBLE: implements the device C-ABI of include/lama_hip.h on top of the CPU oracle.
//
// Purpose: exercise the HOST-side logic of the product (lama::PFSlam2D orchestration, RNG replay, sharding,
// the torch.distributed driver and its particle shipping) on machines without a GPU, e.g. the world_size-2
// gloo tests.  It lives under tests/, links the oracle, and is never shipped, built or loaded by the product:
// it is only reachable through lama_host_set_engine_library(), which exists only in the test-suite's own -DLAMA_TESTING build of the host library.
// "Device buffers" of export/import are plain host pointers here.
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/lama_hip.h"
#include "../../oracle/lama_oracle.hpp"

using namespace orc;

struct lama_hip_ctx {
    lama_hip_cfg cfg;
    std::string error;
    bool init = false;
    std::vector<SE2> poses;
    std::vector<std::shared_ptr<DynamicDistanceMap>> dm;
    std::vector<std::shared_ptr<FrequencyOccupancyMap>> occ;
    std::vector<std::shared_ptr<ProbabilisticOccupancyMap>> pocc;     // cfg.occupancy_policy == 1 (LidarOdometry2D)
    std::unique_ptr<PFSlam2D> tool;   // borrowed for scanMatch / updateParticleMaps bodies
    Scan last_scan;
    lama_hip_counters ctr;
};

namespace {
Scan make_scan(const double* pts, uint32_t n, const double* origin, const double* quat)
{
    Scan s;
    s.points.resize(n);
    for (uint32_t i = 0; i < n; ++i) s.points[i] = V3d{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    for (int i = 0; i < 3; ++i) s.sensor_origin[i] = origin ? origin[i] : 0.0;
    if (quat) for (int i = 0; i < 4; ++i) s.sensor_orientation[i] = quat[i];
    return s;
}
SE2 se2_of(const double* p) { SE2 s; s.c = p[0]; s.s = p[1]; s.tx = p[2]; s.ty = p[3]; return s; }
void se2_to(const SE2& s, double* p) { p[0] = s.c; p[1] = s.s; p[2] = s.tx; p[3] = s.ty; }

template <class M>
std::vector<uint64_t> sorted_ids(const M& m)
{
    std::vector<uint64_t> v;
    for (auto& kv : m.patches) v.push_back(kv.first);
    std::sort(v.begin(), v.end());
    return v;
}
} // namespace

extern "C" {

void lama_hip_default_cfg(lama_hip_cfg* cfg)
{
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->particles = 30; cfg->resolution = 0.05; cfg->patch_size = 32; cfg->l2_max = 0.5; cfg->meas_sigma = 0.05;
    cfg->max_iter = 100; cfg->window_patches = 128; cfg->dm_patch_capacity = 256; cfg->occ_patch_capacity = 256; cfg->queue_capacity = 32768;
}
int32_t lama_hip_device_count(int32_t* n) { if (n) *n = 1; return LAMA_HIP_OK; }      // one pretend device: contexts of a multi-shard object all land on it
const char* lama_hip_last_error(const lama_hip_ctx* c) { return c ? c->error.c_str() : "null"; }

int32_t lama_hip_ctx_create(const lama_hip_cfg* cfg, lama_hip_ctx** out)
{
    auto* c = new lama_hip_ctx;
    c->cfg = *cfg;
    std::memset(&c->ctr, 0, sizeof(c->ctr));
    PFOptions o;
    o.particles = cfg->particles; o.resolution = cfg->resolution; o.patch_size = cfg->patch_size; o.l2_max = cfg->l2_max;
    o.meas_sigma = cfg->meas_sigma; o.max_iter = cfg->max_iter; o.truncated_ray = cfg->truncated_ray; o.truncated_range = cfg->truncated_range;
    o.seed = 1;
    c->tool.reset(new PFSlam2D(o));
    c->poses.resize(cfg->particles);
    c->dm.resize(cfg->particles);
    c->occ.resize(cfg->particles);
    *out = c;
    return LAMA_HIP_OK;
}
void lama_hip_ctx_destroy(lama_hip_ctx* c) { delete c; }

int32_t lama_hip_pf_init(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin, const double* quat, const double* pose0)
{
    Scan s = make_scan(pts, n, origin, quat);
    if (c->cfg.occupancy_policy == 1) {          // one-"particle" log-odds context (LidarOdometry2D)
        c->last_scan = s;
        c->pocc.assign(1, std::make_shared<ProbabilisticOccupancyMap>(c->cfg.resolution, c->cfg.patch_size));
        c->dm[0] = std::make_shared<DynamicDistanceMap>(c->cfg.resolution, c->cfg.patch_size);
        c->dm[0]->setMaxDistance(c->cfg.l2_max);
        c->poses[0] = se2_of(pose0);
        double mn[3], mx[3];
        lidar_update_maps_body(*c->dm[0], *c->pocc[0], s, c->poses[0], mn, mx);
        c->init = true;
        return LAMA_HIP_OK;
    }
    c->tool->stage_set_scan(s);
    Particle p0;
    p0.pose = se2_of(pose0);
    p0.dm = std::make_shared<DynamicDistanceMap>(c->cfg.resolution, c->cfg.patch_size);
    p0.dm->setMaxDistance(c->cfg.l2_max);
    p0.occ = std::make_shared<FrequencyOccupancyMap>(c->cfg.resolution, c->cfg.patch_size);
    c->tool->updateParticleMaps(&p0);
    for (uint32_t i = 0; i < c->cfg.particles; ++i) {
        c->poses[i] = p0.pose;
        c->dm[i] = i == 0 ? p0.dm : std::make_shared<DynamicDistanceMap>(*p0.dm);
        c->occ[i] = i == 0 ? p0.occ : std::make_shared<FrequencyOccupancyMap>(*p0.occ);
    }
    c->init = true;
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_set_poses(lama_hip_ctx* c, const double* poses)
{
    for (uint32_t i = 0; i < c->cfg.particles; ++i) c->poses[i] = se2_of(poses + 4 * i);
    return LAMA_HIP_OK;
}
int32_t lama_hip_pf_get_poses(lama_hip_ctx* c, double* poses)
{
    for (uint32_t i = 0; i < c->cfg.particles; ++i) se2_to(c->poses[i], poses + 4 * i);
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_scan_match(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin, const double* quat,
                               double* poses_out, double* loglik_out, int32_t* iters_out)
{
    if (!c->init) { c->error = "scan_match before init"; return LAMA_HIP_E_STATE; }
    Scan s = make_scan(pts, n, origin, quat);
    c->last_scan = s;
    c->tool->stage_set_scan(s);
    for (uint32_t i = 0; i < c->cfg.particles; ++i) {
        Particle p;
        p.pose = c->poses[i]; p.dm = c->dm[i]; p.occ = c->occ[i];
        if (c->cfg.solver_strategy == 1) {               // Slam2D / Loc2D with strategy "lm"
            MatchSurface2D ms(p.dm.get(), &s, p.pose);
            CauchyWeight cauchy(0.15);
            const SolveStats st = solve_lm(ms, c->cfg.max_iter, cauchy);
            p.pose = ms.state_;
            p.ctr.iterations = st.iterations;
            p.weight = c->tool->calculateLikelihood(p);
        } else
        c->tool->scanMatch(&p);
        c->poses[i] = p.pose;
        if (poses_out) se2_to(p.pose, poses_out + 4 * i);
        if (loglik_out) loglik_out[i] = p.weight;
        if (iters_out) iters_out[i] = (int32_t)p.ctr.iterations;
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_resample(lama_hip_ctx* c, const int32_t* idx)
{
    const uint32_t P = c->cfg.particles;
    std::vector<SE2> np(P);
    std::vector<std::shared_ptr<DynamicDistanceMap>> nd(P);
    std::vector<std::shared_ptr<FrequencyOccupancyMap>> no(P);
    for (uint32_t i = 0; i < P; ++i) {
        np[i] = c->poses[idx[i]];
        nd[i] = std::make_shared<DynamicDistanceMap>(*c->dm[idx[i]]);
        no[i] = std::make_shared<FrequencyOccupancyMap>(*c->occ[idx[i]]);
    }
    c->poses.swap(np); c->dm.swap(nd); c->occ.swap(no);
    return LAMA_HIP_OK;
}

// (internal linkage: the exported names also exist in the real liblama_hip.so, which may be loaded in the same process;
// a call through the PLT could bind to that one)
static int32_t update_maps_impl(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin, const double* quat)
{
    Scan s = pts ? make_scan(pts, n, origin, quat) : c->last_scan;
    if (c->cfg.occupancy_policy == 1) {
        double mn[3], mx[3];
        lidar_update_maps_body(*c->dm[0], *c->pocc[0], s, c->poses[0], mn, mx);
        return LAMA_HIP_OK;
    }
    c->tool->stage_set_scan(s);
    for (uint32_t i = 0; i < c->cfg.particles; ++i) {
        Particle p;
        p.pose = c->poses[i]; p.dm = c->dm[i]; p.occ = c->occ[i];
        c->tool->updateParticleMaps(&p);
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_update_maps(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin, const double* quat)
{
    return update_maps_impl(c, pts, n, origin, quat);
}
int32_t lama_hip_pf_update_maps_begin(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin, const double* quat)
{
    return update_maps_impl(c, pts, n, origin, quat);
}
int32_t lama_hip_sync(lama_hip_ctx*) { return LAMA_HIP_OK; }
int32_t lama_hip_ctx_device(const lama_hip_ctx* c) { return c ? 0 : -1; }

int32_t lama_hip_pf_map_patches(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t* num)
{
    if (!c->init) { *num = 0; return LAMA_HIP_OK; }      // (no map yet, like the device library before its first scan / upload)
    *num = (uint32_t)(kind == LAMA_HIP_MAP_DISTANCE ? c->dm[particle]->patches.size()
                      : (c->cfg.occupancy_policy == 1 ? c->pocc[particle]->patches.size() : c->occ[particle]->patches.size()));
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_download_map(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t cap, uint64_t* ids, uint8_t* cells,
                                 uint64_t* masks, uint32_t* num)
{
    const Map& m = kind == LAMA_HIP_MAP_DISTANCE ? (const Map&)*c->dm[particle]
                   : (c->cfg.occupancy_policy == 1 ? (const Map&)*c->pocc[particle] : (const Map&)*c->occ[particle]);
    std::vector<uint64_t> v = sorted_ids(m);
    if (num) *num = (uint32_t)v.size();
    const size_t cb = (kind == LAMA_HIP_MAP_DISTANCE ? 10 : 4) * 1024;
    for (uint32_t k = 0; k < v.size() && k < cap; ++k) {
        const Container& ct = *m.patches.at(v[k]);
        if (ids) ids[k] = v[k];
        if (cells) std::memcpy(cells + k * cb, ct.data.data(), cb);
        if (masks) std::memcpy(masks + (size_t)k * 16, ct.mask.data(), 128);
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_upload_map(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t n, const uint64_t* ids, const uint8_t* cells, const uint64_t* masks)
{
    if (!c->init) {
        for (uint32_t i = 0; i < c->cfg.particles; ++i) {
            c->dm[i] = std::make_shared<DynamicDistanceMap>(c->cfg.resolution, c->cfg.patch_size);
            c->dm[i]->setMaxDistance(c->cfg.l2_max);
            c->occ[i] = std::make_shared<FrequencyOccupancyMap>(c->cfg.resolution, c->cfg.patch_size);
        }
        c->init = true;
    }
    Map& m = kind == LAMA_HIP_MAP_DISTANCE ? (Map&)*c->dm[particle] : (c->cfg.occupancy_policy == 1 ? (Map&)*c->pocc[particle] : (Map&)*c->occ[particle]);
    const size_t cb = (kind == LAMA_HIP_MAP_DISTANCE ? 10 : 4) * 1024;
    m.patches.clear();
    for (uint32_t k = 0; k < n; ++k) {
        auto ct = std::make_shared<Container>(5u, (uint32_t)(cb / 1024));
        ct->data.assign(cells + k * cb, cells + (k + 1) * cb);
        ct->mask.assign(masks + (size_t)k * 16, masks + (size_t)(k + 1) * 16);
        m.patches[ids[k]] = ct;
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_match_batch(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                             const double* poses, uint32_t B, double* out)
{
    Scan s = make_scan(pts, n, origin, quat);
    c->tool->stage_set_scan(s);
    for (uint32_t b = 0; b < B; ++b) {
        Particle p;
        p.pose = se2_of(poses + 4 * b); p.dm = c->dm[particle];
        out[b] = c->tool->calculateLikelihood(p);
    }
    return LAMA_HIP_OK;
}

// blob: [pose 4 f64][n_dm u64][n_occ u64] then per patch: id u64, cells, mask
int32_t lama_hip_pf_export_particle(lama_hip_ctx* c, uint32_t particle, void* buf, uint64_t cap, uint64_t* bytes)
{
    const auto dmi = sorted_ids(*c->dm[particle]);
    const auto oci = sorted_ids(*c->occ[particle]);
    const uint64_t need = 32 + 16 + dmi.size() * (8 + 10240 + 128) + oci.size() * (8 + 4096 + 128);
    if (bytes) *bytes = need;
    if (!buf) return LAMA_HIP_OK;
    if (cap < need) return LAMA_HIP_E_INVALID;
    uint8_t* o = (uint8_t*)buf;
    se2_to(c->poses[particle], (double*)o); o += 32;
    uint64_t cnt[2] = {dmi.size(), oci.size()};
    std::memcpy(o, cnt, 16); o += 16;
    for (uint64_t id : dmi) {
        const Container& ct = *c->dm[particle]->patches.at(id);
        std::memcpy(o, &id, 8); o += 8; std::memcpy(o, ct.data.data(), 10240); o += 10240; std::memcpy(o, ct.mask.data(), 128); o += 128;
    }
    for (uint64_t id : oci) {
        const Container& ct = *c->occ[particle]->patches.at(id);
        std::memcpy(o, &id, 8); o += 8; std::memcpy(o, ct.data.data(), 4096); o += 4096; std::memcpy(o, ct.mask.data(), 128); o += 128;
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_import_particle(lama_hip_ctx* c, uint32_t particle, const void* buf, uint64_t bytes)
{
    const uint8_t* in = (const uint8_t*)buf;
    (void)bytes;
    c->poses[particle] = se2_of((const double*)in); in += 32;
    uint64_t cnt[2];
    std::memcpy(cnt, in, 16); in += 16;
    auto dm = std::make_shared<DynamicDistanceMap>(c->cfg.resolution, c->cfg.patch_size);
    dm->setMaxDistance(c->cfg.l2_max);
    auto occ = std::make_shared<FrequencyOccupancyMap>(c->cfg.resolution, c->cfg.patch_size);
    for (uint64_t k = 0; k < cnt[0]; ++k) {
        uint64_t id; std::memcpy(&id, in, 8); in += 8;
        auto ct = std::make_shared<Container>(5, 10);
        std::memcpy(ct->data.data(), in, 10240); in += 10240; std::memcpy(ct->mask.data(), in, 128); in += 128;
        dm->patches[id] = ct;
    }
    for (uint64_t k = 0; k < cnt[1]; ++k) {
        uint64_t id; std::memcpy(&id, in, 8); in += 8;
        auto ct = std::make_shared<Container>(5, 4);
        std::memcpy(ct->data.data(), in, 4096); in += 4096; std::memcpy(ct->mask.data(), in, 128); in += 128;
        occ->patches[id] = ct;
    }
    c->dm[particle] = dm; c->occ[particle] = occ;
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_export_particles(lama_hip_ctx* c, uint32_t n, const uint32_t* particles, void* const* bufs, const uint64_t* caps, uint64_t* bytes)
{
    for (uint32_t k = 0; k < n; ++k) {
        uint64_t nb = 0;
        const int32_t rc = lama_hip_pf_export_particle(c, particles[k], bufs ? bufs[k] : nullptr, caps ? caps[k] : 0, &nb);
        if (rc) return rc;
        if (bytes) bytes[k] = nb;
    }
    return LAMA_HIP_OK;
}
int32_t lama_hip_pf_import_particles(lama_hip_ctx* c, uint32_t n, const uint32_t* particles, const void* const* bufs, const uint64_t* bytes)
{
    for (uint32_t k = 0; k < n; ++k) { const int32_t rc = lama_hip_pf_import_particle(c, particles[k], bufs[k], bytes[k]); if (rc) return rc; }
    return LAMA_HIP_OK;
}

// "device" staging buffers of the test double are host memory
int32_t lama_hip_blob_alloc(lama_hip_ctx*, uint64_t bytes, void** buf) { *buf = std::malloc(bytes); return *buf ? LAMA_HIP_OK : LAMA_HIP_E_HIP; }
int32_t lama_hip_blob_free(lama_hip_ctx*, void* buf) { std::free(buf); return LAMA_HIP_OK; }
int32_t lama_hip_blob_copy(lama_hip_ctx*, void* dst, lama_hip_ctx*, const void* src, uint64_t bytes) { std::memcpy(dst, src, bytes); return LAMA_HIP_OK; }

int32_t lama_hip_get_counters(lama_hip_ctx* c, lama_hip_counters* out)
{
    *out = c->ctr;
    uint64_t d = 0, o = 0;
    for (auto& m : c->dm) if (m) d += m->patches.size();
    for (auto& m : c->occ) if (m) o += m->patches.size();
    out->dm_patches = d; out->occ_patches = o;
    return LAMA_HIP_OK;
}
uint32_t lama_hip_counters_bytes(void) { return (uint32_t)sizeof(lama_hip_counters); }
int32_t lama_hip_get_counters_sized(lama_hip_ctx* c, void* out, uint32_t bytes)
{
    lama_hip_counters full;
    const int32_t rc = lama_hip_get_counters(c, &full);
    if (rc) return rc;
    full.struct_bytes = (uint32_t)sizeof(full);
    std::memcpy(out, &full, std::min<size_t>(bytes, sizeof(full)));
    return LAMA_HIP_OK;
}
int32_t lama_hip_reset_counters(lama_hip_ctx* c) { std::memset(&c->ctr, 0, sizeof(c->ctr)); return LAMA_HIP_OK; }


int32_t lama_hip_map_add_obstacles(lama_hip_ctx* c, uint32_t particle, const uint32_t* cells_xy, uint32_t n)
{
    if (!c->init) {
        for (uint32_t i = 0; i < c->cfg.particles; ++i) {
            c->dm[i] = std::make_shared<DynamicDistanceMap>(c->cfg.resolution, c->cfg.patch_size);
            c->dm[i]->setMaxDistance(c->cfg.l2_max);
            c->occ[i] = std::make_shared<FrequencyOccupancyMap>(c->cfg.resolution, c->cfg.patch_size);
        }
        c->init = true;
    }
    for (uint32_t k = 0; k < n; ++k) c->dm[particle]->addObstacle(V3u{cells_xy[2 * k], cells_xy[2 * k + 1], 0});
    c->dm[particle]->update();
    return LAMA_HIP_OK;
}

static int32_t match_solve_impl(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                                double* pose, double* out7, int32_t* iters, int32_t do_solve, int32_t strategy, uint32_t max_iterations);
int32_t lama_hip_match_solve(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                             double* pose, double* out7, int32_t* iters, int32_t do_solve)
{
    return match_solve_impl(c, particle, pts, n, origin, quat, pose, out7, iters, do_solve, (int32_t)c->cfg.solver_strategy, 0);
}
int32_t lama_hip_match_solve_with(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                                  double* pose, double* out7, int32_t* iters, int32_t strategy, uint32_t max_iterations)
{
    return match_solve_impl(c, particle, pts, n, origin, quat, pose, out7, iters, 1, strategy, max_iterations);
}
// MatchSurface2D::eval rows (residuals, column-major n x 3 Jacobian)
int32_t lama_hip_match_eval(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                            const double* pose, double* residuals, double* jacobian)
{
    Scan s = make_scan(pts, n, origin, quat);
    MatchSurface2D ms(c->dm[particle].get(), &s, se2_of(pose));
    std::vector<double> r, J;
    ms.eval(r, jacobian ? &J : nullptr);
    for (uint32_t i = 0; i < n; ++i) {
        residuals[i] = r[i];
        if (jacobian) { jacobian[i] = J[3 * i]; jacobian[(size_t)n + i] = J[3 * i + 1]; jacobian[2 * (size_t)n + i] = J[3 * i + 2]; }
    }
    return LAMA_HIP_OK;
}
// per-beam terms of MatchSurface2D::error: distance of the cell w2m(tf * p_i)
int32_t lama_hip_match_cell_distances(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                                      const double* pose, double* distances)
{
    Scan s = make_scan(pts, n, origin, quat);
    MatchSurface2D ms(c->dm[particle].get(), &s, se2_of(pose));
    ms.cell_distances(distances);
    return LAMA_HIP_OK;
}
static int32_t match_solve_impl(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                                double* pose, double* out7, int32_t* iters, int32_t do_solve, int32_t strategy, uint32_t max_iterations)
{
    const uint32_t max_iter = max_iterations ? max_iterations : c->cfg.max_iter;
    Scan s = make_scan(pts, n, origin, quat);
    MatchSurface2D ms(c->dm[particle].get(), &s, se2_of(pose));
    CauchyWeight cauchy(0.15);
    SolveStats st;
    if (do_solve) st = strategy == 1 ? solve_lm(ms, max_iter, cauchy) : solve_gn(ms, max_iter, cauchy);
    std::vector<double> r, J;
    ms.eval(r, &J);
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, s2 = 0;
    for (size_t i = 0; i < r.size(); ++i) {
        const double w = std::sqrt(cauchy.value(r[i]));
        const double j[3] = {J[3 * i] * w, J[3 * i + 1] * w, J[3 * i + 2] * w};
        for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a][b] += j[a] * j[b];
        s2 += r[i] * r[i];
    }
    se2_to(ms.state_, pose);
    if (out7) { out7[0] = A[0][0]; out7[1] = A[1][0]; out7[2] = A[1][1]; out7[3] = A[2][0]; out7[4] = A[2][1]; out7[5] = A[2][2]; out7[6] = s2; }
    if (iters) *iters = (int32_t)st.iterations;
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_patch_ids(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t cap, uint64_t* ids, uint32_t* num)
{
    const Map& m = kind == LAMA_HIP_MAP_DISTANCE ? (const Map&)*c->dm[particle]
                   : (c->cfg.occupancy_policy == 1 ? (const Map&)*c->pocc[particle] : (const Map&)*c->occ[particle]);
    std::vector<uint64_t> v = sorted_ids(m);
    if (num) *num = (uint32_t)v.size();
    if (ids) for (size_t k = 0; k < v.size() && k < cap; ++k) ids[k] = v[k];
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_delete_patches(lama_hip_ctx* c, uint32_t particle, const uint64_t* ids, uint32_t n, uint32_t* deleted)
{
    if (deleted) *deleted = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const V3u origin = c->dm[particle]->p2m(ids[k]);
        if (c->cfg.occupancy_policy == 1) c->pocc[particle]->deletePatchAt(origin); else c->occ[particle]->deletePatchAt(origin);
        if (c->dm[particle]->deletePatchAt(origin) && deleted) ++*deleted;
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_eval_batch(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin, const double* quat,
                            const double* poses, uint32_t B, double* sqnorm_out, double* loglik_out)
{
    Scan s = pts ? make_scan(pts, n, origin, quat) : c->last_scan;
    c->last_scan = s;
    if (loglik_out) c->tool->stage_set_scan(s);
    for (uint32_t b = 0; b < B; ++b) {
        if (sqnorm_out) {
            MatchSurface2D ms(c->dm[particle].get(), &s, se2_of(poses + 4 * b));
            std::vector<double> r;
            ms.eval(r, nullptr);
            double e = 0;
            for (double v : r) e += v * v;
            sqnorm_out[b] = e;
        }
        if (loglik_out) {
            Particle p;
            p.pose = se2_of(poses + 4 * b); p.dm = c->dm[particle];
            loglik_out[b] = c->tool->calculateLikelihood(p);
        }
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_map_sample_likelihood(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin,
                                       const double* quat, double yaw, const double* xy, uint32_t K, uint32_t point_step, double* l_out)
{
    Scan s = pts ? make_scan(pts, n, origin, quat) : c->last_scan;
    c->last_scan = s;
    const Affine3 mtf = moving_tf(s);
    double Raa[3][3];
    angle_axis_z(yaw, Raa);
    const DynamicDistanceMap& dm = *c->dm[particle];
    for (uint32_t k = 0; k < K; ++k) {
        const double trans[3] = {xy[2 * k], xy[2 * k + 1], 0.0};
        const Affine3 tf = affine_mul(affine_from(trans, Raa), mtf);
        double l = 0.0;
        for (size_t i = 0; i < s.points.size(); i += point_step) {
            const V3d hit = affine_apply(tf, s.points[i]);
            const double dist = dm.distance(dm.w2m(hit));
            const double e = std::exp(-(dist * dist) / 0.01);
            l += e * e * e;
        }
        l_out[k] = l;
    }
    return LAMA_HIP_OK;
}

} // extern "C"
