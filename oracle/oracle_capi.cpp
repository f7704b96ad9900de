// oracle_capi.cpp -- extern "C" surface of the CPU oracle (lama_oracle.hpp) for ctypes.
// TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.  Never linked into the product library.
#include "lama_oracle.hpp"

#include <cstdio>

using namespace orc;

namespace {

struct ScanBox {
    Scan scan;
};

Scan make_scan(const double* pts, int n, const double* origin, const double* quat)
{
    Scan s;
    s.points.resize(n);
    for (int i = 0; i < n; ++i) s.points[i] = V3d{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    for (int i = 0; i < 3; ++i) s.sensor_origin[i] = origin ? origin[i] : 0.0;
    if (quat) for (int i = 0; i < 4; ++i) s.sensor_orientation[i] = quat[i];
    return s;
}

SE2 se2_of(const double* p) { SE2 s; s.c = p[0]; s.s = p[1]; s.tx = p[2]; s.ty = p[3]; return s; }
void se2_to(const SE2& s, double* p) { p[0] = s.c; p[1] = s.s; p[2] = s.tx; p[3] = s.ty; }

template <class M>
int patch_ids(const M* m, uint64_t* ids, int cap)
{
    std::vector<uint64_t> v;
    for (auto& kv : m->patches) v.push_back(kv.first);
    std::sort(v.begin(), v.end());
    if (ids) for (size_t i = 0; i < v.size() && (int)i < cap; ++i) ids[i] = v[i];
    return (int)v.size();
}

template <class M>
int patch_read(const M* m, uint64_t id, uint8_t* cells, uint64_t* mask)
{
    auto it = m->patches.find(id);
    if (it == m->patches.end()) return -1;
    const Container& c = *it->second;
    if (cells) std::memcpy(cells, c.data.data(), c.data.size());
    if (mask) std::memcpy(mask, c.mask.data(), c.mask.size() * sizeof(uint64_t));
    return (int)c.data.size();
}

struct PFBox {
    std::unique_ptr<PFSlam2D> pf;
    Scan scan; // last scan (kept alive: PFSlam2D holds a pointer, like current_surface_)
};

} // namespace

extern "C" {

// ---------------------------------------------------------------- SE2 / misc KAT helpers
void orc_se2_exp(const double* v3, double* out4) { se2_to(se2_exp(v3[0], v3[1], v3[2]), out4); }
void orc_se2_mul(const double* a4, const double* b4, double* out4) { se2_to(se2_mul(se2_of(a4), se2_of(b4)), out4); }
void orc_se2_inverse(const double* a4, double* out4) { se2_to(se2_inverse(se2_of(a4)), out4); }
void orc_se2_from_xyr(double x, double y, double r, double* out4) { se2_to(se2_from_xyr(x, y, r), out4); }
double orc_se2_rotation(const double* a4) { return se2_rotation(se2_of(a4)); }
void orc_pose_minus(const double* a4, const double* b4, double* out4) { se2_to(pose_minus(se2_of(a4), se2_of(b4)), out4); }
double orc_cauchy(double param, double x) { return CauchyWeight(param).value(x); }
void orc_ldlt3_solve(const double* A9, const double* b3, double* x3)
{
    double A[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) A[i][j] = A9[3 * i + j];
    ldlt3_solve(A, b3, x3);
}
int orc_sizeof_distance_t() { return (int)sizeof(distance_t); }
int orc_sizeof_frequency() { return (int)sizeof(frequency); }

// fixed_tf * moving_tf as the 12 numbers [R row-major 9, t 3]
void orc_scan_tf(const double* pose4, const double* origin3, const double* quat4, double* out12)
{
    Scan s = make_scan(nullptr, 0, origin3, quat4);
    Affine3 tf = affine_mul(fixed_tf(se2_of(pose4)), moving_tf(s));
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out12[3 * i + j] = tf.R[i][j];
    for (int i = 0; i < 3; ++i) out12[9 + i] = tf.t[i];
}

// ---------------------------------------------------------------- Map addressing
void orc_w2m(double res, uint32_t patch_size, const double* p3, uint32_t* out3)
{
    Map m(res, 1, patch_size);
    V3u r = m.w2m(V3d{p3[0], p3[1], p3[2]});
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
uint64_t orc_m2p(double res, uint32_t patch_size, const uint32_t* c3)
{
    Map m(res, 1, patch_size);
    return m.m2p(V3u{c3[0], c3[1], c3[2]});
}
uint32_t orc_m2c(double res, uint32_t patch_size, const uint32_t* c3)
{
    Map m(res, 1, patch_size);
    return m.m2c(V3u{c3[0], c3[1], c3[2]});
}
int orc_compute_ray(const uint32_t* from3, const uint32_t* to3, uint32_t* out, int cap)
{
    Map m(0.05, 1, 32);
    int n = 0;
    m.computeRay(V3u{from3[0], from3[1], from3[2]}, V3u{to3[0], to3[1], to3[2]}, [&](const V3u& c) {
        if (n < cap) { out[3 * n] = c.x; out[3 * n + 1] = c.y; out[3 * n + 2] = c.z; }
        ++n;
    });
    return n;
}

// ---------------------------------------------------------------- DynamicDistanceMap
void* orc_dm_new(double res, uint32_t patch_size, double l2_max)
{
    auto* dm = new DynamicDistanceMap(res, patch_size);
    dm->setMaxDistance(l2_max);
    return dm;
}
void* orc_dm_clone(void* h) { return new DynamicDistanceMap(*(DynamicDistanceMap*)h); }
void orc_dm_free(void* h) { delete (DynamicDistanceMap*)h; }
uint32_t orc_dm_max_sqdist(void* h) { return ((DynamicDistanceMap*)h)->max_sqdist(); }
void orc_dm_add_obstacle(void* h, uint32_t x, uint32_t y, uint32_t z) { ((DynamicDistanceMap*)h)->addObstacle(V3u{x, y, z}); }
void orc_dm_remove_obstacle(void* h, uint32_t x, uint32_t y, uint32_t z) { ((DynamicDistanceMap*)h)->removeObstacle(V3u{x, y, z}); }
uint32_t orc_dm_update(void* h) { return ((DynamicDistanceMap*)h)->update(); }
void orc_set_canonical_default(int on) { canonical_default() = on != 0; }
void orc_dm_set_canonical(void* h, int on) { ((DynamicDistanceMap*)h)->canonical = on != 0; }
double orc_dm_distance_cell(void* h, uint32_t x, uint32_t y, uint32_t z)
{
    return ((const DynamicDistanceMap*)h)->distance(V3u{x, y, z});
}
double orc_dm_distance(void* h, const double* p3, double* grad3)
{
    V3d g{0, 0, 0};
    double d = ((const DynamicDistanceMap*)h)->distance(V3d{p3[0], p3[1], p3[2]}, grad3 ? &g : nullptr);
    if (grad3) { grad3[0] = g.x; grad3[1] = g.y; grad3[2] = g.z; }
    return d;
}
int orc_dm_patch_ids(void* h, uint64_t* ids, int cap) { return patch_ids((const DynamicDistanceMap*)h, ids, cap); }
int orc_dm_patch_read(void* h, uint64_t id, uint8_t* cells, uint64_t* mask) { return patch_read((const DynamicDistanceMap*)h, id, cells, mask); }
void orc_dm_stats(void* h, uint64_t* out6 /* 7 values */)
{
    const BrushfireStats& s = ((DynamicDistanceMap*)h)->stats;
    out6[0] = s.raise_pops; out6[1] = s.lower_pops; out6[2] = s.lower_fired; out6[3] = s.pushes; out6[4] = s.max_queue; out6[5] = s.tie_overwrites; out6[6] = s.max_queue_last;
}

// ---------------------------------------------------------------- FrequencyOccupancyMap
void* orc_occ_new(double res, uint32_t patch_size) { return new FrequencyOccupancyMap(res, patch_size); }
void* orc_occ_clone(void* h) { return new FrequencyOccupancyMap(*(FrequencyOccupancyMap*)h); }
void orc_occ_free(void* h) { delete (FrequencyOccupancyMap*)h; }
int orc_occ_set_free(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((FrequencyOccupancyMap*)h)->setFree(V3u{x, y, z}) ? 1 : 0; }
int orc_occ_set_occupied(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((FrequencyOccupancyMap*)h)->setOccupied(V3u{x, y, z}) ? 1 : 0; }
double orc_occ_probability(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((const FrequencyOccupancyMap*)h)->getProbability(V3u{x, y, z}); }
// bit 0 isFree, bit 1 isOccupied, bit 2 isUnknown
int orc_occ_state(void* h, uint32_t x, uint32_t y, uint32_t z)
{
    const FrequencyOccupancyMap* m = (const FrequencyOccupancyMap*)h;
    return (m->isFree(V3u{x, y, z}) ? 1 : 0) | (m->isOccupied(V3u{x, y, z}) ? 2 : 0) | (m->isUnknown(V3u{x, y, z}) ? 4 : 0);
}
// Map::bounds (cells and world) and Map::visit_all_cells of either map class
static void map_bounds(const Map* m, uint32_t* mn3, uint32_t* mx3, double* wmn3, double* wmx3)
{
    V3u a, b; V3d wa, wb;
    m->bounds(a, b); m->bounds(wa, wb);
    mn3[0] = a.x; mn3[1] = a.y; mn3[2] = a.z; mx3[0] = b.x; mx3[1] = b.y; mx3[2] = b.z;
    wmn3[0] = wa.x; wmn3[1] = wa.y; wmn3[2] = wa.z; wmx3[0] = wb.x; wmx3[1] = wb.y; wmx3[2] = wb.z;
}
static int64_t map_cells(const Map* m, uint32_t* xy, uint64_t cap)
{
    uint64_t n = 0;
    m->visit_all_cells([&](const V3u& c) { if (n < cap) { xy[2 * n] = c.x; xy[2 * n + 1] = c.y; } ++n; });
    return (int64_t)n;
}
void orc_occ_bounds(void* h, uint32_t* mn3, uint32_t* mx3, double* wmn3, double* wmx3) { map_bounds((const FrequencyOccupancyMap*)h, mn3, mx3, wmn3, wmx3); }
void orc_dm_bounds(void* h, uint32_t* mn3, uint32_t* mx3, double* wmn3, double* wmx3) { map_bounds((const DynamicDistanceMap*)h, mn3, mx3, wmn3, wmx3); }
int64_t orc_occ_cells(void* h, uint32_t* xy, uint64_t cap) { return map_cells((const FrequencyOccupancyMap*)h, xy, cap); }
int64_t orc_dm_cells(void* h, uint32_t* xy, uint64_t cap) { return map_cells((const DynamicDistanceMap*)h, xy, cap); }
int orc_occ_patch_ids(void* h, uint64_t* ids, int cap) { return patch_ids((const FrequencyOccupancyMap*)h, ids, cap); }
int orc_occ_patch_read(void* h, uint64_t id, uint8_t* cells, uint64_t* mask) { return patch_read((const FrequencyOccupancyMap*)h, id, cells, mask); }

// ---------------------------------------------------------------- single-problem solve / eval
// eval: residuals (n) and optional J (n x 3 row-major)
void orc_eval(void* dm, const double* pts, int n, const double* origin3, const double* quat4,
              const double* pose4, double* r_out, double* J_out)
{
    Scan s = make_scan(pts, n, origin3, quat4);
    MatchSurface2D ms((const DynamicDistanceMap*)dm, &s, se2_of(pose4));
    std::vector<double> r, J;
    ms.eval(r, J_out ? &J : nullptr);
    std::memcpy(r_out, r.data(), sizeof(double) * n);
    if (J_out) std::memcpy(J_out, J.data(), sizeof(double) * 3 * n);
}
// Solve(GN, Cauchy(0.15), max_iter) -> pose (in/out), returns iterations; evals_out optional
int orc_solve(void* dm, const double* pts, int n, const double* origin3, const double* quat4,
              double* pose4, uint32_t max_iter, uint32_t* evals_out)
{
    Scan s = make_scan(pts, n, origin3, quat4);
    MatchSurface2D ms((const DynamicDistanceMap*)dm, &s, se2_of(pose4));
    CauchyWeight cauchy(0.15);
    SolveStats st = solve_gn(ms, max_iter, cauchy);
    se2_to(ms.state_, pose4);
    if (evals_out) *evals_out = st.evals;
    return (int)st.iterations;
}
// Solve(GaussNewton | LevenbergMarquard, Cauchy(0.15)) with the covariance branch of Solver::solve (solver.cpp:109-116, 133-150)
int orc_solve_full(void* dm, const double* pts, int n, const double* origin3, const double* quat4, double* pose4, uint32_t max_iter,
                   int lm, double* cov9)
{
    Scan s = make_scan(pts, n, origin3, quat4);
    MatchSurface2D ms((const DynamicDistanceMap*)dm, &s, se2_of(pose4));
    CauchyWeight cauchy(0.15);
    SolveStats st = lm ? solve_lm(ms, max_iter, cauchy) : solve_gn(ms, max_iter, cauchy);
    se2_to(ms.state_, pose4);
    if (cov9) {
        std::vector<double> r, J;
        ms.eval(r, &J);
        double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (size_t i = 0; i < r.size(); ++i) {
            const double w = std::sqrt(cauchy.value(r[i]));
            J[3 * i] *= w; J[3 * i + 1] *= w; J[3 * i + 2] *= w;
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a][b] += J[3 * i + a] * J[3 * i + b];
        }
        if (colpiv_qr_rank(J, r.size()) == 3) inverse3(A, cov9); else svd_cov3(J, r.size(), cov9);
    }
    return (int)st.iterations;
}
// MatchSurface2D::error (match_surface_2d.cpp:92-116)
double orc_match_error(void* dm, const double* pts, int n, const double* origin3, const double* quat4, const double* pose4)
{
    Scan s = make_scan(pts, n, origin3, quat4);
    MatchSurface2D ms((const DynamicDistanceMap*)dm, &s, se2_of(pose4));
    return ms.error();
}
// PFSlam2D::calculateLikelihood (pf_slam2d.cpp:393-414)
double orc_loglik(void* dm, const double* pts, int n, const double* origin3, const double* quat4,
                  const double* pose4, double meas_sigma)
{
    Scan s = make_scan(pts, n, origin3, quat4);
    Affine3 tf = affine_mul(fixed_tf(se2_of(pose4)), moving_tf(s));
    double l = 0;
    for (int i = 0; i < n; ++i) {
        V3d hit = affine_apply(tf, s.points[i]);
        double d = ((const DynamicDistanceMap*)dm)->distance(hit, nullptr);
        l += -(d * d) / meas_sigma;
    }
    return l;
}

// ---------------------------------------------------------------- PFSlam2D
struct orc_pf_options {
    uint32_t particles;
    double srr, str, stt, srt;
    double meas_sigma, meas_sigma_gain;
    double trans_thresh, rot_thresh;
    double l2_max, truncated_ray, truncated_range, resolution;
    uint32_t patch_size, max_iter;
    int32_t threads;
    uint32_t seed;
};

void orc_pf_default_options(orc_pf_options* o)
{
    PFOptions d;
    o->particles = d.particles; o->srr = d.srr; o->str = d.str; o->stt = d.stt; o->srt = d.srt;
    o->meas_sigma = d.meas_sigma; o->meas_sigma_gain = d.meas_sigma_gain;
    o->trans_thresh = d.trans_thresh; o->rot_thresh = d.rot_thresh;
    o->l2_max = d.l2_max; o->truncated_ray = d.truncated_ray; o->truncated_range = d.truncated_range;
    o->resolution = d.resolution; o->patch_size = d.patch_size; o->max_iter = d.max_iter;
    o->threads = d.threads; o->seed = d.seed;
}

void* orc_pf_new(const orc_pf_options* o)
{
    PFOptions p;
    p.particles = o->particles; p.srr = o->srr; p.str = o->str; p.stt = o->stt; p.srt = o->srt;
    p.meas_sigma = o->meas_sigma; p.meas_sigma_gain = o->meas_sigma_gain;
    p.trans_thresh = o->trans_thresh; p.rot_thresh = o->rot_thresh;
    p.l2_max = o->l2_max; p.truncated_ray = o->truncated_ray; p.truncated_range = o->truncated_range;
    p.resolution = o->resolution; p.patch_size = o->patch_size; p.max_iter = o->max_iter;
    p.threads = o->threads; p.seed = o->seed;
    auto* b = new PFBox;
    b->pf.reset(new PFSlam2D(p));
    return b;
}
void orc_pf_free(void* h) { delete (PFBox*)h; }
void orc_pf_set_prior(void* h, const double* pose4) { ((PFBox*)h)->pf->setPrior(se2_of(pose4)); }
void orc_pf_count_touches(void* h, int on) { ((PFBox*)h)->pf->count_touches = on != 0; }

int orc_pf_update(void* h, const double* pts, int n, const double* origin3, const double* quat4,
                  const double* odom4, double ts)
{
    PFBox* b = (PFBox*)h;
    b->scan = make_scan(pts, n, origin3, quat4);
    return b->pf->update(b->scan, se2_of(odom4), ts) ? 1 : 0;
}
void orc_pf_times(void* h, double* out5, int* resampled)
{
    const UpdateTimes& t = ((PFBox*)h)->pf->last_times;
    out5[0] = t.total; out5[1] = t.solving; out5[2] = t.normalizing; out5[3] = t.resampling; out5[4] = t.mapping;
    if (resampled) *resampled = t.resampled ? 1 : 0;
}
uint32_t orc_pf_num_resamples(void* h) { return ((PFBox*)h)->pf->num_resamples; }
double orc_pf_neff(void* h) { return ((PFBox*)h)->pf->getNeff(); }
int orc_pf_best(void* h) { return (int)((PFBox*)h)->pf->getBestParticleIdx(); }
void orc_pf_get_poses(void* h, double* out /*P x 4*/)
{
    auto& ps = ((PFBox*)h)->pf->particles();
    for (size_t i = 0; i < ps.size(); ++i) se2_to(ps[i].pose, out + 4 * i);
}
void orc_pf_set_poses(void* h, const double* in /*P x 4*/)
{
    auto& ps = ((PFBox*)h)->pf->particles();
    for (size_t i = 0; i < ps.size(); ++i) ps[i].pose = se2_of(in + 4 * i);
}
void orc_pf_get_weights(void* h, double* weight, double* nweight, double* weight_sum)
{
    auto& ps = ((PFBox*)h)->pf->particles();
    for (size_t i = 0; i < ps.size(); ++i) {
        if (weight) weight[i] = ps[i].weight;
        if (nweight) nweight[i] = ps[i].normalized_weight;
        if (weight_sum) weight_sum[i] = ps[i].weight_sum;
    }
}
void orc_pf_set_weights(void* h, const double* weight, const double* weight_sum)
{
    auto& ps = ((PFBox*)h)->pf->particles();
    for (size_t i = 0; i < ps.size(); ++i) {
        if (weight) ps[i].weight = weight[i];
        if (weight_sum) ps[i].weight_sum = weight_sum[i];
    }
}
void* orc_pf_particle_dm(void* h, int i) { return ((PFBox*)h)->pf->particles()[i].dm.get(); }     // borrowed
void* orc_pf_particle_occ(void* h, int i) { return ((PFBox*)h)->pf->particles()[i].occ.get(); }   // borrowed
// per-particle counters of the last update: [iterations, evals, ray_cells, occ_events, bf_processed,
//                                           n_match, n_occ, n_bf, n_match_or_bf]
void orc_pf_counters(void* h, int i, uint64_t* out9)
{
    const ParticleCounters& c = ((PFBox*)h)->pf->particles()[i].ctr;
    out9[0] = c.iterations; out9[1] = c.evals; out9[2] = c.ray_cells; out9[3] = c.occ_events; out9[4] = c.bf_processed;
    out9[5] = c.n_match; out9[6] = c.n_occ; out9[7] = c.n_bf; out9[8] = c.n_match_or_bf;
}
// The checksum lama_hip_pf_map_checksums computes on the device (iris_lama_amd/csrc/lama_kernels.h, k_map_checksum), from the
// checker's maps: sum mod 2^64 over the patches of mix(id ^ c0) and over their cells of mix((id * 1024 + cell) ^ mix(fields)),
// fields = the cell's stored values in the device's packing + the Container mask bit << 48; all-zero cells add nothing.
static uint64_t cks_mix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
}   // extern "C"
template <class FIELDS>
static uint64_t map_checksum(const Map* m, FIELDS fields)
{
    uint64_t acc = 0;
    for (auto& kv : m->patches) {
        const uint64_t id = kv.first;
        acc += cks_mix(id ^ 0x5DEECE66Dull);
        const Container& c = *kv.second;
        for (uint32_t cell = 0; cell < 1024u; ++cell) {
            const uint64_t f = fields(c, cell) | (((c.mask[cell >> 6] >> (cell & 63)) & 1ull) << 48);
            if (f) acc += cks_mix((id * 1024ull + cell) ^ cks_mix(f));
        }
    }
    return acc;
}
extern "C" {
void orc_pf_map_checksums(void* h, int kind /*0 distance, 1 occupancy, 2 distance in the wide library's packing*/, uint64_t* out)
{
    auto& ps = ((PFBox*)h)->pf->particles();
    for (size_t i = 0; i < ps.size(); ++i) {
        if (kind == 0)
            out[i] = map_checksum(ps[i].dm.get(), [](const Container& c, uint32_t cell) {
                distance_t d;
                std::memcpy(&d, c.data.data() + (size_t)cell * sizeof(distance_t), sizeof(d));
                const uint64_t sv = (uint64_t)(d.sqdist & 0x3FFFu) | (d.valid_obstacle ? 0x8000ull : 0ull) | (d.is_queued ? 0x4000ull : 0ull);
                const uint64_t ob = (uint64_t)(uint16_t)d.obstacle[0] | ((uint64_t)(uint16_t)d.obstacle[1] << 16);
                return sv | (ob << 16);
            });
        else if (kind == 2)     // distance map as liblama_hip_wide.so packs it (l2_max beyond 127 cells: sqdist takes all 16 low bits,
                                // the two flags sit above the mask bit)
            out[i] = map_checksum(ps[i].dm.get(), [](const Container& c, uint32_t cell) {
                distance_t d;
                std::memcpy(&d, c.data.data() + (size_t)cell * sizeof(distance_t), sizeof(d));
                const uint64_t ob = (uint64_t)(uint16_t)d.obstacle[0] | ((uint64_t)(uint16_t)d.obstacle[1] << 16);
                return (uint64_t)d.sqdist | (ob << 16) | (d.valid_obstacle ? 1ull << 49 : 0ull) | (d.is_queued ? 1ull << 50 : 0ull);
            });
        else
            out[i] = map_checksum(ps[i].occ.get(), [](const Container& c, uint32_t cell) {
                uint16_t ov[2];                                   // frequency { occupied, visited }
                std::memcpy(ov, c.data.data() + (size_t)cell * 4, 4);
                return (uint64_t)ov[0] | ((uint64_t)ov[1] << 16);
            });
    }
}
int orc_pf_last_sample_idx(void* h, int32_t* out, int cap)
{
    auto& v = ((PFBox*)h)->pf->last_sample_idx;
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) out[i] = v[i];
    return (int)v.size();
}

// stage-wise (teacher-forced) entry points
void orc_pf_stage_set_scan(void* h, const double* pts, int n, const double* origin3, const double* quat4)
{
    PFBox* b = (PFBox*)h;
    b->scan = make_scan(pts, n, origin3, quat4);
    b->pf->stage_set_scan(b->scan);
}
void orc_pf_stage_scan_match(void* h) { ((PFBox*)h)->pf->stage_scan_match_all(); }
void orc_pf_stage_update_maps(void* h) { ((PFBox*)h)->pf->stage_update_maps_all(); }
double orc_pf_stage_normalize(void* h) { ((PFBox*)h)->pf->stage_normalize(); return ((PFBox*)h)->pf->getNeff(); }
void orc_pf_stage_resample_indices(void* h, double u01, int32_t* out)
{
    auto v = ((PFBox*)h)->pf->stage_resample_indices(u01);
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
}
void orc_pf_stage_resample_with(void* h, const int32_t* idx, int n)
{
    std::vector<int32_t> v(idx, idx + n);
    ((PFBox*)h)->pf->stage_resample_with(v);
}
// motion model on an explicit pose (uses and advances the PF's RNG): pose4 in/out
void orc_pf_draw_from_motion(void* h, const double* delta4, double* pose4)
{
    SE2 p = se2_of(pose4);
    ((PFBox*)h)->pf->drawFromMotion(se2_of(delta4), p);
    se2_to(p, pose4);
}


// ---------------------------------------------------------------- Slam2D
struct SlamBox { std::unique_ptr<Slam2D> s; Scan scan; };
void* orc_slam_new(double trans_thresh, double rot_thresh, double l2_max, double truncated_ray, double truncated_range,
                   double resolution, uint32_t patch_size, uint32_t max_iter)
{
    SlamOptions o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.truncated_ray = truncated_ray;
    o.truncated_range = truncated_range; o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter;
    auto* b = new SlamBox;
    b->s.reset(new Slam2D(o));
    return b;
}
void* orc_slam_new2(double trans_thresh, double rot_thresh, double l2_max, double truncated_ray, double truncated_range,
                    double resolution, uint32_t patch_size, uint32_t max_iter, int transient_map)
{
    SlamOptions o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.truncated_ray = truncated_ray;
    o.truncated_range = truncated_range; o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter;
    o.transient_map = transient_map != 0;
    auto* b = new SlamBox;
    b->s.reset(new Slam2D(o));
    return b;
}
void orc_slam_set_lm(void* h, int on) { ((SlamBox*)h)->s->set_lm(on != 0); }
void orc_loc_set_lm(void* h, int on);
uint32_t orc_slam_deleted_last(void* h) { return ((SlamBox*)h)->s->deleted_last; }
void orc_slam_free(void* h) { delete (SlamBox*)h; }
void orc_slam_set_pose(void* h, const double* pose4) { ((SlamBox*)h)->s->setPose(se2_of(pose4)); }
void orc_slam_get_pose(void* h, double* pose4) { se2_to(((SlamBox*)h)->s->getPose(), pose4); }
int orc_slam_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, const double* odom4, double ts)
{
    SlamBox* b = (SlamBox*)h;
    b->scan = make_scan(pts, n, origin3, quat4);
    return b->s->update(b->scan, se2_of(odom4), ts) ? 1 : 0;
}
int orc_slam_enough_motion(void* h, const double* odom4) { return ((SlamBox*)h)->s->enoughMotion(se2_of(odom4)) ? 1 : 0; }
uint32_t orc_slam_processed_cells(void* h) { return ((SlamBox*)h)->s->getNumberOfProcessedCells(); }
uint32_t orc_slam_iterations(void* h) { return ((SlamBox*)h)->s->last_solve.iterations; }
void* orc_slam_dm(void* h) { return &((SlamBox*)h)->s->dm(); }       // borrowed
void* orc_slam_occ(void* h) { return &((SlamBox*)h)->s->occ(); }     // borrowed


// ---------------------------------------------------------------- Loc2D
struct LocBox { std::unique_ptr<Loc2D> l; Scan scan; };
void* orc_loc_new(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t patch_size, uint32_t max_iter)
{
    LocOptions o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter;
    auto* b = new LocBox;
    b->l.reset(new Loc2D(o));
    return b;
}
void orc_loc_free(void* h) { delete (LocBox*)h; }
void* orc_loc_dm(void* h) { return &((LocBox*)h)->l->dm(); }            // borrowed: fill with orc_dm_add_obstacle + orc_dm_update
void orc_loc_set_pose(void* h, const double* pose4) { ((LocBox*)h)->l->setPose(se2_of(pose4)); }
void orc_loc_get_pose(void* h, double* pose4) { se2_to(((LocBox*)h)->l->getPose(), pose4); }
int orc_loc_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, const double* odom4, double ts, int force)
{
    LocBox* b = (LocBox*)h;
    b->scan = make_scan(pts, n, origin3, quat4);
    return b->l->update(b->scan, se2_of(odom4), ts, force != 0) ? 1 : 0;
}
void orc_loc_covar(void* h, double* out9) { std::memcpy(out9, ((LocBox*)h)->l->getCovar(), 72); }
double orc_loc_rmse(void* h) { return ((LocBox*)h)->l->getRMSE(); }
uint32_t orc_loc_iterations(void* h) { return ((LocBox*)h)->l->lastIterations(); }
int32_t orc_loc_rank_deficient(void* h) { return ((LocBox*)h)->l->rankDeficient() ? 1 : 0; }
// global localisation / sampling covariance (src/loc2d.cpp:194-286)
void* orc_loc_new2(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t patch_size, uint32_t max_iter,
                   uint32_t gloc_particles, uint32_t gloc_iters, double gloc_thresh, double cov_blend)
{
    LocOptions o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter;
    o.gloc_particles = gloc_particles; o.gloc_iters = gloc_iters; o.gloc_thresh = gloc_thresh; o.cov_blend = cov_blend;
    auto* b = new LocBox;
    b->l.reset(new Loc2D(o));
    return b;
}
// ---- map formats (src/sdm/map.cpp:489-575, src/sdm/export.cpp:46-95); kind 0 = DynamicDistanceMap, 1 = FrequencyOccupancyMap
int orc_map_write(void* map, const char* file) { return ((Map*)map)->write(file) ? 1 : 0; }
int orc_map_read(void* map, const char* file) { return ((Map*)map)->read(file) ? 1 : 0; }
int orc_map_image(void* map, int kind, uint32_t* w, uint32_t* h, uint8_t* out, uint64_t cap)
{
    ExportImage im;
    if (kind == 0) build_image_dm(*(DynamicDistanceMap*)map, im);
    else build_image_occ(*(FrequencyOccupancyMap*)map, im);
    *w = im.width; *h = im.height;
    if (out && cap >= im.data.size()) std::memcpy(out, im.data.data(), im.data.size());
    return 1;
}
// ---- LidarOdometry2D (src/lidar_odometry_2d.cpp) on a ProbabilisticOccupancyMap
struct LoBox { std::unique_ptr<LidarOdometry2D> l; Scan scan; };
void* orc_lo_new(double resolution, uint32_t max_iter) { auto* b = new LoBox; b->l.reset(new LidarOdometry2D(resolution, max_iter)); return b; }
void orc_lo_free(void* h) { delete (LoBox*)h; }
int orc_lo_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, double ts)
{
    LoBox* b = (LoBox*)h;
    b->scan = make_scan(pts, n, origin3, quat4);
    return b->l->update(b->scan, ts) ? 1 : 0;
}
void orc_lo_get_odom(void* h, double* pose4) { se2_to(((LoBox*)h)->l->odom, pose4); }
void orc_lo_set_odom(void* h, const double* pose4) { ((LoBox*)h)->l->odom = se2_of(pose4); }
void* orc_lo_dm(void* h) { return &((LoBox*)h)->l->dm(); }
void* orc_lo_occ(void* h) { return &((LoBox*)h)->l->occ(); }
uint32_t orc_lo_deleted_last(void* h) { return ((LoBox*)h)->l->deleted_last; }
uint32_t orc_lo_map_updates(void* h) { return ((LoBox*)h)->l->map_updates; }
uint32_t orc_lo_iterations(void* h) { return ((LoBox*)h)->l->last_solve.iterations; }
int orc_pocc_patch_ids(void* h, uint64_t* ids, int cap) { return patch_ids((const ProbabilisticOccupancyMap*)h, ids, cap); }
int orc_pocc_patch_read(void* h, uint64_t id, uint8_t* cells, uint64_t* mask) { return patch_read((const ProbabilisticOccupancyMap*)h, id, cells, mask); }
void orc_pocc_free(void*) {}
void orc_pocc_params(double* out5)
{
    ProbabilisticOccupancyMap m(0.05);
    out5[0] = m.miss_; out5[1] = m.hit_; out5[2] = m.clamp_min_; out5[3] = m.clamp_max_; out5[4] = m.occ_thresh_;
}

// ---- SE2 pose-graph linearisation (minisam linearzationLowerHessian restated, see lama_oracle.hpp)
void orc_pgo_linearize(const double* poses4, uint32_t N, const int32_t* fi, const int32_t* fj, const double* meas4, const double* sqrt_info3,
                       uint32_t F, double* err, double* Hdiag, double* Hoff, double* b, double* chi2)
{
    std::vector<SE2> x(N);
    for (uint32_t i = 0; i < N; ++i) x[i] = se2_of(poses4 + 4 * i);
    std::vector<PgoFactor> fs(F);
    for (uint32_t k = 0; k < F; ++k) {
        fs[k].i = fi[k]; fs[k].j = fj[k]; fs[k].meas = se2_of(meas4 + 4 * k);
        for (int r = 0; r < 3; ++r) fs[k].sqrt_info[r] = sqrt_info3[3 * k + r];
    }
    std::vector<double> e, hd, ho, bb; double c2;
    pgo_linearize(x, fs, e, hd, ho, bb, c2);
    std::memcpy(err, e.data(), e.size() * 8); std::memcpy(Hdiag, hd.data(), hd.size() * 8);
    std::memcpy(Hoff, ho.data(), ho.size() * 8); std::memcpy(b, bb.data(), bb.size() * 8);
    *chi2 = c2;
}
void orc_random_set_seed(uint32_t seed) { orc::random::setSeed(seed); }
double orc_random_uniform() { return orc::random::uniform(); }
double orc_random_normal(double stddev) { return orc::random::normal(stddev); }
// cells: (x, y) map coordinates; state -1 free / 0 unknown / 1 occupied (SimpleOccupancyMap)
void orc_loc_occ_set(void* h, const uint32_t* cells_xy, uint32_t n, int state)
{
    SimpleOccupancyMap& m = ((LocBox*)h)->l->occ();
    for (uint32_t i = 0; i < n; ++i) {
        const V3u c{cells_xy[2 * i], cells_xy[2 * i + 1], 0};
        if (state < 0) m.setFree(c); else if (state > 0) m.setOccupied(c); else m.setUnknown(c);
    }
}
void orc_loc_occ_bounds(void* h, double* out6)
{
    V3d a, b; ((LocBox*)h)->l->occ().bounds(a, b);
    out6[0] = a.x; out6[1] = a.y; out6[2] = a.z; out6[3] = b.x; out6[4] = b.y; out6[5] = b.z;
}
void orc_loc_set_lm(void* h, int on) { ((LocBox*)h)->l->set_lm(on != 0); }
void orc_loc_trigger_gloc(void* h) { ((LocBox*)h)->l->triggerGlobalLocalization(); }
int orc_loc_gloc_active(void* h) { return ((LocBox*)h)->l->globalLocalizationIsActive() ? 1 : 0; }
uint32_t orc_loc_gloc_candidates(void* h, double* poses4, double* errors, uint32_t cap)
{
    Loc2D& l = *((LocBox*)h)->l;
    const uint32_t n = (uint32_t)l.gloc_poses.size();
    for (uint32_t i = 0; i < n && i < cap; ++i) { se2_to(l.gloc_poses[i], poses4 + 4 * i); errors[i] = l.gloc_errors[i]; }
    return n;
}
uint32_t orc_loc_sampling_l(void* h, double* out, uint32_t cap)
{
    Loc2D& l = *((LocBox*)h)->l;
    const uint32_t n = (uint32_t)l.sampling_l.size();
    for (uint32_t i = 0; i < n && i < cap; ++i) out[i] = l.sampling_l[i];
    return n;
}

} // extern "C"
