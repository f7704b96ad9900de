// ref_capi.cpp -- TEST INFRASTRUCTURE.  A C view of the REFERENCE ITSELF: this file is compiled together with the reference's
// own sources, taken where they lie under /root/reference (nothing is copied), into oracle/_ref/liblama_ref.so by
// oracle/Makefile.ref.  Eigen3 is not installed in this image; the reference's sources compile against the small stand-in
// under oracle/ref_shim/ (see mini_eigen.hpp for what that does and does not pin).  The functions mirror the `orc_*` view of the
// CPU oracle (oracle/oracle_capi.cpp) so that tests/test_oracle_vs_reference.py can drive both with the same inputs:
//   poses are handed over as [cos, sin, tx, ty] (SE2d's unit complex + translation), inputs as (x, y, rotation).
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "lama/pose2d.h"
#include "lama/random.h"
#include "lama/match_surface_2d.h"
#include "lama/nlls/solver.h"
#include "lama/nlls/gauss_newton.h"
#include "lama/nlls/levenberg_marquardt.h"
#include "lama/nlls/robust_cost.h"
#include "lama/sdm/dynamic_distance_map.h"
#include "lama/sdm/frequency_occupancy_map.h"
#include "lama/sdm/simple_occupancy_map.h"
#include "lama/pf_slam2d.h"
#include "lama/slam2d.h"
#include "lama/loc2d.h"
#include "lama/lidar_odometry_2d.h"
#include "lama/sdm/probabilistic_occupancy_map.h"
#include "lama/sdm/export.h"
#include "lama/image.h"
#include "lama/image_io.h"

using namespace lama;

namespace {

void pose_to(const Pose2D& p, double* out4)
{
    out4[0] = p.state.so2().unit_complex()[0]; out4[1] = p.state.so2().unit_complex()[1];
    out4[2] = p.state.translation()[0]; out4[3] = p.state.translation()[1];
}

PointCloudXYZ::Ptr make_cloud(const double* pts, int n, const double* origin3, const double* quat4 /*w,x,y,z*/)
{
    PointCloudXYZ::Ptr c(new PointCloudXYZ);
    c->points.reserve((size_t)n);
    for (int i = 0; i < n; ++i) c->points.push_back(Vector3d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
    c->sensor_origin_ = Vector3d(origin3[0], origin3[1], origin3[2]);
    c->sensor_orientation_ = Quaterniond(quat4[0], quat4[1], quat4[2], quat4[3]);
    return c;
}

int patch_ids(const lama::Map* m, uint64_t* ids, int cap)
{
    int n = 0;
    for (auto& kv : m->patches) { if (n < cap) ids[n] = kv.first; ++n; }
    return n;
}
// raw cell records (the reference's own struct layout) and the Container mask of one patch
int patch_read(const lama::Map* m, uint64_t id, uint8_t* cells, uint64_t* mask)
{
    auto it = m->patches.find(id);
    if (it == m->patches.end()) return 0;
    const Container* c = it->second.get();
    std::memcpy(cells, c->data, (size_t)c->SIZE * c->element_size);
    std::memcpy(mask, c->mask, (size_t)c->WORD_COUNT * sizeof(uint64_t));
    return (int)c->SIZE;
}

struct PFBox { std::unique_ptr<PFSlam2D> pf; };

} // namespace

extern "C" {

int ref_sizeof_cell(void* map) { return (int)((lama::Map*)map)->cell_memory_size; }

// ---- SE2 / Pose2D (include/lama/pose2d.h, the vendored Sophus)
void ref_pose_from_xyr(double x, double y, double r, double* out4) { pose_to(Pose2D(x, y, r), out4); }
void ref_pose_plus_xyr(const double* a3, const double* b3, double* out4) { pose_to(Pose2D(a3[0], a3[1], a3[2]) + Pose2D(b3[0], b3[1], b3[2]), out4); }
void ref_pose_minus_xyr(const double* a3, const double* b3, double* out4) { pose_to(Pose2D(a3[0], a3[1], a3[2]) - Pose2D(b3[0], b3[1], b3[2]), out4); }
void ref_se2_exp(const double* v3, double* out4) { Pose2D p(SE2d::exp(Vector3d(v3[0], v3[1], v3[2]))); pose_to(p, out4); }
double ref_pose_rotation(double x, double y, double r) { return Pose2D(x, y, r).rotation(); }
double ref_cauchy(double param, double x) { return CauchyWeight(param).value(x); }

// ---- random (src/random.cpp)
void ref_random_set_seed(uint32_t seed) { random::setSeed(seed); }
double ref_random_uniform() { return random::uniform(); }
double ref_random_normal(double stddev) { return random::normal(stddev); }

// ---- DynamicDistanceMap
void* ref_dm_new(double res, uint32_t patch_size, double l2_max) { auto* m = new DynamicDistanceMap(res, patch_size); m->setMaxDistance(l2_max); return m; }
void* ref_dm_clone(void* h) { return new DynamicDistanceMap(*(DynamicDistanceMap*)h); }
void ref_dm_free(void* h) { delete (DynamicDistanceMap*)h; }
void ref_dm_add_obstacle(void* h, uint32_t x, uint32_t y, uint32_t z) { ((DynamicDistanceMap*)h)->addObstacle(Vector3ui(x, y, z)); }
void ref_dm_remove_obstacle(void* h, uint32_t x, uint32_t y, uint32_t z) { ((DynamicDistanceMap*)h)->removeObstacle(Vector3ui(x, y, z)); }
uint32_t ref_dm_update(void* h) { return ((DynamicDistanceMap*)h)->update(); }
double ref_dm_max_distance(void* h) { return ((DynamicDistanceMap*)h)->maxDistance(); }
double ref_dm_distance_cell(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((const DynamicDistanceMap*)h)->distance(Vector3ui(x, y, z)); }
double ref_dm_distance(void* h, const double* p3, double* grad3)
{
    Vector3d g;
    const double d = ((const DynamicDistanceMap*)h)->distance(Vector3d(p3[0], p3[1], p3[2]), grad3 ? &g : nullptr);
    if (grad3) { grad3[0] = g[0]; grad3[1] = g[1]; grad3[2] = g[2]; }
    return d;
}
int ref_map_patch_ids(void* h, uint64_t* ids, int cap) { return patch_ids((const lama::Map*)(DynamicDistanceMap*)h, ids, cap); }
int ref_map_patch_read(void* h, uint64_t id, uint8_t* cells, uint64_t* mask) { return patch_read((const lama::Map*)(DynamicDistanceMap*)h, id, cells, mask); }
int ref_occ_patch_ids(void* h, uint64_t* ids, int cap) { return patch_ids((const lama::Map*)(FrequencyOccupancyMap*)h, ids, cap); }
int ref_occ_patch_read(void* h, uint64_t id, uint8_t* cells, uint64_t* mask) { return patch_read((const lama::Map*)(FrequencyOccupancyMap*)h, id, cells, mask); }
void ref_dm_w2m(void* h, const double* p3, uint32_t* out3) { Vector3ui c = ((const DynamicDistanceMap*)h)->w2m(Vector3d(p3[0], p3[1], p3[2])); out3[0] = c[0]; out3[1] = c[1]; out3[2] = c[2]; }
void ref_dm_m2w(void* h, const uint32_t* c3, double* out3) { Vector3d w = ((const DynamicDistanceMap*)h)->m2w(Vector3ui(c3[0], c3[1], c3[2])); out3[0] = w[0]; out3[1] = w[1]; out3[2] = w[2]; }
uint64_t ref_dm_m2p(void* h, const uint32_t* c3) { return ((const DynamicDistanceMap*)h)->m2p(Vector3ui(c3[0], c3[1], c3[2])); }
uint32_t ref_dm_m2c(void* h, const uint32_t* c3) { return ((const DynamicDistanceMap*)h)->m2c(Vector3ui(c3[0], c3[1], c3[2])); }
int ref_compute_ray(void* h, const uint32_t* from3, const uint32_t* to3, uint32_t* out, int cap)
{
    VectorVector3ui sink;
    ((DynamicDistanceMap*)h)->computeRay(Vector3ui(from3[0], from3[1], from3[2]), Vector3ui(to3[0], to3[1], to3[2]), sink);
    int n = 0;
    for (auto& c : sink) { if (n < cap) { out[3 * n] = c[0]; out[3 * n + 1] = c[1]; out[3 * n + 2] = c[2]; } ++n; }
    return n;
}

// ---- FrequencyOccupancyMap
void* ref_occ_new(double res, uint32_t patch_size) { return new FrequencyOccupancyMap(res, patch_size); }
void ref_occ_free(void* h) { delete (FrequencyOccupancyMap*)h; }
int ref_occ_set_free(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((FrequencyOccupancyMap*)h)->setFree(Vector3ui(x, y, z)) ? 1 : 0; }
int ref_occ_set_occupied(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((FrequencyOccupancyMap*)h)->setOccupied(Vector3ui(x, y, z)) ? 1 : 0; }
double ref_occ_probability(void* h, uint32_t x, uint32_t y, uint32_t z) { return ((const FrequencyOccupancyMap*)h)->getProbability(Vector3ui(x, y, z)); }

// ---- MatchSurface2D + Solver (src/match_surface_2d.cpp, src/nlls/*.cpp)
void ref_eval(void* dm, const double* pts, int n, const double* origin3, const double* quat4, const double* xyr, double* residuals, double* J /*n x 3 col-major or null*/)
{
    PointCloudXYZ::Ptr c = make_cloud(pts, n, origin3, quat4);
    MatchSurface2D ms((const DynamicDistanceMap*)dm, c, Pose2D(xyr[0], xyr[1], xyr[2]).state);
    VectorXd r; MatrixXd Jm;
    ms.eval(r, J ? &Jm : nullptr);
    for (int i = 0; i < n; ++i) residuals[i] = r[i];
    if (J) for (int j = 0; j < 3; ++j) for (int i = 0; i < n; ++i) J[(size_t)j * n + i] = Jm(i, j);
}
// Solve with the options PFSlam2D::scanMatch / Slam2D / Loc2D use: Cauchy(0.15), GaussNewton or LevenbergMarquard
void ref_solve(void* dm, const double* pts, int n, const double* origin3, const double* quat4, const double* xyr, uint32_t max_iter, int lm, double* pose4, double* cov9)
{
    PointCloudXYZ::Ptr c = make_cloud(pts, n, origin3, quat4);
    MatchSurface2D ms((const DynamicDistanceMap*)dm, c, Pose2D(xyr[0], xyr[1], xyr[2]).state);
    Solver::Options so;
    so.max_iterations = max_iter;
    if (lm) so.strategy.reset(new LevenbergMarquard); else so.strategy.reset(new GaussNewton);
    so.robust_cost.reset(new CauchyWeight(0.15));
    MatrixXd cov;
    Solve(so, ms, cov9 ? &cov : nullptr);
    pose_to(Pose2D(ms.getState()), pose4);
    if (cov9) for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) cov9[3 * i + j] = cov(i, j);
}

// ---- PFSlam2D
struct ref_pf_options {
    uint32_t particles;
    double srr, str, stt, srt;
    double meas_sigma, meas_sigma_gain;
    double trans_thresh, rot_thresh;
    double l2_max, truncated_ray, truncated_range, resolution;
    uint32_t patch_size, max_iter;
    int32_t threads;
    uint32_t seed;
};
void* ref_pf_new(const ref_pf_options* o)
{
    PFSlam2D::Options p;
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
void ref_pf_free(void* h) { delete (PFBox*)h; }
void ref_pf_set_prior(void* h, const double* xyr) { ((PFBox*)h)->pf->setPrior(Pose2D(xyr[0], xyr[1], xyr[2])); }
int ref_pf_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, const double* odom_xyr, double ts)
{
    return ((PFBox*)h)->pf->update(make_cloud(pts, n, origin3, quat4), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts) ? 1 : 0;
}
double ref_pf_neff(void* h) { return ((PFBox*)h)->pf->getNeff(); }
int ref_pf_best(void* h) { return (int)((PFBox*)h)->pf->getBestParticleIdx(); }
void ref_pf_get_pose(void* h, double* out4) { pose_to(((PFBox*)h)->pf->getPose(), out4); }
void ref_pf_get_poses(void* h, double* out /*P x 4*/)
{
    auto& ps = ((PFBox*)h)->pf->getParticles();
    for (size_t i = 0; i < ps.size(); ++i) pose_to(ps[i].pose, out + 4 * i);
}
void ref_pf_get_weights(void* h, double* weight, double* nweight, double* weight_sum)
{
    auto& ps = ((PFBox*)h)->pf->getParticles();
    for (size_t i = 0; i < ps.size(); ++i) {
        if (weight) weight[i] = ps[i].weight;
        if (nweight) nweight[i] = ps[i].normalized_weight;
        if (weight_sum) weight_sum[i] = ps[i].weight_sum;
    }
}
void* ref_pf_particle_dm(void* h, int i) { return ((PFBox*)h)->pf->getParticles()[(size_t)i].dm.get(); }      // borrowed
void* ref_pf_particle_occ(void* h, int i) { return ((PFBox*)h)->pf->getParticles()[(size_t)i].occ.get(); }    // borrowed

// ---- Slam2D
void* ref_slam_new(double trans_thresh, double rot_thresh, double l2_max, double truncated_ray, double truncated_range, double resolution,
                   uint32_t patch_size, uint32_t max_iter, int transient_map, int lm)
{
    Slam2D::Options o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.truncated_ray = truncated_ray; o.truncated_range = truncated_range;
    o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter; o.transient_map = transient_map != 0;
    o.strategy = lm ? "lm" : "gn";
    return new Slam2D(o);
}
void ref_slam_free(void* h) { delete (Slam2D*)h; }
void ref_slam_set_pose(void* h, const double* xyr) { ((Slam2D*)h)->setPose(Pose2D(xyr[0], xyr[1], xyr[2])); }
void ref_slam_get_pose(void* h, double* out4) { pose_to(((Slam2D*)h)->getPose(), out4); }
int ref_slam_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, const double* odom_xyr, double ts)
{
    return ((Slam2D*)h)->update(make_cloud(pts, n, origin3, quat4), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts) ? 1 : 0;
}
void* ref_slam_dm(void* h) { return (void*)((Slam2D*)h)->getDistanceMap(); }
void* ref_slam_occ(void* h) { return (void*)((Slam2D*)h)->getOccupancyMap(); }

// ---- Loc2D
void* ref_loc_new(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t patch_size, uint32_t max_iter, int lm)
{
    Loc2D::Options o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter;
    o.strategy = lm ? "lm" : "gn";
    auto* l = new Loc2D;
    l->Init(o);
    return l;
}
void* ref_loc_new2(double trans_thresh, double rot_thresh, double l2_max, double resolution, uint32_t patch_size, uint32_t max_iter,
                   uint32_t gloc_particles, uint32_t gloc_iters, double gloc_thresh, double cov_blend)
{
    Loc2D::Options o;
    o.trans_thresh = trans_thresh; o.rot_thresh = rot_thresh; o.l2_max = l2_max; o.resolution = resolution; o.patch_size = patch_size; o.max_iter = max_iter;
    o.strategy = "gn"; o.gloc_particles = gloc_particles; o.gloc_iters = gloc_iters; o.gloc_thresh = gloc_thresh; o.cov_blend = cov_blend;
    auto* l = new Loc2D;
    l->Init(o);
    return l;
}
// SimpleOccupancyMap cells of the static map: state -1 free / 1 occupied (occupied cells also become distance-map obstacles)
void ref_loc_occ_set_state(void* h, const uint32_t* cells_xy, uint32_t n, int state)
{
    Loc2D* l = (Loc2D*)h;
    for (uint32_t i = 0; i < n; ++i) {
        const Vector3ui c(cells_xy[2 * i], cells_xy[2 * i + 1], 0);
        if (state > 0) l->occupancy_map->setOccupied(c); else if (state < 0) l->occupancy_map->setFree(c); else l->occupancy_map->setUnknown(c);
    }
}
void ref_loc_trigger_gloc(void* h) { ((Loc2D*)h)->triggerGlobalLocalization(); }
int ref_loc_gloc_active(void* h) { return ((Loc2D*)h)->globalLocalizationIsActive() ? 1 : 0; }
void ref_loc_free(void* h) { delete (Loc2D*)h; }
void* ref_loc_dm(void* h) { return ((Loc2D*)h)->distance_map; }
void ref_loc_occ_set(void* h, const uint32_t* cells_xy, uint32_t n, int occupied)
{
    Loc2D* l = (Loc2D*)h;
    for (uint32_t i = 0; i < n; ++i) {
        const Vector3ui c(cells_xy[2 * i], cells_xy[2 * i + 1], 0);
        if (occupied) { l->occupancy_map->setOccupied(c); l->distance_map->addObstacle(c); } else l->occupancy_map->setFree(c);
    }
    l->distance_map->update();
}
void ref_loc_set_pose(void* h, const double* xyr) { ((Loc2D*)h)->setPose(Pose2D(xyr[0], xyr[1], xyr[2])); }
void ref_loc_get_pose(void* h, double* out4) { pose_to(((Loc2D*)h)->getPose(), out4); }
int ref_loc_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, const double* odom_xyr, double ts, int force)
{
    return ((Loc2D*)h)->update(make_cloud(pts, n, origin3, quat4), Pose2D(odom_xyr[0], odom_xyr[1], odom_xyr[2]), ts, force != 0) ? 1 : 0;
}
void ref_loc_covar(void* h, double* out9) { const Matrix3d& c = ((Loc2D*)h)->getCovar(); for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out9[3 * i + j] = c(i, j); }
double ref_loc_rmse(void* h) { return ((Loc2D*)h)->getRMSE(); }

// ---- LidarOdometry2D (src/lidar_odometry_2d.cpp)
void* ref_lo_new(double resolution, uint32_t max_iter) { LidarOdometry2D::Options o; o.resolution = resolution; o.max_iter = max_iter; return new LidarOdometry2D(o); }
void ref_lo_free(void* h) { delete (LidarOdometry2D*)h; }
int ref_lo_update(void* h, const double* pts, int n, const double* origin3, const double* quat4, double ts) { return ((LidarOdometry2D*)h)->update(make_cloud(pts, n, origin3, quat4), ts) ? 1 : 0; }
void ref_lo_get_odom(void* h, double* out4) { pose_to(((LidarOdometry2D*)h)->odom, out4); }
void* ref_lo_dm(void* h) { return ((LidarOdometry2D*)h)->distance_map; }
void* ref_lo_occ(void* h) { return ((LidarOdometry2D*)h)->occupancy_map; }
int ref_pocc_patch_ids(void* h, uint64_t* ids, int cap) { return patch_ids((const lama::Map*)(ProbabilisticOccupancyMap*)h, ids, cap); }
int ref_pocc_patch_read(void* h, uint64_t id, uint8_t* cells, uint64_t* mask) { return patch_read((const lama::Map*)(ProbabilisticOccupancyMap*)h, id, cells, mask); }

// ---- .sdm files (Map::write / Map::read, src/sdm/map.cpp:489-575) and the export images (src/sdm/export.cpp, src/image_io.cpp)
int ref_dm_write(void* h, const char* file) { return ((DynamicDistanceMap*)h)->write(file) ? 1 : 0; }
int ref_dm_read(void* h, const char* file) { return ((DynamicDistanceMap*)h)->read(file) ? 1 : 0; }
int ref_occ_write(void* h, const char* file) { return ((FrequencyOccupancyMap*)h)->write(file) ? 1 : 0; }
int ref_occ_read(void* h, const char* file) { return ((FrequencyOccupancyMap*)h)->read(file) ? 1 : 0; }
int ref_dm_export_png(void* h, const char* file) { sdm::export_to_png(*(const DistanceMap*)(DynamicDistanceMap*)h, file); return 1; }
int ref_occ_export_png(void* h, const char* file) { sdm::export_to_png(*(const OccupancyMap*)(FrequencyOccupancyMap*)h, file); return 1; }
// decode an image file with the reference's reader (stb): returns 1 and the size; pixels (grey) into out when it is large enough
int ref_image_read(const char* file, uint32_t* w, uint32_t* hgt, uint8_t* out, uint64_t cap)
{
    Image im;
    if (!image_read(im, file)) return 0;
    *w = im.width; *hgt = im.height;
    const uint64_t n = (uint64_t)im.width * im.height;
    if (out && cap >= n) std::memcpy(out, im.data.get(), n);
    return 1;
}

} // extern "C"
