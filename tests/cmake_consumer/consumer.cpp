// Compiles against the mirrored class API the way a node of iris_lama_ros does (pf_slam2d_ros.cpp: fills
// lama::PFSlam2D::Options from parameters, builds a PointCloudXYZ per scan, calls update(cloud, odom, stamp)).
// Without a GPU the constructor must fail loudly (no CPU fallback); that is what this program checks when it runs.
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

// only include paths that exist in the reference's installed header tree (include/lama/...)
#include <lama/loc2d.h>
#include <lama/match_surface_2d.h>
#include <lama/nlls/gauss_newton.h>
#include <lama/nlls/levenberg_marquardt.h>
#include <lama/nlls/robust_cost.h>
#include <lama/nlls/solver.h>
#include <lama/pf_slam2d.h>
#include <lama/pose2d.h>
#include <lama/print.h>
#include <lama/sdm/dynamic_distance_map.h>
#include <lama/sdm/export.h>
#include <lama/sdm/frequency_occupancy_map.h>
#include <lama/sdm/map.h>
#include <lama/sdm/simple_occupancy_map.h>
#include <lama/slam2d.h>
#include <lama/time.h>
#include <lama/types.h>

// what iris_lama_ros' loc2d_ros does with a nav_msgs/OccupancyGrid: fill Loc2D's two public maps cell by cell
// (src/loc2d_ros.cpp InitLoc2DFromOccupancyGridMsg), then distance_map->update()
static int fill_loc2d_maps(lama::Loc2D& loc)
{
    lama::SimpleOccupancyMap* occ = loc.occupancy_map;
    lama::DynamicDistanceMap* dm = loc.distance_map;
    if (!occ || !dm) return 20;
    for (int j = 0; j < 80; ++j)
        for (int i = 0; i < 80; ++i) {
            const lama::Vector3d coords(0.05 * i, 0.05 * j, 0.0);
            const bool wall = i == 0 || j == 0 || i == 79 || j == 79;
            if (wall) { occ->setOccupied(coords); dm->addObstacle(dm->w2m(coords)); }
            else occ->setFree(coords);
        }
    if (!occ->isFree(lama::Vector3d(1.0, 1.0, 0.0)) || !occ->isOccupied(lama::Vector3d(0.0, 1.0, 0.0)) || !occ->isUnknown(lama::Vector3d(9.0, 9.0, 0.0))) return 21;
    lama::Vector3d mn, mx;
    occ->bounds(mn, mx);
    if (!(mx[0] - mn[0] >= 4.0)) return 22;
    return 0;
}

// a user-defined nlls::Problem (include/lama/nlls/problem.h): fit y = a * exp(b * x) to samples -- the generic host loop of
// lama::Solver drives it, nothing of it touches the GPU
struct ExpFit : public lama::Problem {
    double a = 1.0, b = 0.0;
    std::vector<double> xs, ys;
    void eval(lama::VectorXd& r, lama::MatrixXd* J) override
    {
        r.resize(xs.size());
        if (J) J->resize(xs.size(), 2);
        for (size_t i = 0; i < xs.size(); ++i) {
            const double e = std::exp(b * xs[i]);
            r[i] = a * e - ys[i];
            if (J) { (*J)(i, 0) = e; (*J)(i, 1) = a * xs[i] * e; }
        }
    }
    void update(const lama::VectorXd& h) override { a += h[0]; b += h[1]; }
};

static int check_generic_solver()
{
    for (int lm = 0; lm < 2; ++lm) {
        ExpFit fit;
        for (int i = 0; i < 40; ++i) { fit.xs.push_back(0.05 * i); fit.ys.push_back(2.5 * std::exp(-0.7 * 0.05 * i)); }
        lama::Solver::Options so;
        if (lm) so.strategy.reset(new lama::LevenbergMarquard); else so.strategy.reset(new lama::GaussNewton);
        so.robust_cost.reset(new lama::CauchyWeight(1.0));
        lama::MatrixXd cov;
        lama::Solve(so, fit, &cov);
        if (std::fabs(fit.a - 2.5) > 1e-3 || std::fabs(fit.b + 0.7) > 1e-3 || cov.rows() != 2 || !(cov(0, 0) > 0)) {
            std::printf("generic solver (%s): a = %.6f b = %.6f\n", lm ? "lm" : "gn", fit.a, fit.b);
            return 1;
        }
    }
    return 0;
}

int main()
{
    if (check_generic_solver()) return 7;
    lama::PFSlam2D::Options options;
    options.particles = 4;
    options.resolution = 0.05;
    options.l2_max = 0.5;
    options.seed = 7;
    lama::Pose2D prior(1.0, 2.0, 0.25), step(0.5, 0.0, 0.1);
    const lama::Pose2D next = prior + step;                     // SE2 composition, src/pose2d.cpp:76-96
    if (std::fabs(next.rotation() - 0.35) > 1e-12) return 2;
    lama::PointCloudXYZ::Ptr cloud(new lama::PointCloudXYZ);     // a room seen from its middle: 360 returns on a 3 m circle
    for (int k = 0; k < 360; ++k) {
        const double a = k * 3.14159265358979323846 / 180.0;
        cloud->points.push_back(lama::Vector3d(3.0 * std::cos(a), 3.0 * std::sin(a), 0.0));
    }
    {   // host-only part of the Loc2D map API: works without a device
        lama::Loc2D loc;
        loc.Init(lama::Loc2D::Options());
        if (const int rc = fill_loc2d_maps(loc)) return rc;
        lama::Timer timer(true);
        if (timer.elapsed().toSec() < 0.0 || lama::format("%d", 7) != "7") return 24;
    }
    try {
        lama::PFSlam2D slam(options);
        slam.setPrior(prior);
        const bool first = slam.update(cloud, prior, 0.0);
        const bool second = slam.update(cloud, prior + lama::Pose2D(0.0, 0.0, 0.6), 1.0);   // past rot_thresh: a full update
        const lama::Pose2D p = slam.getPose();
        std::printf("device path ran: updates %d %d, %zu particles, pose %.3f %.3f %.3f\n", (int)first, (int)second,
                    slam.getParticles().size(), p.x(), p.y(), p.rotation());
        if (!second || std::fabs(p.x() - 1.0) > 0.2 || std::fabs(p.y() - 2.0) > 0.2) return 3;
        // what pf_slam2d_ros does to publish the map: walk the occupancy map of the best particle
        const lama::FrequencyOccupancyMap* map = slam.getOccupancyMap();
        if (!map) return 4;
        lama::Vector3ui imin, imax;
        map->bounds(imin, imax);
        size_t free_cells = 0, occupied_cells = 0;
        map->visit_all_cells([&](const lama::Vector3ui& c) {
            if (map->isFree(c)) ++free_cells; else if (map->isOccupied(c)) ++occupied_cells;
        });
        const lama::DynamicDistanceMap* dm = slam.getDistanceMap();
        std::printf("map: %u x %u cells, %zu free, %zu occupied; distance at the prior %.3f m\n", imax(0) - imin(0), imax(1) - imin(1),
                    free_cells, occupied_cells, dm ? dm->distance(lama::Vector3d(1.0, 2.0, 0.0)) : -1.0);
        if (free_cells < 1000 || occupied_cells < 100 || !dm) return 5;
        // the scan-matching problem built by hand, as the reference's own classes do (src/slam2d.cpp:170-176): evaluation on the
        // device, Solve() = one fused device launch
        {
            lama::MatchSurface2D problem(dm, cloud, (prior + lama::Pose2D(0.03, -0.02, 0.61)).state);
            lama::VectorXd r;
            lama::MatrixXd J;
            problem.eval(r, &J);
            const double before = std::sqrt(r.squaredNorm() / r.size());
            lama::Solver::Options so;
            so.strategy.reset(new lama::GaussNewton);
            so.robust_cost.reset(new lama::CauchyWeight(0.15));
            lama::MatrixXd cov;
            lama::Solve(so, problem, &cov);
            problem.eval(r, nullptr);
            const double after = std::sqrt(r.squaredNorm() / r.size());
            std::printf("MatchSurface2D: rms residual %.4f -> %.4f m, error() %.4f, var(x) %.2e\n", before, after, problem.error(), cov(0, 0));
            if (!(after < before) || J.rows() != r.size() || J.cols() != 3 || !(cov(0, 0) > 0)) return 8;
            so.robust_cost.reset(new lama::HuberWeight(0.1));
            try { lama::Solve(so, problem); return 9; } catch (const std::invalid_argument&) {}   // no device kernel: must refuse
        }
        // ANY particle's maps, what the reference's public Particle::occ / Particle::dm give (include/lama/pf_slam2d.h:83-84): the best
        // particle's snapshot equals getOccupancyMap() cell for cell, another particle's map is a map of its own
        {
            const size_t best = slam.getBestParticleIdx(), other = (best + 1) % slam.getParticles().size();
            std::shared_ptr<const lama::FrequencyOccupancyMap> pb = slam.getParticleOccupancyMap(best), po = slam.getParticleOccupancyMap(other);
            std::shared_ptr<const lama::DynamicDistanceMap> db = slam.getParticleDistanceMap(best), dother = slam.getParticleDistanceMap(other);
            if (!pb || !po || !db || !dother || slam.getParticleOccupancyMap(slam.getParticles().size())) return 40;
            size_t diff = 0, cells = 0, other_free = 0;
            map->visit_all_cells([&](const lama::Vector3ui& c) { ++cells; if (map->isFree(c) != pb->isFree(c) || map->isOccupied(c) != pb->isOccupied(c)) ++diff; });
            po->visit_all_cells([&](const lama::Vector3ui& c) { if (po->isFree(c)) ++other_free; });
            if (diff != 0 || cells == 0 || other_free < 1000 || pb->patches() != map->patches()) return 41;
            if (db->distance(lama::Vector3d(1.0, 2.0, 0.0)) != dm->distance(lama::Vector3d(1.0, 2.0, 0.0)) || dother->patches() == 0) return 42;
            std::printf("per-particle maps: particle %zu (best) == getOccupancyMap(), particle %zu has %zu free cells\n", best, other, other_free);
        }
        uint64_t occmem = 0, dmmem = 0;
        if (slam.getMemoryUsage(occmem, dmmem) != occmem + dmmem || occmem == 0 || dmmem == 0 || slam.getMemoryUsage() == 0) return 6;
        slam.saveOccImage("/tmp/lama_consumer_occ.png");
        // several GPUs behind the same class (Options::gpus; the shards share the devices that are there): update() is the whole
        // sharded step, the results do not depend on the number of shards
        {
            lama::PFSlam2D::Options mo = options;
            mo.gpus = 2; mo.meas_sigma_gain = 0.01; mo.seed = 42;     // (seed 0 = random_device: two objects would draw different noise)
            lama::PFSlam2D::Options so = mo;
            so.gpus = 1;
            lama::PFSlam2D multi(mo), single(so);
            multi.setPrior(prior); single.setPrior(prior);
            for (int k = 0; k < 4; ++k) {
                const lama::Pose2D od = prior + lama::Pose2D(0.6 * k, 0.0, 0.02 * k);
                const bool u1 = multi.update(cloud, od, (double)k), u2 = single.update(cloud, od, (double)k);
                if (u1 != u2) return 30;
            }
            if (multi.numShards() != 2 || single.numShards() != 0) return 31;
            const lama::Pose2D pm = multi.getPose(), ps = single.getPose();
            if (pm.x() != ps.x() || pm.y() != ps.y() || pm.rotation() != ps.rotation()) return 32;
            if (multi.getBestParticleIdx() != single.getBestParticleIdx() || multi.getNeff() != single.getNeff()) return 33;
            std::printf("multi-GPU object: %zu shards, best particle %zu, pose (%.6f, %.6f, %.6f) == single shard\n", multi.numShards(),
                        multi.getBestParticleIdx(), pm.x(), pm.y(), pm.rotation());
        }
        // the online-SLAM class with its Summary (include/lama/slam2d.h:59-88)
        lama::Slam2D::Options s2o;
        s2o.create_summary = true;
        lama::Slam2D slam2(s2o);
        slam2.setPose(prior);
        slam2.update(cloud, prior, 0.0);
        slam2.update(cloud, prior + lama::Pose2D(0.0, 0.0, 0.6), 1.0);
        if (!slam2.summary || slam2.summary->time.size() != 2 || slam2.summary->time_solving.size() != 1 || slam2.summary->memory.back() <= 0) return 10;
        std::printf("%s", slam2.summary->report().c_str());
        // localisation on a map filled through the public members, as loc2d_ros does
        lama::Loc2D loc;
        lama::Loc2D::Options lo;
        loc.Init(lo);
        if (const int rc = fill_loc2d_maps(loc)) return rc;
        const uint32_t processed = loc.distance_map->update();                   // brushfire on the device
        // the first accessor after update() -- whichever it is -- must already see the device map (a live map refreshes its host
        // copy in every accessor, not only in distance())
        if (loc.distance_map->patches() == 0 || !loc.distance_map->write("/tmp/lama_consumer_loc_dm.sdm")) return 24;
        if (loc.distance_map->update() != 0) return 25;                          // nothing pending: no cells processed by THIS call
        lama::EventFrequency freq(4);
        freq.event(0.0); freq.event(0.5); freq.event(1.0);
        if (!(freq.getFrequency() > 1.99 && freq.getFrequency() < 2.01)) return 26;
        const double dmid = loc.distance_map->distance(lama::Vector3d(2.0, 2.0, 0.0));
        const lama::Matrix3d& cov = loc.getCovar();
        lama::print("Loc2D: %u cells processed, distance at the room centre %.3f m, cov(0,0) %.2f, stamp %.0f\n", processed, dmid,
                    cov(0, 0), lama::Time::now().toSec() > 0 ? 1.0 : 0.0);
        if (processed == 0 || !(dmid > 0.9)) return 23;
        lama::sdm::export_to_png(*loc.occupancy_map, "/tmp/lama_consumer_loc_occ.png");
    } catch (const std::runtime_error& e) {
        std::printf("no device: %s\n", e.what());                // expected on a box without an MI355X
        return 0;
    }
    return 0;
}
