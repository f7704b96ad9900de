// Compiles against the mirrored class API the way a node of iris_lama_ros does (pf_slam2d_ros.cpp: fills
// lama::PFSlam2D::Options from parameters, builds a PointCloudXYZ per scan, calls update(cloud, odom, stamp)).
// Without a GPU the constructor must fail loudly (no CPU fallback); that is what this program checks when it runs.
#include <cstdio>
#include <stdexcept>

#include <lama/loc2d.h>
#include <lama/pf_slam2d.h>
#include <lama/slam2d.h>

int main()
{
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
        uint64_t occmem = 0, dmmem = 0;
        if (slam.getMemoryUsage(occmem, dmmem) != occmem + dmmem || occmem == 0 || dmmem == 0 || slam.getMemoryUsage() == 0) return 6;
        slam.saveOccImage("/tmp/lama_consumer_occ.png");
    } catch (const std::runtime_error& e) {
        std::printf("no device: %s\n", e.what());                // expected on a box without an MI355X
        return 0;
    }
    return 0;
}
