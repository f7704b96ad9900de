// lama/match_surface_2d.h -- lama::MatchSurface2D with the reference's interface (include/lama/match_surface_2d.h:45-82):
// the scan-to-distance-map registration problem.  The distance map lives on the device: `surface` must be a map handed out
// by Slam2D / PFSlam2D::getDistanceMap() (a host snapshot that remembers which device map it was taken from) -- eval() runs
// as a kernel against that live map (lama_hip_match_eval), update() is the SE2 algebra of src/match_surface_2d.cpp:118-122
// on the host, and Solve() on a MatchSurface2D is one fused device launch (see lama/nlls/solver.h).
#pragma once
#include "nlls/problem.h"
#include "pose2d.h"
#include "sdm_maps.h"

namespace lama {

struct MatchSurface2D : public Problem {
    MatchSurface2D(const DynamicDistanceMap* surface, const PointCloudXYZ::Ptr& scan, const SE2d& estimate);

    SE2d getState() const { return state_; }
    // src/match_surface_2d.cpp:42-90: residual_i = interpolated distance at beam i's end point, J_i = [gx, gy, gy*hx - gx*hy]
    void eval(VectorXd& residuals, MatrixXd* J) override;
    // :92-116 root mean squared error
    double error();
    // :118-122 state = SE2::exp(h) * state
    void update(const VectorXd& h) override;

    const DynamicDistanceMap* surface_;
    PointCloudXYZ::Ptr scan_;
    SE2d state_;
};

} // namespace lama
