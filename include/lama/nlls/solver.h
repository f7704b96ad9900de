// lama/nlls/solver.h -- lama::Solver / lama::Solve with the reference's signatures (include/lama/nlls/solver.h:49-85).
//
// What runs where:
//   * a lama::MatchSurface2D problem (the scan-to-map registration of PFSlam2D / Slam2D / Loc2D) with GaussNewton or
//     LevenbergMarquard (default thresholds) and CauchyWeight(0.15) -- the configuration all of the reference's own classes
//     use -- is solved by ONE launch of the fused device solver (lama_hip_match_solve_with: evaluation, robust weights, the
//     3x3 normal equations and the step loop all on the GPU); the covariance follows Solver::calculateCovariance
//     (src/nlls/solver.cpp:133-150) from the weighted J'J the kernel returns;
//   * a MatchSurface2D problem with any other strategy options / weight function is rejected with std::invalid_argument:
//     there is no CPU version of the hot path to fall back to;
//   * any other (user-defined) Problem goes through the generic loop of src/nlls/solver.cpp:53-117 on the host: its eval()
//     is the user's code, the solver is glue around it.
#pragma once
#include <cstdint>

#include "gauss_newton.h"
#include "levenberg_marquardt.h"
#include "problem.h"
#include "robust_cost.h"
#include "strategy.h"

namespace lama {

class Solver {
public:
    struct Options {
        Options();                       // src/nlls/solver.cpp:39-47: 100 iterations, GaussNewton, UnitWeight, quiet
        uint32_t max_iterations;
        Strategy::Ptr strategy;
        RobustCost::Ptr robust_cost;
        bool write_to_stdout;
    };
    Solver(const Options& options = Options()) : options_(options) {}
    void solve(Problem& problem, MatrixXd* cov = 0);
    uint32_t lastIterations() const { return last_iterations_; }

private:
    Options options_;
    uint32_t last_iterations_ = 0;
};

void Solve(const Solver::Options& options, Problem& problem, MatrixXd* cov = 0);

} // namespace lama
