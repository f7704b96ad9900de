// lama/nlls/levenberg_marquardt.h -- include/lama/nlls/levenberg_marquardt.h:41-110, src/nlls/levenberg_marquardt.cpp:38-110
// (the class is spelled LevenbergMarquard in the reference).
#pragma once
#include "strategy.h"

namespace lama {

struct LevenbergMarquard : public Strategy {
    struct Options {
        Options() : eps1(1e-4), eps2(1e-4), tau(1e-4) {}
        double eps1, eps2;
        double tau;       // initial damping = tau * max diag(J'J)
    };
    LevenbergMarquard(const Options& options = Options()) : opt_(options) {}
    void reset() override { mu_ = -1; v_ = 2.0; stop_ = false; }
    VectorXd step(const VectorXd& residuals, const MatrixXd& J) override;
    bool valid(const VectorXd& residuals) override;
    bool stop() override { return stop_; }
    std::string name() const override { return "LevenbergMarquard"; }
    const Options& options() const { return opt_; }

private:
    Options opt_;
    double mu_ = -1, v_ = 2.0, chi2_ = 0.0;
    VectorXd g_, h_;
    bool stop_ = false;
};

} // namespace lama
