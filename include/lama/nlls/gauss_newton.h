// lama/nlls/gauss_newton.h -- include/lama/nlls/gauss_newton.h:42-100, src/nlls/gauss_newton.cpp:38-91.
#pragma once
#include "strategy.h"

namespace lama {

struct GaussNewton : public Strategy {
    struct Options {
        Options() : eps1(1e-4), eps2(1e-4) {}
        double eps1;      // stop when max |J'r| < eps1
        double eps2;      // stop when max |h| < eps2
    };
    GaussNewton(const Options& options = Options()) : opt_(options) {}
    void reset() override { stop_ = false; }
    VectorXd step(const VectorXd& residuals, const MatrixXd& J) override;
    bool valid(const VectorXd& residuals) override;
    bool stop() override { return stop_; }
    std::string name() const override { return "GaussNewton"; }
    const Options& options() const { return opt_; }

private:
    Options opt_;
    double chi2_ = 0.0;
    bool stop_ = false;
};

} // namespace lama
