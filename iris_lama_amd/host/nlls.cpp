// nlls.cpp -- lama::MatchSurface2D, lama::Solver / Solve, GaussNewton, LevenbergMarquard (see the headers under
// include/lama/nlls/ and include/lama/match_surface_2d.h for what runs on the device and what is host glue).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <stdexcept>
#include <vector>

#include "covariance3.hpp"
#include "hip_engine.hpp"
#include "lama/match_surface_2d.h"
#include "lama/nlls/solver.h"

namespace lama {

// ---------------------------------------------------------------------------------------------------------------------
// small dense helpers for the generic (user-defined Problem) path: the parameter dimension m is tiny
// ---------------------------------------------------------------------------------------------------------------------
namespace {

void normal_equations(const VectorXd& r, const MatrixXd& J, std::vector<double>& A, std::vector<double>& g)
{
    const size_t n = J.rows(), m = J.cols();
    A.assign(m * m, 0.0); g.assign(m, 0.0);
    for (size_t a = 0; a < m; ++a) {
        double s = 0;
        for (size_t i = 0; i < n; ++i) s += J(i, a) * r[i];
        g[a] = s;
        for (size_t b = 0; b <= a; ++b) {
            double t = 0;
            for (size_t i = 0; i < n; ++i) t += J(i, a) * J(i, b);
            A[a * m + b] = A[b * m + a] = t;
        }
    }
}

// symmetric A x = b by LDL^T with diagonal pivoting (largest remaining |diagonal|); zero pivots give zero components
// (Eigen LDLT::solve semantics for semidefinite matrices)
std::vector<double> ldlt_solve(std::vector<double> A, const std::vector<double>& b, size_t m)
{
    std::vector<size_t> perm(m);
    for (size_t i = 0; i < m; ++i) perm[i] = i;
    std::vector<double> L(m * m, 0.0), D(m, 0.0);
    auto at = [&](size_t i, size_t j) -> double& { return A[perm[i] * m + perm[j]]; };
    for (size_t k = 0; k < m; ++k) {
        size_t best = k;
        for (size_t c = k + 1; c < m; ++c) if (std::fabs(at(c, c)) > std::fabs(at(best, best))) best = c;
        if (best != k) {
            std::swap(perm[k], perm[best]);
            for (size_t j = 0; j < k; ++j) std::swap(L[k * m + j], L[best * m + j]);
        }
        D[k] = at(k, k);
        L[k * m + k] = 1.0;
        if (D[k] == 0.0) continue;
        for (size_t i = k + 1; i < m; ++i) L[i * m + k] = at(i, k) / D[k];
        for (size_t i = k + 1; i < m; ++i)
            for (size_t j = k + 1; j < m; ++j) at(i, j) -= L[i * m + k] * D[k] * L[j * m + k];
    }
    std::vector<double> y(m), x(m);
    for (size_t i = 0; i < m; ++i) { double s = b[perm[i]]; for (size_t j = 0; j < i; ++j) s -= L[i * m + j] * y[j]; y[i] = s; }
    for (size_t i = 0; i < m; ++i) y[i] = D[i] != 0.0 ? y[i] / D[i] : 0.0;
    for (size_t ii = m; ii-- > 0;) { double s = y[ii]; for (size_t j = ii + 1; j < m; ++j) s -= L[j * m + ii] * x[j]; x[ii] = s; }
    std::vector<double> out(m);
    for (size_t i = 0; i < m; ++i) out[perm[i]] = x[i];
    return out;
}

// A x = b by Cholesky (A positive definite: LevenbergMarquard's damped normal matrix)
std::vector<double> llt_solve(std::vector<double> A, const std::vector<double>& b, size_t m)
{
    for (size_t j = 0; j < m; ++j) {
        double d = A[j * m + j];
        for (size_t k = 0; k < j; ++k) d -= A[j * m + k] * A[j * m + k];
        d = std::sqrt(d);
        A[j * m + j] = d;
        for (size_t i = j + 1; i < m; ++i) {
            double s = A[i * m + j];
            for (size_t k = 0; k < j; ++k) s -= A[i * m + k] * A[j * m + k];
            A[i * m + j] = s / d;
        }
    }
    std::vector<double> y(m), x(m);
    for (size_t i = 0; i < m; ++i) { double s = b[i]; for (size_t k = 0; k < i; ++k) s -= A[i * m + k] * y[k]; y[i] = s / A[i * m + i]; }
    for (size_t ii = m; ii-- > 0;) { double s = y[ii]; for (size_t k = ii + 1; k < m; ++k) s -= A[k * m + ii] * x[k]; x[ii] = s / A[ii * m + ii]; }
    return x;
}

double max_abs(const std::vector<double>& v) { double m = 0; for (double x : v) m = std::max(m, std::fabs(x)); return m; }

VectorXd to_vec(const std::vector<double>& v) { VectorXd o(v.size()); for (size_t i = 0; i < v.size(); ++i) o[i] = v[i]; return o; }

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// GaussNewton (src/nlls/gauss_newton.cpp:53-91) / LevenbergMarquard (src/nlls/levenberg_marquardt.cpp:56-110)
// ---------------------------------------------------------------------------------------------------------------------
VectorXd GaussNewton::step(const VectorXd& residuals, const MatrixXd& J)
{
    std::vector<double> A, g;
    normal_equations(residuals, J, A, g);
    chi2_ = residuals.squaredNorm();
    const size_t m = J.cols();
    if (max_abs(g) < opt_.eps1) { stop_ = true; return VectorXd::Zero(m); }
    std::vector<double> mg(m);
    for (size_t i = 0; i < m; ++i) mg[i] = -g[i];
    const std::vector<double> h = ldlt_solve(A, mg, m);
    if (max_abs(h) < opt_.eps2) stop_ = true;
    return to_vec(h);
}

bool GaussNewton::valid(const VectorXd& residuals)
{
    if (stop_) return true;
    if (chi2_ - residuals.squaredNorm() > 0) return true;
    stop_ = true;
    return false;
}

VectorXd LevenbergMarquard::step(const VectorXd& residuals, const MatrixXd& J)
{
    std::vector<double> A, g;
    normal_equations(residuals, J, A, g);
    chi2_ = residuals.squaredNorm();
    const size_t m = J.cols();
    g_ = to_vec(g);
    if (max_abs(g) < opt_.eps1) { stop_ = true; return VectorXd::Zero(m); }
    if (mu_ < 0) { double d = A[0]; for (size_t i = 1; i < m; ++i) d = std::max(d, A[i * m + i]); mu_ = opt_.tau * d; }
    for (size_t i = 0; i < m; ++i) A[i * m + i] += mu_;
    std::vector<double> mg(m);
    for (size_t i = 0; i < m; ++i) mg[i] = -g[i];
    const std::vector<double> h = llt_solve(A, mg, m);
    h_ = to_vec(h);
    if (max_abs(h) < opt_.eps2) stop_ = true;
    return h_;
}

bool LevenbergMarquard::valid(const VectorXd& residuals)
{
    if (stop_) return true;
    const double dF = chi2_ - residuals.squaredNorm();
    double dL = 0;
    for (size_t i = 0; i < h_.size(); ++i) dL += h_[i] * (mu_ * h_[i] - g_[i]);
    dL *= 0.5;
    if (dL > 0.0 && dF > 0.0) {
        const double q = 2 * (dF / dL) - 1;
        mu_ = mu_ * std::max(1.0 / 3.0, 1 - q * q * q);
        v_ = 2.0;
        return true;
    }
    mu_ = mu_ * v_;
    v_ = 2 * v_;
    return false;
}

// ---------------------------------------------------------------------------------------------------------------------
// MatchSurface2D
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct ScanArrays {
    std::vector<double> pts;
    double o[3], q[4];
    explicit ScanArrays(const PointCloudXYZ& s) : pts(s.points.size() * 3)
    {
        for (size_t i = 0; i < s.points.size(); ++i) { pts[3 * i] = s.points[i].x(); pts[3 * i + 1] = s.points[i].y(); pts[3 * i + 2] = s.points[i].z(); }
        o[0] = s.sensor_origin_.x(); o[1] = s.sensor_origin_.y(); o[2] = s.sensor_origin_.z();
        q[0] = s.sensor_orientation_.w(); q[1] = s.sensor_orientation_.x(); q[2] = s.sensor_orientation_.y(); q[3] = s.sensor_orientation_.z();
    }
};

const DynamicDistanceMap::DeviceBinding& device_of(const MatchSurface2D& m)
{
    const DynamicDistanceMap::DeviceBinding& d = m.surface_->device();
    if (!d.engine || !d.ctx)
        throw std::runtime_error("lama::MatchSurface2D: the distance map does not live on the device (obtain it from "
                                 "Slam2D / PFSlam2D::getDistanceMap()); there is no CPU evaluation path");
    return d;
}

[[noreturn]] void device_fail(const DynamicDistanceMap::DeviceBinding& d, int32_t rc, const char* what)
{
    const char* msg = d.engine->last_error ? d.engine->last_error(d.ctx) : "";
    throw std::runtime_error(std::string("lama::MatchSurface2D: ") + what + " failed (status " + std::to_string(rc) + "): " + (msg ? msg : ""));
}

} // namespace

MatchSurface2D::MatchSurface2D(const DynamicDistanceMap* surface, const PointCloudXYZ::Ptr& scan, const SE2d& estimate)
    : surface_(surface), scan_(scan), state_(estimate)
{
    if (!surface_ || !scan_ || scan_->points.empty()) throw std::invalid_argument("lama::MatchSurface2D: null surface or empty scan");
}

void MatchSurface2D::eval(VectorXd& residuals, MatrixXd* J)
{
    const auto& d = device_of(*this);
    const ScanArrays s(*scan_);
    const uint32_t n = (uint32_t)scan_->points.size();
    double p[4];
    state_.toArray(p);
    residuals.resize(n);
    if (J) J->resize(n, 3);
    const int32_t rc = d.engine->match_eval(d.ctx, d.particle, s.pts.data(), n, s.o, s.q, p, residuals.data(), J ? J->data() : nullptr);
    if (rc) device_fail(d, rc, "lama_hip_match_eval");
}

double MatchSurface2D::error()
{
    const auto& d = device_of(*this);
    const ScanArrays s(*scan_);
    const uint32_t n = (uint32_t)scan_->points.size();
    double p[4];
    state_.toArray(p);
    std::vector<double> dist(n);
    const int32_t rc = d.engine->match_cell_distances(d.ctx, d.particle, s.pts.data(), n, s.o, s.q, p, dist.data());
    if (rc) device_fail(d, rc, "lama_hip_match_cell_distances");
    double ss = 0;
    for (double v : dist) ss += v * v;
    return std::sqrt(ss / (double)n);
}

void MatchSurface2D::update(const VectorXd& h)
{
    if (h.size() != 3) throw std::invalid_argument("lama::MatchSurface2D::update: the step has three components");
    state_ = SE2d::exp(h[0], h[1], h[2]) * state_;
}

// ---------------------------------------------------------------------------------------------------------------------
// Solver
// ---------------------------------------------------------------------------------------------------------------------
Solver::Options::Options() : max_iterations(100), strategy(new GaussNewton), robust_cost(new UnitWeight), write_to_stdout(false) {}

namespace {

// the one configuration of a scan-matching problem the fused device solver implements
int device_strategy(const Solver::Options& o)
{
    const CauchyWeight* cw = dynamic_cast<const CauchyWeight*>(o.robust_cost.get());
    if (!cw || cw->c_ != 1.0 / (0.15 * 0.15)) return -1;
    if (const GaussNewton* gn = dynamic_cast<const GaussNewton*>(o.strategy.get()))
        return (gn->options().eps1 == 1e-4 && gn->options().eps2 == 1e-4) ? 0 : -1;
    if (const LevenbergMarquard* lm = dynamic_cast<const LevenbergMarquard*>(o.strategy.get()))
        return (lm->options().eps1 == 1e-4 && lm->options().eps2 == 1e-4 && lm->options().tau == 1e-4) ? 1 : -1;
    return -1;
}

void cov_from_jacobian(const MatrixXd& J, MatrixXd* cov)
{
    // user-defined problems: (J^T J)^-1 through the LDL^T solves (src/nlls/solver.cpp:141-142, the full-rank branch)
    const size_t n = J.rows(), m = J.cols();
    std::vector<double> A(m * m, 0.0);
    for (size_t a = 0; a < m; ++a)
        for (size_t b = 0; b < m; ++b) { double t = 0; for (size_t i = 0; i < n; ++i) t += J(i, a) * J(i, b); A[a * m + b] = t; }
    cov->resize(m, m);
    for (size_t c = 0; c < m; ++c) {
        std::vector<double> e(m, 0.0);
        e[c] = 1.0;
        const std::vector<double> x = ldlt_solve(A, e, m);
        for (size_t r = 0; r < m; ++r) (*cov)(r, c) = x[r];
    }
}

} // namespace

void Solver::solve(Problem& problem, MatrixXd* cov)
{
    if (!options_.strategy || !options_.robust_cost) throw std::invalid_argument("lama::Solver: strategy and robust_cost must be set");
    if (MatchSurface2D* ms = dynamic_cast<MatchSurface2D*>(&problem)) {
        // scan-to-map registration: one launch of the fused device solver
        const int strategy = device_strategy(options_);
        if (strategy < 0)
            throw std::invalid_argument("lama::Solver: a MatchSurface2D problem runs on the device with GaussNewton or LevenbergMarquard "
                                        "(default thresholds) and CauchyWeight(0.15); this configuration has no device kernel and there is "
                                        "no CPU path to fall back to");
        const auto& d = device_of(*ms);
        const ScanArrays s(*ms->scan_);
        double p[4], out7[7];
        int32_t iters = 0;
        ms->state_.toArray(p);
        const int32_t rc = d.engine->match_solve_with(d.ctx, d.particle, s.pts.data(), (uint32_t)ms->scan_->points.size(), s.o, s.q, p, out7, &iters,
                                                      strategy, options_.max_iterations);
        if (rc) device_fail(d, rc, "lama_hip_match_solve_with");
        ms->state_ = SE2d::fromArray(p);
        last_iterations_ = (uint32_t)iters;
        if (cov) {
            double c9[9];
            detail::covariance_from_normal3(out7, c9);
            cov->resize(3, 3);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) (*cov)(r, c) = c9[3 * r + c];
        }
        return;
    }
    // generic loop, src/nlls/solver.cpp:53-117
    Strategy& strategy = *options_.strategy;
    RobustCost& robust = *options_.robust_cost;
    strategy.reset();
    VectorXd r, ur, h;
    MatrixXd J;
    bool valid = true;
    uint32_t iter = 0;
    auto weigh = [&](VectorXd& res, MatrixXd* jac) {
        for (size_t i = 0; i < res.size(); ++i) {
            const double w = std::sqrt(robust.value(res[i]));
            res[i] *= w;
            if (jac) for (size_t c = 0; c < jac->cols(); ++c) (*jac)(i, c) *= w;
        }
    };
    while (!strategy.stop() && iter < options_.max_iterations) {
        if (valid) { problem.eval(r, &J); weigh(r, &J); }
        h = strategy.step(r, J);
        if (strategy.stop()) break;
        problem.update(h);
        problem.eval(ur, nullptr);
        weigh(ur, nullptr);
        valid = strategy.valid(ur);
        if (!valid) {
            VectorXd mh(h.size());
            for (size_t i = 0; i < h.size(); ++i) mh[i] = -h[i];
            problem.update(mh);
        }
        if (options_.write_to_stdout) std::printf("%s iteration %u: chi2 %.9g%s\n", strategy.name().c_str(), iter, ur.squaredNorm(), valid ? "" : " (reverted)");
        ++iter;
    }
    last_iterations_ = iter;
    if (cov) {
        problem.eval(r, &J);
        for (size_t i = 0; i < r.size(); ++i) {
            const double w = std::sqrt(robust.value(r[i]));
            for (size_t c = 0; c < J.cols(); ++c) J(i, c) *= w;
        }
        cov_from_jacobian(J, cov);
    }
}

void Solve(const Solver::Options& options, Problem& problem, MatrixXd* cov)
{
    Solver solver(options);
    solver.solve(problem, cov);
}

} // namespace lama
