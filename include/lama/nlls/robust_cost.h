// lama/nlls/robust_cost.h -- weight functions of the reference (include/lama/nlls/robust_cost.h:42-80,
// src/nlls/robust_cost.cpp:36-100).  On the device path the scan-matching kernels apply CauchyWeight(0.15), the only
// weight the reference's SLAM / localisation classes use (src/pf_slam2d.cpp:423-427, src/slam2d.cpp:103-106, src/loc2d.cpp:93).
#pragma once
#include <cmath>
#include <memory>

namespace lama {

struct RobustCost {
    typedef std::shared_ptr<RobustCost> Ptr;
    virtual ~RobustCost() {}
    virtual double value(const double& x) = 0;
};
struct UnitWeight : public RobustCost {
    double value(const double&) override { return 1.0; }
};
struct TukeyWeight : public RobustCost {
    TukeyWeight(const double& b = 4.6851f) : bb_(b * b) {}
    double value(const double& x) override
    {
        const double xx = x * x;
        if (xx <= bb_) { const double w = 1.0 - xx / bb_; return w * w; }
        return 0.0;
    }
    double bb_;
};
struct TDistributionWeight : public RobustCost {
    TDistributionWeight(const double& dof) : dof_(dof) {}
    double value(const double& x) override { return ((dof_ + 1.0f) / (dof_ + (x * x))); }
    double dof_;
};
struct CauchyWeight : public RobustCost {
    CauchyWeight(const double& param) : c_(1.0 / (param * param)) {}
    double value(const double& x) override { return (1.0 / (1.0 + x * x * c_)); }
    double c_;
};
struct HuberWeight : public RobustCost {
    HuberWeight(const double& k) : k_(k) {}
    double value(const double& x) override { return (x < k_) ? 1.0 : (k_ / std::fabs(x)); }
    double k_;
};

} // namespace lama
