// lama/nlls/strategy.h -- the reference's optimisation-strategy interface (include/lama/nlls/strategy.h:43-82).
#pragma once
#include <memory>
#include <string>

#include "../types.h"

namespace lama {

struct Strategy {
    typedef std::shared_ptr<Strategy> Ptr;
    virtual ~Strategy() {}
    virtual void reset() = 0;
    virtual VectorXd step(const VectorXd& residuals, const MatrixXd& J) = 0;
    virtual bool valid(const VectorXd& residuals) = 0;
    virtual bool stop() = 0;
    virtual std::string name() const = 0;
};

} // namespace lama
