// lama/nlls/problem.h -- the reference's nlls::Problem interface (include/lama/nlls/problem.h:40-57).
#pragma once
#include "../types.h"

namespace lama {

struct Problem {
    virtual ~Problem() {}
    // residuals and, when J is not null, the Jacobian (residuals x parameters, column major)
    virtual void eval(VectorXd& residuals, MatrixXd* J) = 0;
    // apply the optimisation step h to the internal state
    virtual void update(const VectorXd& h) = 0;
};

} // namespace lama
