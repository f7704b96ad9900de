// ref_pgo_capi.cpp -- TEST INFRASTRUCTURE.  A C view of the reference's pose-graph linearisation: minisam's own sources
// (vendor/minisam/minisam/core/*.cpp, nonlinear/linearization.cpp, nonlinear/SparsityPattern.cpp, slam/PriorFactor.h,
// slam/BetweenFactor.h, geometry/Sophus.h on the vendored Sophus SE2), compiled where they lie by oracle/Makefile.ref against the
// Eigen stand-in of oracle/ref_shim (incl. its SparseMatrix, mini_eigen_sparse.hpp), driven exactly as SimplePGO::optimize builds
// its graph (src/simple_pgo.cpp:48-105: PriorFactor<SE2d> / BetweenFactor<SE2d> with DiagonalLoss, Variables keyed 'x' i) and as
// the optimiser linearises it (nonlinear/NonlinearOptimizer.cpp:117-164: default variable ordering, lower-Hessian sparsity cache,
// internal::linearzationLowerHessian).  tests/test_oracle_vs_reference.py compares the oracle's restatement with it.
#include <cstdint>
#include <memory>
#include <vector>

#include <minisam/core/FactorGraph.h>
#include <minisam/core/LossFunction.h>
#include <minisam/core/VariableOrdering.h>
#include <minisam/core/Variables.h>
#include <minisam/geometry/Sophus.h>
#include <minisam/nonlinear/SparsityPattern.h>
#include <minisam/nonlinear/linearization.h>
#include <minisam/slam/BetweenFactor.h>
#include <minisam/slam/PriorFactor.h>

#include "lama/pose2d.h"

namespace sam = minisam;
using lama::SE2d;

namespace {
SE2d se2_of(const double* p)      // {c, s, tx, ty}
{
    // the state as given, bit for bit: setComplex() / the constructors would normalise the complex number once more, which is not
    // an identity in floating point (the inputs ARE results of the group's own, already normalising, operations)
    SE2d g;
    g.so2().data()[0] = p[0]; g.so2().data()[1] = p[1];
    g.translation() = Eigen::Vector2d(p[2], p[3]);
    return g;
}
} // namespace

extern "C" {

// poses4 [N][4], factors: fi / fj (fj < 0: prior on fi), meas4 [F][4], sqrt_info3 [F][3] (DiagonalLoss::Scales).
// Outputs: H (3N x 3N, row major, FULL symmetric matrix rebuilt from the lower Hessian in POSE order), b = Atb (3N, pose order),
// werr [F][3] the whitened errors (Factor::weightedError).  Returns 0, or < 0 on an exception.
int ref_pgo_linearize(const double* poses4, uint32_t N, const int32_t* fi, const int32_t* fj, const double* meas4, const double* sqrt_info3,
                      uint32_t F, double* H, double* b, double* werr)
{
    try {
        sam::FactorGraph graph;
        for (uint32_t k = 0; k < F; ++k) {
            Eigen::VectorXd s(3);
            s << sqrt_info3[3 * k], sqrt_info3[3 * k + 1], sqrt_info3[3 * k + 2];
            const auto loss = sam::DiagonalLoss::Scales(s);
            if (fj[k] < 0) graph.add(sam::PriorFactor<SE2d>(sam::key('x', (size_t)fi[k]), se2_of(meas4 + 4 * k), loss));
            else graph.add(sam::BetweenFactor<SE2d>(sam::key('x', (size_t)fi[k]), sam::key('x', (size_t)fj[k]), se2_of(meas4 + 4 * k), loss));
        }
        sam::Variables values;
        for (uint32_t i = 0; i < N; ++i) values.add(sam::key('x', (size_t)i), se2_of(poses4 + 4 * i));
        const sam::VariableOrdering ordering = values.defaultVariableOrdering();
        const sam::internal::LowerHessianSparsityPattern sparsity = sam::internal::constructLowerHessianSparsity(graph, values, ordering);
        Eigen::SparseMatrix<double> AtA;
        Eigen::VectorXd Atb;
        sam::internal::linearzationLowerHessian(graph, values, sparsity, AtA, Atb);
        // position of pose i's block in the ordered system
        std::vector<int> col((size_t)N);
        for (uint32_t i = 0; i < N; ++i) col[i] = sparsity.var_col[ordering.searchKey(sam::key('x', (size_t)i))];
        std::vector<int> pose_of((size_t)3 * N), comp_of((size_t)3 * N);
        for (uint32_t i = 0; i < N; ++i) for (int a = 0; a < 3; ++a) { pose_of[(size_t)col[i] + a] = (int)i; comp_of[(size_t)col[i] + a] = a; }
        const size_t n3 = (size_t)3 * N;
        for (size_t k = 0; k < n3 * n3; ++k) H[k] = 0.0;
        const int* outer = AtA.outerIndexPtr(); const int* inner = AtA.innerIndexPtr(); const double* val = AtA.valuePtr();
        for (int j = 0; j < (int)n3; ++j)
            for (int p = outer[j]; p < outer[j + 1]; ++p) {
                const int i = inner[p];
                const size_t r = (size_t)3 * pose_of[(size_t)i] + comp_of[(size_t)i], c = (size_t)3 * pose_of[(size_t)j] + comp_of[(size_t)j];
                H[r * n3 + c] = val[p];
                H[c * n3 + r] = val[p];
            }
        for (uint32_t i = 0; i < N; ++i) for (int a = 0; a < 3; ++a) b[3 * i + a] = Atb((size_t)col[i] + a);
        if (werr)
            for (uint32_t k = 0; k < F; ++k) {
                const Eigen::VectorXd e = graph.factors()[k]->weightedError(values);
                for (int a = 0; a < 3; ++a) werr[3 * k + a] = e(a);
            }
        return 0;
    } catch (...) {
        return -1;
    }
}

} // extern "C"
