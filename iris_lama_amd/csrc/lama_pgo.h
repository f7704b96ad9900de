// lama_pgo.h -- SE2 pose-graph linearisation on the device.
//
// Replaces the per-factor body and the accumulation of minisam's linearzationLowerHessian
// (vendor/minisam/minisam/nonlinear/linearization.cpp:150-272, 290-341) for the graphs built by SimplePGO::optimize
// (src/simple_pgo.cpp:48-105) and GraphSlam2D::optimizePoseGraph (src/graph_slam2d.cpp:394-430): PriorFactor<SE2d>
// (slam/PriorFactor.h:50-62), BetweenFactor<SE2d> (slam/BetweenFactor.h:50-68), DiagonalLoss
// (core/LossFunction.cpp:95-113), Sophus traits (geometry/Sophus.h:45-74).  The sparse Cholesky stays on the host.
//
//   k_pgo_factors : one thread per factor -> whitened error (3), J_i^T J_i, J_j^T J_j, J_i^T J_j (3x3 each),
//                   J_i^T e, J_j^T e.  ~300 B written per factor; 10k poses / 50k edges = 15 MB per linearisation.
//   k_pgo_reduce  : one thread per variable: sums the diagonal blocks and gradient segments of its factors IN
//                   FACTOR ORDER through a CSR incidence list (what the sequential loop of the reference does), so the
//                   result does not depend on scheduling -- no floating-point atomics.
#pragma once
#include "lama_dev.h"

namespace lama_dev {

struct PgoPtrs {
    const double* poses;      // [N][4] c, s, tx, ty
    const int32_t* fi;        // [F]
    const int32_t* fj;        // [F]  (-1: prior)
    const double* meas;       // [F][4]
    const double* sqrt_info;  // [F][3]
    double* err;              // [F][3]
    double* hoff;             // [F][9]
    double* fdi;              // [F][9]  J_i^T J_i
    double* fdj;              // [F][9]  J_j^T J_j
    double* fg;               // [F][6]  J_i^T e, J_j^T e
    const int32_t* inc_ptr;   // [N+1]
    const int32_t* inc;       // [nnz] factor index * 2 + side (0: the variable is the factor's i, 1: its j), factor order
    double* hdiag;            // [N][9]
    double* b;                // [N][3]
    double* chi2_part;        // [gridDim of k_pgo_factors]
};

__device__ inline SE2 pgo_mul(const SE2& a, const SE2& b)          // se2.hpp:154-157,262-265 (operator*= renormalises)
{
    SE2 r;
    r.tx = a.tx + (a.c * b.tx - a.s * b.ty);
    r.ty = a.ty + (a.s * b.tx + a.c * b.ty);
    r.c = a.c * b.c - a.s * b.s;
    r.s = a.c * b.s + a.s * b.c;
    so2_normalize(r.c, r.s);
    return r;
}
__device__ inline SE2 pgo_inverse(const SE2& a)                     // se2.hpp:163-167
{
    SE2 r;
    r.c = a.c; r.s = -a.s;
    so2_normalize(r.c, r.s);
    const double mx = a.tx * -1.0, my = a.ty * -1.0;
    r.tx = r.c * mx - r.s * my;
    r.ty = r.s * mx + r.c * my;
    return r;
}
__device__ inline void pgo_log(const SE2& g, double out[3])         // se2.hpp:519-542
{
    const double theta = atan2(g.s, g.c);
    out[2] = theta;
    const double halftheta = 0.5 * theta;
    const double real_minus_one = g.c - 1.;
    double h;
    if (fabs(real_minus_one) < 1e-10) h = 1. - (1. / 12) * theta * theta;
    else h = -(halftheta * g.s) / (real_minus_one);
    out[0] = h * g.tx + halftheta * g.ty;
    out[1] = -halftheta * g.tx + h * g.ty;
}
__device__ inline void pgo_adj(const SE2& g, double A[3][3])        // se2.hpp:125-133
{
    A[0][0] = g.c; A[0][1] = -g.s; A[0][2] = g.ty;
    A[1][0] = g.s; A[1][1] = g.c;  A[1][2] = -g.tx;
    A[2][0] = 0;   A[2][1] = 0;    A[2][2] = 1;
}

constexpr int PGO_BLOCK = 256;

__global__ __launch_bounds__(PGO_BLOCK) void k_pgo_factors(PgoPtrs g, uint32_t F)
{
    __shared__ double red[PGO_BLOCK / 64];
    const uint32_t k = blockIdx.x * PGO_BLOCK + threadIdx.x;
    double c2 = 0.0;
    if (k < F) {
        const int i = g.fi[k], j = g.fj[k];
        const SE2 z{g.meas[4 * k], g.meas[4 * k + 1], g.meas[4 * k + 2], g.meas[4 * k + 3]};
        const SE2 v1{g.poses[4 * i], g.poses[4 * i + 1], g.poses[4 * i + 2], g.poses[4 * i + 3]};
        double e[3], Ji[3][3], Jj[3][3];
        if (j < 0) {
            pgo_log(pgo_mul(pgo_inverse(z), v1), e);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Ji[r][c] = r == c ? 1.0 : 0.0; Jj[r][c] = 0.0; }
        } else {
            const SE2 v2{g.poses[4 * j], g.poses[4 * j + 1], g.poses[4 * j + 2], g.poses[4 * j + 3]};
            pgo_log(pgo_mul(pgo_inverse(z), pgo_mul(pgo_inverse(v1), v2)), e);
            double Hinv[3][3], Hc[3][3];
            pgo_adj(v1, Hinv);
            pgo_adj(pgo_inverse(v2), Hc);
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
                Ji[r][c] = (Hc[r][0] * -Hinv[0][c] + Hc[r][1] * -Hinv[1][c]) + Hc[r][2] * -Hinv[2][c];
                Jj[r][c] = r == c ? 1.0 : 0.0;
            }
        }
        for (int r = 0; r < 3; ++r) {
            const double w = g.sqrt_info[3 * k + r];
            e[r] = e[r] * w;
            for (int c = 0; c < 3; ++c) { Ji[r][c] *= w; Jj[r][c] *= w; }
            g.err[3 * k + r] = e[r];
            c2 += e[r] * e[r];
        }
        for (int a = 0; a < 3; ++a) {
            g.fg[6 * k + a] = (Ji[0][a] * e[0] + Ji[1][a] * e[1]) + Ji[2][a] * e[2];
            g.fg[6 * k + 3 + a] = (Jj[0][a] * e[0] + Jj[1][a] * e[1]) + Jj[2][a] * e[2];
            for (int c = 0; c < 3; ++c) {
                g.fdi[9 * k + 3 * a + c] = (Ji[0][a] * Ji[0][c] + Ji[1][a] * Ji[1][c]) + Ji[2][a] * Ji[2][c];
                g.fdj[9 * k + 3 * a + c] = (Jj[0][a] * Jj[0][c] + Jj[1][a] * Jj[1][c]) + Jj[2][a] * Jj[2][c];
                g.hoff[9 * k + 3 * a + c] = j < 0 ? 0.0 : (Ji[0][a] * Jj[0][c] + Ji[1][a] * Jj[1][c]) + Ji[2][a] * Jj[2][c];
            }
        }
    }
    // chi^2: fixed-shape reduction (diagnostic only; the per-factor errors are the parity-relevant output)
    for (int o = 32; o > 0; o >>= 1) c2 += __shfl_xor(c2, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c2;
    __syncthreads();
    if (threadIdx.x == 0) { double t = 0; for (int w = 0; w < PGO_BLOCK / 64; ++w) t += red[w]; g.chi2_part[blockIdx.x] = t; }
}

__global__ __launch_bounds__(PGO_BLOCK) void k_pgo_reduce(PgoPtrs g, uint32_t N)
{
    const uint32_t v = blockIdx.x * PGO_BLOCK + threadIdx.x;
    if (v >= N) return;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, bb[3] = {0, 0, 0};
    for (int32_t q = g.inc_ptr[v]; q < g.inc_ptr[v + 1]; ++q) {
        const int32_t code = g.inc[q];
        const int32_t k = code >> 1;
        const double* D = (code & 1) ? g.fdj + 9 * (size_t)k : g.fdi + 9 * (size_t)k;
        const double* G = g.fg + 6 * (size_t)k + ((code & 1) ? 3 : 0);
        for (int t = 0; t < 9; ++t) H[t] += D[t];
        for (int t = 0; t < 3; ++t) bb[t] -= G[t];
    }
    for (int t = 0; t < 9; ++t) g.hdiag[9 * (size_t)v + t] = H[t];
    for (int t = 0; t < 3; ++t) g.b[3 * (size_t)v + t] = bb[t];
}

} // namespace lama_dev
