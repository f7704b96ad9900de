// covariance3.hpp -- Solver::calculateCovariance for the 3-parameter scan-matching problem, shared by Loc2D and lama::Solver.
#pragma once
#include <algorithm>
#include <cmath>

namespace lama {
namespace detail {

// Solver::calculateCovariance (src/nlls/solver.cpp:133-150) from the normal matrix A = J^T J the device returns (the
// weighted Jacobian itself never leaves the GPU).
//   rank: the pivots of ColPivHouseholderQR(J) are the square roots of the pivots of the diagonally pivoted Cholesky
//         factorisation of A (same pivot order: largest remaining column norm); rank() counts those above
//         3 eps x the largest.  Exactly degenerate Jacobians (a zero or repeated column: a featureless corridor,
//         open space) are classified as the reference does; for nearly dependent columns the test sees the squared
//         condition number and keeps "full rank" longer than a QR of J would -- there the reference's own answer is
//         rounding noise either way.
//   full rank: (J^T J)^-1.   else: V diag(sv > 1e-3 ? 1 / sv^2 : 3.0) V^T with sv^2, V the eigen-pairs of A
//         (= singular values / right singular vectors of J), by cyclic Jacobi rotations.
inline int rank_from_normal3(const double L[6] /*00,10,11,20,21,22*/)
{
    double A[3][3] = {{L[0], L[1], L[3]}, {L[1], L[2], L[4]}, {L[3], L[4], L[5]}};
    double piv[3] = {0, 0, 0};
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
        int best = k;
        for (int c = k + 1; c < 3; ++c) if (A[perm[c]][perm[c]] > A[perm[best]][perm[best]]) best = c;
        std::swap(perm[k], perm[best]);
        const int pk = perm[k];
        const double d = A[pk][pk];
        if (!(d > 0.0)) break;
        piv[k] = std::sqrt(d);
        for (int a = k + 1; a < 3; ++a)
            for (int b = k + 1; b < 3; ++b) A[perm[a]][perm[b]] -= A[perm[a]][pk] * A[pk][perm[b]] / d;
    }
    const double maxp = std::max(piv[0], std::max(piv[1], piv[2]));
    const double thr = 2.220446049250313e-16 * 3.0 * maxp;
    int rank = 0;
    for (int k = 0; k < 3; ++k) if (piv[k] > thr) ++rank;
    return rank;
}

inline void inverse_sym3(const double L[6], double out[9])
{
    const double a = L[0], b = L[1], c = L[3], e = L[2], f = L[4], i = L[5];       // [a b c; b e f; c f i]
    const double det = a * (e * i - f * f) - b * (b * i - f * c) + c * (b * f - e * c);
    out[0] = (e * i - f * f) / det; out[1] = (c * f - b * i) / det; out[2] = (b * f - c * e) / det;
    out[3] = out[1];                out[4] = (a * i - c * c) / det; out[5] = (c * b - a * f) / det;
    out[6] = out[2];                out[7] = out[5];                out[8] = (a * e - b * b) / det;
}

inline void deficient_cov3(const double L[6], double out[9])
{
    double A[3][3] = {{L[0], L[1], L[3]}, {L[1], L[2], L[4]}, {L[3], L[4], L[5]}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
        if (off == 0.0) break;
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0 || std::fabs(A[p][q]) <= 2.220446049250313e-16 * std::sqrt(std::fabs(A[p][p] * A[q][q]))) continue;
                rotated = true;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
                for (int k = 0; k < 3; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - s * akq; A[k][q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - s * aqk; A[q][k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
            }
        if (!rotated) break;
    }
    double d[3];
    for (int j = 0; j < 3; ++j) {
        const double sv = std::sqrt(std::max(A[j][j], 0.0));
        d[j] = sv > 1.e-3 ? 1.0 / (sv * sv) : 3.0;                                  // src/nlls/solver.cpp:147-148
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) out[3 * a + b] = V[a][0] * d[0] * V[b][0] + V[a][1] * d[1] * V[b][1] + V[a][2] * d[2] * V[b][2];
}


// out9 (row major 3 x 3) from the lower triangle of the weighted J^T J
inline void covariance_from_normal3(const double L[6], double out9[9])
{
    if (rank_from_normal3(L) == 3) inverse_sym3(L, out9); else deficient_cov3(L, out9);
}

} // namespace detail
} // namespace lama
