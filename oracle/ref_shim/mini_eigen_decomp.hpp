// mini_eigen_decomp.hpp -- TEST INFRASTRUCTURE (see mini_eigen.hpp): ColPivHouseholderQR::rank()/info() and JacobiSVD
// singularValues()/matrixV() as Solver::calculateCovariance (src/nlls/solver.cpp:133-150) uses them.  Same algorithms as
// the restatement in oracle/lama_oracle.hpp (colpiv_qr_rank, svd_cov3), for any column count.
#pragma once
#include "mini_eigen.hpp"

namespace Eigen {

template <class MatrixType>
class ColPivHouseholderQR {
    typedef typename MatrixType::Scalar S;
    Index rank_;
public:
    template <class D> explicit ColPivHouseholderQR(const DenseBase<D>& Jin) : rank_(0)
    {
        Matrix<S, Dynamic, Dynamic> J = Jin.eval();
        const Index n = J.rows(), m = J.cols();
        std::vector<S> r((size_t)m, S(0));
        std::vector<Index> perm((size_t)m);
        for (Index c = 0; c < m; ++c) perm[(size_t)c] = c;
        for (Index k = 0; k < m && k < n; ++k) {
            Index best = k; S bn = S(-1);
            for (Index c = k; c < m; ++c) { S s2 = 0; for (Index i = k; i < n; ++i) s2 += J(i, perm[(size_t)c]) * J(i, perm[(size_t)c]); if (s2 > bn) { bn = s2; best = c; } }
            std::swap(perm[(size_t)k], perm[(size_t)best]);
            const Index c = perm[(size_t)k];
            const S norm = std::sqrt(bn);
            if (norm == S(0)) { r[(size_t)k] = 0; continue; }
            const S x0 = J(k, c);
            const S alpha = x0 > 0 ? -norm : norm;
            std::vector<S> v((size_t)(n - k));
            for (Index i = k; i < n; ++i) v[(size_t)(i - k)] = J(i, c);
            v[0] -= alpha;
            S vn2 = 0; for (S t : v) vn2 += t * t;
            if (vn2 > 0)
                for (Index cc = k; cc < m; ++cc) {
                    const Index col = perm[(size_t)cc];
                    S dot = 0; for (Index i = k; i < n; ++i) dot += v[(size_t)(i - k)] * J(i, col);
                    const S f = S(2) * dot / vn2;
                    for (Index i = k; i < n; ++i) J(i, col) -= f * v[(size_t)(i - k)];
                }
            r[(size_t)k] = std::abs(alpha);
        }
        S maxp = 0; for (S t : r) maxp = std::max(maxp, t);
        const S thr = NumTraits<S>::epsilon() * S(std::min(n, m)) * maxp;      // epsilon * diagonalSize() relative to the largest pivot
        for (S t : r) if (t > thr) ++rank_;
    }
    Index rank() const { return rank_; }
    ComputationInfo info() const { return Success; }
};

template <class MatrixType, int QRPreconditioner = 0>
class JacobiSVD {
    typedef typename MatrixType::Scalar S;
    Matrix<S, Dynamic, 1> sv_; Matrix<S, Dynamic, Dynamic> V_;
public:
    template <class D> JacobiSVD(const DenseBase<D>& Jin, unsigned int = 0)
    {
        Matrix<S, Dynamic, Dynamic> U = Jin.eval();
        const Index n = U.rows(), m = U.cols();
        V_ = Matrix<S, Dynamic, Dynamic>::Identity(m, m);
        for (int sweep = 0; sweep < 60; ++sweep) {
            bool rotated = false;
            for (Index p = 0; p + 1 < m; ++p)
                for (Index q = p + 1; q < m; ++q) {
                    S alpha = 0, beta = 0, gamma = 0;
                    for (Index i = 0; i < n; ++i) { alpha += U(i, p) * U(i, p); beta += U(i, q) * U(i, q); gamma += U(i, p) * U(i, q); }
                    if (gamma == S(0) || std::abs(gamma) <= S(1e-300) + NumTraits<S>::epsilon() * std::sqrt(alpha * beta)) continue;
                    rotated = true;
                    const S zeta = (beta - alpha) / (S(2) * gamma);
                    const S t = (zeta >= 0 ? S(1) : S(-1)) / (std::abs(zeta) + std::sqrt(S(1) + zeta * zeta));
                    const S c = S(1) / std::sqrt(S(1) + t * t), sn = c * t;
                    for (Index i = 0; i < n; ++i) { const S up = U(i, p), uq = U(i, q); U(i, p) = c * up - sn * uq; U(i, q) = sn * up + c * uq; }
                    for (Index i = 0; i < m; ++i) { const S vp = V_(i, p), vq = V_(i, q); V_(i, p) = c * vp - sn * vq; V_(i, q) = sn * vp + c * vq; }
                }
            if (!rotated) break;
        }
        sv_.resize(m);
        for (Index j = 0; j < m; ++j) { S s2 = 0; for (Index i = 0; i < n; ++i) s2 += U(i, j) * U(i, j); sv_(j) = std::sqrt(s2); }
    }
    const Matrix<S, Dynamic, 1>& singularValues() const { return sv_; }
    const Matrix<S, Dynamic, Dynamic>& matrixV() const { return V_; }
};

} // namespace Eigen
