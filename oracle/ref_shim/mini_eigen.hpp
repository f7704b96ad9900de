// mini_eigen.hpp -- TEST INFRASTRUCTURE, not part of the product.
//
// The reference (iris-ua/iris_lama) needs Eigen3, which is not installed in this image, and its build system is
// not to be run.  This header is a small, eager (no expression templates) stand-in for the subset of the Eigen 3.3
// API that the reference's particle-filter path uses, written from scratch, so that the reference's OWN sources
// (src/pf_slam2d.cpp, src/sdm/*.cpp, src/nlls/*.cpp, src/match_surface_2d.cpp, the vendored Sophus headers, ...)
// compile unchanged from /root/reference into oracle/_ref/liblama_ref.so (recipe: oracle/Makefile.ref).  That library
// is what pins oracle/lama_oracle.hpp: tests/test_oracle_vs_reference.py runs both on the same scans.
//
// What is and is not pinned by this: every statement of the reference repository itself is the real code.  The
// arithmetic INSIDE Eigen (summation order of small fixed-size products, AngleAxis/Quaternion -> matrix, pivoted LDLT)
// is restated here with the same conventions as oracle/lama_oracle.hpp (its header lists them), because real Eigen is
// not available to compare with; integer and single-operation floating-point results do not depend on those choices.
#pragma once

#include <cassert>
#include <cstring>
#include <cstdlib>
#include <functional>
#include <iostream>
#include <string>
#include <utility>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <complex>
#include <algorithm>
#include <initializer_list>
#include <limits>
#include <memory>
#include <type_traits>
#include <vector>
#include <ostream>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#define EIGEN_STRONG_INLINE inline
#define EIGEN_ALWAYS_INLINE inline
#define EIGEN_DEVICE_FUNC
#define EIGEN_DEPRECATED
#define EIGEN_WORLD_VERSION 3
#define EIGEN_MAJOR_VERSION 3
#define EIGEN_MINOR_VERSION 7
#define EIGEN_DEFINE_STL_VECTOR_SPECIALIZATION(...)
#define EIGEN_INHERIT_ASSIGNMENT_EQUAL_OPERATOR(Derived) \
    using Base::operator=; \
    Derived& operator=(const Derived& other) { Base::operator=(other); return *this; }

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;
const int Infinity = -1;
enum { ColMajor = 0, RowMajor = 1, AutoAlign = 0, DontAlign = 2 };
enum { Unaligned = 0, Aligned = 16 };
enum { Lower = 1, Upper = 2 };
enum TransformTraits { Isometry = 1, Affine = 2, AffineCompact = 3, Projective = 4 };
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };
enum { ComputeFullU = 0x04, ComputeThinU = 0x08, ComputeFullV = 0x10, ComputeThinV = 0x20 };

template <class T> struct NumTraits {
    static T epsilon() { return std::numeric_limits<T>::epsilon(); }
    static T dummy_precision() { return T(1e-12); }
    static T highest() { return std::numeric_limits<T>::max(); }
    static T lowest() { return std::numeric_limits<T>::lowest(); }
};
template <> inline float NumTraits<float>::dummy_precision() { return 1e-5f; }

template <class T> using aligned_allocator = std::allocator<T>;
template <int O = 0, int I = 0> struct Stride {};

namespace internal {
template <class T> struct traits;              // specialised by Sophus for its own types
template <class T> struct traits<const T> : traits<T> {};
}

template <class S, int R, int C, int Opt = 0, int MR = R, int MC = C> class Matrix;
template <class S, int R, int C> class Block;
template <class D> class ArrayWrapper;
template <class D> class BoolArray;
template <class D> class ArrayLvalue;
template <class S> struct DiagonalWrapper { std::vector<S> d; };
template <class T, int MapOptions = 0, class StrideType = Stride<0, 0> > class Map;

namespace detail {
template <class D> struct xtraits;             // Scalar, Rows, Cols of a dense object
template <class S, int R, int C, int O, int MR, int MC> struct xtraits<Matrix<S, R, C, O, MR, MC> > { typedef S Scalar; enum { Rows = R, Cols = C }; };
template <class S, int R, int C> struct xtraits<Block<S, R, C> > { typedef S Scalar; enum { Rows = R, Cols = C }; };
template <class M, int O, class St> struct xtraits<Map<M, O, St> > : xtraits<typename std::remove_const<M>::type> {};
template <class D> struct xtraits<const D> : xtraits<D> {};
constexpr int prod_dim(int a, int b) { return (a == Dynamic || b == Dynamic) ? Dynamic : a * b; }
constexpr int pick_dim(int a, int b) { return a != Dynamic ? a : b; }

// x0 + x1 + ... in index order (the convention oracle/lama_oracle.hpp uses for Eigen's small sums)
template <class S, class F> inline S seq_sum(Index n, F f) { S acc = S(0); for (Index i = 0; i < n; ++i) acc = (i == 0) ? f(0) : acc + f(i); return acc; }
}

// ------------------------------------------------------------------------------------------------------------------
// DenseBase: everything that only needs rows(), cols(), coeff(i, j)
// ------------------------------------------------------------------------------------------------------------------
template <class D>
class DenseBase {
public:
    typedef typename detail::xtraits<D>::Scalar Scalar;
    enum { RowsAtCompileTime = detail::xtraits<D>::Rows, ColsAtCompileTime = detail::xtraits<D>::Cols,
           SizeAtCompileTime = detail::prod_dim(detail::xtraits<D>::Rows, detail::xtraits<D>::Cols) };
    typedef Matrix<Scalar, RowsAtCompileTime, ColsAtCompileTime> PlainObject;

    const D& derived() const { return *static_cast<const D*>(this); }
    D& derived() { return *static_cast<D*>(this); }
    Index rows() const { return derived().rows_(); }
    Index cols() const { return derived().cols_(); }
    Index size() const { return rows() * cols(); }
    Scalar coeff(Index i, Index j) const { return derived().at_(i, j); }
    Scalar coeff(Index i) const { return cols() == 1 ? coeff(i, 0) : (rows() == 1 ? coeff(0, i) : coeff(i % rows(), i / rows())); }
    Scalar operator()(Index i, Index j) const { return coeff(i, j); }
    Scalar operator()(Index i) const { return coeff(i); }
    Scalar operator[](Index i) const { return coeff(i); }
    Scalar x() const { return coeff(0); }
    Scalar y() const { return coeff(1); }
    Scalar z() const { return coeff(2); }
    Scalar w() const { return coeff(3); }

    PlainObject eval() const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = coeff(i, j); return r; }

    template <class T> Matrix<T, RowsAtCompileTime, ColsAtCompileTime> cast() const
    { Matrix<T, RowsAtCompileTime, ColsAtCompileTime> r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = static_cast<T>(coeff(i, j)); return r; }

    Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> transpose() const
    { Matrix<Scalar, ColsAtCompileTime, RowsAtCompileTime> r; r.resize(cols(), rows()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(j, i) = coeff(i, j); return r; }

    Scalar sum() const { const Index n = size(); return detail::seq_sum<Scalar>(n, [&](Index k) { return coeff(k); }); }
    Scalar mean() const { return sum() / Scalar(size()); }
    Scalar prod() const { Scalar acc = Scalar(1); for (Index k = 0; k < size(); ++k) acc = (k == 0) ? coeff(0) : acc * coeff(k); return acc; }
    Scalar squaredNorm() const { const Index n = size(); return detail::seq_sum<Scalar>(n, [&](Index k) { return coeff(k) * coeff(k); }); }
    Scalar norm() const { return std::sqrt(squaredNorm()); }
    template <int P> Scalar lpNorm() const
    { static_assert(P == Infinity, "only lpNorm<Infinity>"); Scalar m = Scalar(0); for (Index k = 0; k < size(); ++k) { const Scalar a = std::abs(coeff(k)); if (k == 0 || a > m) m = a; } return m; }
    Scalar maxCoeff() const { Scalar m = coeff(0); for (Index k = 1; k < size(); ++k) if (coeff(k) > m) m = coeff(k); return m; }
    Scalar minCoeff() const { Scalar m = coeff(0); for (Index k = 1; k < size(); ++k) if (coeff(k) < m) m = coeff(k); return m; }
    Scalar trace() const { return detail::seq_sum<Scalar>(std::min(rows(), cols()), [&](Index k) { return coeff(k, k); }); }
    template <class O> Scalar dot(const DenseBase<O>& o) const { return detail::seq_sum<Scalar>(size(), [&](Index k) { return coeff(k) * o.coeff(k); }); }
    PlainObject normalized() const { const Scalar n = norm(); PlainObject r = eval(); for (Index k = 0; k < r.size(); ++k) r.data()[k] = r.data()[k] / n; return r; }
    template <class O> PlainObject cwiseProduct(const DenseBase<O>& o) const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = coeff(i, j) * o.coeff(i, j); return r; }
    template <class O> PlainObject cwiseMin(const DenseBase<O>& o) const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = std::min(coeff(i, j), o.coeff(i, j)); return r; }
    template <class O> PlainObject cwiseMax(const DenseBase<O>& o) const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = std::max(coeff(i, j), o.coeff(i, j)); return r; }
    PlainObject cwiseSqrt() const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = std::sqrt(coeff(i, j)); return r; }
    PlainObject cwiseInverse() const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = Scalar(1) / coeff(i, j); return r; }
    PlainObject cwiseAbs() const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = std::abs(coeff(i, j)); return r; }
    PlainObject operator-() const { PlainObject r; r.resize(rows(), cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) r(i, j) = -coeff(i, j); return r; }
    template <class O> bool operator==(const DenseBase<O>& o) const { if (rows() != o.rows() || cols() != o.cols()) return false; for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) if (!(coeff(i, j) == o.coeff(i, j))) return false; return true; }
    template <class O> bool operator!=(const DenseBase<O>& o) const { return !(*this == o); }
    template <class O> bool isApprox(const DenseBase<O>& o, Scalar prec = NumTraits<Scalar>::dummy_precision()) const
    { Scalar d2 = 0, a2 = squaredNorm(), b2 = o.squaredNorm(); for (Index k = 0; k < size(); ++k) { const Scalar d = coeff(k) - o.coeff(k); d2 += d * d; } return d2 <= prec * prec * std::min(a2, b2); }
    bool allFinite() const { for (Index k = 0; k < size(); ++k) if (!std::isfinite((double)coeff(k))) return false; return true; }

    ArrayWrapper<PlainObject> array() const;
    const D& matrix() const { return derived(); }
    PlainObject inverse() const;
    Scalar determinant() const;
    DiagonalWrapper<Scalar> asDiagonal() const;
};

// ------------------------------------------------------------------------------------------------------------------
// storage
// ------------------------------------------------------------------------------------------------------------------
namespace detail {
template <class S, int R, int C, bool Fixed = (R != Dynamic && C != Dynamic)> struct Storage;
template <class S, int R, int C> struct Storage<S, R, C, true> {
    S d[R * C > 0 ? R * C : 1];
    Storage() { for (int k = 0; k < R * C; ++k) d[k] = S(); }     // real Eigen leaves them uninitialised; zero is a legal value of "uninitialised"
    Index rows() const { return R; } Index cols() const { return C; }
    void resize(Index r, Index c) { (void)r; (void)c; assert(r == R && c == C); }
    S* data() { return d; } const S* data() const { return d; }
};
template <class S, int R, int C> struct Storage<S, R, C, false> {
    std::vector<S> d; Index r_, c_;
    Storage() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {}
    Index rows() const { return r_; } Index cols() const { return c_; }
    void resize(Index r, Index c) { if (r != r_ || c != c_) { d.assign((size_t)(r * c), S()); r_ = r; c_ = c; } }   // Eigen: values lost on a size change
    S* data() { return d.data(); } const S* data() const { return d.data(); }
};
}

template <class S, class Owner = void> class CommaInit {
    S* p_; Index ld_, rows_, cols_, k_; Owner* owner_;
public:
    CommaInit(S* p, Index ld, Index rows, Index cols, S first, Owner* owner = nullptr) : p_(p), ld_(ld), rows_(rows), cols_(cols), k_(0), owner_(owner) { put(first); }
    void put(S v) { const Index i = k_ / cols_, j = k_ % cols_; assert(i < rows_); p_[i + j * ld_] = v; ++k_; }     // row by row, like Eigen
    CommaInit& operator,(S v) { put(v); return *this; }
    Owner& finished() { return *owner_; }
};

// writable interface shared by Matrix and Block: needs ref_(i, j)
template <class D>
class DenseWritable : public DenseBase<D> {
public:
    typedef DenseBase<D> Base;
    typedef typename Base::Scalar Scalar;
    using Base::derived; using Base::rows; using Base::cols; using Base::size;
    using Base::operator(); using Base::operator[]; using Base::x; using Base::y; using Base::z; using Base::w;
    Scalar& coeffRef(Index i, Index j) { return derived().ref_(i, j); }
    Scalar& coeffRef(Index i) { return cols() == 1 ? coeffRef(i, 0) : (rows() == 1 ? coeffRef(0, i) : coeffRef(i % rows(), i / rows())); }
    Scalar& operator()(Index i, Index j) { return coeffRef(i, j); }
    Scalar& operator()(Index i) { return coeffRef(i); }
    Scalar& operator[](Index i) { return coeffRef(i); }
    Scalar& x() { return coeffRef(0); }
    Scalar& y() { return coeffRef(1); }
    Scalar& z() { return coeffRef(2); }
    Scalar& w() { return coeffRef(3); }
    template <class O> D& assign_(const DenseBase<O>& o) { typename DenseBase<O>::PlainObject t = o.eval(); assert(t.rows() == rows() && t.cols() == cols()); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = t(i, j); return derived(); }
    template <class O> D& operator+=(const DenseBase<O>& o) { typename DenseBase<O>::PlainObject t = o.eval(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = coeffRef(i, j) + t(i, j); return derived(); }
    template <class O> D& operator-=(const DenseBase<O>& o) { typename DenseBase<O>::PlainObject t = o.eval(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = coeffRef(i, j) - t(i, j); return derived(); }
    D& operator*=(Scalar s) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = coeffRef(i, j) * s; return derived(); }
    D& operator/=(Scalar s) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = coeffRef(i, j) / s; return derived(); }
    D& fill(Scalar v) { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = v; return derived(); }
    D& setConstant(Scalar v) { return fill(v); }
    D& setZero() { return fill(Scalar(0)); }
    D& setOnes() { return fill(Scalar(1)); }
    D& setIdentity() { for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = (i == j) ? Scalar(1) : Scalar(0); return derived(); }
    void normalize() { const Scalar n = this->norm(); for (Index j = 0; j < cols(); ++j) for (Index i = 0; i < rows(); ++i) coeffRef(i, j) = coeffRef(i, j) / n; }
    using Base::array;
    ArrayLvalue<D> array();
    D& noalias() { return derived(); }                                  // evaluation is eager here: nothing can alias
    CommaInit<Scalar, D> operator<<(Scalar first) { return CommaInit<Scalar, D>(&coeffRef(0, 0), derived().ld_(), rows(), cols(), first, &derived()); }

    // sub-blocks
    Block<Scalar, Dynamic, Dynamic> block(Index r0, Index c0, Index nr, Index nc) { return Block<Scalar, Dynamic, Dynamic>(&coeffRef(0, 0) + r0 + c0 * derived().ld_(), derived().ld_(), nr, nc); }
    template <int NR, int NC> Block<Scalar, NR, NC> block(Index r0, Index c0) { return Block<Scalar, NR, NC>(&coeffRef(0, 0) + r0 + c0 * derived().ld_(), derived().ld_(), NR, NC); }
    Block<Scalar, 1, Base::ColsAtCompileTime> row(Index i) { return Block<Scalar, 1, Base::ColsAtCompileTime>(&coeffRef(0, 0) + i, derived().ld_(), 1, cols()); }
    Block<Scalar, Base::RowsAtCompileTime, 1> col(Index j) { return Block<Scalar, Base::RowsAtCompileTime, 1>(&coeffRef(0, 0) + j * derived().ld_(), derived().ld_(), rows(), 1); }
    template <int N> Block<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> head() { return seg_<N>(0); }
    template <int N> Block<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> tail() { return seg_<N>(size() - N); }
    template <int N> Block<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> segment(Index s) { return seg_<N>(s); }
    Block<Scalar, Dynamic, Dynamic> head(Index n) { return segd_(0, n); }
    Block<Scalar, Dynamic, Dynamic> tail(Index n) { return segd_(size() - n, n); }
    Block<Scalar, Dynamic, Dynamic> segment(Index s, Index n) { return segd_(s, n); }
    template <int NR, int NC> Block<Scalar, NR, NC> topLeftCorner() { return block<NR, NC>(0, 0); }
    Block<Scalar, Dynamic, 1> diagonal() { return Block<Scalar, Dynamic, 1>(&coeffRef(0, 0), derived().ld_(), std::min(rows(), cols()), 1, derived().ld_() + 1); }
    Matrix<Scalar, Dynamic, 1> diagonal() const { Matrix<Scalar, Dynamic, 1> m; const Index n = std::min(rows(), cols()); m.resize(n, 1); for (Index i = 0; i < n; ++i) m(i, 0) = this->coeff(i, i); return m; }
    // const versions return copies
    Matrix<Scalar, Dynamic, Dynamic> block(Index r0, Index c0, Index nr, Index nc) const { Matrix<Scalar, Dynamic, Dynamic> m; m.resize(nr, nc); for (Index j = 0; j < nc; ++j) for (Index i = 0; i < nr; ++i) m(i, j) = this->coeff(r0 + i, c0 + j); return m; }
    template <int NR, int NC> Matrix<Scalar, NR, NC> block(Index r0, Index c0) const { Matrix<Scalar, NR, NC> m; for (Index j = 0; j < NC; ++j) for (Index i = 0; i < NR; ++i) m(i, j) = this->coeff(r0 + i, c0 + j); return m; }
    Matrix<Scalar, 1, Base::ColsAtCompileTime> row(Index i) const { Matrix<Scalar, 1, Base::ColsAtCompileTime> m; m.resize(1, cols()); for (Index j = 0; j < cols(); ++j) m(0, j) = this->coeff(i, j); return m; }
    Matrix<Scalar, Base::RowsAtCompileTime, 1> col(Index j) const { Matrix<Scalar, Base::RowsAtCompileTime, 1> m; m.resize(rows(), 1); for (Index i = 0; i < rows(); ++i) m(i, 0) = this->coeff(i, j); return m; }
    template <int N> Matrix<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> head() const { return cseg_<N>(0); }
    template <int N> Matrix<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> tail() const { return cseg_<N>(size() - N); }
    template <int N> Matrix<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> segment(Index s) const { return cseg_<N>(s); }
    Matrix<Scalar, Dynamic, 1> head(Index n) const { Matrix<Scalar, Dynamic, 1> m; m.resize(n, 1); for (Index i = 0; i < n; ++i) m(i, 0) = this->coeff(i); return m; }
    Matrix<Scalar, Dynamic, 1> tail(Index n) const { Matrix<Scalar, Dynamic, 1> m; m.resize(n, 1); for (Index i = 0; i < n; ++i) m(i, 0) = this->coeff(size() - n + i); return m; }
    template <int NR, int NC> Matrix<Scalar, NR, NC> topLeftCorner() const { return block<NR, NC>(0, 0); }
    Matrix<Scalar, Dynamic, 1> segment(Index s, Index n) const { Matrix<Scalar, Dynamic, 1> m; m.resize(n, 1); for (Index i = 0; i < n; ++i) m(i, 0) = this->coeff(s + i); return m; }
private:
    template <int N> Block<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> seg_(Index s)
    {
        typedef Block<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> B;
        if (cols() == 1) return B(&coeffRef(0, 0) + s, derived().ld_(), N, 1);
        return B(&coeffRef(0, 0) + s * derived().ld_(), derived().ld_(), 1, N);
    }
    Block<Scalar, Dynamic, Dynamic> segd_(Index s, Index n)
    {
        if (cols() == 1) return Block<Scalar, Dynamic, Dynamic>(&coeffRef(0, 0) + s, derived().ld_(), n, 1);
        return Block<Scalar, Dynamic, Dynamic>(&coeffRef(0, 0) + s * derived().ld_(), derived().ld_(), 1, n);
    }
    template <int N> Matrix<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> cseg_(Index s) const
    { Matrix<Scalar, (Base::ColsAtCompileTime == 1 ? N : 1), (Base::ColsAtCompileTime == 1 ? 1 : N)> m; for (Index i = 0; i < N; ++i) m(i) = this->coeff(s + i); return m; }
};

// ------------------------------------------------------------------------------------------------------------------
// Matrix
// ------------------------------------------------------------------------------------------------------------------
template <class S, int R, int C, int Opt, int MR, int MC>
class Matrix : public DenseWritable<Matrix<S, R, C, Opt, MR, MC> > {
    detail::Storage<S, R, C> st_;
public:
    typedef S Scalar;
    typedef DenseWritable<Matrix> Base;
    using Base::operator(); using Base::operator[];
    Index rows_() const { return st_.rows(); }
    Index cols_() const { return st_.cols(); }
    Index ld_() const { return st_.rows(); }
    S at_(Index i, Index j) const { assert(i >= 0 && i < rows_() && j >= 0 && j < cols_()); return st_.data()[i + j * st_.rows()]; }
    S& ref_(Index i, Index j) { assert(i >= 0 && i < rows_() && j >= 0 && j < cols_()); return st_.data()[i + j * st_.rows()]; }
    S* data() { return st_.data(); }
    const S* data() const { return st_.data(); }
    void resize(Index r, Index c) { st_.resize(r, c); }
    void resize(Index n) { if (C == 1) st_.resize(n, 1); else st_.resize(1, n); }
    void conservativeResize(Index r, Index c) { Matrix t; t.resize(r, c); for (Index j = 0; j < std::min(c, cols_()); ++j) for (Index i = 0; i < std::min(r, rows_()); ++i) t(i, j) = at_(i, j); *this = t; }

    Matrix() {}
    Matrix(const Matrix&) = default;
    Matrix& operator=(const Matrix&) = default;
    // one integer: size of a dynamic vector; one scalar on a 1x1 ... (only the forms the reference uses)
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    explicit Matrix(T n) { if (R == Dynamic || C == Dynamic) resize((Index)n); else { assert(R * C == 1); st_.data()[0] = (S)n; } }
    template <class T0, class T1, class = typename std::enable_if<std::is_arithmetic<T0>::value && std::is_arithmetic<T1>::value>::type>
    Matrix(T0 a, T1 b) { if (R == Dynamic || C == Dynamic) { if (R == Dynamic && C == Dynamic) st_.resize((Index)a, (Index)b); else if (R * C == -2 || (R == 2 && C == 1) || (R == 1 && C == 2)) { two_(a, b); } else st_.resize((Index)a, (Index)b); } else two_(a, b); }
    Matrix(S a, S b, S c) { assert(this->size() == 3); st_.data()[0] = a; st_.data()[1] = b; st_.data()[2] = c; }
    Matrix(S a, S b, S c, S d) { assert(this->size() == 4); st_.data()[0] = a; st_.data()[1] = b; st_.data()[2] = c; st_.data()[3] = d; }
    template <class O> Matrix(const DenseBase<O>& o) { copy_(o); }
    template <class O> Matrix& operator=(const DenseBase<O>& o) { copy_(o); return *this; }
    template <class M> Matrix(const ArrayWrapper<M>& a);
    template <class M> Matrix& operator=(const ArrayWrapper<M>& a);

    static Matrix Zero() { Matrix m; m.fill(S(0)); return m; }
    static Matrix Zero(Index n) { Matrix m; m.resize(n); m.fill(S(0)); return m; }
    static Matrix Zero(Index r, Index c) { Matrix m; m.resize(r, c); m.fill(S(0)); return m; }
    static Matrix Ones() { Matrix m; m.fill(S(1)); return m; }
    static Matrix Ones(Index n) { Matrix m; m.resize(n); m.fill(S(1)); return m; }
    static Matrix Constant(S v) { Matrix m; m.fill(v); return m; }
    static Matrix Constant(Index n, S v) { Matrix m; m.resize(n); m.fill(v); return m; }
    static Matrix Constant(Index r, Index c, S v) { Matrix m; m.resize(r, c); m.fill(v); return m; }
    static Matrix Identity() { Matrix m; m.setIdentity(); return m; }
    static Matrix Identity(Index r, Index c) { Matrix m; m.resize(r, c); m.setIdentity(); return m; }
    static Matrix UnitX() { Matrix m; m.fill(S(0)); m(0) = S(1); return m; }
    static Matrix UnitY() { Matrix m; m.fill(S(0)); m(1) = S(1); return m; }
    static Matrix UnitZ() { Matrix m; m.fill(S(0)); m(2) = S(1); return m; }

    // A.selfadjointView<Lower>().ldlt().solve(b)  (src/nlls/gauss_newton.cpp:66)
    template <int UpLo> struct SelfAdjointView;
    template <int UpLo> SelfAdjointView<UpLo> selfadjointView() const;
    template <int UpLo> SelfAdjointView<UpLo> selfadjointView();
private:
    template <class T0, class T1> void two_(T0 a, T1 b) { st_.resize(C == 1 ? 2 : 1, C == 1 ? 1 : 2); st_.data()[0] = (S)a; st_.data()[1] = (S)b; }
    template <class O> void copy_(const DenseBase<O>& o)
    {
        // a row may be assigned to a column vector type and vice versa when one dimension is 1 (Eigen allows vectors)
        Index r = o.rows(), c = o.cols();
        if ((R == 1 && C != 1 && c == 1) || (C == 1 && R != 1 && r == 1)) std::swap(r, c);
        const bool tr = (r != o.rows());
        // self-assignment through a Block of myself: evaluate first
        std::vector<S> tmp((size_t)(r * c));
        for (Index j = 0; j < c; ++j) for (Index i = 0; i < r; ++i) tmp[(size_t)(i + j * r)] = tr ? o.coeff(j, i) : o.coeff(i, j);
        st_.resize(r, c);
        for (Index k = 0; k < r * c; ++k) st_.data()[k] = tmp[(size_t)k];
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Block: writable view of a column-major array
// ------------------------------------------------------------------------------------------------------------------
template <class S, int R, int C>
class Block : public DenseWritable<Block<S, R, C> > {
    S* p_; Index ld__, r_, c_, is_;
public:
    typedef S Scalar;
    typedef DenseWritable<Block> Base;
    using Base::operator(); using Base::operator[];
    Block(S* p, Index ld, Index r, Index c, Index inner = 1) : p_(p), ld__(ld), r_(r), c_(c), is_(inner) {}
    Block(const Block&) = default;
    Index rows_() const { return r_; }
    Index cols_() const { return c_; }
    Index ld_() const { return ld__; }
    S at_(Index i, Index j) const { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return p_[i * is_ + j * ld__]; }
    S& ref_(Index i, Index j) { assert(i >= 0 && i < r_ && j >= 0 && j < c_); return p_[i * is_ + j * ld__]; }
    S* data() { return p_; }
    const S* data() const { return p_; }
    Block& operator=(const Block& o) { return this->assign_(o); }
    template <class O> Block& operator=(const DenseBase<O>& o)
    {
        typename DenseBase<O>::PlainObject t = o.eval();
        if (t.rows() == r_ && t.cols() == c_) { for (Index j = 0; j < c_; ++j) for (Index i = 0; i < r_; ++i) ref_(i, j) = t(i, j); }
        else { assert(t.rows() == c_ && t.cols() == r_ && (r_ == 1 || c_ == 1)); for (Index j = 0; j < c_; ++j) for (Index i = 0; i < r_; ++i) ref_(i, j) = t(j, i); }
        return *this;
    }
};

// ------------------------------------------------------------------------------------------------------------------
// Map of a plain matrix type
// ------------------------------------------------------------------------------------------------------------------
template <class M, int MapOptions, class StrideType>
class Map : public DenseWritable<Map<M, MapOptions, StrideType> > {
    typedef typename std::remove_const<M>::type PM;
    typedef typename PM::Scalar S;
    S* p_; Index r_, c_;
public:
    typedef S Scalar;
    typedef DenseWritable<Map> Base;
    using Base::operator(); using Base::operator[];
    Map(const S* p) : p_(const_cast<S*>(p)), r_(PM::RowsAtCompileTime), c_(PM::ColsAtCompileTime) {}
    Map(const S* p, Index n) : p_(const_cast<S*>(p)), r_(PM::ColsAtCompileTime == 1 ? n : 1), c_(PM::ColsAtCompileTime == 1 ? 1 : n) {}
    Map(const S* p, Index r, Index c) : p_(const_cast<S*>(p)), r_(r), c_(c) {}
    Map(const Map&) = default;
    Index rows_() const { return r_; }
    Index cols_() const { return c_; }
    Index ld_() const { return r_; }
    S at_(Index i, Index j) const { return p_[i + j * r_]; }
    S& ref_(Index i, Index j) { return p_[i + j * r_]; }
    S* data() { return p_; }
    const S* data() const { return p_; }
    Map& operator=(const Map& o) { return this->assign_(o); }
    template <class O> Map& operator=(const DenseBase<O>& o) { return this->assign_(o); }
};

// ------------------------------------------------------------------------------------------------------------------
// arithmetic
// ------------------------------------------------------------------------------------------------------------------
#define MINI_EIGEN_CWISE(OP)                                                                                          \
    template <class A, class B>                                                                                       \
    Matrix<typename DenseBase<A>::Scalar, detail::pick_dim(DenseBase<A>::RowsAtCompileTime, DenseBase<B>::RowsAtCompileTime), \
           detail::pick_dim(DenseBase<A>::ColsAtCompileTime, DenseBase<B>::ColsAtCompileTime)>                      \
    operator OP(const DenseBase<A>& a, const DenseBase<B>& b)                                                         \
    {                                                                                                                 \
        Matrix<typename DenseBase<A>::Scalar, detail::pick_dim(DenseBase<A>::RowsAtCompileTime, DenseBase<B>::RowsAtCompileTime), \
               detail::pick_dim(DenseBase<A>::ColsAtCompileTime, DenseBase<B>::ColsAtCompileTime)> r;               \
        assert(a.rows() == b.rows() && a.cols() == b.cols());                                                         \
        r.resize(a.rows(), a.cols());                                                                                 \
        for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r(i, j) = a.coeff(i, j) OP b.coeff(i, j); \
        return r;                                                                                                     \
    }
MINI_EIGEN_CWISE(+)
MINI_EIGEN_CWISE(-)
#undef MINI_EIGEN_CWISE

template <class A, class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
typename DenseBase<A>::PlainObject operator*(const DenseBase<A>& a, T s)
{ typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols()); for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r(i, j) = a.coeff(i, j) * (typename DenseBase<A>::Scalar)s; return r; }
template <class A, class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
typename DenseBase<A>::PlainObject operator*(T s, const DenseBase<A>& a)
{ typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols()); for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r(i, j) = (typename DenseBase<A>::Scalar)s * a.coeff(i, j); return r; }
template <class A, class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
typename DenseBase<A>::PlainObject operator/(const DenseBase<A>& a, T s)
{ typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols()); for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r(i, j) = a.coeff(i, j) / (typename DenseBase<A>::Scalar)s; return r; }

// matrix product: every coefficient is sum_k a(i,k) b(k,j) in index order
template <class A, class B>
Matrix<typename DenseBase<A>::Scalar, DenseBase<A>::RowsAtCompileTime, DenseBase<B>::ColsAtCompileTime>
operator*(const DenseBase<A>& a, const DenseBase<B>& b)
{
    typedef typename DenseBase<A>::Scalar S;
    Matrix<S, DenseBase<A>::RowsAtCompileTime, DenseBase<B>::ColsAtCompileTime> r;
    assert(a.cols() == b.rows());
    r.resize(a.rows(), b.cols());
    const Index K = a.cols();
    for (Index j = 0; j < b.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r(i, j) = detail::seq_sum<S>(K, [&](Index k) { return a.coeff(i, k) * b.coeff(k, j); });
    return r;
}
template <class A>
typename DenseBase<A>::PlainObject operator*(const DenseBase<A>& a, const DiagonalWrapper<typename DenseBase<A>::Scalar>& d)
{ typename DenseBase<A>::PlainObject r; r.resize(a.rows(), a.cols()); for (Index j = 0; j < a.cols(); ++j) for (Index i = 0; i < a.rows(); ++i) r(i, j) = a.coeff(i, j) * d.d[(size_t)j]; return r; }
template <class D> DiagonalWrapper<typename DenseBase<D>::Scalar> DenseBase<D>::asDiagonal() const
{ DiagonalWrapper<Scalar> w; for (Index k = 0; k < size(); ++k) w.d.push_back(coeff(k)); return w; }

template <class D> std::ostream& operator<<(std::ostream& os, const DenseBase<D>& m)
{ for (Index i = 0; i < m.rows(); ++i) { for (Index j = 0; j < m.cols(); ++j) os << (j ? " " : "") << m.coeff(i, j); if (i + 1 < m.rows()) os << "\n"; } return os; }

// ------------------------------------------------------------------------------------------------------------------
// arrays (coefficient-wise world), eager
// ------------------------------------------------------------------------------------------------------------------
template <class M>
class BoolArray {
public:
    Matrix<int, M::RowsAtCompileTime, M::ColsAtCompileTime> b;
    bool any() const { for (Index k = 0; k < b.size(); ++k) if (b.data()[k]) return true; return false; }
    bool all() const { for (Index k = 0; k < b.size(); ++k) if (!b.data()[k]) return false; return true; }
    Index count() const { Index n = 0; for (Index k = 0; k < b.size(); ++k) n += b.data()[k] ? 1 : 0; return n; }
    // (cond).select(then, else) with scalars / matrices / arrays in either position
    template <class T> static typename M::Scalar pick_(const T& t, Index k, typename std::enable_if<std::is_arithmetic<T>::value>::type* = 0) { return (typename M::Scalar)t; }
    template <class T> static typename M::Scalar pick_(const DenseBase<T>& t, Index k) { return t.coeff(k); }
    template <class T> static typename M::Scalar pick_(const ArrayWrapper<T>& t, Index k) { return t.m.coeff(k); }
    template <class T, class E> ArrayWrapper<M> select(const T& t, const E& e) const
    { ArrayWrapper<M> r; r.m.resize(b.rows(), b.cols()); for (Index k = 0; k < b.size(); ++k) r.m.data()[k] = b.data()[k] ? pick_(t, k) : pick_(e, k); return r; }
};

template <class M>
class ArrayWrapper {
public:
    typedef typename M::Scalar Scalar;
    M m;
    ArrayWrapper() {}
    explicit ArrayWrapper(const M& mm) : m(mm) {}
    const M& matrix() const { return m; }
    Index size() const { return m.size(); }
    Scalar operator()(Index i) const { return m.coeff(i); }
    Scalar operator[](Index i) const { return m.coeff(i); }
    template <class T> Matrix<T, M::RowsAtCompileTime, M::ColsAtCompileTime> cast() const { return m.template cast<T>(); }
    template <class F> ArrayWrapper map_(F f) const { ArrayWrapper r; r.m.resize(m.rows(), m.cols()); for (Index k = 0; k < m.size(); ++k) r.m.data()[k] = f(m.data()[k]); return r; }
    template <class F> BoolArray<M> cmp_(F f) const { BoolArray<M> r; r.b.resize(m.rows(), m.cols()); for (Index k = 0; k < m.size(); ++k) r.b.data()[k] = f(m.data()[k]) ? 1 : 0; return r; }
    ArrayWrapper abs() const { return map_([](Scalar v) { return v < Scalar(0) ? Scalar(-v) : v; }); }
    ArrayWrapper square() const { return map_([](Scalar v) { return v * v; }); }
    ArrayWrapper inverse() const { return map_([](Scalar v) { return Scalar(1) / v; }); }
    ArrayWrapper sqrt() const { return map_([](Scalar v) { return (Scalar)std::sqrt(v); }); }
    ArrayWrapper floor() const { return map_([](Scalar v) { return (Scalar)std::floor(v); }); }
    ArrayWrapper round() const { return map_([](Scalar v) { return (Scalar)std::round(v); }); }
    Scalar sum() const { return m.sum(); }
    Scalar mean() const { return m.mean(); }
    Scalar maxCoeff() const { return m.maxCoeff(); }
    Scalar minCoeff() const { return m.minCoeff(); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> ArrayWrapper operator+(T s) const { return map_([s](Scalar v) { return (Scalar)(v + (Scalar)s); }); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> ArrayWrapper operator-(T s) const { return map_([s](Scalar v) { return (Scalar)(v - (Scalar)s); }); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> ArrayWrapper operator*(T s) const { return map_([s](Scalar v) { return (Scalar)(v * (Scalar)s); }); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> ArrayWrapper operator/(T s) const { return map_([s](Scalar v) { return (Scalar)(v / (Scalar)s); }); }
    ArrayWrapper operator+(const ArrayWrapper& o) const { ArrayWrapper r(m); for (Index k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] + o.m.data()[k]; return r; }
    ArrayWrapper operator-(const ArrayWrapper& o) const { ArrayWrapper r(m); for (Index k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] - o.m.data()[k]; return r; }
    ArrayWrapper operator*(const ArrayWrapper& o) const { ArrayWrapper r(m); for (Index k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] * o.m.data()[k]; return r; }
    ArrayWrapper operator/(const ArrayWrapper& o) const { ArrayWrapper r(m); for (Index k = 0; k < m.size(); ++k) r.m.data()[k] = m.data()[k] / o.m.data()[k]; return r; }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> BoolArray<M> operator<(T s) const { return cmp_([s](Scalar v) { return v < (Scalar)s; }); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> BoolArray<M> operator>(T s) const { return cmp_([s](Scalar v) { return v > (Scalar)s; }); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> BoolArray<M> operator<=(T s) const { return cmp_([s](Scalar v) { return v <= (Scalar)s; }); }
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type> BoolArray<M> operator>=(T s) const { return cmp_([s](Scalar v) { return v >= (Scalar)s; }); }
};
template <class D> ArrayWrapper<typename DenseBase<D>::PlainObject> DenseBase<D>::array() const { return ArrayWrapper<PlainObject>(eval()); }
template <class S, int R, int C, int O, int MR, int MC> template <class M> Matrix<S, R, C, O, MR, MC>::Matrix(const ArrayWrapper<M>& a) { copy_(a.m); }
template <class S, int R, int C, int O, int MR, int MC> template <class M> Matrix<S, R, C, O, MR, MC>& Matrix<S, R, C, O, MR, MC>::operator=(const ArrayWrapper<M>& a) { copy_(a.m); return *this; }

// `v.array() += s` on an lvalue: the coefficient-wise view of a writable object (reads like an ArrayWrapper, writes through)
template <class D>
class ArrayLvalue : public ArrayWrapper<typename DenseBase<D>::PlainObject> {
    D& d_;
public:
    typedef typename DenseBase<D>::Scalar Scalar;
    explicit ArrayLvalue(D& d) : ArrayWrapper<typename DenseBase<D>::PlainObject>(d.eval()), d_(d) {}
    template <class T> ArrayLvalue& operator+=(T s) { for (Index j = 0; j < d_.cols(); ++j) for (Index i = 0; i < d_.rows(); ++i) d_(i, j) = (Scalar)(d_(i, j) + (Scalar)s); this->m = d_.eval(); return *this; }
    template <class T> ArrayLvalue& operator-=(T s) { for (Index j = 0; j < d_.cols(); ++j) for (Index i = 0; i < d_.rows(); ++i) d_(i, j) = (Scalar)(d_(i, j) - (Scalar)s); this->m = d_.eval(); return *this; }
    template <class T> ArrayLvalue& operator*=(T s) { for (Index j = 0; j < d_.cols(); ++j) for (Index i = 0; i < d_.rows(); ++i) d_(i, j) = (Scalar)(d_(i, j) * (Scalar)s); this->m = d_.eval(); return *this; }
    template <class T> ArrayLvalue& operator/=(T s) { for (Index j = 0; j < d_.cols(); ++j) for (Index i = 0; i < d_.rows(); ++i) d_(i, j) = (Scalar)(d_(i, j) / (Scalar)s); this->m = d_.eval(); return *this; }
};
template <class D> ArrayLvalue<D> DenseWritable<D>::array() { return ArrayLvalue<D>(derived()); }

// ------------------------------------------------------------------------------------------------------------------
// inverse / determinant (small sizes by cofactors like Eigen; general size by partial-pivot LU)
// ------------------------------------------------------------------------------------------------------------------
template <class D> typename DenseBase<D>::Scalar DenseBase<D>::determinant() const
{
    const Index n = rows();
    PlainObject a = eval();
    if (n == 1) return a(0, 0);
    if (n == 2) return a(0, 0) * a(1, 1) - a(1, 0) * a(0, 1);
    if (n == 3) return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) + a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
    Scalar det = 1;
    for (Index k = 0; k < n; ++k) {
        Index p = k; for (Index i = k + 1; i < n; ++i) if (std::abs(a(i, k)) > std::abs(a(p, k))) p = i;
        if (a(p, k) == Scalar(0)) return Scalar(0);
        if (p != k) { for (Index j = 0; j < n; ++j) std::swap(a(k, j), a(p, j)); det = -det; }
        det *= a(k, k);
        for (Index i = k + 1; i < n; ++i) { const Scalar f = a(i, k) / a(k, k); for (Index j = k; j < n; ++j) a(i, j) -= f * a(k, j); }
    }
    return det;
}
template <class D> typename DenseBase<D>::PlainObject DenseBase<D>::inverse() const
{
    const Index n = rows();
    assert(n == cols());
    PlainObject a = eval(), r; r.resize(n, n);
    if (n == 1) { r(0, 0) = Scalar(1) / a(0, 0); return r; }
    if (n == 2) { const Scalar invdet = Scalar(1) / (a(0, 0) * a(1, 1) - a(1, 0) * a(0, 1)); r(0, 0) = a(1, 1) * invdet; r(1, 0) = -a(1, 0) * invdet; r(0, 1) = -a(0, 1) * invdet; r(1, 1) = a(0, 0) * invdet; return r; }
    if (n == 3 && RowsAtCompileTime == 3) {      // Eigen: cofactors of the first column, determinant from them, then the rest
        auto cof = [&](int i, int j) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3; return a(i1, j1) * a(i2, j2) - a(i1, j2) * a(i2, j1); };
        const Scalar c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
        const Scalar det = (c0 * a(0, 0) + c1 * a(0, 1)) + c2 * a(0, 2);     // cofactors_col0 . row 0 ... (transposed pairing as in Eigen)
        const Scalar invdet = Scalar(1) / det;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = cof(j, i) * invdet;
        return r;
    }
    // general: Gauss-Jordan with partial pivoting
    r.setIdentity();
    for (Index k = 0; k < n; ++k) {
        Index p = k; for (Index i = k + 1; i < n; ++i) if (std::abs(a(i, k)) > std::abs(a(p, k))) p = i;
        if (p != k) for (Index j = 0; j < n; ++j) { std::swap(a(k, j), a(p, j)); std::swap(r(k, j), r(p, j)); }
        const Scalar piv = a(k, k);
        for (Index j = 0; j < n; ++j) { a(k, j) /= piv; r(k, j) /= piv; }
        for (Index i = 0; i < n; ++i) if (i != k) { const Scalar f = a(i, k); if (f != Scalar(0)) for (Index j = 0; j < n; ++j) { a(i, j) -= f * a(k, j); r(i, j) -= f * r(k, j); } }
    }
    return r;
}

// ------------------------------------------------------------------------------------------------------------------
// LDLT (robust Cholesky with pivoting) -- Eigen 3.3 ldlt_inplace<Lower>::unblocked + LDLT::_solve_impl, the same
// restatement as oracle/lama_oracle.hpp ldlt3_solve, for any size
// ------------------------------------------------------------------------------------------------------------------
template <class MatrixType, int UpLo = Lower>
class LDLT {
    typedef typename MatrixType::Scalar S;
    Matrix<S, Dynamic, Dynamic> m_; std::vector<Index> tr_; Index n_;
public:
    LDLT() : n_(0) {}
    template <class D> explicit LDLT(const DenseBase<D>& a) { compute(a); }
    template <class D> LDLT& compute(const DenseBase<D>& ain)
    {
        n_ = ain.rows();
        const Index size = n_;
        m_.resize(size, size);
        for (Index i = 0; i < size; ++i) for (Index j = 0; j < size; ++j) m_(i, j) = (j <= i) ? ain.coeff(i, j) : ain.coeff(j, i);     // lower triangle is the input
        tr_.assign((size_t)size, 0);
        std::vector<S> temp((size_t)size);
        auto& m = m_;
        for (Index k = 0; k < size; ++k) {
            Index big = k; S best = std::abs(m(k, k));
            for (Index i = k + 1; i < size; ++i) if (std::abs(m(i, i)) > best) { best = std::abs(m(i, i)); big = i; }
            tr_[(size_t)k] = big;
            if (k != big) {
                const Index s = size - big - 1;
                for (Index j = 0; j < k; ++j) std::swap(m(k, j), m(big, j));
                for (Index i = 0; i < s; ++i) std::swap(m(size - s + i, k), m(size - s + i, big));
                std::swap(m(k, k), m(big, big));
                for (Index i = k + 1; i < big; ++i) { const S tmp = m(i, k); m(i, k) = m(big, i); m(big, i) = tmp; }
            }
            const Index rs = size - k - 1;
            if (k > 0) {
                for (Index j = 0; j < k; ++j) temp[(size_t)j] = m(j, j) * m(k, j);
                m(k, k) -= detail::seq_sum<S>(k, [&](Index j) { return m(k, j) * temp[(size_t)j]; });
                for (Index i = 0; i < rs; ++i) m(k + 1 + i, k) -= detail::seq_sum<S>(k, [&](Index j) { return m(k + 1 + i, j) * temp[(size_t)j]; });
            }
            const S realAkk = m(k, k);
            const bool pivot_is_valid = std::abs(realAkk) > S(0);
            if (k == 0 && !pivot_is_valid) { for (Index j = 0; j < size; ++j) tr_[(size_t)j] = j; break; }
            if (rs > 0 && pivot_is_valid) for (Index i = 0; i < rs; ++i) m(k + 1 + i, k) /= realAkk;
        }
        return *this;
    }
    template <class D> Matrix<S, Dynamic, 1> solve(const DenseBase<D>& b) const
    {
        const Index size = n_;
        Matrix<S, Dynamic, 1> d; d.resize(size);
        for (Index i = 0; i < size; ++i) d(i) = b.coeff(i);
        for (Index k = 0; k < size; ++k) if (tr_[(size_t)k] != k) std::swap(d(k), d(tr_[(size_t)k]));
        for (Index i = 1; i < size; ++i) d(i) -= detail::seq_sum<S>(i, [&](Index j) { return m_(i, j) * d(j); });
        const S tol = S(1) / NumTraits<S>::highest();
        for (Index i = 0; i < size; ++i) { if (std::abs(m_(i, i)) > tol) d(i) /= m_(i, i); else d(i) = S(0); }
        for (Index i = size - 2; i >= 0; --i) d(i) -= detail::seq_sum<S>(size - 1 - i, [&](Index jj) { const Index j = i + 1 + jj; return m_(j, i) * d(j); });
        for (Index k = size - 1; k >= 0; --k) if (tr_[(size_t)k] != k) std::swap(d(k), d(tr_[(size_t)k]));
        return d;
    }
    ComputationInfo info() const { return Success; }
};

// LLT (standard Cholesky) -- Eigen 3.3 llt_inplace<Scalar, Lower>::unblocked on the lower triangle (an Upper view works on the
// transposed problem, i.e. the same numbers) + solve; the restatement of oracle/lama_oracle.hpp llt3_solve for any size
template <class MatrixType, int UpLo = Lower>
class LLT {
    typedef typename MatrixType::Scalar S;
    Matrix<S, Dynamic, Dynamic> L_; Index n_; bool ok_;
public:
    LLT() : n_(0), ok_(false) {}
    template <class D> explicit LLT(const DenseBase<D>& a) { compute(a); }
    template <class V, class = decltype(std::declval<const V&>().m)> explicit LLT(const V& view) { compute(view.m); }      // LLT<MatrixXd>(A.selfadjointView<Upper>())
    Matrix<S, Dynamic, Dynamic> matrixL() const { return L_; }
    Matrix<S, Dynamic, Dynamic> matrixU() const { Matrix<S, Dynamic, Dynamic> u; u.resize(n_, n_); for (Index i = 0; i < n_; ++i) for (Index j = 0; j < n_; ++j) u(i, j) = L_(j, i); return u; }
    template <class D> LLT& compute(const DenseBase<D>& ain)
    {
        n_ = ain.rows(); ok_ = true;
        const Index size = n_;
        L_.resize(size, size);
        for (Index i = 0; i < size; ++i) for (Index j = 0; j < size; ++j) L_(i, j) = (UpLo == Lower) ? ((j <= i) ? ain.coeff(i, j) : S(0)) : ((j <= i) ? ain.coeff(j, i) : S(0));
        for (Index k = 0; k < size; ++k) {
            const Index rs = size - k - 1;
            S x = L_(k, k);
            if (k > 0) x -= detail::seq_sum<S>(k, [&](Index j) { return L_(k, j) * L_(k, j); });
            if (x <= S(0)) { ok_ = false; return *this; }
            L_(k, k) = x = std::sqrt(x);
            if (k > 0 && rs > 0) for (Index i = 0; i < rs; ++i) L_(k + 1 + i, k) -= detail::seq_sum<S>(k, [&](Index j) { return L_(k + 1 + i, j) * L_(k, j); });
            if (rs > 0) for (Index i = 0; i < rs; ++i) L_(k + 1 + i, k) /= x;
        }
        return *this;
    }
    template <class D> Matrix<S, Dynamic, 1> solve(const DenseBase<D>& b) const
    {
        const Index size = n_;
        Matrix<S, Dynamic, 1> d; d.resize(size);
        for (Index i = 0; i < size; ++i) d(i) = b.coeff(i);
        for (Index i = 0; i < size; ++i) { if (i > 0) d(i) -= detail::seq_sum<S>(i, [&](Index j) { return L_(i, j) * d(j); }); d(i) /= L_(i, i); }
        for (Index i = size - 1; i >= 0; --i) { if (i + 1 < size) d(i) -= detail::seq_sum<S>(size - 1 - i, [&](Index jj) { const Index j = i + 1 + jj; return L_(j, i) * d(j); }); d(i) /= L_(i, i); }
        return d;
    }
    ComputationInfo info() const { return ok_ ? Success : NumericalIssue; }
};
template <class S, int R, int C, int O, int MR, int MC> template <int UpLo>
struct Matrix<S, R, C, O, MR, MC>::SelfAdjointView {
    Matrix m;
    Matrix* target = nullptr;      // set by the non-const selfadjointView(): rankUpdate writes through
    // this += u * u^T on the viewed triangle (Eigen: SelfAdjointView::rankUpdate; sums in index order)
    template <class D> SelfAdjointView& rankUpdate(const DenseBase<D>& u)
    {
        Matrix& t = target ? *target : m;
        for (Index j = 0; j < t.cols(); ++j) for (Index i = 0; i < t.rows(); ++i) {
            if ((UpLo == Lower) ? (i < j) : (i > j)) continue;
            S acc = t(i, j);
            for (Index k = 0; k < u.cols(); ++k) acc += u.coeff(i, k) * u.coeff(j, k);
            t(i, j) = acc;
        }
        return *this;
    }
    LDLT<Matrix<S, Dynamic, Dynamic>, UpLo> ldlt() const { static_assert(UpLo == Lower, "only Lower"); return LDLT<Matrix<S, Dynamic, Dynamic>, UpLo>(m); }
    LLT<Matrix<S, Dynamic, Dynamic>, UpLo> llt() const { return LLT<Matrix<S, Dynamic, Dynamic>, UpLo>(m); }
};
template <class S, int R, int C, int O, int MR, int MC> template <int UpLo>
typename Matrix<S, R, C, O, MR, MC>::template SelfAdjointView<UpLo> Matrix<S, R, C, O, MR, MC>::selfadjointView() const { return SelfAdjointView<UpLo>{*this, nullptr}; }
template <class S, int R, int C, int O, int MR, int MC> template <int UpLo>
typename Matrix<S, R, C, O, MR, MC>::template SelfAdjointView<UpLo> Matrix<S, R, C, O, MR, MC>::selfadjointView() { return SelfAdjointView<UpLo>{*this, this}; }

// ------------------------------------------------------------------------------------------------------------------
// typedefs
// ------------------------------------------------------------------------------------------------------------------
#define MINI_EIGEN_TYPEDEFS(T, Sfx)                                             \
    typedef Matrix<T, 2, 1> Vector2##Sfx; typedef Matrix<T, 3, 1> Vector3##Sfx; typedef Matrix<T, 4, 1> Vector4##Sfx; \
    typedef Matrix<T, Dynamic, 1> VectorX##Sfx; typedef Matrix<T, 1, Dynamic> RowVectorX##Sfx;                       \
    typedef Matrix<T, 1, 2> RowVector2##Sfx; typedef Matrix<T, 1, 3> RowVector3##Sfx;                                \
    typedef Matrix<T, 2, 2> Matrix2##Sfx; typedef Matrix<T, 3, 3> Matrix3##Sfx; typedef Matrix<T, 4, 4> Matrix4##Sfx; \
    typedef Matrix<T, Dynamic, Dynamic> MatrixX##Sfx;
MINI_EIGEN_TYPEDEFS(double, d)
MINI_EIGEN_TYPEDEFS(float, f)
MINI_EIGEN_TYPEDEFS(int, i)
#undef MINI_EIGEN_TYPEDEFS

} // namespace Eigen
