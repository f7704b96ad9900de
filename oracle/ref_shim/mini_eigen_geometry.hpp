// mini_eigen_geometry.hpp -- TEST INFRASTRUCTURE (see mini_eigen.hpp): the Eigen/Geometry subset the reference uses.
// Conventions of the floating-point formulas follow oracle/lama_oracle.hpp (quat_to_matrix, angle_axis_z, affine_mul,
// affine_apply), which restate Eigen 3.3's published algorithms.
#pragma once
#include "mini_eigen.hpp"

namespace Eigen {

template <class S> class AngleAxis;
template <class S, int Dim, int Mode = Affine, int Opt = 0> class Transform;

template <class S, int Opt = 0>
class Quaternion {
    Matrix<S, 4, 1> c_;                          // x, y, z, w like Eigen's coeffs()
public:
    typedef S Scalar;
    Quaternion() {}
    Quaternion(S w, S x, S y, S z) { c_(0) = x; c_(1) = y; c_(2) = z; c_(3) = w; }
    template <class D> explicit Quaternion(const DenseBase<D>& m) { *this = m; }
    explicit Quaternion(const AngleAxis<S>& aa) { *this = aa; }
    static Quaternion Identity() { return Quaternion(S(1), S(0), S(0), S(0)); }
    Quaternion& setIdentity() { *this = Identity(); return *this; }
    S w() const { return c_(3); } S x() const { return c_(0); } S y() const { return c_(1); } S z() const { return c_(2); }
    S& w() { return c_(3); } S& x() { return c_(0); } S& y() { return c_(1); } S& z() { return c_(2); }
    const Matrix<S, 4, 1>& coeffs() const { return c_; }
    Matrix<S, 4, 1>& coeffs() { return c_; }
    Matrix<S, 3, 1> vec() const { return Matrix<S, 3, 1>(x(), y(), z()); }
    S squaredNorm() const { return c_.squaredNorm(); }
    S norm() const { return c_.norm(); }
    void normalize() { c_.normalize(); }
    Quaternion normalized() const { Quaternion q(*this); q.normalize(); return q; }
    Quaternion conjugate() const { return Quaternion(w(), -x(), -y(), -z()); }
    Quaternion inverse() const { const S n2 = squaredNorm(); Quaternion q = conjugate(); q.c_ /= n2; return q; }
    Quaternion operator*(const Quaternion& b) const
    {
        const Quaternion& a = *this;
        return Quaternion(a.w() * b.w() - a.x() * b.x() - a.y() * b.y() - a.z() * b.z(),
                          a.w() * b.x() + a.x() * b.w() + a.y() * b.z() - a.z() * b.y(),
                          a.w() * b.y() + a.y() * b.w() + a.z() * b.x() - a.x() * b.z(),
                          a.w() * b.z() + a.z() * b.w() + a.x() * b.y() - a.y() * b.x());
    }
    // Eigen QuaternionBase::toRotationMatrix
    Matrix<S, 3, 3> toRotationMatrix() const
    {
        Matrix<S, 3, 3> R;
        const S tx = S(2) * x(), ty = S(2) * y(), tz = S(2) * z();
        const S twx = tx * w(), twy = ty * w(), twz = tz * w();
        const S txx = tx * x(), txy = ty * x(), txz = tz * x();
        const S tyy = ty * y(), tyz = tz * y(), tzz = tz * z();
        R(0, 0) = S(1) - (tyy + tzz); R(0, 1) = txy - twz;          R(0, 2) = txz + twy;
        R(1, 0) = txy + twz;          R(1, 1) = S(1) - (txx + tzz); R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy;          R(2, 1) = tyz + twx;          R(2, 2) = S(1) - (txx + tyy);
        return R;
    }
    Matrix<S, 3, 3> matrix() const { return toRotationMatrix(); }
    Matrix<S, 3, 1> operator*(const Matrix<S, 3, 1>& v) const { return toRotationMatrix() * v; }
    Matrix<S, 3, 1> _transformVector(const Matrix<S, 3, 1>& v) const { return toRotationMatrix() * v; }
    Quaternion& operator=(const AngleAxis<S>& aa)
    {
        const S ha = S(0.5) * aa.angle();
        const S sn = std::sin(ha);
        c_(3) = std::cos(ha); c_(0) = sn * aa.axis()(0); c_(1) = sn * aa.axis()(1); c_(2) = sn * aa.axis()(2);
        return *this;
    }
    // Eigen quaternionbase_assign_impl<3,3> (Ken Shoemake)
    template <class D> Quaternion& operator=(const DenseBase<D>& mat)
    {
        S t = mat.coeff(0, 0) + mat.coeff(1, 1) + mat.coeff(2, 2);
        if (t > S(0)) {
            t = std::sqrt(t + S(1.0));
            w() = S(0.5) * t; t = S(0.5) / t;
            x() = (mat.coeff(2, 1) - mat.coeff(1, 2)) * t; y() = (mat.coeff(0, 2) - mat.coeff(2, 0)) * t; z() = (mat.coeff(1, 0) - mat.coeff(0, 1)) * t;
        } else {
            Index i = 0;
            if (mat.coeff(1, 1) > mat.coeff(0, 0)) i = 1;
            if (mat.coeff(2, 2) > mat.coeff(i, i)) i = 2;
            const Index j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(mat.coeff(i, i) - mat.coeff(j, j) - mat.coeff(k, k) + S(1.0));
            c_(i) = S(0.5) * t; t = S(0.5) / t;
            w() = (mat.coeff(k, j) - mat.coeff(j, k)) * t;
            c_(j) = (mat.coeff(j, i) + mat.coeff(i, j)) * t;
            c_(k) = (mat.coeff(k, i) + mat.coeff(i, k)) * t;
        }
        return *this;
    }
    template <class T> Quaternion<T> cast() const { return Quaternion<T>((T)w(), (T)x(), (T)y(), (T)z()); }
};
typedef Quaternion<double> Quaterniond;
typedef Quaternion<float> Quaternionf;

template <class S>
class AngleAxis {
    S angle_; Matrix<S, 3, 1> axis_;
public:
    typedef S Scalar;
    AngleAxis() : angle_(0) { axis_ = Matrix<S, 3, 1>(S(1), S(0), S(0)); }
    template <class D> AngleAxis(S angle, const DenseBase<D>& axis) : angle_(angle), axis_(axis) {}
    explicit AngleAxis(const Quaternion<S>& q) { *this = q; }
    S angle() const { return angle_; } S& angle() { return angle_; }
    const Matrix<S, 3, 1>& axis() const { return axis_; } Matrix<S, 3, 1>& axis() { return axis_; }
    AngleAxis& operator=(const Quaternion<S>& q)
    {
        S n = q.vec().norm();
        if (n < std::numeric_limits<S>::epsilon()) n = std::sqrt(q.vec().squaredNorm());
        if (n != S(0)) { angle_ = S(2) * std::atan2(n, std::abs(q.w())); if (q.w() < S(0)) n = -n; axis_ = q.vec() / n; }
        else { angle_ = S(0); axis_ = Matrix<S, 3, 1>(S(1), S(0), S(0)); }
        return *this;
    }
    // Eigen AngleAxis::toRotationMatrix
    Matrix<S, 3, 3> toRotationMatrix() const
    {
        Matrix<S, 3, 3> res;
        const S sn = std::sin(angle_), c = std::cos(angle_);
        const Matrix<S, 3, 1> sin_axis = sn * axis_;
        const Matrix<S, 3, 1> cos1_axis = (S(1) - c) * axis_;
        S tmp;
        tmp = cos1_axis.x() * axis_.y(); res(0, 1) = tmp - sin_axis.z(); res(1, 0) = tmp + sin_axis.z();
        tmp = cos1_axis.x() * axis_.z(); res(0, 2) = tmp + sin_axis.y(); res(2, 0) = tmp - sin_axis.y();
        tmp = cos1_axis.y() * axis_.z(); res(1, 2) = tmp - sin_axis.x(); res(2, 1) = tmp + sin_axis.x();
        res(0, 0) = cos1_axis.x() * axis_.x() + c;
        res(1, 1) = cos1_axis.y() * axis_.y() + c;
        res(2, 2) = cos1_axis.z() * axis_.z() + c;
        return res;
    }
    Matrix<S, 3, 3> matrix() const { return toRotationMatrix(); }
    Matrix<S, 3, 1> operator*(const Matrix<S, 3, 1>& v) const { return toRotationMatrix() * v; }
};
typedef AngleAxis<double> AngleAxisd;
typedef AngleAxis<float> AngleAxisf;

template <class S>
class Rotation2D {
    S a_;
public:
    Rotation2D() : a_(0) {}
    explicit Rotation2D(S a) : a_(a) {}
    template <class D> explicit Rotation2D(const DenseBase<D>& m) { fromRotationMatrix(m); }
    S angle() const { return a_; } S& angle() { return a_; }
    template <class D> Rotation2D& fromRotationMatrix(const DenseBase<D>& m) { a_ = std::atan2(m.coeff(1, 0), m.coeff(0, 0)); return *this; }
    Matrix<S, 2, 2> toRotationMatrix() const { const S sn = std::sin(a_), c = std::cos(a_); Matrix<S, 2, 2> R; R(0, 0) = c; R(0, 1) = -sn; R(1, 0) = sn; R(1, 1) = c; return R; }
    Matrix<S, 2, 2> matrix() const { return toRotationMatrix(); }
    Matrix<S, 2, 1> operator*(const Matrix<S, 2, 1>& v) const { return toRotationMatrix() * v; }
    Rotation2D inverse() const { return Rotation2D(-a_); }
};
typedef Rotation2D<double> Rotation2Dd;
typedef Rotation2D<float> Rotation2Df;

template <class S> struct UniformScaling { S f; explicit UniformScaling(S s) : f(s) {} S factor() const { return f; } };
inline UniformScaling<float> Scaling(float s) { return UniformScaling<float>(s); }
inline UniformScaling<double> Scaling(double s) { return UniformScaling<double>(s); }

template <class S, int Dim>
class Translation {
    Matrix<S, Dim, 1> v_;
public:
    Translation() {}
    Translation(S x, S y) { v_(0) = x; v_(1) = y; }
    Translation(S x, S y, S z) { v_(0) = x; v_(1) = y; v_(2) = z; }
    template <class D> explicit Translation(const DenseBase<D>& v) : v_(v) {}
    const Matrix<S, Dim, 1>& vector() const { return v_; }
    const Matrix<S, Dim, 1>& translation() const { return v_; }
    S x() const { return v_(0); } S y() const { return v_(1); } S z() const { return v_(2); }
    Transform<S, Dim, Affine> operator*(const UniformScaling<S>& s) const;
    Transform<S, Dim, Affine> operator*(const Quaternion<S>& q) const;
    Transform<S, Dim, Affine> operator*(const AngleAxis<S>& a) const;
    Transform<S, Dim, Affine> operator*(const Rotation2D<S>& r) const;
    template <class D> Transform<S, Dim, Affine> operator*(const DenseBase<D>& lin) const;
    Transform<S, Dim, Affine> operator*(const Transform<S, Dim, Affine>& t) const;
};
typedef Translation<double, 2> Translation2d;
typedef Translation<double, 3> Translation3d;
typedef Translation<float, 2> Translation2f;
typedef Translation<float, 3> Translation3f;

// Transform (Affine / Isometry): linear part + translation; the last row of the homogeneous matrix is implied
template <class S, int Dim, int Mode, int Opt>
class Transform {
    Matrix<S, Dim, Dim> L_; Matrix<S, Dim, 1> t_;
public:
    typedef S Scalar;
    typedef Matrix<S, Dim, Dim> LinearMatrixType;
    typedef Matrix<S, Dim, 1> VectorType;
    typedef Matrix<S, Dim + 1, Dim + 1> MatrixType;
    Transform() { L_.setIdentity(); t_.setZero(); }       // Eigen leaves it uninitialised
    template <int M2> Transform(const Transform<S, Dim, M2>& o) : L_(o.linear()), t_(o.translation()) {}
    Transform(const Translation<S, Dim>& tr) { L_.setIdentity(); t_ = tr.vector(); }
    Transform(const UniformScaling<S>& s) { L_.setIdentity(); L_ *= s.f; t_.setZero(); }
    Transform(const Quaternion<S>& q) { L_ = q.toRotationMatrix(); t_.setZero(); }
    Transform(const AngleAxis<S>& a) { L_ = a.toRotationMatrix(); t_.setZero(); }
    Transform(const Rotation2D<S>& r) { L_ = r.toRotationMatrix(); t_.setZero(); }
    template <class D> explicit Transform(const DenseBase<D>& m) { *this = m; }
    template <class D> Transform& operator=(const DenseBase<D>& m)
    {
        if (m.rows() == Dim) { for (int i = 0; i < Dim; ++i) for (int j = 0; j < Dim; ++j) L_(i, j) = m.coeff(i, j); t_.setZero(); if (m.cols() == Dim + 1) for (int i = 0; i < Dim; ++i) t_(i) = m.coeff(i, Dim); }
        else { for (int i = 0; i < Dim; ++i) { for (int j = 0; j < Dim; ++j) L_(i, j) = m.coeff(i, j); t_(i) = m.coeff(i, Dim); } }
        return *this;
    }
    Transform& operator=(const Quaternion<S>& q) { L_ = q.toRotationMatrix(); t_.setZero(); return *this; }
    Transform& operator=(const AngleAxis<S>& a) { L_ = a.toRotationMatrix(); t_.setZero(); return *this; }
    Transform& operator=(const Translation<S, Dim>& tr) { L_.setIdentity(); t_ = tr.vector(); return *this; }
    static Transform Identity() { return Transform(); }
    Transform& setIdentity() { L_.setIdentity(); t_.setZero(); return *this; }
    const LinearMatrixType& linear() const { return L_; } LinearMatrixType& linear() { return L_; }
    const VectorType& translation() const { return t_; } VectorType& translation() { return t_; }
    LinearMatrixType rotation() const { return L_; }     // exact for isometries, which is all the reference builds
    MatrixType matrix() const
    { MatrixType m; m.setIdentity(); for (int i = 0; i < Dim; ++i) { for (int j = 0; j < Dim; ++j) m(i, j) = L_(i, j); m(i, Dim) = t_(i); } return m; }
    S operator()(Index i, Index j) const { return i < Dim ? (j < Dim ? L_(i, j) : t_(i)) : (j < Dim ? S(0) : S(1)); }
    // Affine * Affine: linear = A.L * B.L ; translation = A.L * B.t + A.t
    Transform operator*(const Transform& o) const { Transform r; r.L_ = L_ * o.L_; r.t_ = (L_ * o.t_) + t_; return r; }
    Transform& operator*=(const Transform& o) { *this = *this * o; return *this; }
    // Affine * point = linear * p + translation
    VectorType operator*(const VectorType& p) const { return (L_ * p) + t_; }
    template <class D> VectorType operator*(const DenseBase<D>& p) const { return (L_ * p.eval()) + t_; }
    Transform operator*(const Translation<S, Dim>& tr) const { Transform r(*this); r.t_ = (L_ * tr.vector()) + t_; return r; }
    Transform operator*(const UniformScaling<S>& s) const { Transform r(*this); r.L_ *= s.f; return r; }
    Transform operator*(const Quaternion<S>& q) const { Transform r(*this); r.L_ = L_ * q.toRotationMatrix(); return r; }
    Transform operator*(const AngleAxis<S>& a) const { Transform r(*this); r.L_ = L_ * a.toRotationMatrix(); return r; }
    Transform operator*(const Rotation2D<S>& a) const { Transform r(*this); r.L_ = L_ * a.toRotationMatrix(); return r; }
    template <class D> Transform& translate(const DenseBase<D>& v) { t_ = (L_ * v.eval()) + t_; return *this; }
    template <class D> Transform& pretranslate(const DenseBase<D>& v) { t_ += v; return *this; }
    Transform& rotate(const Quaternion<S>& q) { L_ = L_ * q.toRotationMatrix(); return *this; }
    Transform& rotate(const AngleAxis<S>& a) { L_ = L_ * a.toRotationMatrix(); return *this; }
    Transform& rotate(const Rotation2D<S>& a) { L_ = L_ * a.toRotationMatrix(); return *this; }
    Transform& scale(S s) { L_ *= s; return *this; }
    // Transform::inverse(Affine): general linear inverse, translation = -(L^-1 t)
    Transform inverse(TransformTraits = (TransformTraits)Mode) const { Transform r; r.L_ = L_.inverse(); r.t_ = -(r.L_ * t_); return r; }
    template <class T> Transform<T, Dim, Mode> cast() const { Transform<T, Dim, Mode> r; r.linear() = L_.template cast<T>(); r.translation() = t_.template cast<T>(); return r; }
};
typedef Transform<double, 2, Affine> Affine2d;
typedef Transform<double, 3, Affine> Affine3d;
typedef Transform<float, 2, Affine> Affine2f;
typedef Transform<float, 3, Affine> Affine3f;
typedef Transform<double, 2, Isometry> Isometry2d;
typedef Transform<double, 3, Isometry> Isometry3d;

template <class S, int Dim> Transform<S, Dim, Affine> Translation<S, Dim>::operator*(const UniformScaling<S>& s) const
{ Transform<S, Dim, Affine> r; r.linear().setIdentity(); r.linear() *= s.f; r.translation() = v_; return r; }
template <class S, int Dim> Transform<S, Dim, Affine> Translation<S, Dim>::operator*(const Quaternion<S>& q) const
{ Transform<S, Dim, Affine> r; r.linear() = q.toRotationMatrix(); r.translation() = v_; return r; }
template <class S, int Dim> Transform<S, Dim, Affine> Translation<S, Dim>::operator*(const AngleAxis<S>& a) const
{ Transform<S, Dim, Affine> r; r.linear() = a.toRotationMatrix(); r.translation() = v_; return r; }
template <class S, int Dim> Transform<S, Dim, Affine> Translation<S, Dim>::operator*(const Rotation2D<S>& a) const
{ Transform<S, Dim, Affine> r; r.linear() = a.toRotationMatrix(); r.translation() = v_; return r; }
template <class S, int Dim> template <class D> Transform<S, Dim, Affine> Translation<S, Dim>::operator*(const DenseBase<D>& lin) const
{ Transform<S, Dim, Affine> r; r.linear() = lin; r.translation() = v_; return r; }
template <class S, int Dim> Transform<S, Dim, Affine> Translation<S, Dim>::operator*(const Transform<S, Dim, Affine>& t) const
{ Transform<S, Dim, Affine> r(t); r.translation() = t.translation() + v_; return r; }

} // namespace Eigen
