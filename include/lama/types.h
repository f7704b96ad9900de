// lama/types.h -- value types of the host-side mirror of the LaMa interface.
//
// The reference's public API is typed on Eigen3 (include/lama/types.h:48-120 of the reference) and on the
// vendored Sophus SE2 (include/lama/sophus/se2.hpp).  Eigen is a hard external dependency there and is NOT
// available in the build image of this repository, so:
//   * when <Eigen/Core> is found, lama::Vector2d/Vector3d/Quaterniond ARE the Eigen types (the configuration
//     an iris_lama_ros build uses; compiled and run by the tests against an Eigen API stand-in -- see INTEGRATION.md);
//   * otherwise the minimal stand-ins below provide exactly the members this path touches
//     (x()/y()/z()/w(), operator[], norm()).  They are not an Eigen replacement.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#if defined(LAMA_USE_EIGEN) || (defined(__has_include) && __has_include(<Eigen/Core>) && !defined(LAMA_NO_EIGEN))
#include <Eigen/Core>
#include <Eigen/Geometry>
#define LAMA_HAVE_EIGEN 1
namespace lama {
using Eigen::Quaterniond;
using Eigen::Vector2d;
using Eigen::Vector3d;
typedef Eigen::Matrix<uint32_t, 3, 1> Vector3ui;      // include/lama/types.h of the reference
using Eigen::MatrixXd;
using Eigen::VectorXd;
using Eigen::Matrix3d;
}
#else
namespace lama {

struct Vector2d {
    double v[2] = {0, 0};
    Vector2d() {}
    Vector2d(double a, double b) { v[0] = a; v[1] = b; }
    double& x() { return v[0]; }
    double& y() { return v[1]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1]); }
};

struct Vector3d {
    double v[3] = {0, 0, 0};
    Vector3d() {}
    Vector3d(double a, double b, double c) { v[0] = a; v[1] = b; v[2] = c; }
    double& x() { return v[0]; }
    double& y() { return v[1]; }
    double& z() { return v[2]; }
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double& operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
    static Vector3d Zero() { return Vector3d(); }
};

struct Vector3ui {
    uint32_t v[3] = {0, 0, 0};
    Vector3ui() {}
    Vector3ui(uint32_t a, uint32_t b, uint32_t c) { v[0] = a; v[1] = b; v[2] = c; }
    uint32_t& operator()(int i) { return v[i]; }
    uint32_t operator()(int i) const { return v[i]; }
    uint32_t& operator[](int i) { return v[i]; }
    uint32_t operator[](int i) const { return v[i]; }
};

// dynamic vector / column-major matrix with the handful of members the nlls interface touches (size / rows / cols / resize /
// element access / data); NOT an Eigen replacement -- with Eigen installed these names are Eigen's
struct VectorXd {
    std::vector<double> v;
    VectorXd() {}
    explicit VectorXd(size_t n) : v(n, 0.0) {}
    static VectorXd Zero(size_t n) { return VectorXd(n); }
    size_t size() const { return v.size(); }
    size_t rows() const { return v.size(); }
    void resize(size_t n) { v.resize(n); }
    double& operator[](size_t i) { return v[i]; }
    double operator[](size_t i) const { return v[i]; }
    double& operator()(size_t i) { return v[i]; }
    double operator()(size_t i) const { return v[i]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
    double squaredNorm() const { double s = 0; for (double x : v) s += x * x; return s; }
};
struct MatrixXd {
    size_t r = 0, c = 0;
    std::vector<double> v;                 // column major
    MatrixXd() {}
    MatrixXd(size_t rows, size_t cols) : r(rows), c(cols), v(rows * cols, 0.0) {}
    size_t rows() const { return r; }
    size_t cols() const { return c; }
    void resize(size_t rows, size_t cols) { r = rows; c = cols; v.resize(rows * cols); }
    double& operator()(size_t i, size_t j) { return v[j * r + i]; }
    double operator()(size_t i, size_t j) const { return v[j * r + i]; }
    double* data() { return v.data(); }
    const double* data() const { return v.data(); }
};

struct Matrix3d {       // 3x3 row-major stand-in for Eigen::Matrix3d (Loc2D::getCovar)
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double operator()(int r, int c) const { return m[3 * r + c]; }
    double& operator()(int r, int c) { return m[3 * r + c]; }
    static Matrix3d Identity() { Matrix3d a; a.m[0] = a.m[4] = a.m[8] = 1.0; return a; }
    static Matrix3d Zero() { return Matrix3d(); }
};

struct Quaterniond {
    double w_ = 1, x_ = 0, y_ = 0, z_ = 0;
    Quaterniond() {}
    Quaterniond(double w, double x, double y, double z) : w_(w), x_(x), y_(y), z_(z) {}
    double w() const { return w_; }
    double x() const { return x_; }
    double y() const { return y_; }
    double z() const { return z_; }
    static Quaterniond Identity() { return Quaterniond(); }
};

} // namespace lama
#endif

namespace lama {

template <class T> using DynamicArray = std::vector<T>;

// include/lama/types.h:111-120 of the reference
struct PointCloudXYZ {
    typedef std::shared_ptr<PointCloudXYZ> Ptr;
    std::vector<Vector3d> points;
    Vector3d sensor_origin_ = Vector3d(0, 0, 0);
    Quaterniond sensor_orientation_ = Quaterniond(1, 0, 0, 0);
};

} // namespace lama
