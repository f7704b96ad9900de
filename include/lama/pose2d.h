// lama/pose2d.h -- Pose2D over an SE2 state (unit complex + translation), host-side mirror of the reference's
// include/lama/pose2d.h:43-76 + the parts of Sophus::SE2d it relies on (include/lama/sophus/so2.hpp:168-176,
// 205-214,322-324,401-404; se2.hpp:154-167,262-265).  Semantics: a + b = a*b, a - b = a^-1 * b,
// every product re-normalises the complex number, rotation() = atan2(s, c).
#pragma once

#include "types.h"

namespace lama {

struct SO2d {
    Vector2d unit_complex_ = Vector2d(1.0, 0.0);
    const Vector2d& unit_complex() const { return unit_complex_; }
    double log() const { return std::atan2(unit_complex_.y(), unit_complex_.x()); }
};

struct SE2d {
    SO2d so2_;
    Vector2d translation_ = Vector2d(0.0, 0.0);
    SE2d() {}
    SE2d(double theta, const Vector2d& t);
    const SO2d& so2() const { return so2_; }
    const Vector2d& translation() const { return translation_; }
    Vector2d& translation() { return translation_; }
    SE2d inverse() const;
    SE2d operator*(const SE2d& o) const;
    SE2d& operator*=(const SE2d& o) { *this = *this * o; return *this; }
    // {c, s, tx, ty}: the layout the C-ABI of include/lama_hip.h uses
    void toArray(double out4[4]) const;
    static SE2d fromArray(const double in4[4]);
    // Sophus SE2::exp of the tangent [vx, vy, theta] (include/lama/sophus/se2.hpp:389-411; small-angle branch below 1e-10)
    static SE2d exp(double vx, double vy, double theta);
};

struct Pose2D {
    Pose2D() {}
    Pose2D(const double& x, const double& y, const double& rotation) : state(rotation, Vector2d(x, y)) {}
    Pose2D(const Vector2d& xy, const double& rotation) : state(rotation, xy) {}
    Pose2D(const SE2d& se2) : state(se2) {}

    Pose2D operator+(const Pose2D& other) const { return Pose2D(state * other.state); }
    Pose2D operator-(const Pose2D& other) const { return Pose2D(state.inverse() * other.state); }
    Pose2D& operator+=(const Pose2D& other) { state *= other.state; return *this; }
    Pose2D& operator-=(const Pose2D& other) { state = state.inverse() * other.state; return *this; }

    double x() const { return state.translation().x(); }
    double y() const { return state.translation().y(); }
    Vector2d xy() const { return state.translation(); }
    double rotation() const { return state.so2().log(); }

    SE2d state;
};

} // namespace lama
