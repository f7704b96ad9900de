// pose2d.cpp -- SE2 arithmetic of the host mirror (see include/lama/pose2d.h for the reference lines).
#include <stdexcept>

#include "lama/pose2d.h"

namespace lama {

static void normalize(double& c, double& s)
{
    // Sophus SO2::normalize: throws SophusException when the complex number is (near) zero
    const double length = std::sqrt(c * c + s * s);
    if (length < 1e-10) throw std::runtime_error("lama::SE2d: complex number is (near) zero");
    c /= length;
    s /= length;
}

SE2d::SE2d(double theta, const Vector2d& t) : translation_(t)
{
    double c = std::cos(theta), s = std::sin(theta);
    normalize(c, s);
    so2_.unit_complex_ = Vector2d(c, s);
}

SE2d SE2d::inverse() const
{
    SE2d r;
    double c = so2_.unit_complex_.x(), s = -so2_.unit_complex_.y();
    normalize(c, s);
    r.so2_.unit_complex_ = Vector2d(c, s);
    const double mx = translation_.x() * -1.0, my = translation_.y() * -1.0;
    r.translation_ = Vector2d(c * mx - s * my, s * mx + c * my);
    return r;
}

SE2d SE2d::operator*(const SE2d& o) const
{
    SE2d r = *this;
    const double c = so2_.unit_complex_.x(), s = so2_.unit_complex_.y();
    r.translation_.x() += c * o.translation_.x() - s * o.translation_.y();
    r.translation_.y() += s * o.translation_.x() + c * o.translation_.y();
    double nc = c * o.so2_.unit_complex_.x() - s * o.so2_.unit_complex_.y();
    double ns = c * o.so2_.unit_complex_.y() + s * o.so2_.unit_complex_.x();
    normalize(nc, ns);
    r.so2_.unit_complex_ = Vector2d(nc, ns);
    return r;
}

SE2d SE2d::exp(double vx, double vy, double theta)
{
    SE2d r;
    double c = std::cos(theta), s = std::sin(theta);
    normalize(c, s);
    r.so2_.unit_complex_ = Vector2d(c, s);
    double a, b;                                   // sin(theta) / theta, (1 - cos(theta)) / theta
    if (std::fabs(theta) < 1e-10) {
        const double theta_sq = theta * theta;
        a = 1. - (1. / 6.) * theta_sq;
        b = 0.5 * theta - (1. / 24.) * theta * theta_sq;
    } else {
        a = s / theta;
        b = (1. - c) / theta;
    }
    r.translation_ = Vector2d(a * vx - b * vy, b * vx + a * vy);
    return r;
}

void SE2d::toArray(double out4[4]) const
{
    out4[0] = so2_.unit_complex_.x(); out4[1] = so2_.unit_complex_.y();
    out4[2] = translation_.x(); out4[3] = translation_.y();
}

SE2d SE2d::fromArray(const double in4[4])
{
    SE2d r;
    r.so2_.unit_complex_ = Vector2d(in4[0], in4[1]);
    r.translation_ = Vector2d(in4[2], in4[3]);
    return r;
}

} // namespace lama
