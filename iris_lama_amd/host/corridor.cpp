// corridor.cpp -- seeded synthetic 1080-beam corridor log (the bench / test workload).
//
// Specification: SURVEY.md section 8(d) "Synthetic input (fixed, seeded)".
//   world   : rectangle x in [0,28] m, y in [0,4] m (closed), 6 square pillars 0.4 x 0.4 m centred at
//             x = 5,9,13,17,21,25 with y alternating 0.8 / 3.2
//   sensor  : `beams` beams over 270 deg (-135 .. +135, step 270/beams), max range 30 m,
//             range noise N(0, 0.01 m) from std::mt19937(1234); points (r cos phi, r sin phi, 0)
//   path    : start (2,2,0), steps of 0.6 m along +x up to x = 26 (k = 40), then back and forth (the robot
//             reverses, triangle wave of period 80); yaw_k = 0.05 sin(0.3 k)
//   odometry: truth composed with per-step drift N(0,0.01 m) on x and N(0,0.002 rad) on yaw,
//             cumulative, std::mt19937(4321)
// This is workload generation, not part of the reference (which ships no data).
#include <cmath>
#include <cstdint>
#include <random>
#include <vector>

#include "lama_host.h"

namespace {

struct Seg { double x0, y0, x1, y1; };

std::vector<Seg> world()
{
    std::vector<Seg> w;
    auto box = [&](double xa, double ya, double xb, double yb) {
        w.push_back({xa, ya, xb, ya});
        w.push_back({xb, ya, xb, yb});
        w.push_back({xb, yb, xa, yb});
        w.push_back({xa, yb, xa, ya});
    };
    box(0.0, 0.0, 28.0, 4.0);
    for (int k = 0; k < 6; ++k) {
        double cx = 5.0 + 4.0 * k;
        double cy = (k % 2 == 0) ? 0.8 : 3.2;
        box(cx - 0.2, cy - 0.2, cx + 0.2, cy + 0.2);
    }
    return w;
}

// distance along the ray (ox,oy)+t(dx,dy) to segment s, or +inf
double hit(const Seg& s, double ox, double oy, double dx, double dy)
{
    double ex = s.x1 - s.x0, ey = s.y1 - s.y0;
    double den = dx * ey - dy * ex;
    if (std::fabs(den) < 1e-14) return INFINITY;
    double t = ((s.x0 - ox) * ey - (s.y0 - oy) * ex) / den;
    double u = ((s.x0 - ox) * dy - (s.y0 - oy) * dx) / den;
    if (t <= 1e-9 || u < 0.0 || u > 1.0) return INFINITY;
    return t;
}

} // namespace

extern "C" int lama_corridor_generate(int steps, int beams, double* pts, double* odom_xyr, double* truth_xyr)
{
    if (steps < 0 || beams <= 0 || !pts || !odom_xyr) return -1;
    const std::vector<Seg> w = world();
    std::mt19937 gen_range(1234), gen_odom(4321);
    std::normal_distribution<double> n_range(0.0, 0.01), n_dx(0.0, 0.01), n_dyaw(0.0, 0.002);
    const double max_range = 30.0;
    const double deg = M_PI / 180.0;

    double ox = 0, oy = 0, oyaw = 0;     // odometry pose
    double px = 0, py = 0, pyaw = 0;     // previous true pose
    for (int k = 0; k <= steps; ++k) {
        const int kk = k % 80;
        const int tri = kk <= 40 ? kk : 80 - kk;
        const double tx = 2.0 + 0.6 * tri, ty = 2.0, tyaw = 0.05 * std::sin(0.3 * k);
        if (truth_xyr) { truth_xyr[3 * k] = tx; truth_xyr[3 * k + 1] = ty; truth_xyr[3 * k + 2] = tyaw; }
        if (k == 0) {
            ox = tx; oy = ty; oyaw = tyaw;
        } else {
            // true relative motion in the previous true frame
            double c = std::cos(pyaw), s = std::sin(pyaw);
            double ddx = c * (tx - px) + s * (ty - py);
            double ddy = -s * (tx - px) + c * (ty - py);
            double ddyaw = tyaw - pyaw;
            ddx += n_dx(gen_odom);
            ddyaw += n_dyaw(gen_odom);
            double co = std::cos(oyaw), so = std::sin(oyaw);
            ox += co * ddx - so * ddy;
            oy += so * ddx + co * ddy;
            oyaw += ddyaw;
        }
        odom_xyr[3 * k] = ox; odom_xyr[3 * k + 1] = oy; odom_xyr[3 * k + 2] = oyaw;
        px = tx; py = ty; pyaw = tyaw;

        for (int i = 0; i < beams; ++i) {
            const double phi = (-135.0 + (270.0 / beams) * i) * deg;
            const double a = tyaw + phi;
            const double dx = std::cos(a), dy = std::sin(a);
            double r = max_range;
            for (const Seg& s : w) r = std::fmin(r, hit(s, tx, ty, dx, dy));
            r += n_range(gen_range);
            double* p = pts + (size_t(k) * beams + i) * 3;
            p[0] = r * std::cos(phi);
            p[1] = r * std::sin(phi);
            p[2] = 0.0;
        }
    }
    return 0;
}
