// lse_check.cpp -- research: runs the corridor log twice through the CPU oracle, once with the faithful
// DynamicDistanceMap::update() and once with the level-synchronous prototype (lse_proto.hpp) hooked in, and compares the
// complete priority-queue operation traces (every pop, every push, in order) and the map checksums.
//   g++ -O2 -std=c++14 -ffp-contract=off -pthread -I oracle -I include tools/research/lse_check.cpp iris_lama_amd/host/corridor.cpp -o /tmp/lse_check
#include <cstdio>
#include <cstdlib>
#include "lse_proto.hpp"
extern "C" int lama_corridor_generate(int steps, int beams, double* pts, double* odom_xyr, double* truth_xyr);
using namespace orc;
static std::vector<std::vector<BfTraceRec>> run(int P, int S, double gain, bool lse_on, std::vector<double>& pts, std::vector<double>& odom, std::vector<double>& truth, std::vector<uint64_t>& sums)
{
    const int beams = 1080;
    bf_update_hook() = lse_on ? &lse::update : nullptr;
    PFOptions o; o.particles = P; o.seed = 42; o.threads = -1; o.meas_sigma_gain = gain;
    PFSlam2D pf(o);
    pf.setPrior(se2_from_xyr(truth[0], truth[1], truth[2]));
    std::vector<std::vector<BfTraceRec>> out;
    std::vector<BfTraceRec> trace;
    bf_trace() = &trace;
    for (int k = 0; k <= S; ++k) {
        Scan s; s.points.resize(beams);
        for (int i = 0; i < beams; ++i) s.points[i] = V3d{pts[((size_t)k * beams + i) * 3], pts[((size_t)k * beams + i) * 3 + 1], 0.0};
        trace.clear();
        pf.update(s, se2_from_xyr(odom[3 * k], odom[3 * k + 1], odom[3 * k + 2]), 0.1 * k);
        out.push_back(trace);
    }
    // checksum of every particle's distance map: all cells + masks
    for (auto& p : pf.particles()) {
        uint64_t acc = 0;
        for (auto& kv : p.dm->patches) {
            uint64_t h = kv.first * 0x9E3779B97F4A7C15ull;
            for (uint8_t b : kv.second->data) h = (h ^ b) * 0x100000001B3ull;
            for (uint64_t w : kv.second->mask) h = (h ^ w) * 0x100000001B3ull;
            acc += h;
        }
        sums.push_back(acc);
    }
    bf_trace() = nullptr; bf_update_hook() = nullptr;
    return out;
}
int main(int argc, char** argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 10, S = argc > 2 ? atoi(argv[2]) : 36;
    const double gain = argc > 3 ? atof(argv[3]) : 3.0;
    const int beams = 1080;
    std::vector<double> pts((size_t)(S + 1) * beams * 3), odom(3 * (S + 1)), truth(3 * (S + 1));
    lama_corridor_generate(S, beams, pts.data(), odom.data(), truth.data());
    std::vector<uint64_t> sa, sb;
    auto A = run(P, S, gain, false, pts, odom, truth, sa);
    auto B = run(P, S, gain, true, pts, odom, truth, sb);
    int bad = 0;
    for (int k = 0; k <= S && !bad; ++k) {
        if (A[k].size() != B[k].size()) { printf("scan %d: trace length %zu vs %zu\n", k, A[k].size(), B[k].size()); }
        const size_t n = std::min(A[k].size(), B[k].size());
        for (size_t i = 0; i < n; ++i) {
            const BfTraceRec &a = A[k][i], &b = B[k][i];
            if (a.op != b.op || a.prio != b.prio || a.x != b.x || a.y != b.y) {
                printf("scan %d rec %zu: ref {%u %u %u %u} lse {%u %u %u %u}\n", k, i, a.op, a.prio, a.x, a.y, b.op, b.prio, b.x, b.y);
                bad = 1; break;
            }
        }
        if (A[k].size() != B[k].size()) bad = 1;
    }
    if (sa != sb) { printf("map checksums differ\n"); bad = 1; }
    const LseStats& st = lse_stats();
    printf("%s  P=%d scans=%d gain=%g\n", bad ? "MISMATCH" : "IDENTICAL traces and maps", P, S, gain);
    printf("levels %lu plans %lu passes %lu vevents %lu pops fast %lu serial %lu | hazards: nonsolid %lu sq %lu dead-target %lu | dup lanes %lu dead lanes %lu big levels %lu\n",
           st.levels, st.plans, st.passes, st.vevents, st.pops_fast, st.pops_serial, st.hz_nonsolid, st.hz_sq, st.hz_dead_target, st.dup_lanes, st.dead_lanes, st.big_levels);
    printf("sift moves/pop %.2f  pushes %lu climbs %lu\n", (double)st.sift_moves / (st.pops_fast ? st.pops_fast : 1), st.pushes, st.push_climbs);
    printf("heap commit: passes %lu sifts %lu dirty(static) %lu (%.1f%%) zone-touching sifts %lu (%.2f%%) passes with a touch %lu (%.1f%%) depth rounds/pass %.2f underflow passes %lu climb passes %lu double climbs %lu\n",
           st.st_passes, st.st_sifts, st.st_dirty, 100.0 * st.st_dirty / st.st_sifts, st.st_touch_sifts, 100.0 * st.st_touch_sifts / st.st_sifts,
           st.st_touch_passes, 100.0 * st.st_touch_passes / st.st_passes, (double)st.st_depth_rounds / st.st_passes, st.st_underflow_passes, st.st_climb_passes, st.st_dbl_climb);
    return bad;
}
