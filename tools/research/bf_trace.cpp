// bf_trace.cpp -- research harness (not product, not a test): runs the CPU oracle's PFSlam2D on the corridor log and dumps the
// priority-queue operation trace of every DynamicDistanceMap::update() (oracle/lama_oracle.hpp bf_trace hook).
//   g++ -O2 -std=c++14 -I oracle -I include tools/research/bf_trace.cpp iris_lama_amd/host/corridor.cpp -o /tmp/bf_trace
//   /tmp/bf_trace <particles> <scans> <out.bin>
#include <cstdio>
#include <cstdlib>
#include "lama_oracle.hpp"
extern "C" int lama_corridor_generate(int steps, int beams, double* pts, double* odom_xyr, double* truth_xyr);
using namespace orc;
int main(int argc, char** argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 30, S = argc > 2 ? atoi(argv[2]) : 36;
    const char* out = argc > 3 ? argv[3] : "/tmp/bf_trace.bin";
    const double gain = argc > 4 ? atof(argv[4]) : 3.0;
    const int beams = 1080;
    std::vector<double> pts((size_t)(S + 1) * beams * 3), odom(3 * (S + 1)), truth(3 * (S + 1));
    lama_corridor_generate(S, beams, pts.data(), odom.data(), truth.data());
    PFOptions o; o.particles = P; o.seed = 42; o.threads = -1; o.meas_sigma_gain = gain;
    PFSlam2D pf(o);
    pf.setPrior(se2_from_xyr(truth[0], truth[1], truth[2]));
    std::vector<BfTraceRec> trace;
    bf_trace() = &trace;
    FILE* f = fopen(out, "wb");
    for (int k = 0; k <= S; ++k) {
        Scan s; s.points.resize(beams);
        for (int i = 0; i < beams; ++i) s.points[i] = V3d{pts[((size_t)k * beams + i) * 3], pts[((size_t)k * beams + i) * 3 + 1], 0.0};
        trace.clear();
        pf.update(s, se2_from_xyr(odom[3 * k], odom[3 * k + 1], odom[3 * k + 2]), 0.1 * k);
        BfTraceRec hdr{100u, (uint32_t)k, (uint32_t)trace.size(), 0};
        fwrite(&hdr, sizeof hdr, 1, f);
        fwrite(trace.data(), sizeof(BfTraceRec), trace.size(), f);
        fprintf(stderr, "scan %d: %zu records\n", k, trace.size());
    }
    fclose(f);
    return 0;
}
