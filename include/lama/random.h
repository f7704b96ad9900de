// lama/random.h -- lama::random of the reference (include/lama/random.h, src/random.cpp:34-73): ONE process-wide
// std::mt19937 seeded from std::random_device unless setSeed() is called; every draw builds a fresh distribution.
// Used by Loc2D::globalLocalization (src/loc2d.cpp:263-270).  (PFSlam2D keeps a generator per instance instead:
// the reference re-seeds the global one in its constructor, so the streams are the same for one instance.)
#pragma once

#include <cstdint>

namespace lama {
namespace random {

uint32_t genSeed();
void setSeed(uint32_t seed);
double uniform();
double uniform(double low, double high);
int32_t uniform(int32_t from, int32_t to);
double normal(double stddev);

} // namespace random
} // namespace lama
