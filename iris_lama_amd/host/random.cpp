// random.cpp -- lama::random (src/random.cpp:34-73): process-wide generator, a fresh distribution per draw.
#include "lama/random.h"

#include <random>

namespace lama {
namespace random {

static std::random_device& device() { static std::random_device rd; return rd; }
static std::mt19937& generator() { static std::mt19937 gen(device()()); return gen; }

uint32_t genSeed() { return device()(); }
void setSeed(uint32_t seed) { generator().seed(seed); }
double uniform() { return std::uniform_real_distribution<double>(0.0, 1.0)(generator()); }
double uniform(double low, double high) { return std::uniform_real_distribution<double>(low, high)(generator()); }
int32_t uniform(int32_t from, int32_t to) { return std::uniform_int_distribution<int32_t>(from, to)(generator()); }
double normal(double stddev) { return std::normal_distribution<double>(0.0, stddev)(generator()); }

} // namespace random
} // namespace lama
