// heap_shim.cpp -- CPU shim around the product's device heap (iris_lama_amd/csrc/lama_heap.h) so the test
// can replay identical push/pop sequences through it and through std::priority_queue with the reference's
// comparator (include/lama/sdm/dynamic_distance_map.h:90-98).
#include <cstdint>
#include <queue>
#include <utility>
#include <vector>

#include "../iris_lama_amd/csrc/lama_heap.h"

namespace {
struct VecStore {
    std::vector<uint64_t> v;
    uint64_t get(uint32_t i) const { return v[i]; }
    void set(uint32_t i, uint64_t x) { if (i >= v.size()) v.resize(i + 1); v[i] = x; }
};
struct V3 { uint32_t x, y, z; };
typedef std::pair<int, V3> qp;
struct cmp { bool operator()(const qp& l, const qp& r) const { return l.first > r.first; } };
}

// ops[i] >= 0: push priority ops[i] with payload i ; ops[i] < 0: pop.  Writes the payloads popped by each
// implementation; returns the number of pops.
extern "C" int heap_replay(const int32_t* ops, int n, uint32_t* out_dev, uint32_t* out_std)
{
    VecStore st;
    uint32_t size = 0;
    std::priority_queue<qp, std::vector<qp>, cmp> pq;
    int k = 0;
    for (int i = 0; i < n; ++i) {
        if (ops[i] >= 0) {
            lama_dev::heap_push(st, size, ((uint64_t)(uint32_t)ops[i] << 48) | (uint32_t)i);
            pq.push({ops[i], V3{(uint32_t)i, 0, 0}});
        } else if (size > 0) {
            out_dev[k] = (uint32_t)(lama_dev::heap_pop(st, size) & 0xFFFFFFFFu);
            out_std[k] = pq.top().second.x;
            pq.pop();
            ++k;
        }
    }
    return k;
}
