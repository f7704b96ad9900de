// heap_shim.cpp -- CPU shim around the product's device heap (iris_lama_amd/csrc/lama_heap.h) so the test
// can replay identical push/pop sequences through it and through std::priority_queue with the reference's
// comparator (include/lama/sdm/dynamic_distance_map.h:90-98).
#include <algorithm>
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

// The top-down form of pop (heap_pop_topdown = scalar statement of the helper wave's lds_pop_topdown) must leave the
// SAME ARRAY as libstdc++'s std::pop_heap after every operation, not just pop in the same order: the array layout decides
// all later ties.  Returns the number of operations after which the two arrays were compared equal, or -(index + 1) of
// the first mismatch.
extern "C" int heap_replay_layout(const int32_t* ops, int n)
{
    VecStore st;
    uint32_t size = 0;
    std::vector<qp> ref;
    cmp c;
    for (int i = 0; i < n; ++i) {
        if (ops[i] >= 0) {
            lama_dev::heap_push(st, size, ((uint64_t)(uint32_t)ops[i] << 48) | (uint32_t)i);
            ref.push_back({ops[i], V3{(uint32_t)i, 0, 0}});
            std::push_heap(ref.begin(), ref.end(), c);
        } else if (size > 0) {
            uint64_t root = 0;
            const uint64_t top = lama_dev::heap_pop_topdown(st, size, &root);
            if ((uint32_t)top != ref.front().second.x) return -(i + 1);
            std::pop_heap(ref.begin(), ref.end(), c);
            ref.pop_back();
            if (size > 0 && (uint32_t)root != ref.front().second.x) return -(i + 1);
        }
        if (size != ref.size()) return -(i + 1);
        for (uint32_t k = 0; k < size; ++k)
            if ((uint32_t)st.v[k] != ref[k].second.x || (int)(st.v[k] >> 48) != ref[k].first) return -(i + 1);
    }
    return n;
}
