// lama_heap.h -- binary min-heap whose pop order (INCLUDING ties) is that of
//   std::priority_queue<std::pair<int,Vector3ui>, std::vector<...>, compare_prio>
// as instantiated by the reference (include/lama/sdm/dynamic_distance_map.h:90-98) on libstdc++.
//
// The brushfire result depends on the order in which equal-priority cells are popped
// (hard part 3), so the device queue replays libstdc++'s heap algorithms
// step by step (bits/stl_heap.h: __push_heap, __adjust_heap, __pop_heap): push = append + sift-up
// with `comp(parent, value)`; pop = move last to a hole at the root, sift the hole DOWN to a leaf
// always taking the child that compares "larger" under comp (ties -> right child), then sift-up.
// compare_prio(l, r) = l.first > r.first, i.e. only the priority is compared, never the location.
//
// Entry encoding: bits 63..48 priority (squared distance), bits 47..32 payload that is NOT compared (the
// obstacle offset the cell carried when it was queued, see lama_kernels.h), bits 31..0 location
// (ry << 16 | rx, window-relative cell coordinates).  Storage is abstracted so the same code runs on an LDS-backed
// array with global-memory overflow on the device and on a plain array in the CPU unit test.
#pragma once
#include <stdint.h>

#ifndef LAMA_HD
#if defined(__HIPCC__)
#define LAMA_HD __host__ __device__ inline
#else
#define LAMA_HD inline
#endif
#endif

namespace lama_dev {

LAMA_HD uint32_t heap_prio(uint64_t e) { return (uint32_t)(e >> 32) >> 16; }      // (of the high word: a 32-bit shift and 32-bit compares on the device)
// compare_prio(left, right): left.first > right.first
LAMA_HD bool heap_comp(uint64_t l, uint64_t r) { return heap_prio(l) > heap_prio(r); }

// Storage concept: uint64_t get(uint32_t i) ; void set(uint32_t i, uint64_t v)
template <class Store>
LAMA_HD void heap_push_hole(Store& st, uint32_t holeIndex, uint32_t topIndex, uint64_t value)
{
    // std::__push_heap
    while (holeIndex > topIndex) {
        uint32_t parent = (holeIndex - 1) / 2;
        uint64_t pv = st.get(parent);
        if (!heap_comp(pv, value)) break;
        st.set(holeIndex, pv);
        holeIndex = parent;
    }
    st.set(holeIndex, value);
}

// priority_queue::push: c.push_back(v); std::push_heap(begin, end, comp)
template <class Store>
LAMA_HD void heap_push(Store& st, uint32_t& size, uint64_t value)
{
    uint32_t hole = size++;
    heap_push_hole(st, hole, 0u, value);
}

// priority_queue::top + pop: std::pop_heap(begin, end, comp); c.pop_back()
template <class Store>
LAMA_HD uint64_t heap_pop(Store& st, uint32_t& size)
{
    const uint64_t top = st.get(0);
    --size;
    if (size > 0) {
        // __pop_heap: value = *(last-1); __adjust_heap(first, 0, len = size, value)
        const uint64_t value = st.get(size);
        const uint32_t len = size;
        uint32_t holeIndex = 0;
        uint32_t secondChild = 0;
        while (secondChild < (len - 1) / 2) {
            secondChild = 2 * (secondChild + 1);
            uint64_t r = st.get(secondChild);
            uint64_t l = st.get(secondChild - 1);
            if (heap_comp(r, l)) { secondChild--; r = l; }
            st.set(holeIndex, r);
            holeIndex = secondChild;
        }
        if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
            secondChild = 2 * (secondChild + 1);
            st.set(holeIndex, st.get(secondChild - 1));
            holeIndex = secondChild - 1;
        }
        heap_push_hole(st, holeIndex, 0u, value);
    }
    return top;
}

// The same pop, computed top-down (scalar statement of lds_pop_topdown in lama_kernels.h, which the helper wave of
// k_brushfire runs 63 nodes at a time).  __adjust_heap moves every entry n_1 .. n_k of the hole's path up one slot,
// __push_heap then moves the entries with prio(n_i) > prio(value) down again, into the slots they came from.  Priorities
// do not decrease along the path, so the net effect is: the prefix of the path with prio(n_i) <= prio(value) moves up,
// `value` takes the slot of its last entry.  Same final array as heap_pop(); also returns the new root through `root`.
template <class Store>
LAMA_HD uint64_t heap_pop_topdown(Store& st, uint32_t& size, uint64_t* root)
{
    const uint64_t top = st.get(0);
    --size;
    if (size == 0) { if (root) *root = 0; return top; }
    const uint64_t value = st.get(size);
    const uint32_t len = size, lim = (len - 1) / 2;
    uint32_t hole = 0;
    bool stopped = false;
    while (hole < lim) {
        const uint32_t right = 2 * hole + 2;
        const uint64_t r = st.get(right), l = st.get(right - 1);
        const uint32_t child = heap_comp(r, l) ? right - 1 : right;
        const uint64_t cv = heap_comp(r, l) ? l : r;
        if (heap_prio(cv) > heap_prio(value)) { stopped = true; break; }
        st.set(hole, cv);
        hole = child;
    }
    if (!stopped && (len & 1) == 0 && hole == (len - 2) / 2) {      // lone left child at the end of the array
        const uint64_t cv = st.get(len - 1);
        if (!(heap_prio(cv) > heap_prio(value))) { st.set(hole, cv); hole = len - 1; }
    }
    st.set(hole, value);
    if (root) *root = st.get(0);
    return top;
}

} // namespace lama_dev
