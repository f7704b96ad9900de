// lse_proto.hpp -- research prototype (CPU, not product, not a test oracle): a LEVEL-SYNCHRONOUS restatement of the lower wave
// of DynamicDistanceMap::update() (src/sdm/dynamic_distance_map.cpp:175-194, 281-330) that is meant to reproduce the
// reference's result INCLUDING the pop order of equal-priority cells out of libstdc++'s binary heap -- the plan the HIP kernel
// k_brushfire's "LSE" lower phase follows.  The phases and array names mirror the kernel.
//
// Facts used (checked on traces by tools/research/bf_heapstats.cpp):
//  * priorities popped by the lower wave never decrease, and a cell that fires at level d only pushes priorities > d;
//  * while d is the minimum, every heap entry of priority d has only priority-d ancestors: the entries of level d are a
//    connected top part T of the heap array ("members");
//  * __adjust_heap prefers the right child among equal priorities, so the level's entries pop in the RIGHT-FIRST PREORDER of
//    T, and the array slot that is vacated (and refilled by sifting the re-inserted last element v down from there) at the
//    i-th pop is the i-th slot of T's right-first POSTORDER -- unless the re-inserted last element is itself a member
//    (a "v-event"), which re-inserts that entry just below the current right spine;
//  * member slots never have to be moved physically: their content is never compared with anything but "priority == d".
#pragma once
#include <algorithm>
#include <cassert>
#include <cstdint>
#include <cstdio>
#include <unordered_map>
#include <vector>
#include "lama_oracle.hpp"

namespace orc {

struct LseStats {
    uint64_t levels = 0, plans = 0, passes = 0, vevents = 0, pops_fast = 0, pops_serial = 0;
    uint64_t hz_nonsolid = 0, hz_sq = 0, hz_dead_target = 0, dup_lanes = 0, dead_lanes = 0, big_levels = 0;
    uint64_t sift_moves = 0, push_climbs = 0, pushes = 0;
    uint64_t st_passes = 0, st_dirty = 0, st_touch_sifts = 0, st_touch_passes = 0, st_sifts = 0, st_depth_rounds = 0, st_underflow_passes = 0, st_climb_passes = 0, st_dbl_climb = 0;
};
inline LseStats& lse_stats() { static LseStats s; return s; }

struct BfResearchAccess {
    typedef std::pair<int, V3u> QP;
    typedef std::priority_queue<QP, std::vector<QP>, DynamicDistanceMap::compare_prio> queue_t;
    struct Q : queue_t { static std::vector<QP>& vec(queue_t& q) { return q.*(&Q::c); } };
    static std::vector<QP>& lower(DynamicDistanceMap& dm) { return Q::vec(dm.lower_); }
    static queue_t& raise_q(DynamicDistanceMap& dm) { return dm.raise_; }
    static queue_t& lower_q(DynamicDistanceMap& dm) { return dm.lower_; }
    static uint32_t max_sq(DynamicDistanceMap& dm) { return dm.max_sqdist_; }

    // the reference's raise wave, verbatim (update() :162-173)
    static uint32_t raise_phase(DynamicDistanceMap& dm)
    {
        uint32_t processed = 0;
        while (!dm.raise_.empty()) {
            bf_trace_add(3, (uint32_t)dm.raise_.top().first, dm.raise_.top().second);
            V3u location = dm.raise_.top().second; dm.raise_.pop();
            distance_t* current = (distance_t*)dm.get(location);
            ++processed; ++dm.stats.raise_pops;
            dm.raise(location, current);
        }
        return processed;
    }
    // one iteration of the reference's lower loop (update() :175-194)
    static void lower_one(DynamicDistanceMap& dm)
    {
        const uint32_t tprio = (uint32_t)dm.lower_.top().first;
        V3u location = dm.lower_.top().second; dm.lower_.pop();
        distance_t* current = (distance_t*)dm.get(location);
        ++dm.stats.lower_pops;
        const uint64_t f0 = dm.stats.lower_fired;
        if (bf_trace()) bf_trace()->push_back({4u, tprio, location.x, location.y});
        const size_t at = bf_trace() ? bf_trace()->size() - 1 : 0;
        if (current->valid_obstacle) {
            V3u obs = DynamicDistanceMap::offs(location, current->obstacle);
            const distance_t* obstacle = (distance_t*)dm.get(obs);
            current = (distance_t*)dm.get(location);
            if (obstacle->sqdist == 0) dm.lower(location, current);
        }
        if (bf_trace() && dm.stats.lower_fired != f0) (*bf_trace())[at].op = 5u;
    }
};

namespace lse {

typedef BfResearchAccess::QP QP;
static const distance_t ZERO_CELL = {{0, 0, 0}, 0, false, false};

// side-effect free read of a cell (no allocation, no mask bit; absent patch == calloc'd zeros)
inline const distance_t& peek(const DynamicDistanceMap& dm, uint32_t x, uint32_t y)
{
    const V3u c{x, y, 0};
    auto it = dm.patches.find(dm.m2p(c));
    if (it == dm.patches.end()) return ZERO_CELL;
    return *(const distance_t*)(it->second->data.data() + size_t(dm.m2c(c)) * sizeof(distance_t));
}
inline bool solid(const distance_t& o) { return o.valid_obstacle && o.sqdist == 0; }

static const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
constexpr int WAVE = 64;

struct Plan {
    uint32_t nl = 0, m = 0;
    std::vector<uint32_t> mpos;                 // member slots, ascending
    std::vector<uint16_t> C;                    // C[pos] = members with slot < pos; C[nl] = m
    std::vector<uint16_t> sByPos, tByPos, remByPos;
    std::vector<uint32_t> holepos;              // holepos[i] = slot vacated at the plan's i-th pop
};

inline int depth_of(uint32_t q) { return 31 - __builtin_clz(q + 1); }

// PLAN: members of level d in H[0, nl), right-first preorder rank t, subtree size s, vacating rank rem = t + s - 1 - depth
inline void make_plan(const std::vector<QP>& H, int d, Plan& P)
{
    const uint32_t nl = (uint32_t)H.size();
    P.nl = nl;
    P.mpos.clear();
    P.C.assign(nl + 1, 0);
    for (uint32_t pos = 0; pos < nl; ++pos) {        // kernel: 64 slots per step, ballot + popcount
        P.C[pos] = (uint16_t)P.mpos.size();
        if (H[pos].first == d) P.mpos.push_back(pos);
    }
    P.m = (uint32_t)P.mpos.size();
    P.C[nl] = (uint16_t)P.m;
    P.sByPos.assign(nl, 0); P.tByPos.assign(nl, 0); P.remByPos.assign(nl, 0xFFFF);
    P.holepos.assign(P.m, 0xFFFFFFFFu);
    for (uint32_t q : P.mpos) {                       // kernel: one member per lane
        uint32_t s = 0;
        for (int j = 0;; ++j) {
            const uint64_t lo = ((uint64_t)(q + 1) << j) - 1;
            if (lo >= nl) break;
            const uint64_t hi = std::min<uint64_t>(((uint64_t)(q + 2) << j) - 1, nl);
            s += (uint32_t)(P.C[hi] - P.C[lo]);
        }
        P.sByPos[q] = (uint16_t)s;
    }
    for (uint32_t q : P.mpos) {
        const int dep = depth_of(q);
        uint32_t t = (uint32_t)dep;
        for (uint32_t c = q; c > 0; c = (c - 1) >> 1)
            if (c & 1u) { const uint32_t r = c + 1; if (r < nl) t += P.sByPos[r]; }     // left child: the right sibling's subtree pops first
        const uint32_t rem = t + P.sByPos[q] - 1 - (uint32_t)dep;
        P.tByPos[q] = (uint16_t)t; P.remByPos[q] = (uint16_t)rem;
        assert(rem < P.m && P.holepos[rem] == 0xFFFFFFFFu);
        P.holepos[rem] = q;
    }
}

struct Lane {
    uint32_t x = 0, y = 0;        // B
    distance_t cur;               // state of B at the start of the pass
    bool fired = false, dup = false;
    int cox = 0, coy = 0;         // obstacle offset of B
    // per direction
    bool away[4] = {false, false, false, false};
    uint32_t nsq[4] = {0, 0, 0, 0};
    bool ok[4] = {false, false, false, false};        // my offer succeeds
    int later_ok[4] = {-1, -1, -1, -1};               // rank of the last successful offer to the same target in this pass
    // all successful ranks for the target, to find the last one <= cutoff
    int succ_rank[4][4]; int nsucc[4] = {0, 0, 0, 0};
    int cnt = 0;
};

// One level of the lower wave.  Returns the number of pops.
inline uint32_t level(DynamicDistanceMap& dm, std::vector<QP>& H, int d)
{
    LseStats& S = lse_stats();
    ++S.levels;
    const uint32_t max_sq = BfResearchAccess::max_sq(dm);
    uint32_t processed = 0;
    Plan P;
    std::vector<QP> Lst;       // remaining entries of the level in pop order
    bool fresh = true;

    auto serial_finish = [&](uint32_t I) {
        // materialise: logical entry of member slot q = Lst[preorder rank of q among the remaining members]
        if (I > 0) {
            std::vector<QP> rest(Lst.begin() + I, Lst.end());
            make_plan(H, d, P); ++S.plans;
            assert(P.m == rest.size());
            Lst.swap(rest);
        }
        for (uint32_t q : P.mpos) H[q] = Lst[P.tByPos[q]];
        while (!H.empty() && H[0].first == d) { BfResearchAccess::lower_one(dm); ++processed; ++S.pops_serial; }
    };

    for (;;) {
        make_plan(H, d, P); ++S.plans;
        if (P.m > 4 * WAVE) ++S.big_levels;
        if (fresh) { Lst.assign(P.m, QP()); for (uint32_t q : P.mpos) Lst[P.tByPos[q]] = H[q]; }
        assert(Lst.size() == P.m);
        uint32_t I = 0;
        bool replan = false;
        while (I < P.m) {
            const uint32_t k = std::min<uint32_t>(WAVE, P.m - I);
            ++S.passes;
            const uint32_t nl_pass = (uint32_t)H.size();
            // ---------------- CELLS (compute only) ----------------
            std::vector<Lane> L(k);
            std::unordered_map<uint64_t, int> table;          // kernel: LDS hash, cell -> lowest firing lane
            auto key = [](uint32_t x, uint32_t y) { return ((uint64_t)y << 32) | x; };
            bool hazard = false;
            int hz_first = -1;          // first firing lane the parallel form cannot mix with others: it gets a pass of its own
            std::vector<int> dead;
            for (uint32_t i = 0; i < k; ++i) {
                Lane& l = L[i];
                l.x = Lst[I + i].second.x; l.y = Lst[I + i].second.y;
                l.cur = peek(dm, l.x, l.y);
                l.cox = l.cur.obstacle[0]; l.coy = l.cur.obstacle[1];
                const distance_t& o = peek(dm, l.x + l.cox, l.y + l.coy);
                l.fired = l.cur.valid_obstacle && o.sqdist == 0 && l.cur.is_queued;
                if (l.fired) {
                    if (!solid(o) || l.cur.sqdist != (uint16_t)d) { if (hz_first < 0) hz_first = (int)i; }
                    auto ins = table.emplace(key(l.x, l.y), (int)i);
                    if (!ins.second) { l.dup = true; l.fired = false; ++S.dup_lanes; }     // same cell, later pop: is_queued is off by then
                } else { dead.push_back((int)i); ++S.dead_lanes; }
            }
            if (!hazard) {
                for (uint32_t i = 0; i < k; ++i) {
                    Lane& l = L[i];
                    if (!l.fired) continue;
                    for (int a = 0; a < 4; ++a) {
                        l.away[a] = !(DX[a] * l.cox > 0 || DY[a] * l.coy > 0);
                        if (!l.away[a]) continue;
                        const uint32_t nx = l.x + DX[a], ny = l.y + DY[a];
                        const int qx = (int)nx - (int)(l.x + l.cox), qy = (int)ny - (int)(l.y + l.coy);
                        l.nsq[a] = (uint32_t)(qx * qx + qy * qy);
                        // every offer to N = (nx, ny) of this pass, in rank order
                        struct Off { int rank; uint32_t nsq; };
                        Off offs[4]; int no = 0;
                        for (int b = 0; b < 4; ++b) {
                            const uint32_t mx = nx - DX[b], my = ny - DY[b];       // the cell that reaches N by direction b
                            auto it = table.find(key(mx, my));
                            if (it == table.end()) continue;
                            const Lane& j = L[it->second];
                            if (DX[b] * j.cox > 0 || DY[b] * j.coy > 0) continue;
                            const int rx = (int)nx - (int)(mx + j.cox), ry = (int)ny - (int)(my + j.coy);
                            offs[no++] = Off{it->second, (uint32_t)(rx * rx + ry * ry)};
                        }
                        // closed form of "apply the offers in pop order" (kernel): mine succeeds iff it would succeed on the
                        // pass-start state and no EARLIER pop offers a candidate <= mine; it owns the cell's final state iff in
                        // addition no LATER pop (up to the cut) offers a candidate < mine.
                        const distance_t& n0 = peek(dm, nx, ny);
                        const bool valid0 = n0.valid_obstacle; const uint32_t sq0 = n0.sqdist;
                        const bool solid0 = solid(peek(dm, nx + n0.obstacle[0], ny + n0.obstacle[1]));
                        const uint32_t cmp0 = valid0 ? sq0 : max_sq;
                        const uint32_t mine = l.nsq[a];
                        bool ok = mine < cmp0 || (mine == sq0 && (!valid0 || !solid0));
                        int later = 255;
                        for (int t = 0; t < no; ++t) {
                            if (offs[t].rank < (int)i && offs[t].nsq <= mine) ok = false;
                            if (offs[t].rank > (int)i && offs[t].nsq < mine) later = std::min(later, offs[t].rank);
                        }
                        l.ok[a] = ok;
                        l.later_ok[a] = later;
                        if (l.ok[a]) ++l.cnt;
                    }
                }
            }
            // P4: a successful offer of an EARLIER lane that lands on the cell of a lane that did not fire may change what that lane
            // does: cut the pass right before that lane (it is evaluated again on the updated map)
            uint32_t dcut = 64;
            if (!hazard) {
                for (int j : dead) {
                    bool hit = false;
                    for (uint32_t i = 0; i < (uint32_t)j && !hit; ++i)
                        for (int a = 0; a < 4; ++a)
                            if (L[i].ok[a] && L[i].x + DX[a] == L[j].x && L[i].y + DY[a] == L[j].y) hit = true;
                    if (hit) { dcut = (uint32_t)j; ++S.hz_dead_target; break; }
                }
            }
            if (hazard) { serial_finish(I); return processed; }
            // ---------------- v-event: the first pop whose re-inserted last element is a member ----------------
            uint32_t last = k - 1; bool vevent = false;
            {
                uint32_t pushed = 0;
                for (uint32_t i = 0; i < k; ++i) {
                    const uint32_t n_i = nl_pass - i + pushed;
                    const uint32_t pos = n_i - 1;
                    if (pos < nl_pass && H[pos].first == d && P.remByPos[pos] >= I + i) { vevent = true; last = i; break; }
                    pushed += (uint32_t)L[i].cnt;
                }
            }
            if (hz_first == 0) { if (last > 0) { last = 0; vevent = false; } if (hz_first == 0) ++S.hz_sq; }
            else if (hz_first > 0 && (uint32_t)hz_first < dcut) dcut = (uint32_t)hz_first;
            if (dcut <= last) { last = dcut - 1; vevent = false; }
            // ---------------- COMMIT cells, lanes 0 .. last ----------------
            for (uint32_t i = 0; i <= last; ++i) {
                Lane& l = L[i];
                if (bf_trace()) bf_trace()->push_back({l.fired ? 5u : 4u, (uint32_t)d, l.x, l.y});
                ++dm.stats.lower_pops; ++processed; ++S.pops_fast;
                if (l.dup) {
                    // the reference get()s the cell and its obstacle: both exist with their mask bits on -> nothing to do
                    continue;
                }
                if (!l.fired) continue;
                ++dm.stats.lower_fired;
                for (int a = 0; a < 4; ++a) {
                    if (!l.away[a]) continue;
                    const V3u nl_{l.x + DX[a], l.y + DY[a], 0};
                    distance_t* n = (distance_t*)dm.get(nl_);                    // allocation + mask bit (map.cpp:371-412)
                    if (!l.ok[a]) continue;
                    bf_trace_add(1, l.nsq[a], nl_);
                    ++dm.stats.pushes;
                    // the last successful offer with rank <= last owns the cell's final state
                    if (!(l.later_ok[a] <= (int)last)) {
                        n->sqdist = (uint16_t)l.nsq[a]; n->valid_obstacle = true; n->is_queued = true;
                        n->obstacle[0] = (int16_t)((int)(l.x + l.cox) - (int)nl_.x);
                        n->obstacle[1] = (int16_t)((int)(l.y + l.coy) - (int)nl_.y);
                        n->obstacle[2] = 0;
                    }
                }
                ((distance_t*)dm.get(V3u{l.x, l.y, 0}))->is_queued = false;
            }
            // ---------------- COMMIT heap, pops 0 .. last, serial and exact ----------------
            // statistics for a parallel form: the zone of slots the tail of the array can interact with during this pass
            uint32_t z_lo = (uint32_t)H.size(), z_hi = (uint32_t)H.size();
            {
                uint32_t sz = (uint32_t)H.size();
                for (uint32_t i = 0; i <= last; ++i) { sz -= 1; z_lo = std::min(z_lo, sz); sz += (uint32_t)L[i].cnt; z_hi = std::max(z_hi, sz); }
            }
            const uint32_t p_lo = z_lo ? (z_lo - 1) / 2 : 0, p_hi = z_hi ? (z_hi - 1) / 2 : 0;
            auto in_zone = [&](uint32_t pos) { return pos >= z_lo || (pos >= p_lo && pos <= p_hi); };
            auto is_anc_of_zone = [&](uint32_t h) {       // h is an ancestor-or-self of a slot in [p_lo, p_hi] or [z_lo, z_hi]
                for (int j = 0; j < 20; ++j) {
                    const uint64_t a = ((uint64_t)(h + 1) << j) - 1, b = ((uint64_t)(h + 2) << j) - 2;
                    if (a > z_hi) break;
                    if ((a <= p_hi && b >= p_lo) || (a <= z_hi && b >= z_lo)) return true;
                }
                return false;
            };
            ++S.st_passes;
            bool touch_pass = false, climb_pass = false;
            if (z_lo + 1 < nl_pass) ++S.st_underflow_passes;
            { uint32_t dm = 0; for (uint32_t i = 0; i <= last; ++i) if (!(vevent && i == last)) dm |= 1u << depth_of(P.holepos[I + i]); S.st_depth_rounds += (uint64_t)__builtin_popcount(dm); }
            // Deferred form (what the two-wave / lane-parallel kernel relies on): the sift of a pop whose vacated slot is NOT an
            // ancestor of any slot of the tail zone [z_lo, z_hi] ("clean") commutes with everything the tail does, so it may run
            // any time before the next "dirty" sift (one whose slot is an ancestor of the zone) or the end of the pass.
            auto is_dirty = [&](uint32_t h) {
                for (int j = 0; j < 20; ++j) {
                    const uint64_t a = ((uint64_t)(h + 1) << j) - 1, b = ((uint64_t)(h + 2) << j) - 2;
                    if (a > z_hi) break;
                    if (a <= z_hi && b >= z_lo) return true;
                }
                return false;
            };
            struct Pend { uint32_t hole; QP v; };
            std::vector<Pend> pend;
            auto do_sift = [&](uint32_t hole, const QP& v, uint32_t len) {
                bool touched = in_zone(hole);
                for (;;) {
                    const uint32_t c2 = 2 * hole + 2;
                    uint32_t c;
                    if (c2 < len) { c = (H[c2].first > H[c2 - 1].first) ? c2 - 1 : c2; if (in_zone(c2) || in_zone(c2 - 1)) touched = true; }
                    else if (c2 == len) { c = len - 1; touched = true; }
                    else break;
                    if (H[c].first > v.first) break;
                    H[hole] = H[c]; hole = c; ++S.sift_moves;
                }
                H[hole] = v;
                return touched;
            };
            auto flush = [&]() {
                // deepest slots first (a slot's member descendants are vacated before it; equal depths are disjoint subtrees)
                std::stable_sort(pend.begin(), pend.end(), [](const Pend& p, const Pend& q) { return depth_of(p.hole) > depth_of(q.hole); });
                for (const Pend& pe : pend) { const bool t = do_sift(pe.hole, pe.v, z_lo); assert(!t); (void)t; }
                pend.clear();
            };
            for (uint32_t i = 0; i <= last; ++i) {
                const uint32_t n = (uint32_t)H.size();
                if (vevent && i == last) {
                    H.pop_back();                              // the member at the tail leaves its slot; it re-enters below the right spine
                } else {
                    const QP v = H[n - 1]; H.pop_back();
                    const uint32_t len = n - 1;
                    if (len > 0) {
                        uint32_t hole = P.holepos[I + i];
                        assert(hole < len);
                        ++S.st_sifts;
                        if (is_dirty(hole)) {
                            ++S.st_dirty;
                            flush();
                            if (do_sift(hole, v, len)) { ++S.st_touch_sifts; touch_pass = true; }
                        } else pend.push_back(Pend{hole, v});
                    }
                }
                const Lane& l = L[i];
                for (int a = 0; a < 4; ++a) {
                    if (!l.ok[a]) continue;
                    const QP x{(int)l.nsq[a], V3u{l.x + DX[a], l.y + DY[a], 0}};
                    H.push_back(x);
                    uint32_t hole = (uint32_t)H.size() - 1;
                    int climbs = 0;
                    while (hole > 0) {
                        const uint32_t parent = (hole - 1) / 2;
                        if (!(H[parent].first > x.first)) break;
                        H[hole] = H[parent]; hole = parent; ++S.push_climbs; ++climbs;
                    }
                    if (climbs) climb_pass = true;
                    if (climbs > 1) ++S.st_dbl_climb;
                    H[hole] = x; ++S.pushes;
                }
            }
            flush();
            if (touch_pass) ++S.st_touch_passes;
            if (climb_pass) ++S.st_climb_passes;
            if (vevent) {
                ++S.vevents;
                // list edit: e_v moves from its place to the slot right behind the current right spine
                const uint32_t n_before = nl_pass;  (void)n_before;
                // which slot was the tail?  recompute as in the detection loop
                uint32_t pushed = 0; for (uint32_t i = 0; i < last; ++i) pushed += (uint32_t)L[i].cnt;
                const uint32_t qv = nl_pass - last + pushed - 1;
                const uint32_t jv = P.tByPos[qv] - (I + last + 1);      // index of e_v among the entries still to pop
                std::vector<QP> R(Lst.begin() + I + last + 1, Lst.end());
                const QP ev = R.empty() ? QP() : R[jv];
                Plan P2; make_plan(H, d, P2); ++S.plans;
                assert(P2.m == R.size());
                std::vector<QP> NL;
                if (P2.m > 0) {
                    // the entries still to pop, without e_v, keep their order; e_v now sits in the last slot of the right spine
                    // (depth kdep), i.e. pops after the kdep entries above it
                    const uint32_t kdep = (uint32_t)depth_of(P2.holepos[0]);
                    for (uint32_t x = 0; x < R.size(); ++x) if (x != jv) NL.push_back(R[x]);
                    assert(kdep <= NL.size());
                    NL.insert(NL.begin() + kdep, ev);
                }
                Lst.swap(NL);
                // the re-inserted entry is still part of the level: it sits (logically) in slot P2.holepos[0]
                fresh = false; replan = true;
                break;
            }
            I += last + 1;
        }
        if (!replan) break;
        if (H.empty() || H[0].first != d) break;
    }
    return processed;
}

inline uint32_t update(DynamicDistanceMap& dm)
{
    bf_trace_add(0, 0, V3u{0, 0, 0});
    uint32_t processed = BfResearchAccess::raise_phase(dm);
    std::vector<QP>& H = BfResearchAccess::lower(dm);
    while (!H.empty()) processed += level(dm, H, H[0].first);
    bf_trace_add(6, 0, V3u{0, 0, 0});
    return processed;
}

} // namespace lse
} // namespace orc
