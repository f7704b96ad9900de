// lama_brushfire_lse.h -- the LOWER wave of DynamicDistanceMap::update() (src/sdm/dynamic_distance_map.cpp:175-194, lower()
// :281-330) processed one priority LEVEL at a time, bit-identical to the reference's sequential loop INCLUDING the order in
// which libstdc++'s binary heap pops cells of equal priority.  Included by lama_kernels.h; runs on the main wave of
// k_brushfire after the raise wave.
//
// Why a level can be processed at once (each statement was checked op by op against the reference's queue trace,
// tools/research/lse_check.cpp, before this kernel was written):
//  * the lower wave pops non-decreasing priorities, and a cell that fires at level d only pushes priorities > d;
//  * while d is the minimum every entry of priority d has only priority-d ancestors, so the level's entries ("members") are a
//    connected top part T of the heap array;
//  * std::__adjust_heap steps to the right child unless comp(right, left): among equal priorities it prefers the right child.
//    The members therefore pop in the RIGHT-FIRST PREORDER of T, and the array slot that is vacated at the i-th pop -- where the
//    re-inserted last element v is sifted down from -- is the i-th slot of T's right-first POSTORDER:
//        t(q) = depth(q) + sum of |subtree(right sibling)| over the left-child steps of the path root -> q
//        vacated at pop  t(q) + |subtree(q)| - 1 - depth(q)
//    Both follow from the member positions alone (subtree sizes = member counts of the contiguous slot range of every depth);
//  * member slots never move physically: their content is only ever compared as "priority == d", so a pop is just
//    "sift v down from the vacated slot" (2 moves on average instead of a walk from the root);
//  * the exception is a pop whose re-inserted last element v is itself a member (the array's tail has priority d, a
//    "v-event"): that entry re-enters at the end of the right spine of T.  The pass stops there, the list of entries still to
//    pop is edited (one element moves) and the rest of the level is planned again;
//  * what the pops of one level do to the map commutes except for offers to the same neighbour cell; those are resolved per
//    target in pop order (at most four offers, found through an LDS hash of the firing cells).  Configurations where the order
//    could matter in another way (a firing cell whose obstacle is not a live obstacle, a stale queue entry that an offer of the
//    same level lands on, ...) are detected and the rest of the level is replayed one pop at a time by a plain serial loop.
#pragma once

namespace lama_dev {

constexpr uint32_t LSE_HEMPTY = 0xFFFFFFFFu;
constexpr int LSE_HSIZE = 256;

template <int LQ>
struct LseLds {
    uint64_t lst[2][LQ];           // entries of the level still to pop, in pop order (two buffers: a v-event edits the list)
    uint16_t C[LQ + 64];           // C[pos] = members with slot < pos
    uint16_t sByPos[LQ + 64];      // |subtree(q)| of a member slot, 0 elsewhere
    uint16_t tByPos[LQ];           // pop rank of the member slot
    uint16_t remByPos[LQ];         // rank of the pop that vacates the member slot
    uint16_t holepos[LQ];          // inverse of remByPos
    uint16_t mpos[LQ];             // member slots, ascending
    uint32_t hkey[LSE_HSIZE];      // firing cells of the pass: (ry << 13 | rx) << 6 | lane, open addressing
    uint64_t pushlist[64 * 4];     // entries pushed by the pass, pop by pop, in neighbour order
    uint32_t laneobs[64];          // obstacle offset of each firing lane's cell
#ifdef LAMA_PROFILE_LSE
    uint64_t prof[8];
#endif
};

__device__ __forceinline__ void lse_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#ifdef LAMA_PROFILE_LSE            // developer build (tools/prof_lse.py): cycles per phase of the lower wave in prm.dbg[8 p + k]
__device__ __forceinline__ uint64_t lse_now() { uint64_t t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
// raw event log of particle 0: (event << 56) | timestamp, appended by lane 0 to prm.dbg + 16 P (1 MiB reserved by the host)
#define LSET(k) do { const uint64_t t_ = lse_now(); if (prof_p == 0 && lane == 0 && lognum < (1u << 17) - 16u) prm.dbg[16 * (size_t)prm.P + lognum] = ((uint64_t)(k) << 56) | (t_ & 0xFFFFFFFFFFFFFFull); ++lognum; } while (0)
#define LSEC(k, n) do {} while (0)
#else
#define LSET(k) do {} while (0)
#define LSEC(k, n) do {} while (0)
#endif
__device__ __forceinline__ int lse_dx(int a) { return a == 0 ? 1 : (a == 2 ? -1 : 0); }
__device__ __forceinline__ int lse_dy(int a) { return a == 1 ? 1 : (a == 3 ? -1 : 0); }
__device__ __forceinline__ uint32_t lse_hash(uint32_t key26) { return (key26 * 0x9E3779B1u) >> 24; }

// PLAN -- members of level d in H[0, nl): pop rank, vacating rank (see the header).  `fresh`: the array is consistent (start of
// a level) and the entries are taken from it; otherwise the caller carries the list of entries over.  Returns the member count.
template <int LQ>
__device__ __noinline__ uint32_t lse_plan(const uint64_t* H, uint32_t nl, uint32_t d, LseLds<LQ>& L, int lane, bool fresh, uint64_t* lst_dst)
{
    uint32_t running = 0;
    for (uint32_t pos0 = 0; pos0 <= nl; pos0 += 64) {
        const uint32_t pos = pos0 + (uint32_t)lane;
        const bool isM = pos < nl && heap_prio(H[pos < nl ? pos : 0]) == d;
        const unsigned long long mk = __ballot(isM);
        const uint32_t rank = (uint32_t)__popcll(mk & ((1ull << lane) - 1ull));
        L.C[pos] = (uint16_t)(running + rank);
        L.sByPos[pos] = 0;
        if (isM) L.mpos[running + rank] = (uint16_t)pos;
        running += (uint32_t)__popcll(mk);
    }
    const uint32_t m = running;
    lse_lds_sync();
    for (uint32_t idx = (uint32_t)lane; idx < m; idx += 64) {
        const uint32_t q = L.mpos[idx];
        // |subtree(q)| = members in the slot range of every depth below q (all reads independent: no early exit)
        uint32_t s = 0;
        #pragma unroll
        for (int j = 0; j < 14; ++j) {
            const uint32_t lo = ((q + 1u) << j) - 1u;
            uint32_t hi = ((q + 2u) << j) - 1u;
            hi = hi < nl ? hi : nl;
            const uint32_t lo_c = lo < nl ? lo : nl;
            s += (uint32_t)L.C[hi] - (uint32_t)L.C[lo_c];
        }
        L.sByPos[q] = (uint16_t)s;
    }
    lse_lds_sync();
    for (uint32_t idx = (uint32_t)lane; idx < m; idx += 64) {
        const uint32_t q = L.mpos[idx];
        const uint32_t dep = 31u - (uint32_t)__clz((int)(q + 1u));
        uint32_t t = dep;
        uint32_t c = q;
        #pragma unroll
        for (int j = 0; j < 14; ++j) {                                   // the path q -> root, all reads independent
            const uint32_t r = c + 1u;
            const bool use = c > 0 && (c & 1u) && r < nl;
            t += use ? (uint32_t)L.sByPos[use ? r : 0u] : 0u;
            c = c > 0 ? (c - 1u) >> 1 : 0u;
        }
        const uint32_t rem = t + L.sByPos[q] - 1u - dep;
        L.tByPos[q] = (uint16_t)t;
        L.remByPos[q] = (uint16_t)rem;
        L.holepos[rem] = (uint16_t)q;
        if (fresh) lst_dst[t] = H[q];
    }
    lse_lds_sync();
    return m;
}

// The lower wave.  One wave (all 64 lanes call); H = the lower queue's heap array in LDS with nl entries (a consistent libstdc++
// heap).  On return either nl == 0 or `spill` is set (the heap no longer fits the LDS window: the array is consistent again and
// the caller hands the particle to the next stage).
template <int LQ>
__device__ __forceinline__ void lse_lower(const DevParams& prm, uint64_t* H, uint32_t& nl, LseLds<LQ>& L, const DirCache& dc, int16_t* dir,
                          uint16_t* sv, uint32_t* obs, uint64_t* mask, int& count, uint64_t& processed, bool& spill, const int lane, const uint64_t anc, const int prof_p = 0)
{
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
#ifdef LAMA_PROFILE_LSE
    uint32_t lognum = 1;
#endif
    while (nl > 0 && !spill) {
        const uint32_t d = heap_prio(H[0]);
        LSET(6);
        int cur = 0;                       // list buffer in use
        bool fresh = true;
        uint32_t m = lse_plan<LQ>(H, nl, d, L, lane, true, L.lst[0]);
        LSET(0); LSEC(7, 1ull << 20);
        uint32_t I = 0;                    // pops done under the current plan
        bool level_done = false;
        while (!level_done) {
            if (I >= m) { level_done = true; break; }
            const uint32_t k = (m - I) < 64u ? (m - I) : 64u;
            const bool act = (uint32_t)lane < k;
            const uint32_t nl_pass = nl;
            // ------------------------------------------------------------------ CELLS (no stores to the map yet)
            const uint64_t e = act ? L.lst[cur][I + (uint32_t)lane] : 0ull;
            const int x = q_rx(e), y = q_ry(e);
            int slot_[5]; uint32_t ci_[5], pidx_[5]; bool inw_[5];
            uint16_t s_[5]; uint32_t o_[5];
            #pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int cx = x + (c < 4 ? lse_dx(c) : 0), cy = y + (c < 4 ? lse_dy(c) : 0);
                inw_[c] = act && (uint32_t)cx < prm.WC && (uint32_t)cy < prm.WC;
                pidx_[c] = ((uint32_t)cy >> 5) * prm.W + ((uint32_t)cx >> 5);
                ci_[c] = ((uint32_t)cx & 31u) | (((uint32_t)cy & 31u) << 5);
                slot_[c] = inw_[c] ? dc.lookup(pidx_[c]) : -1;
            }
            #pragma unroll
            for (int c = 0; c < 5; ++c) {
                s_[c] = 0; o_[c] = 0;
                if (slot_[c] >= 0) { s_[c] = sv[slot_[c] * 1024 + (int)ci_[c]]; o_[c] = obs[slot_[c] * 1024 + (int)ci_[c]]; }
            }
            // second round: the cell my own offset points to, and for every neighbour the cell ITS offset points to (tie test)
            const int cox = obs_x(o_[4]), coy = obs_y(o_[4]);
            uint16_t t_[5];
            #pragma unroll
            for (int c = 0; c < 5; ++c) {
                const int cx = x + (c < 4 ? lse_dx(c) : 0) + obs_x(o_[c]), cy = y + (c < 4 ? lse_dy(c) : 0) + obs_y(o_[c]);
                t_[c] = 0;
                const bool need = slot_[c] >= 0 && (s_[c] & SV_VALID) && (uint32_t)cx < prm.WC && (uint32_t)cy < prm.WC;
                const int ts = need ? dc.lookup(((uint32_t)cy >> 5) * prm.W + ((uint32_t)cx >> 5)) : -1;
                if (ts >= 0) t_[c] = sv[ts * 1024 + (int)(((uint32_t)cx & 31u) | (((uint32_t)cy & 31u) << 5))];
            }
            const uint16_t s0 = s_[4], os0 = t_[4];
            LSET(1); LSEC(7, 1);
            // update() :183-192 + lower() :283
            bool fired = act && slot_[4] >= 0 && (s0 & SV_VALID) && (os0 & SV_SQMASK) == 0 && (s0 & SV_QUEUED);
            // the parallel form needs: the obstacle is a live obstacle, the cell really is at level d
            bool hazard = fired && (!(os0 & SV_VALID) || (uint32_t)(s0 & SV_SQMASK) != d);
            // hash of the firing cells (lowest lane wins a duplicate: the later pop of the same cell finds is_queued off)
            const uint32_t key26 = ((uint32_t)y << 13) | (uint32_t)x;
            #pragma unroll
            for (int j = 0; j < LSE_HSIZE / 64; ++j) L.hkey[j * 64 + lane] = LSE_HEMPTY;
            lse_lds_sync();
            if (fired) {
                const uint32_t word = (key26 << 6) | (uint32_t)lane;
                uint32_t hs = lse_hash(key26);
                for (;;) {
                    const uint32_t old = atomicCAS(&L.hkey[hs], LSE_HEMPTY, word);
                    if (old == LSE_HEMPTY) break;
                    if ((old >> 6) == key26) { atomicMin(&L.hkey[hs], word); break; }
                    hs = (hs + 1u) & (LSE_HSIZE - 1);
                }
                L.laneobs[lane] = o_[4];
            }
            lse_lds_sync();
            auto find = [&](uint32_t k26) -> int {
                uint32_t hs = lse_hash(k26);
                for (;;) {
                    const uint32_t w = L.hkey[hs];
                    if (w == LSE_HEMPTY) return -1;
                    if ((w >> 6) == k26) return (int)(w & 63u);
                    hs = (hs + 1u) & (LSE_HSIZE - 1);
                }
            };
            bool dup = false;
            if (fired) { dup = find(key26) != lane; if (dup) fired = false; }
            // offers: per direction a the <= 4 offers to N = B + delta_a of this pass, applied in pop order
            uint32_t okm = 0;                       // my successful offers, bit a
            uint32_t awm = 0;                       // directions away from my obstacle
            uint32_t nsq_[4];
            // who else offers to my targets?  The cell that reaches N = B + delta_a by direction b is N - delta_b (b == a: me).
            // All 12 first probes of the hash are issued together, then the obstacle offsets of the lanes found.
            uint32_t pk_[4][4], pw_[4][4];
            #pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bool away = fired && !(lse_dx(a) * cox > 0 || lse_dy(a) * coy > 0);
                #pragma unroll
                for (int b = 0; b < 4; ++b) {
                    pk_[a][b] = LSE_HEMPTY; pw_[a][b] = LSE_HEMPTY;
                    if (b == a) continue;
                    const int mx = x + lse_dx(a) - lse_dx(b), my = y + lse_dy(a) - lse_dy(b);
                    if (away && inw_[a] && (uint32_t)mx < prm.WC && (uint32_t)my < prm.WC) {
                        pk_[a][b] = ((uint32_t)my << 13) | (uint32_t)mx;
                        pw_[a][b] = L.hkey[lse_hash(pk_[a][b])];
                    }
                }
            }
            int pj_[4][4]; uint32_t po_[4][4];
            #pragma unroll
            for (int a = 0; a < 4; ++a) {
                #pragma unroll
                for (int b = 0; b < 4; ++b) {
                    pj_[a][b] = -1; po_[a][b] = 0;
                    if (b == a) continue;
                    const uint32_t w = pw_[a][b];
                    if (w != LSE_HEMPTY) {
                        if ((w >> 6) == pk_[a][b]) pj_[a][b] = (int)(w & 63u);
                        else pj_[a][b] = find(pk_[a][b]);                       // collision: walk the probe sequence
                    }
                    if (pj_[a][b] >= 0) po_[a][b] = L.laneobs[pj_[a][b]];
                }
            }
            // Closed form of "apply the <= 4 offers to N in pop order" (dynamic_distance_map.cpp:303-326): mine succeeds iff it
            // would succeed on the state N has at the start of the pass and no EARLIER pop offers a candidate <= mine (an earlier
            // offer either took the cell with a smaller-or-equal distance or failed against an even smaller one; an equal candidate
            // now ties with a live obstacle); it owns N's final state iff moreover no LATER pop up to the cut offers a candidate
            // < mine (successful offers to one cell have strictly decreasing candidates).
            uint32_t later_[4];                     // lowest later pop with a smaller candidate, 255 = none
            #pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int dx = lse_dx(a), dy = lse_dy(a);
                const bool away = fired && !(dx * cox > 0 || dy * coy > 0);
                const int qx = dx - cox, qy = dy - coy;
                nsq_[a] = (uint32_t)(qx * qx + qy * qy);
                const bool valid0 = (s_[a] & SV_VALID) != 0;
                const uint32_t sq0 = (uint32_t)(s_[a] & SV_SQMASK);
                const bool solid0 = (t_[a] & SV_VALID) && (t_[a] & SV_SQMASK) == 0;
                const uint32_t cmp0 = valid0 ? sq0 : prm.max_sqdist;
                bool ok = away && inw_[a] && (nsq_[a] < cmp0 || (nsq_[a] == sq0 && (!valid0 || !solid0)));      // :308-317
                uint32_t later = 255u;
                #pragma unroll
                for (int b = 0; b < 4; ++b) {
                    if (b == a) continue;
                    if (pj_[a][b] >= 0) {
                        const int jcx = obs_x(po_[a][b]), jcy = obs_y(po_[a][b]);
                        if (!(lse_dx(b) * jcx > 0 || lse_dy(b) * jcy > 0)) {                 // that cell offers to N too
                            const int rx_ = lse_dx(b) - jcx, ry_ = lse_dy(b) - jcy;
                            const uint32_t nq = (uint32_t)(rx_ * rx_ + ry_ * ry_);
                            const uint32_t rj = (uint32_t)pj_[a][b];
                            if (rj < (uint32_t)lane && nq <= nsq_[a]) ok = false;
                            if (rj > (uint32_t)lane && nq < nsq_[a] && rj < later) later = rj;
                        }
                    }
                }
                later_[a] = later;
                if (ok) okm |= 1u << a;
                if (away) awm |= 1u << a;
                if (away && !inw_[a]) atomicOr(prm.err, ERR_WINDOW);
            }
            // A successful offer of an EARLIER pop that lands on the cell of a pop that did not fire (a stale queue entry) may
            // change what that pop does: the pass is cut right before it and the rest is evaluated again on the updated map.
            uint32_t dcut = 64u;
            {
                unsigned long long deadm = __ballot(act && !fired && !dup);
                while (deadm) {
                    const int jd = __ffsll((long long)deadm) - 1;
                    deadm &= deadm - 1ull;
                    const int dxq = __builtin_amdgcn_readlane(x, jd), dyq = __builtin_amdgcn_readlane(y, jd);
                    bool hit = false;
                    #pragma unroll
                    for (int a = 0; a < 4; ++a) if (lane < jd && ((okm >> a) & 1u) && x + lse_dx(a) == dxq && y + lse_dy(a) == dyq) hit = true;
                    if (__ballot(hit)) { dcut = (uint32_t)jd; break; }
                }
            }
            const uint32_t cnt = (uint32_t)__popc(okm);
            const uint32_t pbase = (uint32_t)__popcll(__ballot(cnt & 1u) & lt_mask) + 2u * (uint32_t)__popcll(__ballot(cnt & 2u) & lt_mask) +
                                   4u * (uint32_t)__popcll(__ballot(cnt & 4u) & lt_mask);
            // v-event: the first pop whose re-inserted last element is a member
            const uint32_t n_i = nl_pass - (uint32_t)lane + pbase;
            bool vev = false;
            if (act) {
                const uint32_t pos = n_i - 1u;
                if (pos < nl_pass) vev = heap_prio(H[pos]) == d && (uint32_t)L.remByPos[pos] >= I + (uint32_t)lane;
            }
            const unsigned long long vm = __ballot(vev);
            uint32_t last = vm ? (uint32_t)(__ffsll((long long)vm) - 1) : k - 1u;
            bool vevent = vm != 0ull;
            // a firing pop the parallel form must not mix with others (its obstacle is not a live obstacle, or the cell is not at
            // level d: a stale entry of a cell that was re-entered) gets a pass of its own -- one pop at a time IS the reference
            const unsigned long long hzm = __ballot(hazard);
            if (hzm) {
                const uint32_t jh = (uint32_t)(__ffsll((long long)hzm) - 1);
                if (jh == 0u) { if (last > 0u) { last = 0u; vevent = false; } }
                else if (jh < dcut) dcut = jh;
            }
            if (dcut <= last) { last = dcut - 1u; vevent = false; }        // dcut >= 1: an earlier lane made the offer
            LSET(2);
            const uint32_t total_push = (uint32_t)__builtin_amdgcn_readlane((int)(pbase + cnt), (int)last);
            const bool overflow = nl_pass + total_push + 4u > (uint32_t)LQ;
            if (overflow) {
                // the heap would outgrow its LDS window: make the array consistent again (logical entry of member slot q =
                // list[pop rank of q among the remaining members]) and hand the particle to the next stage
                if (I > 0) {
                    m = lse_plan<LQ>(H, nl, d, L, lane, false, nullptr);
                    for (uint32_t idx = (uint32_t)lane; idx < m; idx += 64) { const uint32_t q = L.mpos[idx]; H[q] = L.lst[cur][I + (uint32_t)L.tByPos[q]]; }
                } else {
                    for (uint32_t idx = (uint32_t)lane; idx < m; idx += 64) { const uint32_t q = L.mpos[idx]; H[q] = L.lst[cur][(uint32_t)L.tByPos[q]]; }
                }
                lse_lds_sync();
                spill = true;
                return;
            }
            // ------------------------------------------------------------------ COMMIT cells, lanes 0 .. last
            const bool mine = act && (uint32_t)lane <= last;
            processed += (uint64_t)(last + 1u);
            #pragma unroll
            for (int a = 0; a < 4; ++a) {
                const bool touch = mine && ((awm >> a) & 1u) && inw_[a];                  // get(): allocation + mask bit
                const bool freshp = touch && slot_[a] < 0;
                if (__ballot(freshp)) { const int ns_ = coop_slot(dc, dir, count, (int)prm.dm_cap, freshp, pidx_[a], ERR_DM_CAP, prm.err); if (freshp) slot_[a] = ns_; }
                if (touch && slot_[a] >= 0) {
                    if (freshp || !(s_[a] & (SV_VALID | SV_QUEUED)))
                        atomicOr((unsigned long long*)(mask + (size_t)slot_[a] * 16 + (ci_[a] >> 6)), 1ull << (ci_[a] & 63));
                    if ((okm >> a) & 1u) {
                        if (!(later_[a] <= last)) {             // the last successful offer of a pop <= last owns the cell's state
                            sv[slot_[a] * 1024 + (int)ci_[a]] = (uint16_t)(SV_VALID | SV_QUEUED | (nsq_[a] & SV_SQMASK));
                            obs[slot_[a] * 1024 + (int)ci_[a]] = pack_obs(cox - lse_dx(a), coy - lse_dy(a));
                        }
                        L.pushlist[pbase + (uint32_t)__popc(okm & ((1u << a) - 1u))] =
                            q_entry(nsq_[a], x + lse_dx(a), y + lse_dy(a), cox - lse_dx(a), coy - lse_dy(a));
                    }
                }
            }
            if (mine && fired) sv[slot_[4] * 1024 + (int)ci_[4]] = (uint16_t)(s0 & ~SV_QUEUED);      // :329
            lse_lds_sync();
            LSET(3);
            // ------------------------------------------------------------------ COMMIT heap, pops 0 .. last (serial, exact)
            uint32_t qv_slot = 0;
            const uint32_t hp_lane = act ? (uint32_t)L.holepos[I + (uint32_t)lane] : 0u;       // the slot my pop vacates
            bool tail_known = false;                       // the array's last entry is in registers (the last entry just pushed)
            uint64_t tail_val = 0;
            #pragma unroll 1
            for (uint32_t i = 0; i <= last; ++i) {
                const uint32_t ci_cnt = (uint32_t)__builtin_amdgcn_readlane((int)cnt, (int)i);
                const uint32_t ci_base = (uint32_t)__builtin_amdgcn_readlane((int)pbase, (int)i);
                if (vevent && i == last) {
                    --nl;                                   // the member at the tail leaves its slot (it re-enters below the right spine)
                    qv_slot = nl;
                } else {
                    const uint64_t v = tail_known ? tail_val : H[nl - 1u];
                    --nl;
                    if (nl > 0) {
                        (void)lds_sift_topdown(H, nl, (uint32_t)__builtin_amdgcn_readlane((int)hp_lane, (int)i), v, lane, anc);
                    }
                }
                tail_known = false;
                if (ci_cnt) {
                    const uint64_t lastent = L.pushlist[ci_base + ci_cnt - 1u];
                    tail_known = lds_push_list(H, nl, L.pushlist + ci_base, ci_cnt, lane);
                    tail_val = lastent;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      // this pass's map stores before the next pass's loads
            lse_lds_sync();
            LSET(4);
            if (vevent) {
                // e_v moves from its place in the list to right behind the entries of the (new) right spine
                const uint32_t r0 = I + last + 1u;                      // first entry still to pop
                const uint32_t mR = m - r0;
                const uint32_t jv = (uint32_t)L.tByPos[qv_slot] - r0;
                const uint32_t m2 = lse_plan<LQ>(H, nl, d, L, lane, false, nullptr);
                (void)m2;
                if (mR > 0) {
                    const uint32_t hp0 = L.holepos[0];
                    const uint32_t kdep = 31u - (uint32_t)__clz((int)(hp0 + 1u));
                    const uint64_t* R = L.lst[cur] + r0;
                    uint64_t* NL = L.lst[cur ^ 1];
                    const uint64_t ev = R[jv];
                    for (uint32_t xq = (uint32_t)lane; xq < mR; xq += 64) {
                        uint64_t val;
                        if (xq == kdep) val = ev;
                        else { const uint32_t w = xq - (xq > kdep ? 1u : 0u); val = R[w + (w >= jv ? 1u : 0u)]; }
                        NL[xq] = val;
                    }
                    lse_lds_sync();
                }
                cur ^= 1; m = mR; I = 0; fresh = false;
                LSET(5); LSEC(7, 1ull << 20);
            } else {
                I += last + 1u;
            }
        }
        (void)fresh;
    }
#ifdef LAMA_PROFILE_LSE
    if (prof_p == 0 && lane == 0) prm.dbg[16 * (size_t)prm.P] = lognum;
#endif
}

} // namespace lama_dev
