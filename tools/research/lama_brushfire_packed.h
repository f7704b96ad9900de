// RESEARCH ARCHIVE (round 3; not compiled into the product).  Bit-exact on the MI355X (all 80 GPU parity tests incl. the randomised
// rooms ran it, as cfg.brushfire_waves = 3) and under tests/sim -- and NOT faster, which is why it was removed again:
//
//   brushfire per scan, teacher-forced corridor scans 5-14 (2,840 pops per particle-scan), tools/bf_waves_sweep.py:
//     particles      one wave / particle   wave pair / particle (product)   two particles / wave pair (this file)
//        30               5.54 ms                  3.92 ms                          4.91 ms
//       300               6.31                     4.61                             5.88
//      1000               7.51                     5.81                             7.06
//      2000               8.67                     7.21                             7.90
//      3000               9.15                     8.01                             8.37
//   wave-instructions per launch at 3000 particles (rocprofv3 --pmc SQ_INSTS_*): wave pair 1.91 G VALU + 2.29 G SALU + 0.15 G LDS;
//   packed 1.56 G VALU + 1.31 G SALU + 0.14 G LDS.
//
// Why: packing halves the work that was already scalar (SALU, per wave) but turns every per-particle decision -- queue lengths,
// the next top, "fires", the sift state -- into per-HALF values that live in vector registers, so the VALU stream per iteration
// nearly doubles (365 VALU per two-particle iteration against 224 per pop): -18 % VALU per particle-pop, while the latency chain
// of an iteration grows by 25 % (ds_bpermute instead of readlane, three 5-level sift rounds instead of two 6-level ones, both
// halves' paths executed).  And even at 3000 particles the kernel is only partly issue bound (it takes 2x the 30-particle time for
// 100x the particles): the chain dominates.  See DESIGN.md 4c.
// lama_brushfire_packed.h -- the exact brushfire (DynamicDistanceMap::update / raise / lower, src/sdm/dynamic_distance_map.cpp:
// 160-197, 244-279, 281-330) with TWO particles per wave pair, for contexts with many particles.
//
// k_brushfire (lama_kernels.h) spends a wave pair on one particle and uses 6 of the main wave's 64 lanes; that is the right
// trade while waves are scarce (a pop is a latency chain), but with thousands of particles every SIMD holds several waves and
// the kernel is bound by the instructions it issues (measured at 3000 particles: 3.7 G wave-instructions, one issued every 3.8
// cycles per SIMD).  Here the two 32-lane halves of each wave work for two different particles with ONE instruction stream:
// lanes 0..5 of a half are the role lanes of that half's particle (neighbours +x +y -x -y, the popped cell, its obstacle cell),
// the helper wave's halves own the two particles' heaps (a 5-level subtree per sift round instead of 6).  Everything a half
// decides is a per-half value in vector registers (queue lengths, the next top, "fires", ...); control flow is wave-uniform
// ("does ANY half ...") with per-lane predicates, cross-lane traffic stays inside a half (ds_bpermute / quad DPP / half of a
// ballot).  The sequence of map and heap operations per particle is exactly that of k_brushfire, so the result is bit-identical
// (the GPU parity tests run both; tests/sim runs this source lane by lane against the oracle).
//
// Stage structure: this kernel is the first stage only (queues of at most LQ / RQ entries in LDS).  A particle whose queue does
// not fit, or outgrows it, is handed over with its state intact (prm.slow[p] = 1, queues written back) to the resume stage of
// k_brushfire and from there to k_brushfire_slow, exactly like the unpacked first stage does.
#pragma once

namespace lama_dev {

template <int LQ, int RQ>
struct BfLdsP {
    uint64_t lower[LQ];
    uint64_t raise[RQ];
    uint32_t dc[DC_SIZE];
    // mailboxes, double buffered by iteration parity
    uint64_t pl_e[2][4]; uint32_t pl_n[2];      // main -> helper: entries to push into the LOWER queue, in neighbour order
    uint64_t pr_e[2][4]; uint32_t pr_n[2];      // main -> helper: entries raise() pushes into the RAISE queue
    uint64_t topq[2];                           // helper -> main: the popped heap's root after pop() (before the pushes)
    uint64_t own_e[2]; uint32_t own_k[2];       // main -> main across the barrier: its best own push (entry, priority << 2 | neighbour)
};

__device__ __forceinline__ uint32_t pk_half(unsigned long long m, int hf) { return (uint32_t)(m >> (hf * 32)); }
// value of lane k of MY half
__device__ __forceinline__ uint32_t pk_bcast(uint32_t v, int lane, int k) { return (uint32_t)__shfl((int)v, (lane & 32) | k, 64); }
template <int CTRL>
__device__ __forceinline__ uint32_t pk_quad(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }

// ancestors of relative heap position hl (0 = the hole) inside a half's 5-level subtree, as a mask of relative positions
__device__ __forceinline__ uint32_t pk_ancestors(int hl)
{
    uint32_t anc = 0;
    if (hl < 31) for (int a = hl; a > 0; a = (a - 1) >> 1) anc |= 1u << a;
    return anc;
}

// pop() of a half's heap (the top-down form of std::__adjust_heap + std::__push_heap, see lds_sift_topdown in lama_kernels.h):
// position hl of a half is the node at relative position hl below the hole; the entry that ends up at the root is published in
// *topq_slot by whichever lane knows it.  `act`: this half pops in this iteration.
__device__ __forceinline__ void pk_pop(uint64_t* h, uint32_t& size, bool act, uint64_t* topq_slot, int hl, int hf, uint32_t anc)
{
    LAMA_LOCKSTEP();
    if (act) --size;
    const uint32_t len = size;
    const bool go0 = act && len > 0;
    const uint64_t value = go0 ? h[len] : 0ull;                       // the array's last entry is re-inserted from the root
    const uint32_t vprio = heap_prio(value);
    const uint32_t lim = go0 ? (len - 1) / 2 : 0u;                    // nodes below `lim` have both children
    const int d = 31 - __clz(hl + 1);
    const bool is_left = (hl & 1) != 0;
    uint32_t H = 0;
    bool stopped = false, root_moved = false;
    for (;;) {
        const bool go = go0 && !stopped && H < lim;
        if (__ballot(go) == 0ull) break;
        const uint32_t idx = (H << d) + (uint32_t)hl;
        const uint32_t left = is_left ? idx : idx - 1;
        const uint32_t parent = (left - 1) >> 1;
        const bool cand = go && hl >= 1 && hl < 31 && parent < lim;
        const uint32_t la = cand ? left : 1u;
        const uint64_t vl = h[la], vr = h[la + 1];
        const bool take_left = heap_prio(vr) > heap_prio(vl);         // __adjust_heap: right unless comp(right, left)
        const bool step_to_me = cand && (is_left == take_left);
        const uint32_t okm = pk_half(__ballot(step_to_me), hf);
        const bool onpath = cand && (okm & anc) == anc;
        const uint64_t mine = is_left ? vl : vr;
        const bool moves = onpath && !(heap_prio(mine) > vprio);
        const unsigned long long pb = __ballot(onpath), mb = __ballot(moves);
        const uint32_t pathm = pk_half(pb, hf), mvm = pk_half(mb, hf);
        if (moves) {
            h[parent] = mine;
            if (parent == 0) *topq_slot = mine;                       // only in the first round (H == 0): the new root
        }
        if (go && mvm) {
            if (H == 0) root_moved = true;
            const int rel = 31 - __clz((int)mvm);                     // deepest moved entry: its old slot is the new hole
            H = (H << (31 - __clz(rel + 1))) + (uint32_t)rel;
        }
        if (go && mvm != pathm) stopped = true;
    }
    LAMA_LOCKSTEP();
    const bool tail = go0 && !stopped && (len & 1) == 0 && H == (len - 2) / 2;    // the hole has a lone left child, the array's last entry
    const uint64_t c = tail ? h[len - 1] : 0ull;
    const bool up = tail && heap_prio(c) <= vprio;
    if (up) {
        if (hl == 0) { h[H] = c; if (H == 0) *topq_slot = c; }
        if (H == 0) root_moved = true;
        H = len - 1;
    }
    if (go0 && hl == 0) { h[H] = value; if (!root_moved) *topq_slot = value; }
    LAMA_LOCKSTEP();
}

// one std::push_heap on a half's heap; per-lane control flow, no cross-lane traffic (every lane of the half walks, hl == 0 stores)
__device__ __forceinline__ void pk_push1(uint64_t* h, uint32_t& size, uint64_t value, bool doit, bool writer)
{
    if (!doit) return;
    uint32_t hole = size++;
    while (hole > 0) {
        const uint32_t parent = (hole - 1) / 2;
        const uint64_t pv = h[parent];
        if (!heap_comp(pv, value)) break;
        if (writer) h[hole] = pv;
        hole = parent;
    }
    if (writer) h[hole] = value;
}

// push_heap of a half's `cnt` (<= 4) mailbox entries, in order.  Fast path: one gather of all would-be parents; if no new entry has
// to move up (parent priority <= its priority: the normal case in a Dijkstra wave) they are appended, which is what the sequential
// push_heap calls would have done.
__device__ __forceinline__ void pk_pushes(uint64_t* heap, uint32_t& n, const uint64_t* ent, uint32_t cnt, int hl, int hf)
{
    LAMA_LOCKSTEP();
    const uint32_t l4 = (uint32_t)hl & 3u;
    const uint64_t entry = ent[l4];
    const uint32_t pos = n + l4;
    const uint32_t pprio = heap_prio(heap[n >= 4 ? (pos - 1) / 2 : 0]);
    const bool mine = (uint32_t)hl < cnt;
    const bool up = mine && pprio > heap_prio(entry);
    const uint32_t upm = pk_half(__ballot(up), hf);
    const bool fast = cnt > 0 && n >= 4 && upm == 0;
    if (fast) { if (mine) heap[pos] = entry; n += cnt; }
    const bool slow = cnt > 0 && !fast;
    if (__ballot(slow)) {
        #pragma unroll 1
        for (uint32_t i = 0; i < 4; ++i) pk_push1(heap, n, ent[i], slow && i < cnt, hl == 0);
    }
    LAMA_LOCKSTEP();
}

// the non-const Map::get of a half's role lanes (src/sdm/map.cpp:371-412): every lane with `want` gets the slot of window patch
// `pidx` of ITS particle, missing patches are allocated once per distinct patch.  All 64 lanes call; `count` is the half's
// (wave-half-uniform) patch counter.
__device__ inline int pk_coop_slot(const DirCache& dc, int16_t* dir, int& count, int cap, bool want, uint32_t pidx, int errbit, int32_t* err, int lane)
{
    int slot = want ? dc.lookup(pidx) : 0;
    bool need = want && slot < 0;
    for (int h2 = 0; h2 < 2; ++h2) {
        const bool mineh = (lane >> 5) == h2;
        unsigned long long m = __ballot(need && mineh);
        while (m) {
            const int leader = __ffsll((long long)m) - 1;
            const uint32_t lp = (uint32_t)__shfl((int)pidx, leader, 64);
            if (mineh) {
                int ns;
                if (count < cap) { ns = count; ++count; } else { ns = -1; }
                if (lane == leader) {
                    if (ns >= 0) { dir[lp] = (int16_t)ns; dc.update(lp, ns); } else atomicOr(err, errbit);
                }
                if (need && pidx == lp) { slot = ns; need = false; }
            }
            m = __ballot(need && mineh);
        }
    }
    return slot;
}

template <int LQ, int RQ>
__global__ __launch_bounds__(2 * UM_BLOCK) void k_brushfire_packed(DevParams prm, int first_particle, int num_particles)
{
    __shared__ BfLdsP<LQ, RQ> shp[2];
    const int tid = threadIdx.x, lane = tid & 63, hf = lane >> 5, hl = lane & 31;
    const bool helper = tid >= UM_BLOCK;
    // both particles' queue lengths are known to every thread (the loop and exit decisions are wave-uniform)
    uint32_t qn[2][2];
    bool work2[2];
    #pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
        const int pi2 = 2 * (int)blockIdx.x + h2;
        const bool v2 = pi2 < num_particles;
        qn[h2][0] = v2 ? prm.qsizes[2 * (first_particle + pi2)] : 0u;
        qn[h2][1] = v2 ? prm.qsizes[2 * (first_particle + pi2) + 1] : 0u;
        const bool big2 = qn[h2][0] + 4 > (uint32_t)LQ || qn[h2][1] + 4 > (uint32_t)RQ;
        work2[h2] = v2 && (qn[h2][0] | qn[h2][1]) != 0 && !big2;
        // hand-over flag: a queue that does not fit goes to the resume stage untouched
        if (v2 && tid == 0) prm.slow[first_particle + pi2] = big2 ? 1u : 0u;
    }
    if (!work2[0] && !work2[1]) return;

    const int pi = 2 * (int)blockIdx.x + hf;
    const bool work = work2[hf];
    const int p = first_particle + (pi < num_particles ? pi : 0);
    BfLdsP<LQ, RQ>& sh = shp[hf];
    const size_t WW = (size_t)prm.W * prm.W;
    int16_t* dir = prm.dm_dir + (size_t)p * WW;
    uint16_t* sv = prm.dm_sv + (size_t)p * prm.dm_cap * 1024;
    uint32_t* obs = prm.dm_obs + (size_t)p * prm.dm_cap * 1024;
    uint64_t* mask = prm.dm_mask + (size_t)p * prm.dm_cap * 16;
    int count = prm.counts[2 * p];
    uint32_t nl = work ? qn[hf][0] : 0u, nr = work ? qn[hf][1] : 0u;

    #pragma unroll 1
    for (int h2 = 0; h2 < 2; ++h2) {
        if (!work2[h2]) continue;
        const int p2 = first_particle + 2 * (int)blockIdx.x + h2;
        const uint64_t* gl = prm.q_lower + (size_t)p2 * prm.qcap;
        const uint64_t* gr = prm.q_raise + (size_t)p2 * prm.qcap;
        for (int k = tid; k < DC_SIZE; k += 2 * UM_BLOCK) shp[h2].dc[k] = DC_EMPTY;
        for (uint32_t k = tid; k < qn[h2][0]; k += 2 * UM_BLOCK) shp[h2].lower[k] = gl[k];
        for (uint32_t k = tid; k < qn[h2][1]; k += 2 * UM_BLOCK) shp[h2].raise[k] = gr[k];
    }
    __syncthreads();                                           // S0: heaps and directory caches in LDS

    const bool any_raise = (work2[0] && qn[0][1] > 0) || (work2[1] && qn[1][1] > 0);
    bool run = work;                                           // false once the half's particle was handed to the next stage

    if (helper) {
        // ---- helper wave: its halves own the two particles' heaps.  While the raise queue is not empty it is the one popped
        // (dynamic_distance_map.cpp:162-173), then the lower queue (:175-194); the two halves go through the phases together.
        const uint32_t anc = pk_ancestors(hl);
        uint32_t it = 0;
        if (any_raise) {
            for (;;) {
                const bool act = run && nr > 0;
                if (__ballot(act) == 0ull) break;
                const uint32_t b = it & 1u;
                pk_pop(sh.raise, nr, act, &sh.topq[b], hl, hf, anc);
                lds_barrier();                                 // D
                const uint32_t cr = act ? sh.pr_n[b] : 0u, cl = act ? sh.pl_n[b] : 0u;
                pk_pushes(sh.raise, nr, sh.pr_e[b], cr, hl, hf);   // raise() is the only producer of raise entries
                pk_pushes(sh.lower, nl, sh.pl_e[b], cl, hl, hf);
                if (act && (nl + 4 > (uint32_t)LQ || (nr > 0 && nr + 4 > (uint32_t)RQ))) run = false;
                ++it;
            }
            lds_barrier();                                     // X: the last pushes of the raise phase are in the lower queues
        }
        for (;;) {
            const bool act = run && nl > 0;
            if (__ballot(act) == 0ull) break;
            const uint32_t b = it & 1u;
            pk_pop(sh.lower, nl, act, &sh.topq[b], hl, hf, anc);
            lds_barrier();                                     // D
            const uint32_t cl = act ? sh.pl_n[b] : 0u;
            pk_pushes(sh.lower, nl, sh.pl_e[b], cl, hl, hf);
            if (act && nl + 4 > (uint32_t)LQ) run = false;
            ++it;
        }
        __syncthreads();                                       // F: last pushes applied
        return;
    }

    // ---- main wave: cells, raise() / lower() decisions, map stores; derives every next top itself (see k_brushfire)
    const DirCache dc{sh.dc, dir, prm.W};
    const int ddx = hl == 0 ? 1 : (hl == 2 ? -1 : 0), ddy = hl == 1 ? 1 : (hl == 3 ? -1 : 0);
    const bool is_cur = hl == 4, is_nb = hl < 4;
    const uint32_t below = (1u << (hl & 3)) - 1u;
    uint64_t processed = 0;
    uint64_t e_next = 0;
    if (work) e_next = nr > 0 ? sh.raise[0] : sh.lower[0];
    uint32_t it = 0;

    // my best own push into the popped queue goes through LDS across the barrier: smallest (priority, neighbour index) of the
    // four neighbour lanes (quad DPP), the owning lane stores its entry
    #define PK_OWN_BEST(PUSHES_IT, ENTRY, B)                                                                 \
        {                                                                                                   \
            uint32_t key_ = (is_nb && (PUSHES_IT)) ? ((heap_prio(ENTRY) << 2) | (uint32_t)hl) : 0xFFFFFFFFu; \
            uint32_t o_ = pk_quad<0xB1>(key_); key_ = o_ < key_ ? o_ : key_;     /* quad_perm [1,0,3,2] */    \
            o_ = pk_quad<0x4E>(key_); key_ = o_ < key_ ? o_ : key_;              /* quad_perm [2,3,0,1] */    \
            if (is_nb && (PUSHES_IT) && (key_ & 3u) == (uint32_t)hl) sh.own_e[B] = (ENTRY);                  \
            if (hl == 0) sh.own_k[B] = key_;                                                                \
        }

    // ---- raise wave ------------------------------------------------------------------------- :162-173
    if (any_raise) {
        for (;;) {
            const bool act = run && nr > 0;
            if (__ballot(act) == 0ull) break;
            const uint32_t b = it & 1u;
            const uint64_t e = e_next;
            const int rx = q_rx(e), ry = q_ry(e);
            if (act) ++processed;
            // round A: every role lane loads its own cell (side-effect free)
            const int x = rx + ddx, y = ry + ddy;
            const bool role = act && hl < 5;
            const bool inwin = (uint32_t)x < prm.WC && (uint32_t)y < prm.WC;
            const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);
            const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
            int slot = (role && inwin) ? dc.lookup(pidx) : -1;
            uint16_t s = 0; uint32_t ob = 0;
            if (slot >= 0) { s = sv[slot * 1024 + (int)ci]; ob = obs[slot * 1024 + (int)ci]; }
            if (act) --nr;                                     // the helper wave pops
            // round B: the cell my offset points to (obstacle cell); offset 0 -> myself
            const int ox = x + obs_x(ob), oy = y + obs_y(ob);
            uint16_t os = 0;
            {
                const bool oin = role && slot >= 0 && (uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC;
                const int oslot = oin ? dc.lookup(((uint32_t)oy >> 5) * prm.W + ((uint32_t)ox >> 5)) : -1;
                if (oslot >= 0) os = sv[oslot * 1024 + (int)(((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5))];
            }
            // neighbours: get() = allocate + mask bit (all four)
            if (act && is_nb && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool nb = act && is_nb && inwin;
            const bool fresh = nb && slot < 0;
            if (__ballot(fresh)) { const int ns_ = pk_coop_slot(dc, dir, count, (int)prm.dm_cap, fresh, pidx, ERR_DM_CAP, prm.err, lane); if (fresh) slot = ns_; }
            const bool nbok = nb && slot >= 0;
            if (nbok) {
                const uint64_t bit = 1ull << (ci & 63);
                if (fresh || !(s & (SV_VALID | SV_QUEUED))) atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);
            }
            // :253  skip queued or invalid neighbours
            const bool cand = nbok && !(s & SV_QUEUED) && (s & SV_VALID);
            bool ovalid = (os & SV_VALID) != 0;
            // sequential semantics: neighbour i is handled before j > i; if i gets cleared and j's offset points at i, j must see i as
            // invalid (only possible through stale offsets; replayed here to stay exact).  Quad broadcasts of neighbour i's cell / flag.
            bool clear = cand && !ovalid;
            {
                const uint32_t cx0 = pk_quad<0x00>((uint32_t)x), cy0 = pk_quad<0x00>((uint32_t)y), cl0 = pk_quad<0x00>(clear ? 1u : 0u);
                if (hl > 0 && is_nb && cand && cl0 && ox == (int)cx0 && oy == (int)cy0) { ovalid = false; clear = true; }
                const uint32_t cx1 = pk_quad<0x55>((uint32_t)x), cy1 = pk_quad<0x55>((uint32_t)y), cl1 = pk_quad<0x55>(clear ? 1u : 0u);
                if (hl > 1 && is_nb && cand && cl1 && ox == (int)cx1 && oy == (int)cy1) { ovalid = false; clear = true; }
                const uint32_t cx2 = pk_quad<0xAA>((uint32_t)x), cy2 = pk_quad<0xAA>((uint32_t)y), cl2 = pk_quad<0xAA>(clear ? 1u : 0u);
                if (hl > 2 && is_nb && cand && cl2 && ox == (int)cx2 && oy == (int)cy2) { ovalid = false; clear = true; }
            }
            const bool to_raise = cand && !ovalid;             // :262-268
            const bool to_lower = cand && ovalid;              // :269-272
            const uint32_t rm = pk_half(__ballot(to_raise), hf) & 15u, lm = pk_half(__ballot(to_lower), hf) & 15u;
            const uint64_t entry_r = q_entry((uint32_t)(s & SV_SQMASK), x, y);
            if (to_raise) sh.pr_e[b][__popc(rm & below)] = entry_r;
            if (to_lower) sh.pl_e[b][__popc(lm & below)] = q_entry((uint32_t)(s & SV_SQMASK), x, y, obs_x(ob), obs_y(ob));
            const uint32_t cnt_r = act ? (uint32_t)__popc(rm) : 0u, cnt_l = act ? (uint32_t)__popc(lm) : 0u;
            if (hl == 0) { sh.pr_n[b] = cnt_r; sh.pl_n[b] = cnt_l; }
            if (to_raise) { sv[slot * 1024 + (int)ci] = SV_QUEUED; obs[slot * 1024 + (int)ci] = 0u; }
            if (to_lower) sv[slot * 1024 + (int)ci] = (uint16_t)(s | SV_QUEUED);
            if (act && is_cur && slot >= 0) sv[slot * 1024 + (int)ci] = (uint16_t)(s & ~SV_QUEUED);      // :278
            PK_OWN_BEST(to_raise, entry_r, b)
            lds_barrier();                                     // D
            {
                const uint64_t root_ = sh.topq[b];
                const uint32_t kb_ = sh.own_k[b];
                const uint64_t oe_ = sh.own_e[b];
                const bool own_wins_ = kb_ != 0xFFFFFFFFu && (!(nr > 0) || (kb_ >> 2) < heap_prio(root_));
                if (act) {
                    e_next = own_wins_ ? oe_ : root_;
                    nr += cnt_r; nl += cnt_l;
                    if (nl + 4 > (uint32_t)LQ || (nr > 0 && nr + 4 > (uint32_t)RQ)) run = false;
                }
            }
            ++it;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
        lds_barrier();                                         // X
        if (run && nl > 0) e_next = sh.lower[0];
    }

    // ---- lower wave ------------------------------------------------------------------------- :175-194
    for (;;) {
        const bool act = run && nl > 0;
        if (__ballot(act) == 0ull) break;
        const uint32_t b = it & 1u;
        const uint64_t e = e_next;
        const int rx = q_rx(e), ry = q_ry(e);
        if (act) ++processed;
        // ONE load round: lanes 0..4 of the half their own cell, lane 5 the obstacle cell the entry says the popped cell points to
        const bool is_oc = hl == 5;
        const int x = rx + (is_oc ? q_ox(e) : ddx), y = ry + (is_oc ? q_oy(e) : ddy);
        const bool role = act && hl < 6;
        const bool inwin = (uint32_t)x < prm.WC && (uint32_t)y < prm.WC;
        const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);
        const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
        int slot = (role && inwin) ? dc.lookup(pidx) : -1;
        uint16_t s = 0; uint32_t ob = 0;
        if (slot >= 0) { s = sv[slot * 1024 + (int)ci]; if (!is_oc) ob = obs[slot * 1024 + (int)ci]; }
        if (act) --nl;                                         // the helper wave pops
        const uint16_t cs = (uint16_t)pk_bcast((uint32_t)s, lane, 4);
        const uint32_t cob = pk_bcast(ob, lane, 4);
        uint16_t cos_ = (uint16_t)pk_bcast((uint32_t)s, lane, 5);
        const bool stale = act && (obs_x(cob) != q_ox(e) || obs_y(cob) != q_oy(e));
        if (__ballot(stale)) {
            // stale entry (the cell was overwritten after it was queued): fetch the obstacle cell it points to now
            if (stale) {
                const int ox2 = rx + obs_x(cob), oy2 = ry + obs_y(cob);
                uint16_t t = 0;
                if ((uint32_t)ox2 < prm.WC && (uint32_t)oy2 < prm.WC) {
                    const int os2 = dc.lookup(((uint32_t)oy2 >> 5) * prm.W + ((uint32_t)ox2 >> 5));
                    if (os2 >= 0) t = sv[os2 * 1024 + (int)(((uint32_t)ox2 & 31u) | (((uint32_t)oy2 & 31u) << 5))];
                }
                cos_ = t;
            }
        }
        // :183-192  valid, its obstacle still has sqdist 0 (valid NOT tested), and lower() :283 still queued
        const bool fire = act && (cs & SV_VALID) && (cos_ & SV_SQMASK) == 0 && (cs & SV_QUEUED);
        uint32_t cnt_l = 0;
        uint64_t entry_l = 0;
        bool over = false;
        if (__ballot(fire)) {
            const int cox = obs_x(cob), coy = obs_y(cob);
            const int obx = rx + cox, oby = ry + coy;
            const bool away = fire && is_nb && !(ddx * cox > 0 || ddy * coy > 0);            // :296
            if (away && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool nb = away && inwin;
            const bool fresh = nb && slot < 0;
            if (__ballot(fresh)) { const int ns_ = pk_coop_slot(dc, dir, count, (int)prm.dm_cap, fresh, pidx, ERR_DM_CAP, prm.err, lane); if (fresh) slot = ns_; }
            const bool nbok = nb && slot >= 0;
            if (nbok) {
                const uint64_t bit = 1ull << (ci & 63);
                if (fresh || !(s & (SV_VALID | SV_QUEUED))) atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);   // see raise()
            }
            const int qx = x - obx, qy = y - oby;
            const uint32_t new_sq = (uint32_t)(qx * qx + qy * qy);
            const uint32_t cmp = (s & SV_VALID) ? (uint32_t)(s & SV_SQMASK) : prm.max_sqdist;
            over = nbok && new_sq < cmp;
            const bool tie = nbok && !over && new_sq == (uint32_t)(s & SV_SQMASK);           // :311-317
            if (__ballot(tie)) {
                // the neighbour's own obstacle cell: usually the very cell the popped cell points to (both were reached from the same
                // obstacle), whose state the half already holds -- no second load round then
                const int ox = x + obs_x(ob), oy = y + obs_y(ob);
                const bool same = ox == obx && oy == oby;
                uint16_t os = cos_;
                if (tie && !same) {
                    os = 0;
                    if ((uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC) {
                        const int oslot = dc.lookup(((uint32_t)oy >> 5) * prm.W + ((uint32_t)ox >> 5));
                        if (oslot >= 0) os = sv[oslot * 1024 + (int)(((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5))];
                    }
                }
                if (tie && (!(s & SV_VALID) || !((os & SV_VALID) && (os & SV_SQMASK) == 0))) over = true;
            }
            if (over) {
                sv[slot * 1024 + (int)ci] = (uint16_t)(SV_VALID | SV_QUEUED | (new_sq & SV_SQMASK));
                obs[slot * 1024 + (int)ci] = pack_obs(obx - x, oby - y);
            }
            if (fire && is_cur && slot >= 0) sv[slot * 1024 + (int)ci] = (uint16_t)(cs & ~SV_QUEUED);   // :329
            // pushes in neighbour order: the helper wave applies them
            const uint32_t om = pk_half(__ballot(over), hf) & 15u;
            entry_l = q_entry(new_sq, x, y, obx - x, oby - y);
            if (over) sh.pl_e[b][__popc(om & below)] = entry_l;
            cnt_l = fire ? (uint32_t)__popc(om) : 0u;
        }
        if (hl == 0) sh.pl_n[b] = cnt_l;
        PK_OWN_BEST(over, entry_l, b)
        lds_barrier();                                         // D
        {
            const uint64_t root_ = sh.topq[b];
            const uint32_t kb_ = sh.own_k[b];
            const uint64_t oe_ = sh.own_e[b];
            const bool own_wins_ = kb_ != 0xFFFFFFFFu && (!(nl > 0) || (kb_ >> 2) < heap_prio(root_));
            if (act) {
                e_next = own_wins_ ? oe_ : root_;
                nl += cnt_l;
                if (nl + 4 > (uint32_t)LQ) run = false;
            }
        }
        ++it;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    #undef PK_OWN_BEST
    __syncthreads();                                           // F: the helper wave has applied the last pushes
    const bool spilled = work && !run;
    if (spilled) {      // hand the particle over, state intact, to the resume stage of k_brushfire
        uint64_t* gl = prm.q_lower + (size_t)p * prm.qcap;
        uint64_t* gr = prm.q_raise + (size_t)p * prm.qcap;
        for (uint32_t k = hl; k < nl; k += 32) gl[k] = sh.lower[k];
        for (uint32_t k = hl; k < nr; k += 32) gr[k] = sh.raise[k];
    }
    if (work && hl == 0) {
        prm.counts[2 * p] = count;
        prm.stats[4 * p + 3] += processed;
        if (spilled) { prm.qsizes[2 * p] = nl; prm.qsizes[2 * p + 1] = nr; prm.slow[p] = 1; }
    }
}

} // namespace lama_dev
