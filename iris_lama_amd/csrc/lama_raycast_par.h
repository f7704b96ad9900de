// lama_raycast_par.h -- parallel EXACT ray-cast of PFSlam2D::updateParticleMaps (src/pf_slam2d.cpp:458-505).
//
// k_raycast (lama_kernels.h) walks the 1080 beams of a particle one after the other because the reference's
// result is order dependent in one respect only: WHICH visit of a cell crosses the 0.25 occupancy threshold
// decides where the add/remove-obstacle event lands in the brushfire queues.  The uint16 counters themselves
// commute.  This file splits the work accordingly:
//
//   k_ray_hits   (thread / beam)   : the hit cells of this scan are flagged (one bit per cell) and queued as
//                                    "active" visits;
//   k_ray_patches (workgroup / occupancy patch, lama_raycast_patch.h; the round-1 beam-centric form with LDS-aggregated
//                                    atomics is in the history: tools/research/README.md):
//                                    every free-cell visit (beam, t) in parallel.  A visit is INERT when its cell
//                                    is not hit in this scan and is already free (4*occupied < visited: a miss can
//                                    never raise an event, src/sdm/frequency_occupancy_map.cpp:65-74) or brand new
//                                    (visited == 0: its first miss always raises removeObstacle, which is a no-op
//                                    on the distance map apart from get()'s patch allocation + mask bit): inert
//                                    visits are one atomicAdd.  All other visits (cells hit in this scan, cells at
//                                    or above the threshold) are ACTIVE: appended to a list, the cell untouched;
//   k_ray_replay (workgroup / particle): sorts the active list by (cell, beam, step), replays every active cell's
//                                    visits in the reference's order from the untouched (occupied, visited) state
//                                    -- one thread per cell, cells in parallel -- applies add/removeObstacle to the
//                                    distance-map cell, then sorts the resulting events by (beam, step) and writes
//                                    the lower/raise queues exactly as the sequential code would have pushed them.
//
// Exactness caveat (documented in DESIGN.md): a cell whose uint16 `visited` counter wraps to 0 inside a scan
// (65,536 visits) is treated like the reference does for the counters, but a wrap can make an "inert" cell active
// mid-scan; the sequential kernel (cfg.sequential_raycast) is exact there too.
#pragma once
#include "lama_dev.h"

namespace lama_dev {


constexpr int RP_BLOCK_SMALL = 256, RP_BLOCK_LARGE = 1024;   // k_ray_replay workgroup: 1024 threads while the chip is not full
#ifdef LAMA_TEST_SMALL_QUEUES       // tests/sim only: a first replay stage so small that a corridor scan is handed to the resume stage
constexpr int RP_SORT_SMALL = 512;
#else
constexpr int RP_SORT_SMALL = 2048; // active visits the first replay stage sorts in LDS (the resume stage: 8192)
#endif

// ---- directory entry (int16 inside an aligned 32-bit word) with lock-free allocation ----------------------
// -1 = absent, -2 = being allocated, -3 = allocation failed (arena full), >= 0 = slot (never changes afterwards)
//
// dir_alloc_one is the allocation proper, for ONE lane of a wave at a time: it may spin on an entry another WAVE is allocating.
// It must never be entered by two lanes of the same wave that want the same entry -- the winner's critical section and the
// losers' spin are one SIMT instruction stream, and wherever the compiler places the critical section after the spin loop's exit
// (it does, depending on the call site) the losers spin forever on a lock their own wave holds.  dir_get_or_alloc therefore
// elects, among the lanes that are active at the call, one leader per distinct entry; the others take the leader's result.
__device__ inline int dir_alloc_one(int16_t* dir, uint32_t pidx, int32_t* count, int cap, int errbit, int32_t* err)
{
    uint32_t* w = reinterpret_cast<uint32_t*>(dir) + (pidx >> 1);
    const int sh = (int)(pidx & 1u) * 16;
    for (uint32_t spins = 0;; ++spins) {
        const uint32_t v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int s = (int)(int16_t)((v >> sh) & 0xFFFFu);
        if (s >= 0) return s;
        if (s == -3) return -1;
        if (spins > (1u << 22)) { atomicOr(err, errbit); return -1; }       // never seen; a reported error beats a hung device
        if (s == -1) {
            const uint32_t locked = (v & ~(0xFFFFu << sh)) | (0xFFFEu << sh);
            if (atomicCAS(w, v, locked) == v) {
                int ns = atomicAdd(count, 1);
                // (a failed allocation leaves the counter incremented: after a cleanly aborted allocation phase it tells the host how
                // many patches the particle WANTED, and its region is grown by exactly that -- recover_update resets the count)
                if (ns >= cap) { atomicOr(err, errbit); ns = -3; }
                for (;;) {      // publish our half (the other half may change concurrently)
                    const uint32_t cur = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t nv = (cur & ~(0xFFFFu << sh)) | (((uint32_t)(uint16_t)(int16_t)ns) << sh);
                    if (atomicCAS(w, cur, nv) == cur) break;
                }
                return ns >= 0 ? ns : -1;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}
__device__ inline int dir_get_or_alloc(int16_t* dir, uint32_t pidx, int32_t* count, int cap, int errbit, int32_t* err)
{
    int s = dir[pidx];
    if (s >= 0) return s;
#ifdef LAMA_WAVE_SIM      // tests/sim (lane-level simulator): lanes are cooperative fibers, a lane's critical section cannot be
    return dir_alloc_one(dir, pidx, count, cap, errbit, err);     // interleaved with another lane's spin -- no election needed
#endif
    // slow path: one leader per distinct (directory, entry) among the lanes that got here together
    const int lane = (int)(threadIdx.x & 63u);
    const uint64_t key = ((uint64_t)(uintptr_t)dir << 20) ^ (uint64_t)pidx;       // directories are >= 2 KB apart, pidx < 2^16
    int res = -1;
    bool pending = true;
    for (;;) {
        const unsigned long long m = __ballot(pending);              // only the active lanes vote
        if (m == 0ull) break;
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t klo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)key, leader);
        const uint32_t khi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(key >> 32), leader);
        int r = 0;
        if (lane == leader) r = dir_alloc_one(dir, pidx, count, cap, errbit, err);
        r = __builtin_amdgcn_readlane(r, leader);
        if (pending && (uint32_t)key == klo && (uint32_t)(key >> 32) == khi) { res = r; pending = false; }
    }
    return res;
}

struct BeamGeom {
    uint32_t mhx, mhy, msx, msy;   // hit / start cell (map coordinates)
    uint32_t a0, a1, nn;           // |delta| per axis, n = max over the three axes
    int s0, s1;
    bool mark_hit;
    int steps;
    uint64_t magic;
};

// hit / start cells of one beam (src/pf_slam2d.cpp:463-493), identical arithmetic to k_raycast
__device__ inline BeamGeom beam_geometry(const DevParams& prm, const double* T, double px, double py, double pz)
{
    double hx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[9];
    double hy = ((T[3] * px + T[4] * py) + T[5] * pz) + T[10];
    double hz = ((T[6] * px + T[7] * py) + T[8] * pz) + T[11];
    double sx = T[9], sy = T[10], sz = T[11];
    double abx = 0, aby = 0, abz = 0, ray_length = 1.0;
    BeamGeom g;
    g.mark_hit = true;
    if (prm.trunc_range > 0.0) {
        abx = hx - sx; aby = hy - sy; abz = hz - sz;
        ray_length = sqrt((abx * abx + aby * aby) + abz * abz);
        if (prm.trunc_range < ray_length) {
            hx = sx + abx / ray_length * prm.trunc_range;
            hy = sy + aby / ray_length * prm.trunc_range;
            hz = sz + abz / ray_length * prm.trunc_range;
            g.mark_hit = false;
        }
    }
    if (g.mark_hit && prm.trunc_ray > 0.0) {
        if (prm.trunc_range == 0.0) {
            abx = hx - sx; aby = hy - sy; abz = hz - sz;
            ray_length = sqrt((abx * abx + aby * aby) + abz * abz);
        }
        if (prm.trunc_ray < ray_length) {
            sx = hx - abx / ray_length * prm.trunc_ray;
            sy = hy - aby / ray_length * prm.trunc_ray;
            sz = hz - abz / ray_length * prm.trunc_ray;
        }
    }
    g.mhx = w2m(prm, hx); g.mhy = w2m(prm, hy);
    g.msx = w2m(prm, sx); g.msy = w2m(prm, sy);
    const uint32_t mhz = w2m(prm, hz), msz = w2m(prm, sz);
    const int64_t d0 = (int64_t)g.mhx - (int64_t)g.msx, d1 = (int64_t)g.mhy - (int64_t)g.msy, d2 = (int64_t)mhz - (int64_t)msz;
    const int64_t a0 = d0 < 0 ? -d0 : d0, a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
    const int64_t nn = a0 > a1 ? (a0 > a2 ? a0 : a2) : (a1 > a2 ? a1 : a2);
    g.s0 = d0 < 0 ? -1 : 1; g.s1 = d1 < 0 ? -1 : 1;
    g.a0 = (uint32_t)a0; g.a1 = (uint32_t)a1; g.nn = (uint32_t)(nn > 0xFFFFFFFFll ? 0xFFFFFFFFll : nn);
    g.steps = nn > 0 ? (int)g.nn - 1 : 0;
    if (nn >= 8192) g.steps = -1;                       // longer than any window
    g.magic = (1ull << 42) / (uint64_t)(2 * (nn > 0 && nn < 8192 ? nn : 1)) + 1ull;
    return g;
}

// active-visit key: (cellkey << 26) | seq, cellkey = ry << 15 | rx (window-relative cell: 15 bits each, a window of up to 1016
// patches), seq = beam << 13 | t (13 bits each: scans of up to 8192 points, rays of up to 8191 cells; t = 0: the hit)
constexpr int ACT_SEQ_BITS = 26, ACT_XY_BITS = 15;
__device__ inline uint64_t act_key(uint32_t rx, uint32_t ry, uint32_t beam, uint32_t t)
{
    return ((uint64_t)((ry << ACT_XY_BITS) | rx) << ACT_SEQ_BITS) | (uint64_t)((beam << 13) | t);
}

__device__ inline void act_append(const DevParams& prm, int p, uint64_t key)
{
    const uint32_t k = atomicAdd(prm.act_count + p, 1u);
    if (k < prm.act_cap) prm.act[(size_t)p * prm.act_cap + k] = key;
    else atomicOr(prm.err, ERR_QUEUE);
}

// ------------------------------------------------------------------------------------------------
// With `rec_out` (the patch-centric form, lama_raycast_patch.h) the thread also stores its beam's ray record and the bounding box
// of the cells the ray visits; k_ray_alloc_walk then allocates the occupancy patches the rays cross, so that the patch pass finds
// every patch in the directory.
struct RayRec;
struct RayChunk;
constexpr uint64_t RAY_BBOX_EMPTY = 0x0000FFFF0000FFFFull;   // x0 = y0 = 0xFFFF > x1 = y1 = 0
__device__ inline uint64_t ray_hits_record(const DevParams& prm, const BeamGeom& g, int p, int i, int n, RayRec* rec_out, uint64_t* bbox_out);
__device__ inline void ray_chunk_record(const DevParams& prm, const BeamGeom& g, uint64_t bb, int lane, RayChunk* chunks, size_t index);

// alloc_only: the beam-sequential ray-cast (k_raycast) follows -- only the allocation phase is wanted here (ray records for the
// allocation walk, the hit cells' patches); no hit bits, no active-visit list.
__global__ __launch_bounds__(256) void k_ray_hits(DevParams prm, const double* __restrict__ pts, int n,
                                                   const double* __restrict__ tfs, int first_particle,
                                                   RayRec* __restrict__ rec_out = nullptr, uint64_t* __restrict__ bbox_out = nullptr, int alloc_only = 0,
                                                   RayChunk* __restrict__ chunk_out = nullptr)
{
    const int p = first_particle + blockIdx.x;
    if (p >= (int)prm.P) return;                    // (the particle dimension of the grid is rounded up to a multiple of 8: xcd_grid)
    const int i = blockIdx.y * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool live = i < n;
    const int ic = live ? i : n - 1;
    double T[12];
    uload_f64_w<12>(tfs + 12 * (size_t)p, T);                 // (rewritten by the host before every update: coherent uniform loads, lama_dev.h)
    const BeamGeom g = beam_geometry(prm, T, pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]);
    uint64_t bb = RAY_BBOX_EMPTY;
    if (rec_out && live) bb = ray_hits_record(prm, g, p, i, n, rec_out, bbox_out);
    if (chunk_out && (i & ~63) < n) ray_chunk_record(prm, g, bb, lane, chunk_out, (size_t)p * ((n + 63) / 64) + (i >> 6));   // wave-uniform condition
    if (live && g.steps < 0) atomicOr(prm.err, ERR_WINDOW);
    // ray cells of the scan (statistics): one atomic per wave, not one per beam on the particle's single counter
    if (!alloc_only) {
        uint32_t st = (live && g.steps > 0) ? (uint32_t)g.steps : 0u;
        for (int off = 32; off > 0; off >>= 1) st += (uint32_t)__shfl_xor((int)st, off, 64);
        if (lane == 0 && st) atomicAdd((unsigned long long*)(prm.stats + 4 * p + 2), (unsigned long long)st);
    }
    bool hit = live && g.steps >= 0 && g.mark_hit;
    const uint32_t rx = g.mhx - prm.wx0, ry = g.mhy - prm.wy0;
    if (hit && (rx >= prm.WC || ry >= prm.WC)) { atomicOr(prm.err, ERR_WINDOW); hit = false; }
    const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5), ci = (rx & 31u) | ((ry & 31u) << 5);
    int slot = -1;
    const PV pv = pview(prm, p);
    if (hit) slot = dir_get_or_alloc(pv.occ_dir, pidx, prm.counts + 2 * p + 1, (int)pv.occ_cap, ERR_OCC_CAP, prm.err);
    hit = hit && slot >= 0 && !alloc_only;
    if (hit) atomicOr((unsigned long long*)(pv.occ_hit + (size_t)slot * 16 + (ci >> 6)), 1ull << (ci & 63));
    // the hits are order-sensitive visits (t = 0): appended with one counter update per wave (their order in the list is free,
    // k_ray_replay sorts)
    const unsigned long long hm = __ballot(hit);
    if (hm) {
        uint32_t base = 0;
        if (lane == (__ffsll((long long)hm) - 1)) base = atomicAdd(prm.act_count + p, (uint32_t)__popcll(hm));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)hm) - 1, 64);
        if (hit) {
            const uint32_t k = base + (uint32_t)__popcll(hm & ((1ull << lane) - 1ull));
            if (k < prm.act_cap) prm.act[(size_t)p * prm.act_cap + k] = act_key(rx, ry, (uint32_t)i, 0u);
            else atomicOr(prm.err, ERR_QUEUE);
        }
    }
}

// ---- in-LDS bitonic sort (ascending) of m = 2^k keys by the RP_BLOCK threads of the workgroup ---------------------------------
template <int RP_BLOCK>
__device__ inline void bitonic_sort(uint64_t* a, uint32_t m)
{
    for (uint32_t k = 2; k <= m; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < m; i += RP_BLOCK) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const uint64_t x = a[i], y = a[ixj];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { a[i] = y; a[ixj] = x; }
                }
            }
            __syncthreads();
        }
    }
}

// The same sort with E keys per thread held in registers (m = RP_BLOCK * E keys; key i lives in thread i / E, register i % E).  A
// bitonic network exchanges key i with key i ^ j; bit b of the index is used by log2(m) - b stages, so the bits that are used most
// are the cheapest here: the low log2(E) bits are register-to-register (no memory at all), the next six are lanes of one wave (a
// cross-lane move, no barrier), and only the top bits -- 3 of the 66 stages for 2,048 keys on 256 threads -- go through LDS with a
// barrier pair.  The LDS form above pays two LDS reads, two conditional writes and a barrier for every one of the 66 stages.
template <int RP_BLOCK, int E>
__device__ inline void bitonic_sort_regs(uint64_t* a)
{
    constexpr uint32_t M = (uint32_t)RP_BLOCK * E;
    const uint32_t base = threadIdx.x * (uint32_t)E;
    uint64_t x[E];
#pragma unroll
    for (int r = 0; r < E; ++r) x[r] = a[base + r];
    for (uint32_t k = 2; k <= M; k <<= 1) {
        uint32_t j = k >> 1;
        for (; j >= 64u * E; j >>= 1) {                        // partners in another wave: through LDS
            __syncthreads();
#pragma unroll
            for (int r = 0; r < E; ++r) a[base + r] = x[r];
            __syncthreads();
            const uint32_t pb = base ^ j;
            const bool keep_min = ((base & j) == 0) == ((base & k) == 0);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint64_t y = a[pb + r];
                x[r] = ((y < x[r]) == keep_min) ? y : x[r];
            }
        }
        for (; j >= (uint32_t)E; j >>= 1) {                    // partners in another lane of this wave
            const int lj = (int)(j / (uint32_t)E);
            const bool keep_min = ((base & j) == 0) == ((base & k) == 0);
#pragma unroll
            for (int r = 0; r < E; ++r) {
                const uint64_t y = __shfl_xor(x[r], lj, 64);
                x[r] = ((y < x[r]) == keep_min) ? y : x[r];
            }
        }
#pragma unroll
        for (int jr = E / 2; jr > 0; jr >>= 1) {               // partners in this thread's registers
            if ((uint32_t)jr >= k) continue;
#pragma unroll
            for (int r = 0; r < E; ++r) {
                if (r & jr) continue;
                const bool up = ((base + (uint32_t)r) & k) == 0;
                const uint64_t lo = x[r], hi = x[r | jr];
                const bool sw = (lo > hi) == up;
                x[r] = sw ? hi : lo; x[r | jr] = sw ? lo : hi;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < E; ++r) a[base + r] = x[r];
    __syncthreads();
}

// sorts the first `m` keys of `a` (m a power of two; the caller padded with ~0): the register form whenever a thread's share fits
// eight register pairs, else (the resume stage's 8,192 keys on 256 threads) the LDS form.  Returns with the keys in LDS, barrier done.
template <int RP_BLOCK>
__device__ inline void sort_keys(uint64_t* a, uint32_t m)
{
    if (m < (uint32_t)RP_BLOCK) { if (m > 1) bitonic_sort<RP_BLOCK>(a, m); }
    else if (m == (uint32_t)RP_BLOCK) bitonic_sort_regs<RP_BLOCK, 1>(a);
    else if (m == 2u * RP_BLOCK) bitonic_sort_regs<RP_BLOCK, 2>(a);
    else if (m == 4u * RP_BLOCK) bitonic_sort_regs<RP_BLOCK, 4>(a);
    else if (m == 8u * RP_BLOCK) bitonic_sort_regs<RP_BLOCK, 8>(a);
    else bitonic_sort<RP_BLOCK>(a, m);
}

template <int SORT_CAP, int EV_CAP>
struct ReplayLds {
    uint64_t keys[SORT_CAP];
    uint64_t ev[EV_CAP];
    uint32_t ev_n;
    uint32_t nadd, nrem;
};

// event: (seq << 32) | (is_add << 31) | cellkey
// the ordered replay of ONE particle's active visits by the calling workgroup (body of k_ray_replay)
template <int SORT_CAP, int EV_CAP, bool RESUME, int RP_BLOCK>
__device__ __forceinline__ void ray_replay_particle(const DevParams& prm, const int p, ReplayLds<SORT_CAP, EV_CAP>& sh)
{
    const uint32_t n = prm.act_count[p];
    if (RESUME && prm.slow[p] == 0) return;
    if (n > prm.act_cap) return;                              // overflow already reported by act_append
    if (n > (uint32_t)SORT_CAP) {                             // hand over to the next (bigger) stage: flag + list entry
        if (threadIdx.x == 0) {
            if (RESUME) atomicOr(prm.err, ERR_QUEUE);
            else { const uint32_t seg = prm.lane ? 3u : 1u; prm.slow[p] = 1; prm.slow_list[(size_t)seg * prm.P + atomicAdd(prm.slow_n + seg, 1u)] = (uint32_t)p; }
        }
        return;
    }
    const PV pv = pview_w(prm, p);
    int16_t* occ_dir = pv.occ_dir;
    int16_t* dm_dir = pv.dm_dir;
    uint32_t* occ = pv.occ;
    sv_t* dm_sv = pv.dm_sv;
    uint32_t* dm_obs = pv.dm_obs;
    uint64_t* q_lower = prm.q_lower + (size_t)p * prm.qcap;
    uint64_t* q_raise = prm.q_raise + (size_t)p * prm.qcap;

    uint32_t m = 1;
    while (m < n) m <<= 1;
    const uint64_t* src = prm.act + (size_t)p * prm.act_cap;
    for (uint32_t i = threadIdx.x; i < m; i += RP_BLOCK) sh.keys[i] = i < n ? src[i] : ~0ull;
    if (threadIdx.x == 0) { sh.ev_n = 0; sh.nadd = 0; sh.nrem = 0; }
    __syncthreads();
    sort_keys<RP_BLOCK>(sh.keys, m);

    // ---- per-cell replay in the reference's visit order (one thread per active cell) ----
    for (uint32_t i = threadIdx.x; i < n; i += RP_BLOCK) {
        const uint64_t k0 = sh.keys[i];
        const uint32_t ck = (uint32_t)(k0 >> ACT_SEQ_BITS);
        if (i > 0 && (uint32_t)(sh.keys[i - 1] >> ACT_SEQ_BITS) == ck) continue;      // not the first visit of its cell
        const uint32_t rx = ck & 32767u, ry = ck >> ACT_XY_BITS;
        const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5), ci = (rx & 31u) | ((ry & 31u) << 5);
        const uint64_t bit = 1ull << (ci & 63);
        const int slot = occ_dir[pidx];                                     // allocated by k_ray_hits / k_ray_alloc_walk
        if (slot < 0) continue;
        uint32_t* cell = occ + (size_t)slot * 1024 + ci;
        const uint32_t v = *cell;
        uint32_t o = v & 0xFFFFu, vis = v >> 16;
        bool dm_loaded = false, dirty = false, had_hit = false;
        int dslot = -1;
        sv_t s = 0;
        for (uint32_t j = i; j < n; ++j) {
            const uint64_t kj = sh.keys[j];
            if ((uint32_t)(kj >> ACT_SEQ_BITS) != ck) break;
            const uint32_t seq = (uint32_t)(kj & 0x3FFFFFFu);
            const bool is_hit = (seq & 8191u) == 0;
            bool changed;
            if (is_hit) {                                                   // setOccupied (frequency_occupancy_map.cpp:81-91)
                had_hit = true;
                const bool occupied = vis != 0 && 4u * o > vis;
                o = (o + 1) & 0xFFFFu; vis = (vis + 1) & 0xFFFFu;
                changed = !occupied && (vis != 0 && 4u * o > vis);
            } else {                                                        // setFree (:65-74)
                const bool was_free = vis != 0 && 4u * o < vis;
                vis = (vis + 1) & 0xFFFFu;
                changed = !was_free && (vis != 0 && 4u * o < vis);
            }
            if (vis == 0) atomicOr((unsigned long long*)(pv.occ_mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);
            if (!changed) continue;
            if (!dm_loaded) {                                               // add/removeObstacle: get() = allocate + mask bit
                dslot = dir_get_or_alloc(dm_dir, pidx, prm.counts + 2 * p, (int)pv.dm_cap, ERR_DM_CAP, prm.err);
                dm_loaded = true;
                if (dslot >= 0) {
                    atomicOr((unsigned long long*)(pv.dm_mask + (size_t)dslot * 16 + (ci >> 6)), (unsigned long long)bit);
                    s = dm_sv[dslot * 1024 + (int)ci];
                }
            }
            if (dslot < 0) continue;
            const bool is_obstacle = (s & SV_VALID) && (s & SV_SQMASK) == 0;
            bool emit = false;
            if (is_hit && !is_obstacle) { s = (sv_t)(SV_VALID | SV_QUEUED); emit = true; }     // addObstacle :212-226
            if (!is_hit && is_obstacle) { s = SV_QUEUED; emit = true; }                            // removeObstacle :228-242
            if (emit) {
                dirty = true;
                const uint32_t e = atomicAdd(&sh.ev_n, 1u);
                if (e < (uint32_t)EV_CAP) sh.ev[e] = ((uint64_t)seq << 32) | ((uint64_t)(is_hit ? 1u : 0u) << 31) | ck;
                else atomicOr(prm.err, ERR_QUEUE);
            }
        }
        *cell = o | (vis << 16);
        if (dirty) { dm_sv[dslot * 1024 + (int)ci] = s; dm_obs[dslot * 1024 + (int)ci] = 0; }
        if (had_hit) atomicAnd((unsigned long long*)(pv.occ_hit + (size_t)slot * 16 + (ci >> 6)), ~(unsigned long long)bit);
    }
    __syncthreads();

    // ---- events in (beam, step) order -> lower (adds) / raise (removes) queues ----
    const uint32_t ne = sh.ev_n < (uint32_t)EV_CAP ? sh.ev_n : (uint32_t)EV_CAP;
    uint32_t me = 1;
    while (me < ne) me <<= 1;
    for (uint32_t i = ne + threadIdx.x; i < me; i += RP_BLOCK) sh.ev[i] = ~0ull;
    __syncthreads();
    if (me > 1) bitonic_sort<RP_BLOCK>(sh.ev, me);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t nadd = 0, nrem = 0;
        for (uint32_t base = 0; base < ne; base += 64) {
            const uint32_t i = base + lane;
            const bool act = i < ne;
            const uint64_t e = act ? sh.ev[i] : 0ull;
            const bool is_add = act && ((e >> 31) & 1ull);
            const bool is_rem = act && !is_add;
            const unsigned long long ma = __ballot(is_add), mr = __ballot(is_rem);
            const uint32_t ck = (uint32_t)(e & 0x3FFFFFFFu);                 // (bit 31: add; the cell key has 30 bits)
            const int rx = (int)(ck & 32767u), ry = (int)(ck >> ACT_XY_BITS);
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (is_add) { const uint32_t k = nadd + (uint32_t)__popcll(ma & lt); if (k < prm.qcap) q_lower[k] = q_entry(0, rx, ry); else atomicOr(prm.err, ERR_QUEUE); }
            if (is_rem) { const uint32_t k = nrem + (uint32_t)__popcll(mr & lt); if (k < prm.qcap) q_raise[k] = q_entry(0, rx, ry); else atomicOr(prm.err, ERR_QUEUE); }
            nadd += (uint32_t)__popcll(ma);
            nrem += (uint32_t)__popcll(mr);
        }
        if (lane == 0) {
            prm.qsizes[2 * p] = nadd < prm.qcap ? nadd : prm.qcap;
            prm.qsizes[2 * p + 1] = nrem < prm.qcap ? nrem : prm.qcap;
            prm.act_count[p] = 0;
            prm.slow[p] = 0;
        }
    }
}


// First stage: a workgroup per particle.  Resume stage (128 KB of LDS: one workgroup per CU): a small grid walks the list of the
// particles the first stage handed over -- usually none (see k_brushfire).
template <int SORT_CAP, int EV_CAP, bool RESUME, int RP_BLOCK>
__global__ __launch_bounds__(RP_BLOCK) void k_ray_replay(DevParams prm, int first_particle)
{
    __shared__ ReplayLds<SORT_CAP, EV_CAP> sh;
    if (map_update_aborted(prm)) return;                      // the allocation phase failed: the maps stay untouched (ERR_CLEAN_ABORT)
    if (!RESUME) {
        const int p = lane_particle(prm, first_particle, (int)blockIdx.x);
        if (p >= 0) ray_replay_particle<SORT_CAP, EV_CAP, RESUME, RP_BLOCK>(prm, p, sh);
        return;
    }
    const uint32_t seg = prm.lane ? 3u : 1u;                  // the lane's own hand-over segment
    const uint32_t n = uload_u32(prm.slow_n + seg);           // (zeroed by the host's memset at the start of the update: coherent uniform loads)
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        ray_replay_particle<SORT_CAP, EV_CAP, RESUME, RP_BLOCK>(prm, (int)uload_u32(prm.slow_list + (size_t)seg * prm.P + i), sh);
        __syncthreads();
    }
}

} // namespace lama_dev
