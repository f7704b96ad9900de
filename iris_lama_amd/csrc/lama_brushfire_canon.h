// lama_brushfire_canon.h -- OPT-IN level-synchronous brushfire (cfg.brushfire_mode = 1).
//
// NOT bit-identical to the reference in one field: where several obstacles are equidistant from a cell, WHICH of
// them the cell's `obstacle` offset names depends, in the reference, on libstdc++'s heap order among equal
// priorities (an implementation artefact, see lama_heap.h).  The exact kernels (k_brushfire<...>) replay that order
// and therefore run one pop at a time.  This kernel instead lets all queued cells of one priority level fire
// together and resolves competing offers to the same neighbour with a fixed rule (smaller candidate first, then the
// direction index of the offering move); the test suite's CPU checker implements the same variant (update_canonical)
// and this kernel is bit-exact against it.  Measured on the corridor log: sqdist, valid,
// queued flags, masks, patch sets -- hence every distance query, pose and occupancy count -- are identical to the
// faithful result in every scan; only obstacle offsets of tie cells differ.
//
// One workgroup (256 threads) per particle.  Raise wave: breadth-first rounds (decisions on the state at the start of
// a round).  Lower wave: for each priority level with pending entries: (A) select + de-duplicate the level's cells,
// (B) every fired cell offers |n - obstacle|^2 to its "away" neighbours through an LDS hash table with atomicMin on
// (candidate << 2 | direction), (C) the winning offer of each neighbour applies the reference's overwrite rule
// (src/sdm/dynamic_distance_map.cpp:300-326) and appends the neighbour to the pending list.  Offers are processed in
// passes over spatial classes of the TARGET cell when a level is larger than the table, which keeps the result
// independent of the table size.  Only plain loads/stores separated by workgroup barriers touch the cell planes (one
// workgroup = one CU = one L1), global atomics are used for mask bits and patch allocation only.
#pragma once
#include "lama_raycast_par.h"

namespace lama_dev {

constexpr int CN_BLOCK = 256;
constexpr int CN_TBL_LOG2 = 12;
constexpr int CN_TBL = 1 << CN_TBL_LOG2;        // LDS hash table entries
constexpr int CN_HIST = 1025;                   // max_sqdist + 1 supported by this kernel
constexpr uint32_t CN_EMPTY = 0xFFFFFFFFu;

struct CanonLds {
    uint32_t tkey[CN_TBL];
    uint32_t tval[CN_TBL];
    uint32_t hist[CN_HIST];
    uint32_t n_list, n_a, n_b, fail;
};
// The frontier of the raise wave / the fired cells of a level live in the particle's (otherwise idle) raise-queue
// region in HBM, split in two halves fa / fb of qcap/2 entries: low word = loc (ry << 16 | rx), high word = the four
// 2-bit raise decisions, resp. the obstacle offset of the fired cell.

__device__ inline void cn_table_clear(CanonLds& sh)
{
    for (int i = threadIdx.x; i < CN_TBL; i += CN_BLOCK) { sh.tkey[i] = CN_EMPTY; sh.tval[i] = CN_EMPTY; }
}
// slot of `key` (inserted if absent); -1 when the table is full
__device__ inline int cn_slot(CanonLds& sh, uint32_t key)
{
    uint32_t h = (key * 2654435761u) >> (32 - CN_TBL_LOG2);
    for (int probe = 0; probe < CN_TBL; ++probe) {
        const uint32_t cur = sh.tkey[h];
        if (cur == key) return (int)h;
        if (cur == CN_EMPTY) {
            const uint32_t old = atomicCAS(&sh.tkey[h], CN_EMPTY, key);
            if (old == CN_EMPTY || old == key) return (int)h;
        }
        h = (h + 1) & (CN_TBL - 1);
    }
    return -1;
}

struct CnMap {
    const DevParams& prm;
    int p;
    int16_t* dir; sv_t* sv; uint32_t* obs; uint64_t* mask; int cap;
    // side-effect free cell index (slot*1024 + ci) or -1 when the patch does not exist / outside the window
    __device__ inline int peek(int x, int y) const
    {
        if ((uint32_t)x >= prm.WC || (uint32_t)y >= prm.WC) return -1;
        const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);
        int slot = dir[pidx];
        if (slot < 0) {      // may be a stale L1 line: patches are allocated with L2 atomics (dir_get_or_alloc)
            const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(dir) + (pidx >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            slot = (int)(int16_t)((w >> ((pidx & 1u) * 16)) & 0xFFFFu);
        }
        if (slot < 0) return -1;
        return slot * 1024 + (int)(((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5));
    }
    // non-const Map::get (src/sdm/map.cpp:371-412): allocate the patch, set the mask bit
    __device__ inline int get(int x, int y) const
    {
        if ((uint32_t)x >= prm.WC || (uint32_t)y >= prm.WC) { atomicOr(prm.err, ERR_WINDOW); return -1; }
        const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);
        const int slot = dir_get_or_alloc(dir, pidx, prm.counts + 2 * p, cap, ERR_DM_CAP, prm.err);
        if (slot < 0) return -1;
        const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
        atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), 1ull << (ci & 63));
        return slot * 1024 + (int)ci;
    }
};

__global__ __launch_bounds__(CN_BLOCK) void k_brushfire_canon(DevParams prm, int first_particle)
{
    __shared__ CanonLds sh;
    const int p = first_particle + blockIdx.x;
    const int tid = threadIdx.x;
    if (map_update_aborted(prm)) return;
    const PV pv = pview(prm, p);
    const CnMap M{prm, p, pv.dm_dir, pv.dm_sv, pv.dm_obs, pv.dm_mask, (int)pv.dm_cap};
    uint64_t* list = prm.q_lower + (size_t)p * prm.qcap;          // pending lower entries, appended in place
    uint64_t* fa = prm.q_raise + (size_t)p * prm.qcap;
    const uint32_t FCAP = prm.qcap / 2;
    uint64_t* fb = fa + FCAP;
    const uint32_t nl0 = prm.qsizes[2 * p], nr0 = prm.qsizes[2 * p + 1];
    if (nl0 == 0 && nr0 == 0) return;
    // what this kernel cannot hold is left, untouched, to the exact kernels that are launched after it
    if (prm.max_sqdist + 1 > (uint32_t)CN_HIST || nr0 > FCAP) return;
    const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};

    for (int i = tid; i < CN_HIST; i += CN_BLOCK) sh.hist[i] = 0;
    for (uint32_t i = tid; i < nr0; i += CN_BLOCK) fa[i] = fa[i] & 0xFFFFFFFFull;      // q_entry -> loc
    if (tid == 0) { sh.n_list = nl0; sh.n_a = nr0; sh.n_b = 0; sh.fail = 0; }
    __syncthreads();
    if (tid == 0) sh.hist[0] = nl0;
    uint32_t processed = 0;

    // ------------------------------------------------------------------ raise wave, breadth-first rounds
    while (sh.n_a > 0) {
        const uint32_t nf = sh.n_a;
        cn_table_clear(sh);
        __syncthreads();
        // A1: decisions on the state at the start of the round (src/sdm/dynamic_distance_map.cpp:244-279)
        for (uint32_t f = tid; f < nf; f += CN_BLOCK) {
            const uint32_t loc = (uint32_t)fa[f];
            const int rx = (int)(loc & 0xFFFFu), ry = (int)(loc >> 16);
            ++processed;
            uint32_t decs = 0;
            for (int i = 0; i < 4; ++i) {
                uint32_t d = 0;
                const int nx = rx + DX[i], ny = ry + DY[i];
                const int nc = M.get(nx, ny);
                if (nc >= 0) {
                    const sv_t s = M.sv[nc];
                    if (!(s & SV_QUEUED) && (s & SV_VALID)) {
                        const uint32_t o = M.obs[nc];
                        const int oc = M.peek(nx + obs_x(o), ny + obs_y(o));
                        const bool ovalid = oc >= 0 && (M.sv[oc] & SV_VALID);
                        d = ovalid ? 2 : 1;
                    }
                }
                decs |= d << (2 * i);
            }
            fa[f] = (uint64_t)loc | ((uint64_t)decs << 32);
        }
        __syncthreads();
        // A2: apply, each neighbour once
        for (uint32_t f = tid; f < nf; f += CN_BLOCK) {
            const uint64_t fe = fa[f];
            const int rx = (int)(fe & 0xFFFFu), ry = (int)((fe >> 16) & 0xFFFFu);
            for (int i = 0; i < 4; ++i) {
                const uint32_t d = (uint32_t)(fe >> (32 + 2 * i)) & 3u;
                if (!d) continue;
                const int nx = rx + DX[i], ny = ry + DY[i];
                const uint32_t nloc = ((uint32_t)ny << 16) | (uint32_t)nx;
                const int ts = cn_slot(sh, nloc);
                if (ts < 0) { sh.fail = 1; continue; }
                if (atomicExch(&sh.tval[ts], 0u) != CN_EMPTY) continue;          // another frontier cell owns it
                const int nc = M.peek(nx, ny);
                const sv_t s = M.sv[nc];
                if (d == 1) {                                                    // its obstacle is gone: clear, raise further
                    M.sv[nc] = SV_QUEUED; M.obs[nc] = 0;
                    const uint32_t k = atomicAdd(&sh.n_b, 1u);
                    if (k < FCAP) fb[k] = nloc; else sh.fail = 1;
                } else {                                                         // still has a live obstacle: re-lower from it
                    M.sv[nc] = (sv_t)(s | SV_QUEUED);
                    const uint32_t k = atomicAdd(&sh.n_list, 1u);
                    const uint32_t o = M.obs[nc];
                    if (k < prm.qcap) list[k] = q_entry(s & SV_SQMASK, nx, ny, obs_x(o), obs_y(o)); else sh.fail = 1;
                    atomicAdd(&sh.hist[s & SV_SQMASK], 1u);
                }
            }
            const int cc = M.peek(rx, ry);
            if (cc >= 0) M.sv[cc] = (sv_t)(M.sv[cc] & ~SV_QUEUED);            // :278
        }
        __syncthreads();
        const uint32_t nn = sh.n_b < FCAP ? sh.n_b : FCAP;
        for (uint32_t i = tid; i < nn; i += CN_BLOCK) fa[i] = fb[i];
        __syncthreads();
        if (tid == 0) { sh.n_a = nn; sh.n_b = 0; }
        __syncthreads();
        if (sh.fail) break;
    }

    // ------------------------------------------------------------------ lower wave, one priority level at a time
    for (uint32_t lev = 0; lev < prm.max_sqdist && !sh.fail; ++lev) {
        if (sh.hist[lev] == 0) continue;                      // uniform: hist is only modified between barriers
        const uint32_t n_list = sh.n_list;
        cn_table_clear(sh);
        if (tid == 0) sh.n_a = 0;
        __syncthreads();
        // A: cells queued at this level that still satisfy update()'s conditions (:183-192, :283), each once
        for (uint32_t i = tid; i < n_list; i += CN_BLOCK) {
            const uint64_t e = list[i];
            if (heap_prio(e) != lev) continue;
            const int rx = q_rx(e), ry = q_ry(e);
            const uint32_t loc = ((uint32_t)ry << 16) | (uint32_t)rx;
            const int ts = cn_slot(sh, loc);
            if (ts < 0) { sh.fail = 1; continue; }
            if (atomicExch(&sh.tval[ts], 0u) != CN_EMPTY) continue;              // duplicate entry of the same cell
            ++processed;
            const int cc = M.peek(rx, ry);
            if (cc < 0) continue;
            const sv_t s = M.sv[cc];
            if (!(s & SV_VALID) || !(s & SV_QUEUED)) continue;
            const uint32_t o = M.obs[cc];
            const int oc = M.peek(rx + obs_x(o), ry + obs_y(o));
            if (oc < 0 || (M.sv[oc] & SV_SQMASK) != 0) continue;                 // :191 (valid NOT tested)
            const uint32_t k = atomicAdd(&sh.n_a, 1u);
            if (k < FCAP) fa[k] = (uint64_t)loc | ((uint64_t)o << 32); else sh.fail = 1;
            M.sv[cc] = (sv_t)(s & ~SV_QUEUED);                               // :329 (nothing reads it inside this level)
        }
        __syncthreads();
        if (tid == 0) sh.hist[lev] = 0;
        const uint32_t nf = sh.n_a < FCAP ? sh.n_a : FCAP;
        const uint32_t passes = (nf * 4u + (CN_TBL / 2 - 1)) / (CN_TBL / 2);
        for (uint32_t pass = 0; pass < passes && !sh.fail; ++pass) {
            cn_table_clear(sh);
            __syncthreads();
            // B: offers (only towards cells of this pass' spatial class, so that all offers to one cell meet)
            for (uint32_t f = tid; f < nf; f += CN_BLOCK) {
                const uint64_t fe = fa[f];
                const int rx = (int)(fe & 0xFFFFu), ry = (int)((fe >> 16) & 0xFFFFu);
                const int cox = obs_x((uint32_t)(fe >> 32)), coy = obs_y((uint32_t)(fe >> 32));
                for (int i = 0; i < 4; ++i) {
                    if (DX[i] * cox > 0 || DY[i] * coy > 0) continue;            // only away from the obstacle (:296)
                    const int nx = rx + DX[i], ny = ry + DY[i];
                    if (passes > 1 && (uint32_t)((nx >> 3) + 5 * (ny >> 3)) % passes != pass) continue;
                    if (M.get(nx, ny) < 0) continue;
                    const int qx = nx - (rx + cox), qy = ny - (ry + coy);
                    const uint32_t cand = (uint32_t)(qx * qx + qy * qy);
                    const int ts = cn_slot(sh, ((uint32_t)ny << 16) | (uint32_t)nx);
                    if (ts < 0) { sh.fail = 1; continue; }
                    atomicMin(&sh.tval[ts], (cand << 2) | (uint32_t)i);
                }
            }
            __syncthreads();
            // C: the winning offer of every neighbour applies the overwrite rule (:300-326)
            for (uint32_t f = tid; f < nf; f += CN_BLOCK) {
                const uint64_t fe = fa[f];
                const int rx = (int)(fe & 0xFFFFu), ry = (int)((fe >> 16) & 0xFFFFu);
                const int cox = obs_x((uint32_t)(fe >> 32)), coy = obs_y((uint32_t)(fe >> 32));
                const int obx = rx + cox, oby = ry + coy;
                for (int i = 0; i < 4; ++i) {
                    if (DX[i] * cox > 0 || DY[i] * coy > 0) continue;
                    const int nx = rx + DX[i], ny = ry + DY[i];
                    if (passes > 1 && (uint32_t)((nx >> 3) + 5 * (ny >> 3)) % passes != pass) continue;
                    const int nc = M.peek(nx, ny);
                    if (nc < 0) continue;
                    const int qx = nx - obx, qy = ny - oby;
                    const uint32_t cand = (uint32_t)(qx * qx + qy * qy);
                    const int ts = cn_slot(sh, ((uint32_t)ny << 16) | (uint32_t)nx);
                    if (ts < 0 || sh.tval[ts] != ((cand << 2) | (uint32_t)i)) continue;
                    const sv_t ns = M.sv[nc];
                    const uint32_t cmp = (ns & SV_VALID) ? (uint32_t)(ns & SV_SQMASK) : prm.max_sqdist;
                    bool over = cand < cmp;
                    if (!over && cand == (uint32_t)(ns & SV_SQMASK)) {
                        const uint32_t no = M.obs[nc];
                        const int oc = M.get(nx + obs_x(no), ny + obs_y(no));
                        const sv_t os = oc >= 0 ? M.sv[oc] : (sv_t)0;
                        if (!(ns & SV_VALID) || !((os & SV_VALID) && (os & SV_SQMASK) == 0)) over = true;
                    }
                    if (!over) continue;
                    M.sv[nc] = (sv_t)(SV_VALID | SV_QUEUED | (cand & SV_SQMASK));
                    M.obs[nc] = pack_obs(obx - nx, oby - ny);
                    const uint32_t k = atomicAdd(&sh.n_list, 1u);
                    if (k < prm.qcap) list[k] = q_entry(cand, nx, ny, obx - nx, oby - ny); else sh.fail = 1;
                    if (cand < (uint32_t)CN_HIST) atomicAdd(&sh.hist[cand], 1u);
                }
            }
            __syncthreads();
        }
    }
    // processed cells of this particle (not comparable with the reference's count: duplicates are not re-popped)
    atomicAdd((unsigned long long*)(prm.stats + 4 * p + 3), (unsigned long long)processed);
    if (tid == 0) {
        if (sh.fail) atomicOr(prm.err, ERR_QUEUE);
        prm.qsizes[2 * p] = 0;
        prm.qsizes[2 * p + 1] = 0;
    }
}

} // namespace lama_dev
