// lama_raycast_patch.h -- the free-cell visits of PFSlam2D::updateParticleMaps (src/pf_slam2d.cpp:495-505: computeRay +
// FrequencyOccupancyMap::setFree, src/sdm/map.cpp:198-227, src/sdm/frequency_occupancy_map.cpp:65-74) gathered PER OCCUPANCY
// PATCH instead of scattered per beam: the variant of k_ray_visits (lama_raycast_par.h) without global atomics.
//
// k_ray_visits walks the beams and adds every visit to its cell with a device-scope atomic (aggregated per workgroup in an LDS
// hash table first): 3.6x the algorithmic traffic and bound by L2 atomic throughput once the chip is full of particles.  Here a
// workgroup OWNS one 32 x 32 patch of one particle for the whole launch:
//   1. the patch (4 KB of `occupied | visited << 16` cells + its hit bits) is read once, coalesced, and classified: a cell is
//      ACTIVE when it was hit in this scan or is not already free / brand new -- exactly k_ray_visits' rule;
//   2. every beam of the scan is tested against the patch: first the bounding box of its ray cells (8 B per beam), then the exact
//      range [t_lo, t_hi] of Bresenham steps whose cell lies inside the patch -- the closed form of Map::computeRay,
//      steps_j(t) = floor((2 t |d_j| + n) / (2 n)), is monotone in t, so the range follows from two integer divisions per axis;
//   3. the cells of those ranges are walked (a half-wave per beam: a ray crosses at most 33 cells of a patch): visits of
//      inert cells are counted in an LDS array of 1024 counters, visits of active cells are appended to the particle's active
//      list with their (beam, step) for k_ray_replay -- unchanged;
//   4. the counters are added to the cells the workgroup still holds in registers and the patch is written back once, coalesced.
//      The first miss of a brand-new cell is removeObstacle() on a cell that cannot be an obstacle = get() on the distance map
//      (patch allocation + Container mask bit, src/sdm/dynamic_distance_map.cpp:228-234): one OR per mask word.
// The uint16 counters commute, so the result is the one k_ray_visits (and the reference's beam-by-beam loop) produces; the host
// still routes scans in which a counter could wrap to the beam-sequential kernel.
//
// k_ray_hits (lama_raycast_par.h) provides what this needs: it stores every beam's ray record and bounding box and walks the ray
// once at patch granularity to allocate the occupancy patches it crosses (the lock-free directory CAS only ever runs for a
// missing patch); k_occ_reverse_dir then lists the particle's patches (arena slot -> directory position).
#pragma once
#include "lama_raycast_par.h"

namespace lama_dev {

// ray record of one (particle, beam): start cell in map coordinates, |delta| of the x / y axes, n = max |delta| over the three
// axes, direction signs, the magic of the closed form.  valid = the ray lies inside the window and has cells to visit.
struct RayRec {
    uint32_t msx, msy;
    uint32_t a01;            // a0 | a1 << 16
    uint32_t nnf;            // nn | (s0 < 0) << 16 | (s1 < 0) << 17 | valid << 18
    uint64_t magic;
};
constexpr uint64_t RAY_BBOX_EMPTY = 0x0000FFFF0000FFFFull;   // x0 = y0 = 0xFFFF > x1 = y1 = 0

__device__ inline RayRec ray_rec(const BeamGeom& g)
{
    RayRec r;
    r.msx = g.msx; r.msy = g.msy;
    r.a01 = (g.a0 & 0xFFFFu) | ((g.a1 & 0xFFFFu) << 16);
    r.nnf = (g.nn & 0xFFFFu) | (g.s0 < 0 ? 1u << 16 : 0u) | (g.s1 < 0 ? 1u << 17 : 0u) | (g.steps > 0 ? 1u << 18 : 0u);
    r.magic = g.magic;
    return r;
}
// window-relative cell of step t
__device__ __forceinline__ void ray_cell(const RayRec& r, uint32_t wx0, uint32_t wy0, uint32_t t, uint32_t& rx, uint32_t& ry)
{
    const uint32_t a0 = r.a01 & 0xFFFFu, a1 = r.a01 >> 16, nn = r.nnf & 0xFFFFu;
    const uint32_t st0 = (uint32_t)(((uint64_t)(2u * t * a0 + nn) * r.magic) >> 42);
    const uint32_t st1 = (uint32_t)(((uint64_t)(2u * t * a1 + nn) * r.magic) >> 42);
    const uint32_t cx = (r.nnf & (1u << 16)) ? r.msx - st0 : r.msx + st0;
    const uint32_t cy = (r.nnf & (1u << 17)) ? r.msy - st1 : r.msy + st1;
    rx = cx - wx0; ry = cy - wy0;
}

// steps t in [1, steps] whose coordinate m + s * floor((2 t a + nn) / (2 nn)) lies in [lo, lo + 31] (all window-relative); returns
// false when there is none
__device__ __forceinline__ bool ray_axis_range(int m, bool neg, uint32_t a, uint32_t nn, int lo, uint32_t steps, uint32_t& t_lo, uint32_t& t_hi)
{
    int ka = neg ? m - (lo + 31) : lo - m, kb = neg ? m - lo : lo + 31 - m;        // k = number of steps the axis has made
    if (ka < 0) ka = 0;
    if (kb > (int)a) kb = (int)a;
    if (ka > kb) return false;
    t_lo = 1u; t_hi = steps;
    if (a == 0u) return true;                                                       // the axis never moves
    const uint32_t d = 2u * a;
    if (ka > 0) t_lo = (2u * nn * (uint32_t)ka - nn + d - 1u) / d;                  // first t with floor(..) >= ka
    if (kb < (int)a) t_hi = (2u * nn * (uint32_t)(kb + 1) - nn + d - 1u) / d - 1u;  // last t with floor(..) <= kb
    if (t_lo < 1u) t_lo = 1u;
    if (t_hi > steps) t_hi = steps;
    return t_lo <= t_hi;
}

// what k_ray_hits stores for the patch pass (see k_ray_hits)
__device__ inline void ray_hits_record(const DevParams& prm, const BeamGeom& g, int p, int i, int n, RayRec* rec_out, uint64_t* bbox_out)
{
    RayRec r = ray_rec(g);
    uint64_t bb = RAY_BBOX_EMPTY;
    bool ok = g.steps > 0;
    if (ok) {
        // cells visited: t = 1 .. steps, every coordinate between the start cell and start + s * a (window-relative)
        const int64_t xa = (int64_t)g.msx - (int64_t)prm.wx0, ya = (int64_t)g.msy - (int64_t)prm.wy0;
        const int64_t xb = xa + (int64_t)g.s0 * (int64_t)g.a0, yb = ya + (int64_t)g.s1 * (int64_t)g.a1;
        const int64_t x0 = xa < xb ? xa : xb, x1 = xa < xb ? xb : xa, y0 = ya < yb ? ya : yb, y1 = ya < yb ? yb : ya;
        if (x0 < 0 || y0 < 0 || x1 >= (int64_t)prm.WC || y1 >= (int64_t)prm.WC) { atomicOr(prm.err, ERR_WINDOW); ok = false; }
        else bb = (uint64_t)x0 | ((uint64_t)x1 << 16) | ((uint64_t)y0 << 32) | ((uint64_t)y1 << 48);
    }
    if (!ok) r.nnf &= ~(1u << 18);
    rec_out[(size_t)p * n + i] = r;
    bbox_out[(size_t)p * n + i] = bb;
}

// allocation walk: afterwards the occupancy patch of every ray cell of the scan exists.  RW_SEG threads per beam (many while the
// chip is not full: short chains; few when it is: fewer threads to launch), each walks one stretch of the ray (one directory read
// per patch change; the lock-free CAS only runs for a missing patch).
__global__ __launch_bounds__(256) void k_ray_alloc_walk(DevParams prm, const RayRec* __restrict__ recs, int n, int first_particle, int RW_SEG)
{
    const int p = first_particle + blockIdx.x;
    const int g = blockIdx.y * 256 + threadIdx.x;
    const int i = g / RW_SEG, seg = g % RW_SEG;
    if (i >= n) return;
    const RayRec r = recs[(size_t)p * n + i];
    if (!(r.nnf & (1u << 18))) return;
    const uint32_t steps = (r.nnf & 0xFFFFu) - 1u;
    const uint32_t t0 = 1u + (uint32_t)(((uint64_t)steps * (uint32_t)seg) / RW_SEG), t1 = (uint32_t)(((uint64_t)steps * (uint32_t)(seg + 1)) / RW_SEG);
    const size_t WW = (size_t)prm.W * prm.W;
    int16_t* occ_dir = prm.occ_dir + (size_t)p * WW;
    if (t0 > t1) return;
    // Map::computeRay (src/sdm/map.cpp:198-227) in its incremental form from step t0 on: k_j = floor((2 t a_j + n) / (2 n)) steps
    // made by axis j, rem_j the remainder; every step adds 2 a_j and carries at 2 n
    const uint32_t a0 = r.a01 & 0xFFFFu, a1 = r.a01 >> 16, nn = r.nnf & 0xFFFFu, n2 = 2u * nn;
    uint32_t k0 = (uint32_t)(((uint64_t)(2u * t0 * a0 + nn) * r.magic) >> 42), k1 = (uint32_t)(((uint64_t)(2u * t0 * a1 + nn) * r.magic) >> 42);
    uint32_t rem0 = 2u * t0 * a0 + nn - k0 * n2, rem1 = 2u * t0 * a1 + nn - k1 * n2;
    const bool neg0 = (r.nnf >> 16) & 1u, neg1 = (r.nnf >> 17) & 1u;
    const uint32_t bx = r.msx - prm.wx0, by = r.msy - prm.wy0;
    uint32_t last = 0xFFFFFFFFu;
    for (uint32_t t = t0; t <= t1; ++t) {
        const uint32_t rx = neg0 ? bx - k0 : bx + k0, ry = neg1 ? by - k1 : by + k1;
        const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5);
        if (pidx != last) { (void)dir_get_or_alloc(occ_dir, pidx, prm.counts + 2 * p + 1, (int)prm.occ_cap, ERR_OCC_CAP, prm.err); last = pidx; }
        rem0 += 2u * a0; if (rem0 >= n2) { rem0 -= n2; ++k0; }
        rem1 += 2u * a1; if (rem1 >= n2) { rem1 -= n2; ++k1; }
    }
}

// arena slot -> directory position of every occupancy patch of the particle (rebuilt per scan: the directories change with
// allocation, resampling, window shifts and patch deletion; 32 KB of directory per particle).
//
// The same pass closes the ALLOCATION phase of the update: every occupancy patch the scan touches exists by now (k_ray_hits,
// k_ray_alloc_walk), and every distance-map patch the update can allocate -- first misses and hits in those patches, the
// brushfire's neighbours at most sqrt(max_sqdist) + 1 cells from a changed obstacle -- lies within `guard_r` patches of an
// occupancy patch.  The number of window positions without a distance-map patch but with an occupancy patch that close bounds what
// is still to come; the last workgroup of the particle compares it with the free slots and raises ERR_DM_CAP BEFORE any map cell
// is modified, so that the host can grow the arena and run the update again.
constexpr int RD_MARK_WORDS = (248 * 248 + 31) / 32;       // one bit per window position (window_patches <= 248)

__global__ __launch_bounds__(256) void k_occ_reverse_dir(DevParams prm, int32_t* __restrict__ rev, int first_particle)
{
    __shared__ uint32_t mark[RD_MARK_WORDS];                  // window positions within guard_r patches of an occupancy patch
    __shared__ uint32_t need_s;
    const int p = first_particle + blockIdx.x;
    const uint32_t W = prm.W, WW = W * W;
    const int tid = threadIdx.x, r = (int)prm.guard_r;
    const int16_t* occ_dir = prm.occ_dir + (size_t)p * WW;
    const int16_t* dm_dir = prm.dm_dir + (size_t)p * WW;
    for (uint32_t i = tid; i < (WW + 31u) / 32u; i += 256u) mark[i] = 0;
    if (tid == 0) need_s = 0;
    __syncthreads();
    // eight directory entries per thread and round (W is a multiple of 8: a run never leaves its row, 16-byte aligned)
    for (uint32_t w0 = (uint32_t)tid * 8u; w0 < WW; w0 += 256u * 8u) {
        const uint4 q = *reinterpret_cast<const uint4*>(occ_dir + w0);
        if ((q.x & q.y & q.z & q.w) == 0xFFFFFFFFu) continue;                 // eight absent patches (-1): the common case
        const uint32_t ww[4] = {q.x, q.y, q.z, q.w};
        const int wy = (int)(w0 / W), x0 = (int)(w0 % W);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int slot = (int)(int16_t)((ww[k >> 1] >> ((k & 1) * 16)) & 0xFFFFu);
            if (slot < 0) continue;
            rev[(size_t)p * prm.occ_cap + slot] = (int32_t)(w0 + (uint32_t)k);
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    const int x = x0 + k + dx, y = wy + dy;
                    if ((uint32_t)x < W && (uint32_t)y < W) { const uint32_t n = (uint32_t)y * W + (uint32_t)x; atomicOr(&mark[n >> 5], 1u << (n & 31u)); }
                }
        }
    }
    __syncthreads();
    uint32_t need = 0;
    for (uint32_t i = tid; i < (WW + 31u) / 32u; i += 256u) {
        uint32_t m = mark[i];
        while (m) {
            const uint32_t n = i * 32u + (uint32_t)(__ffs((int)m) - 1);
            if (dm_dir[n] < 0) ++need;
            m &= m - 1u;
        }
    }
    if (need) atomicAdd(&need_s, need);
    __syncthreads();
    if (tid == 0 && (uint64_t)prm.counts[2 * p] + need_s > prm.dm_cap) atomicOr(prm.err, ERR_DM_CAP);
}

constexpr int RPT_CHUNK = 512;        // beams tested per round: the records of those that cross the patch wait in LDS

__global__ __launch_bounds__(256) void k_ray_patches(DevParams prm, const RayRec* __restrict__ recs, const uint64_t* __restrict__ bbox,
                                                      const int32_t* __restrict__ rev, int n, int first_particle)
{
    __shared__ uint32_t cnt[1024];
    __shared__ RayRec lrec[RPT_CHUNK];           // beams of the round that cross the patch ...
    __shared__ uint32_t lbt[RPT_CHUNK];          // ... their step range t_lo | t_hi << 16 ...
    __shared__ uint16_t lbeam[RPT_CHUNK];        // ... and beam index
    __shared__ uint32_t actw[32], newm[32], wrapm[32];
    __shared__ uint32_t list_n;
    const int p = first_particle + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first map-modifying kernel of the update: nothing is touched when the allocation phase failed (the host grows and retries)
    if (map_update_aborted(prm)) { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicOr(prm.err, ERR_CLEAN_ABORT); return; }
    const int count = prm.counts[2 * p + 1];
    const size_t WW = (size_t)prm.W * prm.W;
    uint32_t* occ = prm.occ + (size_t)p * prm.occ_cap * 1024;
    const RayRec* prec = recs + (size_t)p * n;
    const uint64_t* pbb = bbox + (size_t)p * n;
    for (int slot = blockIdx.y; slot < count; slot += gridDim.y) {
        const uint32_t pidx = (uint32_t)rev[(size_t)p * prm.occ_cap + slot];
        const int px = (int)((pidx % prm.W) * 32u), py = (int)((pidx / prm.W) * 32u);       // window-relative origin of the patch
        // ---- 1. the patch and the classification of its cells
        uint32_t v[4];
        const uint64_t* hitw = prm.occ_hit + ((size_t)p * prm.occ_cap + slot) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = tid + 256 * j;
            v[j] = occ[(size_t)slot * 1024 + ci];
            cnt[ci] = 0;
            const uint32_t o0 = v[j] & 0xFFFFu, v0 = v[j] >> 16;
            const bool hit = (hitw[ci >> 6] >> (ci & 63)) & 1ull;
            const bool active = hit || !(v0 == 0 ? o0 == 0 : 4u * o0 < v0);
            const unsigned long long am = __ballot(active);                // cells 256 j + 64 wave .. + 63
            if (lane == 0) { actw[8 * j + 2 * wave] = (uint32_t)am; actw[8 * j + 2 * wave + 1] = (uint32_t)(am >> 32); }
        }
        if (tid < 32) { newm[tid] = 0; wrapm[tid] = 0; }
        for (int b0 = 0; b0 < n; b0 += RPT_CHUNK) {
            if (tid == 0) list_n = 0;
            __syncthreads();
            // ---- 2. which beams of the round cross the patch, and in which steps
            for (int b = b0 + tid; b < n && b < b0 + RPT_CHUNK; b += 256) {
                const uint64_t bb = pbb[b];
                const int x0 = (int)(bb & 0xFFFFu), x1 = (int)((bb >> 16) & 0xFFFFu), y0 = (int)((bb >> 32) & 0xFFFFu), y1 = (int)(bb >> 48);
                if (x1 < px || x0 > px + 31 || y1 < py || y0 > py + 31) continue;      // (an invalid ray has an empty box)
                const RayRec r = prec[b];
                const uint32_t nn = r.nnf & 0xFFFFu, steps = nn - 1u;
                {   // Every ray cell lies within half a cell of the segment start -> start + (s0 a0, s1 a1) (each axis is the rounded
                    // position t a / n): a ray whose segment keeps the patch, grown by one cell, strictly on one side cannot touch it
                    const int sx = (int)(r.msx - prm.wx0), sy = (int)(r.msy - prm.wy0);
                    const int dx = ((r.nnf >> 16) & 1u) ? -(int)(r.a01 & 0xFFFFu) : (int)(r.a01 & 0xFFFFu);
                    const int dy = ((r.nnf >> 17) & 1u) ? -(int)(r.a01 >> 16) : (int)(r.a01 >> 16);
                    const int ax = px - 1 - sx, bx = px + 32 - sx, ay = py - 1 - sy, by = py + 32 - sy;
                    const int c00 = dx * ay - dy * ax, c10 = dx * ay - dy * bx, c01 = dx * by - dy * ax, c11 = dx * by - dy * bx;
                    if ((c00 > 0 && c10 > 0 && c01 > 0 && c11 > 0) || (c00 < 0 && c10 < 0 && c01 < 0 && c11 < 0)) continue;
                }
                uint32_t xl, xh, yl, yh;
                if (!ray_axis_range((int)(r.msx - prm.wx0), (r.nnf >> 16) & 1u, r.a01 & 0xFFFFu, nn, px, steps, xl, xh)) continue;
                if (!ray_axis_range((int)(r.msy - prm.wy0), (r.nnf >> 17) & 1u, r.a01 >> 16, nn, py, steps, yl, yh)) continue;
                const uint32_t tl = xl > yl ? xl : yl, th = xh < yh ? xh : yh;
                if (tl > th) continue;
                const uint32_t k = atomicAdd(&list_n, 1u);
                lrec[k] = r; lbt[k] = tl | (th << 16); lbeam[k] = (uint16_t)b;
            }
            __syncthreads();
            // ---- 3. walk the ranges: eight lanes per beam (a ray crosses at most 33 cells of a patch, a dozen on average)
            const uint32_t ln = list_n;
            const int hw = tid >> 3, hl = tid & 7;
            for (uint32_t e = (uint32_t)hw; e < ln; e += 32u) {
                const RayRec r = lrec[e];
                const uint32_t b = lbeam[e], tl = lbt[e] & 0xFFFFu, th = lbt[e] >> 16;
                for (uint32_t t = tl + (uint32_t)hl; t <= th; t += 8u) {
                    uint32_t rx, ry;
                    ray_cell(r, prm.wx0, prm.wy0, t, rx, ry);
                    if ((int)(rx & ~31u) != px || (int)(ry & ~31u) != py) continue;     // cannot happen: the range is exact
                    const uint32_t ci = (rx & 31u) | ((ry & 31u) << 5);
                    if ((actw[ci >> 5] >> (ci & 31u)) & 1u) act_append(prm, p, act_key(rx, ry, b, t));
                    else atomicAdd(&cnt[ci], 1u);
                }
            }
            __syncthreads();
        }
        // ---- 4. visited += count, one coalesced write of the cells that changed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ci = tid + 256 * j;
            const uint32_t c = cnt[ci];
            if (c) {
                const uint32_t o0 = v[j] & 0xFFFFu, v0 = v[j] >> 16, nv = v0 + c;
                if (v0 == 0) atomicOr(&newm[ci >> 5], 1u << (ci & 31));            // first miss of a brand-new cell
                if (nv > 0xFFFFu) atomicOr(&wrapm[ci >> 5], 1u << (ci & 31));      // uint16 wrap: keep the Container mask bit
                occ[(size_t)slot * 1024 + ci] = o0 | ((nv & 0xFFFFu) << 16);
            }
        }
        __syncthreads();
        if (tid < 16) {
            const uint64_t nm = (uint64_t)newm[2 * tid] | ((uint64_t)newm[2 * tid + 1] << 32);
            const uint64_t wm = (uint64_t)wrapm[2 * tid] | ((uint64_t)wrapm[2 * tid + 1] << 32);
            if (nm) {   // removeObstacle on a cell that cannot be an obstacle = get(): patch allocation + mask bit (:228-234)
                const int ds = dir_get_or_alloc(prm.dm_dir + (size_t)p * WW, pidx, prm.counts + 2 * p, (int)prm.dm_cap, ERR_DM_CAP, prm.err);
                if (ds >= 0) atomicOr((unsigned long long*)(prm.dm_mask + ((size_t)p * prm.dm_cap + ds) * 16 + tid), (unsigned long long)nm);
            }
            if (wm) atomicOr((unsigned long long*)(prm.occ_mask + ((size_t)p * prm.occ_cap + slot) * 16 + tid), (unsigned long long)wm);
        }
        __syncthreads();
    }
}

} // namespace lama_dev
