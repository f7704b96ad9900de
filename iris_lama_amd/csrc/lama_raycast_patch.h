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
// 64 consecutive beams of one particle (one wave of k_ray_hits): what k_ray_patches needs to skip all of them at once.  All rays of
// a scan start in the sensor's cell, so the beams of a chunk fan out inside a cone; when the fan is narrower than 180 degrees (a
// LIDAR's beams come in angular order: 64 of them span a few degrees) the chunk stores its two bounding directions.  A ray's cells
// lie within half a cell of the segment start -> start + direction, so a patch that -- grown by one cell -- lies strictly clockwise
// of the clockwise bound (or counter-clockwise of the other one) is not touched by any ray of the chunk.  Nothing depends on the
// order of the points: a chunk without such a cone (cone = 0) is simply tested beam by beam.
struct RayChunk {
    int32_t cwx, cwy;        // direction (cells) of the clockwise-most ray ...
    int32_t ccwx, ccwy;      // ... and of the counter-clockwise-most one
    uint64_t bbox;           // union of the rays' boxes (RAY_BBOX_EMPTY: no valid ray in the chunk)
    uint32_t sxy;            // the common start cell, window-relative: x | y << 16
    uint32_t cone;           // 1 = every valid ray of the chunk lies between the two directions and starts in (sx, sy)
};

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

// what k_ray_hits stores for the patch pass (see k_ray_hits); returns the box of the ray's cells (RAY_BBOX_EMPTY: nothing to visit)
__device__ inline uint64_t ray_hits_record(const DevParams& prm, const BeamGeom& g, int p, int i, int n, RayRec* rec_out, uint64_t* bbox_out)
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
    return bb;
}

// the chunk record of the calling wave (64 consecutive beams; every lane of the wave calls this, `bb` = its ray's box or
// RAY_BBOX_EMPTY for a lane without a ray)
__device__ inline void ray_chunk_record(const DevParams& prm, const BeamGeom& g, uint64_t bb, int lane, RayChunk* chunks, size_t index)
{
    const bool val = bb != RAY_BBOX_EMPTY;
    const unsigned long long vm = __ballot(val);
    uint32_t x0 = (uint32_t)(bb & 0xFFFFu), x1 = (uint32_t)((bb >> 16) & 0xFFFFu), y0 = (uint32_t)((bb >> 32) & 0xFFFFu), y1 = (uint32_t)(bb >> 48);
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t ox0 = (uint32_t)__shfl_xor((int)x0, off, 64), ox1 = (uint32_t)__shfl_xor((int)x1, off, 64);
        const uint32_t oy0 = (uint32_t)__shfl_xor((int)y0, off, 64), oy1 = (uint32_t)__shfl_xor((int)y1, off, 64);
        x0 = ox0 < x0 ? ox0 : x0; x1 = ox1 > x1 ? ox1 : x1; y0 = oy0 < y0 ? oy0 : y0; y1 = oy1 > y1 ? oy1 : y1;
    }
    const int lf = vm ? __ffsll((long long)vm) - 1 : 0, ll = vm ? 63 - __clzll((long long)vm) : 0;
    const int dx = g.s0 < 0 ? -(int)g.a0 : (int)g.a0, dy = g.s1 < 0 ? -(int)g.a1 : (int)g.a1;       // |d| < 2^13: products fit 32 bits
    const int sx = (int)(g.msx - prm.wx0), sy = (int)(g.msy - prm.wy0);
    const int fx = __shfl(dx, lf, 64), fy = __shfl(dy, lf, 64), lx = __shfl(dx, ll, 64), ly = __shfl(dy, ll, 64);
    const int fsx = __shfl(sx, lf, 64), fsy = __shfl(sy, lf, 64);
    const int c_f = fx * dy - fy * dx, c_l = dx * ly - dy * lx;          // cross(first, d), cross(d, last)
    const bool front = fx * dx + fy * dy > 0 && lx * dx + ly * dy > 0 && sx == fsx && sy == fsy;
    const bool all_ccw = __ballot(val && !(front && c_f >= 0 && c_l >= 0)) == 0ull;    // first = clockwise end, last = counter-clockwise end
    const bool all_cw = __ballot(val && !(front && c_f <= 0 && c_l <= 0)) == 0ull;     // a scan that turns the other way
    if (lane == 0) {
        RayChunk c;
        c.bbox = vm ? ((uint64_t)x0 | ((uint64_t)x1 << 16) | ((uint64_t)y0 << 32) | ((uint64_t)y1 << 48)) : RAY_BBOX_EMPTY;
        c.sxy = ((uint32_t)fsx & 0xFFFFu) | ((uint32_t)fsy << 16);
        c.cone = (vm && (all_ccw || all_cw)) ? 1u : 0u;
        c.cwx = all_ccw ? fx : lx; c.cwy = all_ccw ? fy : ly;
        c.ccwx = all_ccw ? lx : fx; c.ccwy = all_ccw ? ly : fy;
        chunks[index] = c;
    }
}

// allocation walk: afterwards the occupancy patch of every ray cell of the scan exists.  RW_SEG threads per beam (many while the
// chip is not full: short chains; few when it is: fewer threads to launch), each takes one stretch [t0, t1] of the ray's steps.
// Round 4: the stretch is walked PATCH BY PATCH, not cell by cell.  Map::computeRay (src/sdm/map.cpp:198-227) puts step t of axis j
// at start_j + s_j * floor((2 t a_j + n) / (2 n)): both coordinates are monotone in t and move by at most one cell per step, so the
// patch changes exactly at the steps where an axis makes its k-th move for a k that takes it across a multiple of 32 -- the first
// such step is t = ceil((2 n k - n) / (2 a_j)).  Two merged sequences of such steps (one division per patch boundary) replace ~20
// cell steps per patch; one directory read per patch, the lock-free CAS only for a missing one.
__global__ __launch_bounds__(256) void k_ray_alloc_walk(DevParams prm, const RayRec* __restrict__ recs, int n, int first_particle, int RW_SEG)
{
    const int p = first_particle + blockIdx.x;
    const int g = blockIdx.y * 256 + threadIdx.x;
    const int i = g / RW_SEG, seg = g % RW_SEG;
    if (i >= n || p >= (int)prm.P) return;          // (the particle dimension of the grid is rounded up to a multiple of 8: xcd_grid)
    const RayRec r = recs[(size_t)p * n + i];
    if (!(r.nnf & (1u << 18))) return;
    const uint32_t steps = (r.nnf & 0xFFFFu) - 1u;
    const uint32_t t0 = 1u + (uint32_t)(((uint64_t)steps * (uint32_t)seg) / RW_SEG), t1 = (uint32_t)(((uint64_t)steps * (uint32_t)(seg + 1)) / RW_SEG);
    const PV pv = pview(prm, p);
    int16_t* occ_dir = pv.occ_dir;
    if (t0 > t1) return;
    const uint32_t a0 = r.a01 & 0xFFFFu, a1 = r.a01 >> 16, nn = r.nnf & 0xFFFFu;
    const uint32_t k0 = (uint32_t)(((uint64_t)(2u * t0 * a0 + nn) * r.magic) >> 42), k1 = (uint32_t)(((uint64_t)(2u * t0 * a1 + nn) * r.magic) >> 42);
    const bool neg0 = (r.nnf >> 16) & 1u, neg1 = (r.nnf >> 17) & 1u;
    const uint32_t bx = r.msx - prm.wx0, by = r.msy - prm.wy0;
    const uint32_t rx = neg0 ? bx - k0 : bx + k0, ry = neg1 ? by - k1 : by + k1;          // the cell of step t0
    uint32_t X = rx >> 5, Y = ry >> 5;
    // the number of moves after which the axis stands in the next patch, and the first step at which it has made them (n < 8192
    // and k <= a < 8192: the products fit 32 bits); an axis that never gets there: no such step
    uint32_t kx = neg0 ? k0 + (rx & 31u) + 1u : k0 + 32u - (rx & 31u), ky = neg1 ? k1 + (ry & 31u) + 1u : k1 + 32u - (ry & 31u);
    auto step_of = [nn](uint32_t k, uint32_t a) { return k > a ? 0xFFFFFFFFu : (2u * nn * k - nn + 2u * a - 1u) / (2u * a); };
    uint32_t tx = step_of(kx, a0), ty = step_of(ky, a1);
    (void)dir_get_or_alloc(occ_dir, Y * prm.W + X, prm.counts + 2 * p + 1, (int)pv.occ_cap, ERR_OCC_CAP, prm.err);
    for (;;) {
        const uint32_t t = tx < ty ? tx : ty;
        if (t > t1) break;
        if (tx == t) { X = neg0 ? X - 1u : X + 1u; kx += 32u; tx = step_of(kx, a0); }
        if (ty == t) { Y = neg1 ? Y - 1u : Y + 1u; ky += 32u; ty = step_of(ky, a1); }
        (void)dir_get_or_alloc(occ_dir, Y * prm.W + X, prm.counts + 2 * p + 1, (int)pv.occ_cap, ERR_OCC_CAP, prm.err);
    }
}

// arena slot -> directory position of every occupancy patch of the particle (rebuilt per scan: the directories change with
// allocation, resampling, window shifts and patch deletion; 32 KB of directory per particle).
//
// The same pass closes the ALLOCATION phase of the update: every occupancy patch the scan touches exists by now (k_ray_hits,
// k_ray_alloc_walk), and every distance-map patch the update can allocate -- first misses and hits in those patches, the
// brushfire's neighbours at most sqrt(max_sqdist) + 1 cells from a changed obstacle -- lies within `guard_r` patches of an
// occupancy patch THE SCAN CAN TOUCH: one whose patch meets the box sensor +- (largest point distance + 2 cells) (counting
// around every occupancy patch of the map, free space far behind the robot included, doubled the arenas of all particles without
// need in open environments).  The number of window positions without a distance-map patch but with such an occupancy patch that
// close bounds what is still to come; the particle's workgroup compares it with the free slots and raises ERR_DM_CAP BEFORE any
// map cell is modified, so that the host can grow the arena and run the update again.  (`guard_off`: the arenas are at their hard
// limit and the bound did not fit -- the update runs unguarded, as it did before the guard existed: it fails only if it REALLY
// runs out of patches.)
__device__ inline int imin(int a, int b) { return a < b ? a : b; }
__device__ inline int imax(int a, int b) { return a > b ? a : b; }
constexpr int RD_MARK_SIDE = 256;                            // the marked region (scan box grown by guard_r) is at most this many patches wide
constexpr int RD_MARK_WORDS = RD_MARK_SIDE * RD_MARK_SIDE / 32;

__global__ __launch_bounds__(256) void k_occ_reverse_dir(DevParams prm, int first_particle,
                                                          const double* __restrict__ tfs /*[P][12]*/, int reach_cells, int guard_off)
{
    __shared__ uint32_t mark[RD_MARK_WORDS];                  // positions of the region within guard_r patches of an occupancy patch in reach
    __shared__ uint32_t need_s;
    const int p = first_particle + blockIdx.x;
    const uint32_t W = prm.W, WW = W * W;
    const int tid = threadIdx.x, r = (int)prm.guard_r;
    const PV pv = pview_w(prm, p);
    const int16_t* occ_dir = pv.occ_dir;
    const int16_t* dm_dir = pv.dm_dir;
    // the patches the scan can touch: sensor origin (tf.translation(), src/pf_slam2d.cpp:452) +- reach, window-relative patch units
    const int scx = (int)(w2m(prm, uload_f64(tfs + 12 * (size_t)p + 9)) - prm.wx0), scy = (int)(w2m(prm, uload_f64(tfs + 12 * (size_t)p + 10)) - prm.wy0);
    const int bx0 = imax((scx - reach_cells) >> 5, 0), bx1 = imin((scx + reach_cells) >> 5, (int)W - 1);
    const int by0 = imax((scy - reach_cells) >> 5, 0), by1 = imin((scy + reach_cells) >> 5, (int)W - 1);
    // the marked region: that box grown by guard_r, clipped to the window; too large for the bitmap -> no bound (as guard_off)
    const int mx0 = imax(bx0 - r, 0), my0 = imax(by0 - r, 0), mw = imin(bx1 + r, (int)W - 1) - mx0 + 1, mh = imin(by1 + r, (int)W - 1) - my0 + 1;
    const bool bounded = mw > 0 && mh > 0 && mw <= RD_MARK_SIDE && mh <= RD_MARK_SIDE;
    const uint32_t nbits = bounded ? (uint32_t)(mw * mh) : 0u;
    for (uint32_t i = tid; i < (nbits + 31u) / 32u; i += 256u) mark[i] = 0;
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
            pv.rev[slot] = (int32_t)(w0 + (uint32_t)k);
            if (!bounded || x0 + k < bx0 || x0 + k > bx1 || wy < by0 || wy > by1) continue;       // out of the scan's reach: nothing changes there
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) {
                    const int x = x0 + k + dx - mx0, y = wy + dy - my0;
                    if ((uint32_t)x < (uint32_t)mw && (uint32_t)y < (uint32_t)mh) { const uint32_t n = (uint32_t)y * (uint32_t)mw + (uint32_t)x; atomicOr(&mark[n >> 5], 1u << (n & 31u)); }
                }
        }
    }
    __syncthreads();
    uint32_t need = 0;
    for (uint32_t i = tid; i < (nbits + 31u) / 32u; i += 256u) {
        uint32_t m = mark[i];
        while (m) {
            const uint32_t n = i * 32u + (uint32_t)(__ffs((int)m) - 1);
            const uint32_t wx = (uint32_t)mx0 + n % (uint32_t)mw, wy = (uint32_t)my0 + n / (uint32_t)mw;
            if (dm_dir[wy * W + wx] < 0) ++need;
            m &= m - 1u;
        }
    }
    if (need) atomicAdd(&need_s, need);
    __syncthreads();
    // (the bound itself is left for the host: a particle whose region is too small is grown by what it asks for -- guard[p])
    if (tid == 0) prm.guard[p] = bounded ? need_s : 0u;
    if (tid == 0 && bounded && !guard_off && (uint64_t)prm.counts[2 * p] + need_s > pv.dm_cap) atomicOr(prm.err, ERR_DM_CAP);
}

// developer build (-DLAMA_PROFILE_RAY, tools/prof_ray.py): event counts and per-phase cycles of k_ray_patches, summed over the launch
// into the spare part of prm.dbg (16 words behind the per-particle blocks)
#ifdef LAMA_PROFILE_RAY
#ifdef LAMA_PROFILE_RAY_COUNT         // the event counts cost a device atomic each: a build of its own, its cycles mean nothing
#define RPC(k, v) atomicAdd((unsigned long long*)(prm.dbg + 16 * (size_t)prm.P + (k)), (unsigned long long)(v))
#else
#define RPC(k, v) do {} while (0)
#endif
// cycles of thread 0 per phase, kept in registers and stored once per workgroup (the first 64 particles only: 8 words per workgroup
// behind the 16 event counters)
#define RPT_T(k) do { if (tid == 0) { const uint64_t t_ = __builtin_readcyclecounter(); tacc[(k) - 8] += t_ - tprev; tprev = t_; } } while (0)
#else
#define RPC(k, v) do {} while (0)
#define RPT_T(k) do {} while (0)
#endif
constexpr int RPT_CHUNK = 256;        // candidate beams tested per round: the records of those that cross the patch wait in LDS
#ifndef LAMA_RPT_WALK
#define LAMA_RPT_WALK 4
#endif
constexpr int RPT_WALK = LAMA_RPT_WALK;   // lanes that share the cells of one (beam, patch) crossing

__global__ __launch_bounds__(256, 6) void k_ray_patches(DevParams prm, const RayRec* __restrict__ recs, const uint64_t* __restrict__ bbox,
                                                      const RayChunk* __restrict__ chunks, int n, int first_particle)
{
    static_assert(RPT_CHUNK == 256, "one candidate beam per thread and round");
    __shared__ uint32_t cnt[1024];
    __shared__ RayRec lrec[RPT_CHUNK];           // beams of the round that cross the patch ...
    __shared__ uint32_t lbt[RPT_CHUNK];          // ... their step range t_lo | t_hi << 16 ...
    __shared__ uint16_t lbeam[RPT_CHUNK];        // ... and beam index
    __shared__ uint32_t actw[32], newm[32], wrapm[32];
    __shared__ uint32_t list_n[2];               // by round parity: the idle one is cleared while the other one is in use
    const int p = lane_particle(prm, first_particle, (int)blockIdx.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // first map-modifying kernel of the update: nothing is touched when the allocation phase failed (the host grows and retries)
    if (map_update_aborted(prm)) { if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) atomicOr(prm.err, ERR_CLEAN_ABORT); return; }
    if (p < 0) return;                                             // (another lane's particle)
    const int count = prm.counts[2 * p + 1];
    const PVOcc pv = pview_occ_w(prm, p);
    uint32_t* occ = pv.occ;
    const RayRec* prec = recs + (size_t)p * n;
    const uint64_t* pbb = bbox + (size_t)p * n;
    const int nck = (n + 63) / 64;                                 // <= 64: lane c of a wave holds chunk c (scans of up to 4096 points)
    // The kernel is a chain of memory round trips per patch unless they are taken out of the chain: the chunk records do not depend
    // on the patch (lane c of EVERY wave keeps chunk c in registers: each wave decides for itself which chunks a patch keeps, no
    // list, no barrier), the directory position of the next patch is fetched one patch ahead, and the cells of the patch, its hit
    // bits and the records of the first 256 candidate beams are requested together.
    RayChunk ck;
    ck.bbox = RAY_BBOX_EMPTY; ck.cone = 0; ck.cwx = ck.cwy = ck.ccwx = ck.ccwy = 0; ck.sxy = 0;
    if (lane < nck) ck = chunks[(size_t)p * nck + lane];
    if (tid < 2) list_n[tid] = 0;
    int slot = blockIdx.y;
    uint32_t pidx_next = slot < count ? (uint32_t)pv.rev[slot] : 0u;
    uint32_t par = 0;                                              // parity of the round
#ifdef LAMA_PROFILE_RAY
    uint64_t tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
#endif
    __syncthreads();
    for (; slot < count; slot += gridDim.y) {
        const uint32_t pidx = pidx_next;
        if (slot + (int)gridDim.y < count) pidx_next = (uint32_t)pv.rev[slot + gridDim.y];
        const int px = (int)((pidx % prm.W) * 32u), py = (int)((pidx / prm.W) * 32u);       // window-relative origin of the patch
#ifdef LAMA_PROFILE_RAY
        tprev = __builtin_readcyclecounter();
        ++tacc[5];
        if (tid == 0) RPC(0, 1);
#endif
        // ---- 2a. which chunks of 64 beams can cross the patch at all: the box of the chunk's rays, then its cone
        bool keep = false;
        if (lane < nck) {
            const int x0 = (int)(ck.bbox & 0xFFFFu), x1 = (int)((ck.bbox >> 16) & 0xFFFFu), y0 = (int)((ck.bbox >> 32) & 0xFFFFu), y1 = (int)(ck.bbox >> 48);
            keep = !(x1 < px || x0 > px + 31 || y1 < py || y0 > py + 31);              // (a chunk without rays has an empty box)
            if (keep && ck.cone) {
                const int sx = (int)(ck.sxy & 0xFFFFu), sy = (int)(ck.sxy >> 16);
                const int ax = px - 1 - sx, bx = px + 32 - sx, ay = py - 1 - sy, by = py + 32 - sy;   // the patch grown by one cell
                const int a00 = ck.cwx * ay - ck.cwy * ax, a10 = ck.cwx * ay - ck.cwy * bx, a01 = ck.cwx * by - ck.cwy * ax, a11 = ck.cwx * by - ck.cwy * bx;
                const int b00 = ck.ccwx * ay - ck.ccwy * ax, b10 = ck.ccwx * ay - ck.ccwy * bx, b01 = ck.ccwx * by - ck.ccwy * ax, b11 = ck.ccwx * by - ck.ccwy * bx;
                if ((a00 < 0 && a10 < 0 && a01 < 0 && a11 < 0) || (b00 > 0 && b10 > 0 && b01 > 0 && b11 > 0)) keep = false;
            }
        }
        const unsigned long long km = __ballot(keep);              // chunks 0 .. 63, the same in every wave
        const int ncand = __popcll(km) * 64;
        if (tid == 0) RPC(2, __popcll(km));
        if (km == 0ull) continue;                                  // a patch out of this scan's reach (the same decision in every wave): untouched
        // ---- 1. the patch: its cells and hit bits are requested now, classified below (after the candidates' loads are out too)
        uint32_t v[4];
        uint64_t hw[4];
        const uint64_t* hitw = pv.occ_hit + (size_t)slot * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = occ[(size_t)slot * 1024 + tid + 256 * j]; hw[j] = hitw[wave + 4 * j]; }
        bool has_act = false;
        for (int c0 = 0; c0 == 0 || c0 < ncand; c0 += RPT_CHUNK, par ^= 1u) {
            // ---- 2b. which beams of the kept chunks cross the patch, and in which steps: one candidate per thread
            const int jc = (c0 >> 6) + wave;                       // this wave's candidates are the beams of the jc-th kept chunk
            int b = -1;
            if (jc * 64 < ncand) {
                unsigned long long m = km;
                for (int i = 0; i < jc; ++i) m &= m - 1ull;
                b = (__ffsll((long long)m) - 1) * 64 + lane;
                if (b >= n) b = -1;
            }
            uint64_t bb = RAY_BBOX_EMPTY;
            RayRec r;
            r.msx = r.msy = r.a01 = r.nnf = 0; r.magic = 0;
            if (b >= 0) { bb = pbb[b]; r = prec[b]; RPC(3, 1); }
            if (c0 == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ci = tid + 256 * j;
                    cnt[ci] = 0;
                    const uint32_t o0 = v[j] & 0xFFFFu, v0 = v[j] >> 16;
                    const bool hit = (hw[j] >> lane) & 1ull;
                    const bool active = hit || !(v0 == 0 ? o0 == 0 : 4u * o0 < v0);
                    const unsigned long long am = __ballot(active);            // cells 256 j + 64 wave .. + 63
                    if (lane == 0) { actw[8 * j + 2 * wave] = (uint32_t)am; actw[8 * j + 2 * wave + 1] = (uint32_t)(am >> 32); }
                }
                if (tid < 32) { newm[tid] = 0; wrapm[tid] = 0; }
                RPT_T(8);
            }
            {
                const int x0 = (int)(bb & 0xFFFFu), x1 = (int)((bb >> 16) & 0xFFFFu), y0 = (int)((bb >> 32) & 0xFFFFu), y1 = (int)(bb >> 48);
                bool cross = !(x1 < px || x0 > px + 31 || y1 < py || y0 > py + 31);        // (an invalid ray has an empty box)
                if (cross) RPC(4, 1);
                const uint32_t nn = r.nnf & 0xFFFFu, steps = nn - 1u;
                if (cross) {
                    // Every ray cell lies within half a cell of the segment start -> start + (s0 a0, s1 a1) (each axis is the rounded
                    // position t a / n): a ray whose segment keeps the patch, grown by one cell, strictly on one side cannot touch it
                    const int sx = (int)(r.msx - prm.wx0), sy = (int)(r.msy - prm.wy0);
                    const int dx = ((r.nnf >> 16) & 1u) ? -(int)(r.a01 & 0xFFFFu) : (int)(r.a01 & 0xFFFFu);
                    const int dy = ((r.nnf >> 17) & 1u) ? -(int)(r.a01 >> 16) : (int)(r.a01 >> 16);
                    const int ax = px - 1 - sx, bx = px + 32 - sx, ay = py - 1 - sy, by = py + 32 - sy;
                    const int c00 = dx * ay - dy * ax, c10 = dx * ay - dy * bx, c01 = dx * by - dy * ax, c11 = dx * by - dy * bx;
                    if ((c00 > 0 && c10 > 0 && c01 > 0 && c11 > 0) || (c00 < 0 && c10 < 0 && c01 < 0 && c11 < 0)) cross = false;
                }
                if (cross) {
                    RPC(5, 1);
                    uint32_t xl = 0, xh = 0, yl = 0, yh = 0;
                    cross = ray_axis_range((int)(r.msx - prm.wx0), (r.nnf >> 16) & 1u, r.a01 & 0xFFFFu, nn, px, steps, xl, xh) &&
                            ray_axis_range((int)(r.msy - prm.wy0), (r.nnf >> 17) & 1u, r.a01 >> 16, nn, py, steps, yl, yh);
                    const uint32_t tl = xl > yl ? xl : yl, th = xh < yh ? xh : yh;
                    if (cross && tl <= th) {
                        const uint32_t k = atomicAdd(&list_n[par], 1u);
                        lrec[k] = r; lbt[k] = tl | (th << 16); lbeam[k] = (uint16_t)b;
                        RPC(6, 1); RPC(7, th - tl + 1u);
                    }
                }
            }
            __syncthreads();
            RPT_T(10);
            if (c0 == 0) has_act = __ballot(actw[lane & 31] != 0u) != 0ull;        // most patches of free space have no active cell at all
            if (tid == 0) list_n[par ^ 1u] = 0;
            // ---- 3. walk the ranges: RPT_WALK lanes per crossing, each a contiguous stretch of its steps in the incremental form of
            // Map::computeRay (k_j = floor((2 t a_j + n) / (2 n)) steps made by axis j, rem_j the remainder: every step adds 2 a_j
            // and carries at 2 n -- the closed form is evaluated once per lane, not once per cell)
            const uint32_t ln = list_n[par];
            const int hq = tid / RPT_WALK, hl = tid % RPT_WALK;
            for (uint32_t e = (uint32_t)hq; e < ln; e += 256u / RPT_WALK) {
                const uint32_t tl = lbt[e] & 0xFFFFu, len = (lbt[e] >> 16) - tl + 1u;
                const uint32_t t0 = tl + (len * (uint32_t)hl) / RPT_WALK, t1 = tl + (len * (uint32_t)(hl + 1)) / RPT_WALK;     // [t0, t1)
                if (t0 >= t1) continue;
                const RayRec q = lrec[e];
                const uint32_t qb = lbeam[e];
                const uint32_t a0 = q.a01 & 0xFFFFu, a1 = q.a01 >> 16, nn = q.nnf & 0xFFFFu, n2 = 2u * nn;
                const uint32_t k0 = (uint32_t)(((uint64_t)(2u * t0 * a0 + nn) * q.magic) >> 42), k1 = (uint32_t)(((uint64_t)(2u * t0 * a1 + nn) * q.magic) >> 42);
                uint32_t rem0 = 2u * t0 * a0 + nn - k0 * n2, rem1 = 2u * t0 * a1 + nn - k1 * n2;
                const bool neg0 = (q.nnf >> 16) & 1u, neg1 = (q.nnf >> 17) & 1u;
                const uint32_t rx = neg0 ? q.msx - prm.wx0 - k0 : q.msx - prm.wx0 + k0, ry = neg1 ? q.msy - prm.wy0 - k1 : q.msy - prm.wy0 + k1;
                if ((int)(rx & ~31u) != px || (int)(ry & ~31u) != py) continue;             // cannot happen: the range is exact
                int ci = (int)((rx & 31u) | ((ry & 31u) << 5));                             // the cell inside the patch; a step moves it by +-1 / +-32
                const int d0 = neg0 ? -1 : 1, d1 = neg1 ? -32 : 32;
                for (uint32_t t = t0; t < t1; ++t) {
                    const uint32_t cu = (uint32_t)ci & 1023u;
                    if (has_act && ((actw[cu >> 5] >> (cu & 31u)) & 1u)) act_append(prm, p, act_key((uint32_t)px + (cu & 31u), (uint32_t)py + (cu >> 5), qb, t));
                    else atomicAdd(&cnt[cu], 1u);
                    rem0 += 2u * a0; if (rem0 >= n2) { rem0 -= n2; ci += d0; }
                    rem1 += 2u * a1; if (rem1 >= n2) { rem1 -= n2; ci += d1; }
                }
            }
            __syncthreads();
            RPT_T(11);
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
                const PV dv = pview(prm, p);                        // (the distance-map side only where it is needed: rare)
                const int ds = dir_get_or_alloc(dv.dm_dir, pidx, prm.counts + 2 * p, (int)dv.dm_cap, ERR_DM_CAP, prm.err);
                if (ds >= 0) atomicOr((unsigned long long*)(dv.dm_mask + (size_t)ds * 16 + tid), (unsigned long long)nm);
            }
            if (wm) atomicOr((unsigned long long*)(pv.occ_mask + (size_t)slot * 16 + tid), (unsigned long long)wm);
        }
        __syncthreads();
        RPT_T(12);
    }
#ifdef LAMA_PROFILE_RAY
    if (tid == 0 && blockIdx.x < 64) {
        uint64_t* o = prm.dbg + 16 * (size_t)prm.P + 16 + 8 * ((size_t)blockIdx.x * gridDim.y + blockIdx.y);
        for (int k = 0; k < 8; ++k) o[k] = tacc[k];
    }
#endif
}

} // namespace lama_dev
