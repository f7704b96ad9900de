// RESEARCH ARCHIVE (not compiled into the product): the round-1 beam-centric visit kernel of the parallel ray-cast.
// It was superseded in round 2 by the patch-centric k_ray_patches (iris_lama_amd/csrc/lama_raycast_patch.h: no global
// atomics on the counters, 3.6x -> 1.0x algorithmic traffic) and removed from the product library in round 3.
// To build it again: include this file after lama_raycast_par.h inside lama_kernels.h and launch it in place of
// k_ray_alloc_walk / k_occ_reverse_dir / k_ray_patches (k_ray_hits without record outputs).
#pragma once
namespace lama_dev {
constexpr int RV_TABLE = 2048;               // LDS aggregation table (entries); 16 KB -> ~10 workgroups per CU

// Neighbouring beams share most of their cells near the sensor (0.25 deg apart: one cell at 11 m), and agent-scope
// atomics are executed memory-side (the line leaves the L2 every time, ~45 B of HBM traffic per atomic).  The plain
// "visited++" of an already-free cell -- the overwhelming majority of the visits -- is therefore first counted per
// workgroup (64 adjacent beams) in an LDS hash table keyed by the cell's arena index and flushed as ONE atomicAdd per
// distinct cell.  The counters commute, so the result is unchanged.
// `bpw` = beams per wave (<= 16; a workgroup takes 4 * bpw adjacent beams): many beams per workgroup aggregate better (fewer
// flush atomics: what counts when the chip is full), few beams per wave give short dependent chains (what counts when it is not).
__global__ __launch_bounds__(256) void k_ray_visits(DevParams prm, const double* __restrict__ pts, int n,
                                                     const double* __restrict__ tfs, int first_particle, int bpw)
{
    __shared__ uint32_t tkey[RV_TABLE];      // window-relative cell (ry << 13 | rx), 0xFFFFFFFF = empty
    __shared__ uint32_t tval[RV_TABLE];      // visits counted so far
    for (int k = threadIdx.x; k < RV_TABLE; k += 256) { tkey[k] = 0xFFFFFFFFu; tval[k] = 0; }
    __syncthreads();
    const int p = first_particle + blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t WW = (size_t)prm.W * prm.W;
    int16_t* occ_dir = prm.occ_dir + (size_t)p * WW;
    int16_t* dm_dir = prm.dm_dir + (size_t)p * WW;
    uint32_t* occ = prm.occ + (size_t)p * prm.occ_cap * 1024;
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = tfs[12 * (size_t)p + k];
    const int b0 = (blockIdx.y * 4 + wave) * bpw;
    // the geometry of this wave's 16 beams (fp64 transform, one 64-bit division each) is computed by 16 lanes at once and
    // broadcast beam by beam
    BeamGeom mine;
    {
        const int ib = b0 + (lane < bpw ? lane : 0);
        const int ic = ib < n ? ib : (n - 1);
        mine = beam_geometry(prm, T, pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2]);
    }
    for (int bi = 0; bi < bpw; ++bi) {
        const int i = b0 + bi;
        if (i >= n) break;
        BeamGeom g;
        g.msx = (uint32_t)__shfl((int)mine.msx, bi, 64); g.msy = (uint32_t)__shfl((int)mine.msy, bi, 64);
        g.a0 = (uint32_t)__shfl((int)mine.a0, bi, 64); g.a1 = (uint32_t)__shfl((int)mine.a1, bi, 64);
        g.nn = (uint32_t)__shfl((int)mine.nn, bi, 64);
        g.s0 = __shfl(mine.s0, bi, 64); g.s1 = __shfl(mine.s1, bi, 64);
        g.steps = __shfl(mine.steps, bi, 64);
        g.magic = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(mine.magic >> 32), bi, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)mine.magic, bi, 64);
        for (int base = 0; base < g.steps; base += 64) {
            const int t = base + lane + 1;
            if (t > g.steps) continue;
            const uint32_t st0 = (uint32_t)(((uint64_t)(2u * (uint32_t)t * g.a0 + g.nn) * g.magic) >> 42);
            const uint32_t st1 = (uint32_t)(((uint64_t)(2u * (uint32_t)t * g.a1 + g.nn) * g.magic) >> 42);
            const uint32_t cx = g.msx + (uint32_t)(g.s0 * (int)st0), cy = g.msy + (uint32_t)(g.s1 * (int)st1);
            const uint32_t rx = cx - prm.wx0, ry = cy - prm.wy0;
            if (rx >= prm.WC || ry >= prm.WC) { atomicOr(prm.err, ERR_WINDOW); continue; }
            // a cell this workgroup has already classified as "plain visited++" needs no global access at all
            const uint32_t key = (ry << 13) | rx;
            const uint32_t h0 = (key * 2654435761u) >> 21;                   // 11 bits
            {
                bool found = false;
#pragma unroll
                for (int tr = 0; tr < 4 && !found; ++tr) {
                    const uint32_t hh = (h0 + (uint32_t)tr) & (RV_TABLE - 1);
                    if (tkey[hh] == key) { atomicAdd(&tval[hh], 1u); found = true; }
                }
                if (found) continue;
            }
            const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5), ci = (rx & 31u) | ((ry & 31u) << 5);
            const int slot = dir_get_or_alloc(occ_dir, pidx, prm.counts + 2 * p + 1, (int)prm.occ_cap, ERR_OCC_CAP, prm.err);
            if (slot < 0) continue;
            const uint64_t bit = 1ull << (ci & 63);
            const bool hitcell = (prm.occ_hit[((size_t)p * prm.occ_cap + slot) * 16 + (ci >> 6)] & bit) != 0;
            uint32_t* cell = occ + (size_t)slot * 1024 + ci;
            const uint32_t v = *cell;
            const uint32_t o0 = v & 0xFFFFu, v0 = v >> 16;
            const bool inert = !hitcell && (v0 == 0 ? o0 == 0 : 4u * o0 < v0);
            if (inert && v0 != 0 && v0 < 0xF000u) {
                // visited++ (setFree): no event, no wrap possible -> count it in the LDS table (linear probing, 4 tries)
                uint32_t h = h0;
                bool done = false;
#pragma unroll
                for (int tr = 0; tr < 4 && !done; ++tr) {
                    const uint32_t old = atomicCAS(&tkey[h], 0xFFFFFFFFu, key);
                    if (old == 0xFFFFFFFFu || old == key) { atomicAdd(&tval[h], 1u); done = true; }
                    h = (h + 1) & (RV_TABLE - 1);
                }
                if (!done) atomicAdd(cell, 0x10000u);
            } else if (inert) {
                const uint32_t old = atomicAdd(cell, 0x10000u);              // visited++ (setFree, no event possible ...)
                if (old == 0) {                                              // ... except the first miss of a new cell:
                    // removeObstacle on a cell that cannot be an obstacle = get(): patch allocation + mask bit (:228-234)
                    const int ds = dir_get_or_alloc(dm_dir, pidx, prm.counts + 2 * p, (int)prm.dm_cap, ERR_DM_CAP, prm.err);
                    if (ds >= 0) atomicOr((unsigned long long*)(prm.dm_mask + ((size_t)p * prm.dm_cap + ds) * 16 + (ci >> 6)), (unsigned long long)bit);
                }
                if ((old >> 16) == 0xFFFFu)                                  // uint16 wrap: keep the Container mask bit
                    atomicOr((unsigned long long*)(prm.occ_mask + ((size_t)p * prm.occ_cap + slot) * 16 + (ci >> 6)), (unsigned long long)bit);
            } else {
                act_append(prm, p, act_key(rx, ry, (uint32_t)i, (uint32_t)t));
            }
        }
    }
    // flush: one atomicAdd per distinct cell of this workgroup
    __syncthreads();
    uint32_t* occ_base = prm.occ + (size_t)p * prm.occ_cap * 1024;
    for (int k = threadIdx.x; k < RV_TABLE; k += 256) {
        const uint32_t cnt = tval[k];
        if (cnt) {      // the patch exists (the cell was classified through it): plain directory read
            const uint32_t key = tkey[k], rx = key & 0x1FFFu, ry = key >> 13;
            const int slot = occ_dir[(ry >> 5) * prm.W + (rx >> 5)];
            atomicAdd(occ_base + (uint32_t)slot * 1024u + ((rx & 31u) | ((ry & 31u) << 5)), cnt << 16);
        }
    }
}

} // namespace lama_dev
