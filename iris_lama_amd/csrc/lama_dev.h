// lama_dev.h -- device-side data model and scalar math of the MI355X scan-matching path.
//
// HBM layout (ONE particle set; resample() is done in place, see below):
//
//   per HOME h (a directory slot; P of them):
//     dm_dir  int16 [W*W]         window directory: patch (wy*W+wx) -> slot in the particle's DM region, -1 = absent
//     occ_dir int16 [W*W]         same for the occupancy region
//   pooled planes, shared by all particles (a pool is a list of CHUNKS -- it grows by another chunk, nothing is ever copied to
//   grow it); a particle owns one contiguous REGION of `cap` patches of each map kind inside one chunk
//   (PartRec: home, capacities, and the region's address in every plane -- per particle):
//     dm_sv   uint16[cap][1024]   DM plane A: bit15 valid_obstacle | bit14 is_queued | bits13..0 sqdist
//     dm_obs  uint32[cap][1024]   DM plane B: int16 obstacle.x | int16 obstacle.y << 16
//     dm_mask uint64[cap][16]     Container::mask of the DM patch
//     occ     uint32[cap][1024]   frequency cell: uint16 occupied | uint16 visited << 16
//     occ_mask uint64[cap][16]
//   per particle p (logical index, the reference's particles_[current][p]):
//     counts  int32 [2]           allocated DM / occupancy slots
//
// The reference keeps two particle sets and copies patch POINTERS on resample (src/pf_slam2d.cpp:558-574,
// include/lama/cow_ptr.h:86-118).  Here a resample is a permutation of the PartRec table: a particle that survives keeps its
// home and its regions (nothing moves), only the second and further copies of a multiply drawn particle are copied -- into the
// homes and regions of particles that died.  Regions have per-particle capacities and grow one particle at a time (a host-side
// region allocator over the pools), so HBM follows what the maps use instead of P x the largest particle.
//
// The reference's records are distance_t (10 B AoS) and frequency (4 B); the split DM planes keep the
// 2 bytes the match kernel gathers (sqdist+valid) apart from the 4 bytes only the brushfire needs.
// A cell is addressed by window-relative cell coordinates (rx, ry) = map coordinate - window origin;
// the map coordinate itself is the reference's (offset by 42,275,904 cells) and is only
// ever formed in fp64 / uint32, never fp32.  Unused region slots and free pool space are always all-zero (calloc semantics
// of Container::alloc, src/sdm/container.cpp:76-95).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>

// Two instantiations of the library (iris_lama_amd/Makefile): the default one, and -- compiled with -DLAMA_WIDE_DM into
// liblama_hip_wide.so -- the one for a distance map whose l2_max lies beyond 127 cells, up to the 255 cells that the reference's own
// records can hold (distance_t::sqdist is a uint16_t, include/lama/sdm/dynamic_distance_map.h:74-88: 255^2 = 65,025).  What differs:
//   * plane A of the distance map is 4 bytes per cell instead of 2 (sv_t: flags in bits 31 / 30, the squared distance below);
//   * a brushfire queue entry carries the obstacle offset in 9 + 9 bits and the cell in 15 + 15 (q_entry below).
// Everything else -- kernels, queues, the heap (16-bit priorities), the host side -- is the same source; the kernels of the wide
// build live in a namespace of their own so that both libraries can be loaded into one process.
#ifdef LAMA_WIDE_DM
#define lama_dev lama_dev_wide
#endif

#include "lama_heap.h"

// The lanes of a wave execute in lock step on the device: "every lane reads X, then lane 0 overwrites X" needs no synchronisation
// there.  The lane-level simulator of tests/sim (test infrastructure; this source compiled for the host) runs the lanes as
// cooperative fibers and needs that ordering spelled out as a wave rendezvous.
#ifdef LAMA_WAVE_SIM
#define LAMA_LOCKSTEP() ((void)__ballot(true))
#else
#define LAMA_LOCKSTEP() ((void)0)
#endif

namespace lama_dev {

#ifdef LAMA_WIDE_DM
typedef uint32_t sv_t;
constexpr uint32_t SV_VALID = 0x80000000u;
constexpr uint32_t SV_QUEUED = 0x40000000u;
constexpr uint32_t SV_SQMASK = 0x3FFFFFFFu;
constexpr int SV_VALID_BIT = 31;
constexpr uint32_t MAX_OFFSET_CELLS = 255;        // |obstacle offset| a queue entry can carry
#else
typedef uint16_t sv_t;
constexpr uint16_t SV_VALID = 0x8000;
constexpr uint16_t SV_QUEUED = 0x4000;
constexpr uint16_t SV_SQMASK = 0x3FFF;
constexpr int SV_VALID_BIT = 15;
constexpr uint32_t MAX_OFFSET_CELLS = 127;
#endif
constexpr uint32_t SV_BYTES = (uint32_t)sizeof(sv_t);         // per cell of plane A ...
constexpr uint32_t SV_PATCH_BYTES = 1024u * SV_BYTES;         // ... and per patch

constexpr int ERR_WINDOW = 1;
constexpr int ERR_DM_CAP = 2;
constexpr int ERR_OCC_CAP = 4;
constexpr int ERR_QUEUE = 8;
constexpr int ERR_NUMERIC = 16;
// set by the first map-modifying kernel of an update when the ALLOCATION phase before it (hit cells, ray patches, the bound on the
// distance-map patches the update can need) reported a capacity / window error: every modifying kernel then returns at once, the
// maps are untouched, and the host can grow the arenas and run the update again (lama_hip.hip, recover_update)
constexpr int ERR_CLEAN_ABORT = 32;

struct Affine {            // rows 0..2 of [R | t]
    double R[3][3];
    double t[3];
};

// where logical particle p keeps its maps: the home of its two directories, the capacities of its two regions (patches) and
// where each plane of them starts (the host's region allocator places a region inside one chunk of the pool)
struct PartRec {
    uint32_t home, dm_cap, occ_cap, r0;
    sv_t* dm_sv; uint32_t* dm_obs; uint64_t* dm_mask;
    uint32_t* occ; uint64_t* occ_mask; uint64_t* occ_hit;
    int32_t* rev;          // [occ_cap] region slot -> directory position, rebuilt per scan (k_occ_reverse_dir)
    uint64_t r1;
};

struct DevParams {
    uint32_t P;            // particles in this context
    uint32_t W;            // window side in patches
    uint32_t WC;           // window side in cells (W*32)
    uint32_t wx0, wy0;     // window origin in map cells
    uint32_t qcap;
    const PartRec* part;   // [P] logical particle -> home / regions
    uint32_t max_sqdist;
    uint32_t max_iter;
    double scale, off, resolution, maxdist, meas_sigma;
    double trunc_ray, trunc_range;
    // directories [home][W*W] (the planes are reached through PartRec)
    int16_t* dm_dir;
    int16_t* occ_dir;
    int32_t* counts;       // [P][2] (logical particle)
    // shared
    double* poses;         // [P][4]
    uint64_t* q_lower;     // [P][qcap]
    uint64_t* q_raise;     // [P][qcap]
    uint32_t* qsizes;      // [P][2] entries handed from k_raycast to k_brushfire (lower, raise)
    uint32_t* slow;        // [P] 1 = a stage handed this particle to the next (bigger / slower) stage
    uint32_t* slow_list;   // [4][P] ([2]: the routed particles, [3]: replay hand-overs of the early lane) particles the first stage of the brushfire ([0]) / of the ordered replay ([1]) handed to its resume stage ...
    uint32_t* slow_n;      // [5]    ... and how many; [2]: particles routed to the big-queue stage before the brushfire started (k_bf_route); [3]: early lane's replay hand-overs; [4]: early-lane particles
    uint8_t* heavy;        // [P]    1 = routed: the first brushfire stage skips the particle (nullptr: routing is off)
    // early lane (k_early_list): the particles that were routed in the PREVIOUS update are usually the long chains again; their
    // modifying ray-cast kernels and their brushfire run on a stream of their own, ahead of everybody else's
    const uint32_t* elist;   // [cap] this launch belongs to the early lane: its particles ... (nullptr: main lane)
    const uint32_t* elist_n; //       ... and how many
    const uint8_t* early;    // [P]   main lane: 1 = the particle is in the early lane, skip it (nullptr: there is none)
    uint32_t lane;           //       0 main / 1 early: which hand-over segment the ordered replay uses
    uint64_t* act;         // [P][act_cap] active visits of the parallel ray-cast (lama_raycast_par.h)
    uint32_t* act_count;   // [P]
    uint32_t act_cap;
    uint64_t* stats;       // [P][4] iterations, evals, ray_cells, bf_cells of the last call
    int32_t* err;
    uint64_t* dbg;         // [P][8] cycle counters of the profiling build (LAMA_PROFILE_BF), else unused
    uint32_t* guard;       // [P] upper bound of the distance-map patches the update may still allocate (k_occ_reverse_dir); read by the host
    uint32_t guard_r;      // patches around an occupancy patch the brushfire can reach: ceil((sqrt(max_sqdist) + 1) / 32)
    // occupancy cell policy / ray rule (cfg.occupancy_policy, cfg.ray_rule)
    uint32_t occ_policy, ray_rule;
    uint32_t strategy;     // 0 = GaussNewton, 1 = LevenbergMarquard (cfg.solver_strategy; Slam2D / Loc2D "lm")
    double lo_miss, lo_hit, lo_min, lo_max;   // ProbabilisticOccupancyMap parameters (float-rounded, as the reference stores them)
};

// Particle p's view of the maps: its directories, its regions of the pooled planes (slot-relative addressing inside), capacities.
struct PV {
    int16_t* dm_dir; int16_t* occ_dir;
    sv_t* dm_sv; uint32_t* dm_obs; uint64_t* dm_mask;
    uint32_t* occ; uint64_t* occ_mask; uint64_t* occ_hit;
    int32_t* counts; int32_t* rev;
    uint32_t dm_cap, occ_cap;
};
// The table is written by the host between launches and never by a kernel.  It is read with agent-scope atomic loads (coherent at
// the L2: no non-coherent cache can serve a record the host has rewritten since -- a move of the particle's region, a resample) and,
// the particle index being wave-uniform, broadcast into SGPRs with v_readfirstlane: the regions' base addresses and capacities feed
// scalar address arithmetic and the buffer resources of the brushfire's straight-line pop.  (Round 5 first read it through the
// constant address space -- s_load through the scalar data cache.  With eight contexts on one device, 8 % of the runs then gave a
// few particles a distance map that differed from the other runs': a record served from a stale scalar-cache line sends a
// kernel to the region the particle has left.  tools: the run-to-run determinism loop of DESIGN.md section 2.)
#ifdef LAMA_WAVE_SIM
__device__ inline uint32_t part_u32(const uint32_t* q) { return *q; }
__device__ inline uint64_t part_u64(const uint64_t* q) { return *q; }
#else
__device__ inline uint32_t part_u32(const uint32_t* q)
{
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ inline uint64_t part_u64(const uint64_t* q)
{
    const uint64_t v = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
#endif
#if defined(LAMA_KC_PROBE) && !defined(LAMA_WAVE_SIM)
// Experiment build (tools/kc_probe.py, DESIGN.md section 8 "scalar-cache hazard"): every reader of the particle table ALSO fetches its
// record the way round 5 first did -- through the constant address space, s_load through the scalar data cache -- and with a plain
// vector load, compares both with the agent-scope load the product uses, and logs what differs.  LAMA_KC_PROBE=2: an s_dcache_inv
// precedes the scalar loads (is the stale copy in the scalar cache, or behind it?).
__device__ uint32_t g_kc_n;
__device__ uint64_t g_kc_ev[64][8];
__device__ __noinline__ void kc_probe(const DevParams& prm, int p)
{
    const uint64_t* q = reinterpret_cast<const uint64_t*>(prm.part + p);
#if LAMA_KC_PROBE == 2
    __builtin_amdgcn_s_dcache_inv();
#endif
    const unsigned long long act = __ballot(1);
    const bool rec = (int)(threadIdx.x & 63u) == __ffsll((long long)act) - 1;
    for (int k = 0; k < 9; ++k) {
        const uint64_t* qa = q + k;
        const uint64_t ua = (uint64_t)qa;
        const uint64_t sa = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ua >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ua);
        uint64_t s, v;
        asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(s) : "s"(sa) : "memory");
        asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(qa) : "memory");
        const uint64_t a = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((s != a || v != a) && rec) {
            const uint32_t i = atomicAdd(&g_kc_n, 1u);
            if (i < 64u) {
                g_kc_ev[i][0] = (uint64_t)(uint32_t)p | ((uint64_t)k << 32) | ((uint64_t)(s != a) << 40) | ((uint64_t)(v != a) << 41);
                g_kc_ev[i][1] = s; g_kc_ev[i][2] = v; g_kc_ev[i][3] = a;
                g_kc_ev[i][4] = (uint64_t)gridDim.x | ((uint64_t)gridDim.y << 24) | ((uint64_t)blockDim.x << 48);
                g_kc_ev[i][5] = (uint64_t)blockIdx.x | ((uint64_t)blockIdx.y << 32);
                g_kc_ev[i][6] = (uint64_t)prm.part; g_kc_ev[i][7] = __builtin_readcyclecounter();
            }
        }
    }
}
#define LAMA_KC_PROBE_CALL(prm, p) kc_probe(prm, p)
#else
#define LAMA_KC_PROBE_CALL(prm, p) do {} while (0)
#endif
// (p must be wave-uniform: every caller takes it from blockIdx or from a list entry all lanes read)
#if defined(LAMA_KC_OLD) && !defined(LAMA_WAVE_SIM)
#if LAMA_KC_OLD == 4
__device__ uint32_t g_kc_n;
__device__ uint64_t g_kc_ev[64][8];
#endif
// Experiment build (tools/determinism_sweep.py with this library in place): the table read as round 5 first read it.  1: through the
// constant address space (s_load, scalar data cache); 2: the same behind an s_dcache_inv; 3: a plain const pointer (the compiler's choice).
__device__ inline PV pview(const DevParams& prm, int p)
{
#if LAMA_KC_OLD == 3
    const PartRec* t = prm.part + p;      // (does not build: the brushfire's buffer resources need SGPRs)
#else
    typedef const __attribute__((address_space(4))) PartRec* PartTablePtr;
    PartTablePtr t = (PartTablePtr)prm.part + p;
#endif
#if LAMA_KC_OLD == 2
    asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    const size_t WW = (size_t)prm.W * prm.W;
    const uint32_t home = t->home;
    PV v;
    v.dm_dir = prm.dm_dir + (size_t)home * WW;
    v.occ_dir = prm.occ_dir + (size_t)home * WW;
    v.dm_sv = t->dm_sv; v.dm_obs = t->dm_obs; v.dm_mask = t->dm_mask;
    v.occ = t->occ; v.occ_mask = t->occ_mask; v.occ_hit = t->occ_hit; v.rev = t->rev;
    v.counts = prm.counts + 2 * (size_t)p;
    v.dm_cap = t->dm_cap; v.occ_cap = t->occ_cap;
#if LAMA_KC_OLD == 4
    {   // the values the kernel goes on to USE (the scalar loads above) against agent-scope loads of the same words, logged when they differ
        const uint64_t* q = reinterpret_cast<const uint64_t*>(prm.part + p);
        const uint64_t used[9] = {(uint64_t)home | ((uint64_t)v.dm_cap << 32), (uint64_t)v.occ_cap, (uint64_t)v.dm_sv, (uint64_t)v.dm_obs, (uint64_t)v.dm_mask,
                                  (uint64_t)v.occ, (uint64_t)v.occ_mask, (uint64_t)v.occ_hit, (uint64_t)v.rev};
        const unsigned long long act = __ballot(1);
        const bool rec = (int)(threadIdx.x & 63u) == __ffsll((long long)act) - 1;
        for (int k = 0; k < 9; ++k) {
            uint64_t a = __hip_atomic_load(q + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (k == 1) a &= 0xFFFFFFFFull;
            if (used[k] != a && rec) {
                const uint32_t i = atomicAdd(&g_kc_n, 1u);
                if (i < 64u) {
                    g_kc_ev[i][0] = (uint64_t)(uint32_t)p | ((uint64_t)k << 32) | (1ull << 40);
                    g_kc_ev[i][1] = used[k]; g_kc_ev[i][2] = 0; g_kc_ev[i][3] = a;
                    g_kc_ev[i][4] = (uint64_t)gridDim.x | ((uint64_t)gridDim.y << 24) | ((uint64_t)blockDim.x << 48);
                    g_kc_ev[i][5] = (uint64_t)blockIdx.x | ((uint64_t)blockIdx.y << 32);
                    g_kc_ev[i][6] = (uint64_t)prm.part; g_kc_ev[i][7] = __builtin_readcyclecounter();
                }
            }
        }
    }
#endif
    return v;
}
#define LAMA_PVIEW_OLD 1
#else
__device__ inline PV pview(const DevParams& prm, int p)
{
    LAMA_KC_PROBE_CALL(prm, p);
    const PartRec* t = prm.part + p;
    const uint64_t* q = reinterpret_cast<const uint64_t*>(t);       // 10 quadwords: {home, dm_cap} {occ_cap, r0} 7 addresses, r1
    const size_t WW = (size_t)prm.W * prm.W;
    const uint64_t w0 = part_u64(q), w1 = part_u64(q + 1);
    const uint32_t home = (uint32_t)w0;
    PV v;
    v.dm_dir = prm.dm_dir + (size_t)home * WW;
    v.occ_dir = prm.occ_dir + (size_t)home * WW;
    v.dm_sv = (sv_t*)part_u64(q + 2); v.dm_obs = (uint32_t*)part_u64(q + 3); v.dm_mask = (uint64_t*)part_u64(q + 4);
    v.occ = (uint32_t*)part_u64(q + 5); v.occ_mask = (uint64_t*)part_u64(q + 6); v.occ_hit = (uint64_t*)part_u64(q + 7);
    v.rev = (int32_t*)part_u64(q + 8);
    v.counts = prm.counts + 2 * (size_t)p;
    v.dm_cap = (uint32_t)(w0 >> 32); v.occ_cap = (uint32_t)w1;
    return v;
}
#endif
static_assert(sizeof(PartRec) == 80 && offsetof(PartRec, dm_sv) == 16 && offsetof(PartRec, rev) == 64, "pview reads PartRec as ten quadwords");

// The same for callers whose wave is COMPLETE at the call (kernel prologues, behind wave-uniform returns only): lane k < 10 fetches
// quadword k, ten v_readlane pairs distribute them -- one load instruction and two VGPRs instead of ten loads and twenty (which cost
// k_ray_patches its sixth wave per SIMD).  Not for code behind a per-lane return: a lane that is gone leaves stale register contents.
__device__ inline PV pview_w(const DevParams& prm, int p)
{
#if defined(LAMA_WAVE_SIM) || defined(LAMA_PVIEW_OLD)
    return pview(prm, p);
#else
    LAMA_KC_PROBE_CALL(prm, p);
    const uint64_t* q = reinterpret_cast<const uint64_t*>(prm.part + p);
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint64_t w = __hip_atomic_load(q + (lane < 10u ? lane : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lo = (int)(uint32_t)w, hi = (int)(uint32_t)(w >> 32);
#define LAMA_PART_Q(k) (((uint64_t)(uint32_t)__builtin_amdgcn_readlane(hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, k))
    const size_t WW = (size_t)prm.W * prm.W;
    const uint64_t w0 = LAMA_PART_Q(0), w1 = LAMA_PART_Q(1);
    const uint32_t home = (uint32_t)w0;
    PV v;
    v.dm_dir = prm.dm_dir + (size_t)home * WW;
    v.occ_dir = prm.occ_dir + (size_t)home * WW;
    v.dm_sv = (sv_t*)LAMA_PART_Q(2); v.dm_obs = (uint32_t*)LAMA_PART_Q(3); v.dm_mask = (uint64_t*)LAMA_PART_Q(4);
    v.occ = (uint32_t*)LAMA_PART_Q(5); v.occ_mask = (uint64_t*)LAMA_PART_Q(6); v.occ_hit = (uint64_t*)LAMA_PART_Q(7);
    v.rev = (int32_t*)LAMA_PART_Q(8);
#undef LAMA_PART_Q
    v.counts = prm.counts + 2 * (size_t)p;
    v.dm_cap = (uint32_t)(w0 >> 32); v.occ_cap = (uint32_t)w1;
    return v;
#endif
}
// Only the occupancy side (k_ray_patches keeps a patch, its hit bits and a list of beams in registers: every scalar it does not hold
// is a VGPR it does not lose to spilled SGPRs).  Complete waves only, like pview_w.
struct PVOcc { uint32_t* occ; uint64_t* occ_mask; uint64_t* occ_hit; int32_t* rev; };
__device__ inline PVOcc pview_occ_w(const DevParams& prm, int p)
{
#if defined(LAMA_WAVE_SIM) || defined(LAMA_PVIEW_OLD)
    const PV v = pview(prm, p);
    return PVOcc{v.occ, v.occ_mask, v.occ_hit, v.rev};
#else
    LAMA_KC_PROBE_CALL(prm, p);
    const uint64_t* q = reinterpret_cast<const uint64_t*>(prm.part + p) + 5;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint64_t w = __hip_atomic_load(q + (lane < 4u ? lane : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lo = (int)(uint32_t)w, hi = (int)(uint32_t)(w >> 32);
#define LAMA_PART_Q(k) (((uint64_t)(uint32_t)__builtin_amdgcn_readlane(hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, k))
    PVOcc v{(uint32_t*)LAMA_PART_Q(0), (uint64_t*)LAMA_PART_Q(1), (uint64_t*)LAMA_PART_Q(2), (int32_t*)LAMA_PART_Q(3)};
#undef LAMA_PART_Q
    return v;
#endif
}

// ------------------------------------------------------------------------------------------------
// Wave-uniform reads of device data that the HOST rewrites between launches (job lists, shipping descriptors, the scan transforms,
// poses, candidate lists) or that a kernel of ANOTHER stream of the context produced (the early / routed lists and their counts).
// The rule of this library ("scalar-cache hazard"; enforced on the ISA by tools/check_scalar_loads.py): such data
// is not read with s_load through the scalar data cache but with agent-scope loads -- coherent at the L2 -- and, the address being
// wave-uniform, broadcast into SGPRs with v_readfirstlane.  A uniform plain load of `const __restrict__` memory becomes an s_load.
// ------------------------------------------------------------------------------------------------
#ifdef LAMA_WAVE_SIM
__device__ inline uint32_t uload_u32(const void* q) { return *reinterpret_cast<const uint32_t*>(q); }
__device__ inline uint64_t uload_u64(const void* q) { return *reinterpret_cast<const uint64_t*>(q); }
#else
__device__ inline uint32_t uload_u32(const void* q) { return part_u32(reinterpret_cast<const uint32_t*>(q)); }
__device__ inline uint64_t uload_u64(const void* q) { return part_u64(reinterpret_cast<const uint64_t*>(q)); }
#endif
__device__ inline int32_t uload_i32(const void* q) { return (int32_t)uload_u32(q); }
__device__ inline double uload_f64(const void* q) { return __longlong_as_double((long long)uload_u64(q)); }
template <class T> __device__ inline T* uload_ptr(const void* q) { return reinterpret_cast<T*>(uload_u64(q)); }
// the same without the broadcast, for a read that one lane does (thread 0 fetching a pose): coherent at the L2, never an s_load
#ifdef LAMA_WAVE_SIM
__device__ inline double cload_f64(const double* q) { return *q; }
#else
__device__ inline double cload_f64(const double* q)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const uint64_t*>(q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
#endif
// N <= 64 consecutive doubles for a COMPLETE wave (kernel prologues): lane k fetches element k, v_readlane distributes them -- one
// load instruction instead of N (the scan transform of a particle: 12 doubles the host rewrites before every map update)
template <int N> __device__ inline void uload_f64_w(const double* q, double (&out)[N])
{
#ifdef LAMA_WAVE_SIM
    for (int k = 0; k < N; ++k) out[k] = q[k];
#else
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint64_t w = __hip_atomic_load(reinterpret_cast<const uint64_t*>(q) + (lane < (uint32_t)N ? lane : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lo = (int)(uint32_t)w, hi = (int)(uint32_t)(w >> 32);
#pragma unroll
    for (int k = 0; k < N; ++k)
        out[k] = __longlong_as_double((long long)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(hi, k) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, k)));
#endif
}
// a record of N quadwords (8-byte aligned) into a by-value struct
template <class T> __device__ inline T uload_rec(const T* q)
{
    static_assert(sizeof(T) % 8 == 0 && alignof(T) <= 8, "uload_rec: whole quadwords");
    union { T v; uint64_t w[sizeof(T) / 8]; } u;
    for (unsigned k = 0; k < sizeof(T) / 8; ++k) u.w[k] = uload_u64(reinterpret_cast<const uint64_t*>(q) + k);
    return u.v;
}

// the particle of workgroup `bx` of a per-particle launch: early-lane launches walk their list, main-lane launches skip the early
// lane's particles; -1: nothing to do
__device__ inline int lane_particle(const DevParams& prm, int first_particle, int bx)
{
    if (prm.elist) return bx < (int)uload_u32(prm.elist_n) ? (int)uload_u32(prm.elist + bx) : -1;     // (bx is the workgroup's index; the list is another stream's kernel's)
    const int p = first_particle + bx;
    if (p >= (int)prm.P) return -1;                 // (two-dimensional launches round their particle dimension up to a multiple of 8: see xcd_grid)
    return (prm.early && prm.early[p]) ? -1 : p;
}


// ------------------------------------------------------------------------------------------------
// SE2 (unit complex + translation): include/lama/sophus/so2.hpp:168-176,205-214,322-324;
// se2.hpp:154-157,262-265,389-411
// ------------------------------------------------------------------------------------------------
struct SE2 { double c, s, tx, ty; };

__device__ inline bool so2_normalize(double& c, double& s)
{
    double length = sqrt(c * c + s * s);
    if (length < 1e-10) return false;
    c /= length;
    s /= length;
    return true;
}

// state' = SE2::exp(h) * state   (MatchSurface2D::update, src/match_surface_2d.cpp:118-122)
__device__ inline SE2 se2_exp_mul(const double h[3], const SE2& st, bool& ok)
{
    const double theta = h[2];
    double ec, es;
    sincos(theta, &es, &ec);                    // (one range reduction for both: the step runs on a single lane)
    ok = so2_normalize(ec, es) && ok;
    double a, b;   // sin(theta)/theta, (1-cos(theta))/theta
    if (fabs(theta) < 1e-10) {
        double theta_sq = theta * theta;
        a = 1. - (1. / 6.) * theta_sq;
        b = 0.5 * theta - (1. / 24.) * theta * theta_sq;
    } else {
        a = es / theta;
        b = (1. - ec) / theta;
    }
    const double etx = a * h[0] - b * h[1];
    const double ety = b * h[0] + a * h[1];
    SE2 r;
    r.tx = etx + (ec * st.tx - es * st.ty);
    r.ty = ety + (es * st.tx + ec * st.ty);
    r.c = ec * st.c - es * st.s;
    r.s = ec * st.s + es * st.c;
    ok = so2_normalize(r.c, r.s) && ok;
    return r;
}

// fixed_tf(state) * moving_tf : Translation(x,y,0) * AngleAxis(atan2(s,c), Z) * (Translation(origin) * q)
// (src/match_surface_2d.cpp:49-58; Eigen AngleAxis::toRotationMatrix restated for the Z axis)
__device__ inline Affine scan_tf(const SE2& st, const Affine& m)
{
    const double theta = atan2(st.s, st.c);
    double sn, cs;
    sincos(theta, &sn, &cs);
    const double F[3][3] = {{cs, 0.0 - sn, 0.0}, {sn, cs, 0.0}, {0.0, 0.0, (1.0 - cs) + cs}};
    const double ft[3] = {st.tx, st.ty, 0.0};
    Affine r;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) r.R[i][j] = (F[i][0] * m.R[0][j] + F[i][1] * m.R[1][j]) + F[i][2] * m.R[2][j];
        r.t[i] = ((F[i][0] * m.t[0] + F[i][1] * m.t[1]) + F[i][2] * m.t[2]) + ft[i];
    }
    return r;
}

// Eigen 3.3 LDLT<Matrix3d, Lower> with pivoting + solve (src/nlls/gauss_newton.cpp:66).
// A: lower triangle used, indices [row][col].
__device__ inline void ldlt3_solve(const double A[3][3], const double b[3], double x[3])
{
    double m[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) m[i][j] = (j <= i) ? A[i][j] : A[j][i];
    int tr[3] = {0, 1, 2};
    double temp[3];
    for (int k = 0; k < 3; ++k) {
        int big = k;
        double best = fabs(m[k][k]);
        for (int i = k + 1; i < 3; ++i)
            if (fabs(m[i][i]) > best) { best = fabs(m[i][i]); big = i; }
        tr[k] = big;
        if (k != big) {
            const int s = 3 - big - 1;
            for (int j = 0; j < k; ++j) { double t = m[k][j]; m[k][j] = m[big][j]; m[big][j] = t; }
            for (int i = 0; i < s; ++i) { double t = m[3 - s + i][k]; m[3 - s + i][k] = m[3 - s + i][big]; m[3 - s + i][big] = t; }
            { double t = m[k][k]; m[k][k] = m[big][big]; m[big][big] = t; }
            for (int i = k + 1; i < big; ++i) { double t = m[i][k]; m[i][k] = m[big][i]; m[big][i] = t; }
        }
        const int rs = 3 - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double acc = m[k][0] * temp[0];
            for (int j = 1; j < k; ++j) acc = acc + m[k][j] * temp[j];
            m[k][k] -= acc;
            for (int i = 0; i < rs; ++i) {
                double a2 = m[k + 1 + i][0] * temp[0];
                for (int j = 1; j < k; ++j) a2 = a2 + m[k + 1 + i][j] * temp[j];
                m[k + 1 + i][k] -= a2;
            }
        }
        const double akk = m[k][k];
        const bool pivot_ok = fabs(akk) > 0.0;
        if (k == 0 && !pivot_ok) { tr[0] = 0; tr[1] = 1; tr[2] = 2; break; }
        if (rs > 0 && pivot_ok)
            for (int i = 0; i < rs; ++i) m[k + 1 + i][k] /= akk;
    }
    double d[3] = {b[0], b[1], b[2]};
    for (int k = 0; k < 3; ++k)
        if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
    d[1] -= m[1][0] * d[0];
    d[2] -= (m[2][0] * d[0] + m[2][1] * d[1]);
    const double tol = 1.0 / 1.7976931348623157e308;
    for (int i = 0; i < 3; ++i) { if (fabs(m[i][i]) > tol) d[i] /= m[i][i]; else d[i] = 0.0; }
    d[1] -= m[2][1] * d[2];
    d[0] -= (m[1][0] * d[1] + m[2][0] * d[2]);
    for (int k = 2; k >= 0; --k)
        if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
    x[0] = d[0]; x[1] = d[1]; x[2] = d[2];
}

// Eigen LLT on selfadjointView<Upper> + solve for a 3x3 (LevenbergMarquard::step, src/nlls/levenberg_marquardt.cpp:76-77;
// published algorithm: llt_inplace::unblocked and the two triangular solves)
__device__ inline void llt3_solve(const double A[3][3] /*upper used*/, const double b[3], double x[3])
{
    const double l00 = sqrt(A[0][0]);
    const double l10 = A[0][1] / l00, l20 = A[0][2] / l00;
    const double l11 = sqrt(A[1][1] - l10 * l10);
    const double l21 = (A[1][2] - l20 * l10) / l11;
    const double l22 = sqrt(A[2][2] - (l20 * l20 + l21 * l21));
    const double y0 = b[0] / l00;
    const double y1 = (b[1] - l10 * y0) / l11;
    const double y2 = (b[2] - (l20 * y0 + l21 * y1)) / l22;
    x[2] = y2 / l22;
    x[1] = (y1 - l21 * x[2]) / l11;
    x[0] = (y0 - (l10 * x[1] + l20 * x[2])) / l00;
}

// ------------------------------------------------------------------------------------------------
// map addressing
// ------------------------------------------------------------------------------------------------
// Map::w2m_nocast / w2m (include/lama/sdm/map.h:125-138): tf_ * v = scale*v + off
__device__ inline double w2m_nocast(const DevParams& prm, double v) { return prm.scale * v + prm.off; }
__device__ inline uint32_t w2m(const DevParams& prm, double v) { return (uint32_t)(w2m_nocast(prm, v) + 0.5); }

// DynamicDistanceMap::distance(Vector3ui) (src/sdm/dynamic_distance_map.cpp:140-147) through the const
// Map::get (src/sdm/map.cpp:414-455): absent patch / mask bit off / !valid_obstacle -> max distance.
// An off-bit cell is still all-zero, hence reads as !valid: the mask needs no separate test.
__device__ inline double dm_distance_cell(const DevParams& prm, const int16_t* __restrict__ dir,
                                          const sv_t* __restrict__ sv, uint32_t x, uint32_t y)
{
    const uint32_t rx = x - prm.wx0, ry = y - prm.wy0;
    if (rx >= prm.WC || ry >= prm.WC) return prm.maxdist;
    const int slot = dir[(ry >> 5) * prm.W + (rx >> 5)];
    if (slot < 0) return prm.maxdist;
    const sv_t v = sv[(uint32_t)slot * 1024u + ((rx & 31u) | ((ry & 31u) << 5))];
    if (!(v & SV_VALID)) return prm.maxdist;
    return sqrt((double)(v & SV_SQMASK)) * prm.resolution;
}

// DynamicDistanceMap::distance(Vector3d, Vector3d*) 2-D branch (src/sdm/dynamic_distance_map.cpp:66-91)
__device__ inline double dm_distance(const DevParams& prm, const int16_t* __restrict__ dir,
                                     const sv_t* __restrict__ sv, double wx, double wy, double* gx, double* gy)
{
    const double mx = w2m_nocast(prm, wx), my = w2m_nocast(prm, wy);
    const uint32_t dx = (uint32_t)mx, dy = (uint32_t)my;
    const double mu0 = mx - (double)dx, mu1 = my - (double)dy;
    const double muinv0 = 1.0 - mu0, muinv1 = 1.0 - mu1;
    const double v0 = dm_distance_cell(prm, dir, sv, dx, dy);
    const double v1 = dm_distance_cell(prm, dir, sv, dx + 1, dy);
    const double v2 = dm_distance_cell(prm, dir, sv, dx, dy + 1);
    const double v3 = dm_distance_cell(prm, dir, sv, dx + 1, dy + 1);
    const double dist = v0 * muinv0 * muinv1 + v1 * muinv1 * mu0 + v2 * muinv0 * mu1 + v3 * mu0 * mu1;
    if (gx) {
        *gx = -((v0 - v1) * muinv1 + (v2 - v3) * mu1) * prm.scale;
        *gy = -((v0 - v2) * muinv0 + (v1 - v3) * mu0) * prm.scale;
    }
    return dist;
}

// CauchyWeight(0.15)::value (src/nlls/robust_cost.cpp:66-73)
__device__ inline double cauchy015(double x)
{
    const double c_ = 1.0 / (0.15 * 0.15);
    return (1.0 / (1.0 + x * x * c_));
}

// ------------------------------------------------------------------------------------------------
// brushfire queue storage: first `lds_cap` entries in LDS, the rest in the particle's HBM region
// ------------------------------------------------------------------------------------------------
struct HybridStore {
    uint64_t* lds;
    uint64_t* glb;
    uint32_t lds_cap;
    __device__ inline uint64_t get(uint32_t i) const { return i < lds_cap ? lds[i] : glb[i]; }
    __device__ inline void set(uint32_t i, uint64_t v) const { if (i < lds_cap) lds[i] = v; else glb[i] = v; }
};

// queue entry: [63:48] priority | [47:40] oy+128 | [39:32] ox+128 | [31:16] ry | [15:0] rx.  (ox, oy) = the obstacle
// offset the cell had when it was queued: lets a pop fetch the obstacle cell in the same round as the cell itself.
// Wide build: [63:48] priority | [47:39] oy+256 | [38:30] ox+256 | [29:15] ry | [14:0] rx (a window is at most 1016 patches =
// 32,512 cells a side: 15 bits).
#ifdef LAMA_WIDE_DM
__device__ inline uint64_t q_entry(uint32_t prio, int rx, int ry, int ox = 0, int oy = 0)
{
    return ((uint64_t)prio << 48) | ((uint64_t)(uint32_t)((oy + 256) & 0x1FF) << 39) | ((uint64_t)(uint32_t)((ox + 256) & 0x1FF) << 30) |
           ((uint64_t)(uint32_t)ry << 15) | (uint32_t)rx;
}
__device__ inline int q_ox(uint64_t e) { return (int)((e >> 30) & 0x1FFu) - 256; }
__device__ inline int q_oy(uint64_t e) { return (int)((e >> 39) & 0x1FFu) - 256; }
__device__ inline int q_rx(uint64_t e) { return (int)(e & 0x7FFFu); }
__device__ inline int q_ry(uint64_t e) { return (int)((e >> 15) & 0x7FFFu); }
#else
__device__ inline uint64_t q_entry(uint32_t prio, int rx, int ry, int ox = 0, int oy = 0)
{
    return ((uint64_t)prio << 48) | ((uint64_t)(uint32_t)((oy + 128) & 0xFF) << 40) | ((uint64_t)(uint32_t)((ox + 128) & 0xFF) << 32) |
           ((uint32_t)ry << 16) | (uint32_t)rx;
}
__device__ inline int q_ox(uint64_t e) { return (int)((e >> 32) & 0xFFu) - 128; }
__device__ inline int q_oy(uint64_t e) { return (int)((e >> 40) & 0xFFu) - 128; }
__device__ inline int q_rx(uint64_t e) { return (int)(e & 0xFFFFu); }
__device__ inline int q_ry(uint64_t e) { return (int)((e >> 16) & 0xFFFFu); }
#endif

// True when the allocation phase of this map update failed (see ERR_CLEAN_ABORT): the calling kernel must not touch the maps.
__device__ inline bool map_update_aborted(const DevParams& prm)
{
    return (__hip_atomic_load(prm.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & (ERR_WINDOW | ERR_DM_CAP | ERR_OCC_CAP | ERR_CLEAN_ABORT)) != 0;
}

__device__ inline uint32_t pack_obs(int ox, int oy) { return ((uint32_t)(uint16_t)(int16_t)ox) | (((uint32_t)(uint16_t)(int16_t)oy) << 16); }
__device__ inline int obs_x(uint32_t o) { return (int)(int16_t)(o & 0xFFFFu); }
__device__ inline int obs_y(uint32_t o) { return (int)(int16_t)(o >> 16); }

} // namespace lama_dev
