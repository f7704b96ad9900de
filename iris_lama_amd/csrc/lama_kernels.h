// lama_kernels.h -- hand-written HIP kernels (gfx950, wave64) of the particle-filter scan-matching path.
//
//   k_scan_match   : region 1 of PFSlam2D::update (src/pf_slam2d.cpp:254-266): one workgroup per particle,
//                    whole Gauss-Newton loop on the device, wave-shuffle reduction of the 10 normal-
//                    equation scalars, fp64 (map coordinates are ~4.2e7 cells: fp32 is not an option).
//   k_update_maps  : region 2 (src/pf_slam2d.cpp:292-302): ray-cast + occupancy counters + obstacle events
//                    + exact dynamic brushfire, one wave per particle.
//   k_copy_particles: resample() / first-scan cloning (src/pf_slam2d.cpp:204-216, 558-574).
//   k_loglik_batch : calculateLikelihood for B poses on one map (src/pf_slam2d.cpp:393-414).
//
// All of this is HBM/latency-bound gather/scatter + integer RMW work: no MFMA anywhere.
#pragma once
#include "lama_dev.h"

namespace lama_dev {

constexpr int SM_BLOCK = 256;       // scan-match workgroup: 4 waves
constexpr int UM_BLOCK = 64;        // update-maps workgroup: 1 wave
constexpr int LQ_LDS = 3072;        // lower-queue entries kept in LDS (24 KiB)
constexpr int RQ_LDS = 1024;        // raise-queue entries kept in LDS (8 KiB)

// ------------------------------------------------------------------------------------------------
// wave / block reductions (fixed shape => results do not depend on how particles are sharded)
// ------------------------------------------------------------------------------------------------
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

template <int NV>
__device__ inline void block_sum(double (&acc)[NV], double* sh /*[SM_BLOCK/64][NV]*/, double* out /*[NV]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        double v = wave_sum(acc[k]);
        if (lane == 0) sh[wave * NV + k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = sh[threadIdx.x];
        for (int w = 1; w < SM_BLOCK / 64; ++w) s += sh[w * NV + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// MatchSurface2D::eval (src/match_surface_2d.cpp:42-90) fused with the robust weighting of
// Solver::solve (src/nlls/solver.cpp:74-79, 92-96) and the J^T J / J^T r / chi2 products of
// GaussNewton::step (src/nlls/gauss_newton.cpp:55-56, 64).
//   acc[0..5] = lower triangle of A (00,10,11,20,21,22), acc[6..8] = g, acc[9] = chi2
// ------------------------------------------------------------------------------------------------
__device__ inline void eval_beams_jac(const DevParams& prm, const int16_t* dir, const uint16_t* sv,
                                      const double* __restrict__ pts, int n, const Affine& tf, double (&acc)[10])
{
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < n; i += SM_BLOCK) {
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const double hx = ((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0];
        const double hy = ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1];
        double gx, gy;
        double r = dm_distance(prm, dir, sv, hx, hy, &gx, &gy);
        const double w = sqrt(cauchy015(r));
        r *= w;
        const double j0 = gx * w, j1 = gy * w, j2 = (gy * hx - gx * hy) * w;
        acc[0] += j0 * j0; acc[1] += j1 * j0; acc[2] += j1 * j1;
        acc[3] += j2 * j0; acc[4] += j2 * j1; acc[5] += j2 * j2;
        acc[6] += j0 * r;  acc[7] += j1 * r;  acc[8] += j2 * r;
        acc[9] += r * r;
    }
}

// residual-only evaluation: acc[0] = sum (w r)^2 (validation, solver.cpp:90-96),
//                           acc[1] = sum -(d*d)/meas_sigma (calculateLikelihood, pf_slam2d.cpp:393-414)
__device__ inline void eval_beams_res(const DevParams& prm, const int16_t* dir, const uint16_t* sv,
                                      const double* __restrict__ pts, int n, const Affine& tf, double (&acc)[2])
{
    acc[0] = 0.0; acc[1] = 0.0;
    for (int i = threadIdx.x; i < n; i += SM_BLOCK) {
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const double hx = ((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0];
        const double hy = ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1];
        const double d = dm_distance(prm, dir, sv, hx, hy, nullptr, nullptr);
        const double wr = d * sqrt(cauchy015(d));
        acc[0] += wr * wr;
        acc[1] += -(d * d) / prm.meas_sigma;
    }
}

struct SMShared {
    double red[(SM_BLOCK / 64) * 10];
    double tot[10];
    Affine tf;
    SE2 state;
    double h[3];
    int ctl;      // 0 = continue, 1 = stop
};

// ------------------------------------------------------------------------------------------------
// k_scan_match: PFSlam2D::scanMatch (src/pf_slam2d.cpp:416-437) for every particle of the shard.
// Solver::solve loop (src/nlls/solver.cpp:67-107) with GaussNewton (gauss_newton.cpp:53-91).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SM_BLOCK) void k_scan_match(DevParams prm, const double* __restrict__ pts, int n, Affine mtf,
                                                          double* __restrict__ loglik_out, int32_t* __restrict__ iters_out)
{
    __shared__ SMShared sh;
    const int p = blockIdx.x;
    const int16_t* dir = prm.dm_dir + (size_t)p * prm.W * prm.W;
    const uint16_t* sv = prm.dm_sv + (size_t)p * prm.dm_cap * 1024;
    const double eps1 = 1e-4, eps2 = 1e-4;

    if (threadIdx.x == 0) {
        const double* q = prm.poses + 4 * p;
        sh.state = SE2{q[0], q[1], q[2], q[3]};
        sh.tf = scan_tf(sh.state, mtf);
        sh.ctl = 0;
    }
    __syncthreads();

    uint32_t iter = 0, evals = 0;
    bool numeric_ok = true;
    while (iter < prm.max_iter) {
        // 1. residuals + Jacobian at the current state, weighted, reduced
        double acc[10];
        {
            const Affine tf = sh.tf;
            eval_beams_jac(prm, dir, sv, pts, n, tf, acc);
        }
        block_sum<10>(acc, sh.red, sh.tot);
        ++evals;
        // 2. Gauss-Newton step (one thread; 3x3)
        if (threadIdx.x == 0) {
            const double* t = sh.tot;
            const double g[3] = {t[6], t[7], t[8]};
            const double max_abs_g = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
            int stop = 0;
            if (max_abs_g < eps1) {
                stop = 1;                                   // h = 0, not applied
            } else {
                const double A[3][3] = {{t[0], 0, 0}, {t[1], t[2], 0}, {t[3], t[4], t[5]}};
                const double mg[3] = {-g[0], -g[1], -g[2]};
                double h[3];
                ldlt3_solve(A, mg, h);
                const double max_abs_h = fmax(fabs(h[0]), fmax(fabs(h[1]), fabs(h[2])));
                if (max_abs_h < eps2) stop = 1;             // returned but NOT applied (solver.cpp:84-86)
                sh.h[0] = h[0]; sh.h[1] = h[1]; sh.h[2] = h[2];
                if (!stop) {
                    bool ok = true;
                    sh.state = se2_exp_mul(h, sh.state, ok);    // problem.update(h)
                    if (!ok) atomicOr(prm.err, ERR_NUMERIC);
                    sh.tf = scan_tf(sh.state, mtf);
                }
            }
            sh.ctl = stop;
        }
        __syncthreads();
        if (sh.ctl) break;
        // 3. validation: residuals at the updated state
        double a2[2];
        {
            const Affine tf = sh.tf;
            eval_beams_res(prm, dir, sv, pts, n, tf, a2);
        }
        block_sum<2>(a2, sh.red, sh.tot + 0);   // tot[9] (chi2) is overwritten? no: tot[0..1] only
        ++evals;
        if (threadIdx.x == 0) {
            // NB: block_sum<2> wrote tot[0..1]; chi2 of step 1 was tot[9] (untouched)
            const double dF = sh.tot[9] - sh.tot[0];
            int stop = 0;
            if (!(dF > 0)) {
                stop = 1;                                   // invalid: revert (solver.cpp:99-102)
                const double mh[3] = {-sh.h[0], -sh.h[1], -sh.h[2]};
                bool ok = true;
                sh.state = se2_exp_mul(mh, sh.state, ok);
                if (!ok) atomicOr(prm.err, ERR_NUMERIC);
                sh.tf = scan_tf(sh.state, mtf);
            }
            sh.ctl = stop;
        }
        ++iter;
        __syncthreads();
        if (sh.ctl) break;
    }
    (void)numeric_ok;
    // likelihood at the final state (pf_slam2d.cpp:433-436)
    double a2[2];
    {
        const Affine tf = sh.tf;
        eval_beams_res(prm, dir, sv, pts, n, tf, a2);
    }
    block_sum<2>(a2, sh.red, sh.tot);
    ++evals;
    if (threadIdx.x == 0) {
        double* q = prm.poses + 4 * p;
        q[0] = sh.state.c; q[1] = sh.state.s; q[2] = sh.state.tx; q[3] = sh.state.ty;
        loglik_out[p] = sh.tot[1];
        iters_out[p] = (int32_t)iter;
        prm.stats[4 * p + 0] = iter;
        prm.stats[4 * p + 1] = evals;
    }
}

// calculateLikelihood for B poses against particle `particle`'s distance map
__global__ __launch_bounds__(SM_BLOCK) void k_loglik_batch(DevParams prm, int particle, const double* __restrict__ pts, int n,
                                                            Affine mtf, const double* __restrict__ poses, double* __restrict__ out)
{
    __shared__ double red[(SM_BLOCK / 64) * 2];
    __shared__ double tot[2];
    __shared__ Affine tfs;
    const int b = blockIdx.x;
    const int16_t* dir = prm.dm_dir + (size_t)particle * prm.W * prm.W;
    const uint16_t* sv = prm.dm_sv + (size_t)particle * prm.dm_cap * 1024;
    if (threadIdx.x == 0) {
        const double* q = poses + 4 * b;
        tfs = scan_tf(SE2{q[0], q[1], q[2], q[3]}, mtf);
    }
    __syncthreads();
    double a2[2];
    const Affine tf = tfs;
    eval_beams_res(prm, dir, sv, pts, n, tf, a2);
    block_sum<2>(a2, red, tot);
    if (threadIdx.x == 0) out[b] = tot[1];
}

// ------------------------------------------------------------------------------------------------
// k_update_maps -- PFSlam2D::updateParticleMaps (src/pf_slam2d.cpp:439-509), one wave per particle.
//
// Exactness contract: the uint16 counters AND the order of add/remove obstacle events AND the order in
// which the brushfire pops equal-priority cells are those of the reference's sequential code, so the
// resulting maps are bit-identical (cells, obstacle offsets, flags, masks, patch sets).
//   * beams are processed in order; within a beam the hit cell first (lane 0), then the Bresenham cells
//     64 at a time (cells of one ray are distinct => plain RMW, no atomics); the closed form
//     steps_j(t) = floor((2 t |d_j| + n) / (2 n)) replays Map::computeRay (src/sdm/map.cpp:198-227);
//   * events are appended to the raise/lower queues in (beam, step) order via ballot ranks (priority 0
//     => libstdc++ push_heap leaves them at the end);
//   * DynamicDistanceMap::update (src/sdm/dynamic_distance_map.cpp:160-197) runs on lane 0 with the
//     libstdc++-exact heap of lama_heap.h.
// ------------------------------------------------------------------------------------------------
struct UMState {
    const DevParams& prm;
    int p;
    int16_t* dm_dir; int16_t* occ_dir;
    uint16_t* dm_sv; uint32_t* dm_obs; uint64_t* dm_mask;
    uint32_t* occ; uint64_t* occ_mask;
    int dm_count, occ_count;       // wave-uniform
};

// Wave-cooperative "non-const Map::get" patch lookup (src/sdm/map.cpp:371-412): every lane with `want`
// gets the slot of window patch `pidx`, allocating missing patches once per distinct patch.
// Must be called by all 64 lanes.  Returns -1 on capacity overflow.
__device__ inline int coop_slot(int16_t* dir, int& count, int cap, bool want, uint32_t pidx, int errbit, int32_t* err)
{
    int slot = want ? (int)dir[pidx] : 0;
    bool need = want && slot < 0;
    unsigned long long m = __ballot(need);
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t lp = __shfl(pidx, leader, 64);
        int ns;
        if (count < cap) { ns = count; ++count; } else { ns = -1; }
        if ((int)(threadIdx.x & 63) == leader) {
            if (ns >= 0) dir[lp] = (int16_t)ns; else atomicOr(err, errbit);
        }
        if (need && pidx == lp) { slot = ns; need = false; }
        m = __ballot(need);
    }
    return slot;
}

struct BfCtx {
    const DevParams& prm;
    int16_t* dir; uint16_t* sv; uint32_t* obs; uint64_t* mask;
    int& count;
    HybridStore lower, raise;
    uint32_t nl, nr;     // queue sizes
    uint64_t processed;
};

// serial (single-lane) non-const get on the DM: allocate + set mask bit; returns slot*1024+cell or -1
__device__ inline int bf_get(BfCtx& c, int rx, int ry)
{
    if ((uint32_t)rx >= c.prm.WC || (uint32_t)ry >= c.prm.WC) { atomicOr(c.prm.err, ERR_WINDOW); return -1; }
    const uint32_t pidx = ((uint32_t)ry >> 5) * c.prm.W + ((uint32_t)rx >> 5);
    int slot = c.dir[pidx];
    if (slot < 0) {
        if (c.count >= (int)c.prm.dm_cap) { atomicOr(c.prm.err, ERR_DM_CAP); return -1; }
        slot = c.count++;
        c.dir[pidx] = (int16_t)slot;
    }
    const uint32_t ci = ((uint32_t)rx & 31u) | (((uint32_t)ry & 31u) << 5);
    uint64_t* w = c.mask + (size_t)slot * 16 + (ci >> 6);
    const uint64_t bit = 1ull << (ci & 63);
    const uint64_t cur = *w;
    if (!(cur & bit)) *w = cur | bit;
    return slot * 1024 + (int)ci;
}

__device__ inline void bf_push(BfCtx& c, bool to_lower, uint32_t prio, int rx, int ry)
{
    uint32_t& n = to_lower ? c.nl : c.nr;
    if (n >= c.prm.qcap) { atomicOr(c.prm.err, ERR_QUEUE); return; }
    if (to_lower) heap_push(c.lower, c.nl, q_entry(prio, rx, ry));
    else heap_push(c.raise, c.nr, q_entry(prio, rx, ry));
}

// DynamicDistanceMap::raise (src/sdm/dynamic_distance_map.cpp:244-279)
__device__ inline void bf_raise(BfCtx& c, int rx, int ry, int cur)
{
    const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int nx = rx + DX[i], ny = ry + DY[i];
        const int nc = bf_get(c, nx, ny);
        if (nc < 0) continue;
        const uint16_t s = c.sv[nc];
        if ((s & SV_QUEUED) || !(s & SV_VALID)) continue;
        const uint32_t no = c.obs[nc];
        const int oc = bf_get(c, nx + obs_x(no), ny + obs_y(no));
        if (oc < 0) continue;
        if (!(c.sv[oc] & SV_VALID)) {
            bf_push(c, false, s & SV_SQMASK, nx, ny);
            c.sv[nc] = SV_QUEUED;          // sqdist 0, !valid, queued
            c.obs[nc] = 0;
        } else {                            // (!is_queued already known)
            bf_push(c, true, s & SV_SQMASK, nx, ny);
            c.sv[nc] = s | SV_QUEUED;
        }
    }
    c.sv[cur] &= (uint16_t)~SV_QUEUED;
}

// DynamicDistanceMap::lower (src/sdm/dynamic_distance_map.cpp:281-330)
__device__ inline void bf_lower(BfCtx& c, int rx, int ry, int cur)
{
    const uint16_t s = c.sv[cur];
    if (!(s & SV_QUEUED)) return;
    const uint32_t co = c.obs[cur];
    const int cox = obs_x(co), coy = obs_y(co);
    const int obx = rx + cox, oby = ry + coy;      // absolute (window) position of the carried obstacle
    const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (DX[i] * cox > 0 || DY[i] * coy > 0) continue;          // only away from the obstacle (:296)
        const int nx = rx + DX[i], ny = ry + DY[i];
        const int nc = bf_get(c, nx, ny);
        if (nc < 0) continue;
        const uint16_t ns = c.sv[nc];
        const int ddx = nx - obx, ddy = ny - oby;
        const uint32_t new_sq = (uint32_t)(ddx * ddx + ddy * ddy);
        const uint32_t cmp = (ns & SV_VALID) ? (uint32_t)(ns & SV_SQMASK) : c.prm.max_sqdist;
        bool over = new_sq < cmp;
        if (!over && new_sq == (uint32_t)(ns & SV_SQMASK)) {         // :311-317
            const uint32_t nobs = c.obs[nc];
            const int oc = bf_get(c, nx + obs_x(nobs), ny + obs_y(nobs));
            if (oc >= 0) {
                const uint16_t os = c.sv[oc];
                if (!(ns & SV_VALID) || !((os & SV_VALID) && (os & SV_SQMASK) == 0)) over = true;
            }
        }
        if (over) {
            bf_push(c, true, new_sq, nx, ny);
            c.sv[nc] = (uint16_t)(SV_VALID | SV_QUEUED | (new_sq & SV_SQMASK));
            c.obs[nc] = pack_obs(obx - nx, oby - ny);
        }
    }
    c.sv[cur] &= (uint16_t)~SV_QUEUED;
}

// DynamicDistanceMap::update (src/sdm/dynamic_distance_map.cpp:160-197)
__device__ inline void bf_update(BfCtx& c)
{
    while (c.nr > 0) {
        const uint64_t e = heap_pop(c.raise, c.nr);
        const int rx = q_rx(e), ry = q_ry(e);
        const int cur = bf_get(c, rx, ry);
        ++c.processed;
        if (cur < 0) continue;
        bf_raise(c, rx, ry, cur);
    }
    while (c.nl > 0) {
        const uint64_t e = heap_pop(c.lower, c.nl);
        const int rx = q_rx(e), ry = q_ry(e);
        const int cur = bf_get(c, rx, ry);
        ++c.processed;
        if (cur < 0) continue;
        const uint16_t s = c.sv[cur];
        if (s & SV_VALID) {
            const uint32_t o = c.obs[cur];
            const int oc = bf_get(c, rx + obs_x(o), ry + obs_y(o));
            if (oc < 0) continue;
            if ((c.sv[oc] & SV_SQMASK) == 0) bf_lower(c, rx, ry, cur);   // :191 (valid NOT tested)
        }
    }
}

__global__ __launch_bounds__(UM_BLOCK) void k_update_maps(DevParams prm, const double* __restrict__ pts, int n,
                                                           const double* __restrict__ tfs /*[P][12]*/, int first_particle)
{
    __shared__ uint64_t lds_lower[LQ_LDS];
    __shared__ uint64_t lds_raise[RQ_LDS];
    const int p = first_particle + blockIdx.x;
    const int lane = threadIdx.x;
    const size_t WW = (size_t)prm.W * prm.W;
    int16_t* dm_dir = prm.dm_dir + (size_t)p * WW;
    int16_t* occ_dir = prm.occ_dir + (size_t)p * WW;
    uint16_t* dm_sv = prm.dm_sv + (size_t)p * prm.dm_cap * 1024;
    uint32_t* dm_obs = prm.dm_obs + (size_t)p * prm.dm_cap * 1024;
    uint64_t* dm_mask = prm.dm_mask + (size_t)p * prm.dm_cap * 16;
    uint32_t* occ = prm.occ + (size_t)p * prm.occ_cap * 1024;
    uint64_t* occ_mask = prm.occ_mask + (size_t)p * prm.occ_cap * 16;
    int dm_count = prm.counts[2 * p], occ_count = prm.counts[2 * p + 1];

    HybridStore lower{lds_lower, prm.q_lower + (size_t)p * prm.qcap, (uint32_t)LQ_LDS};
    HybridStore raise{lds_raise, prm.q_raise + (size_t)p * prm.qcap, (uint32_t)RQ_LDS};
    uint32_t nl = 0, nr = 0;            // wave-uniform queue sizes
    uint64_t ray_cells = 0;

    // tf = fixed_tf * moving_tf of this particle (12 doubles, computed on the host with libm exactly
    // like the reference does on the CPU, so the integer cell coordinates below are reproducible)
    double T[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) T[k] = tfs[12 * (size_t)p + k];
    const double wsx = T[9], wsy = T[10], wsz = T[11];   // wso = tf.translation()

    for (int i = 0; i < n; ++i) {
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        double hx = ((T[0] * px + T[1] * py) + T[2] * pz) + T[9];
        double hy = ((T[3] * px + T[4] * py) + T[5] * pz) + T[10];
        double hz = ((T[6] * px + T[7] * py) + T[8] * pz) + T[11];
        double sx = wsx, sy = wsy, sz = wsz;
        double abx = 0, aby = 0, abz = 0, ray_length = 1.0;
        bool mark_hit = true;
        if (prm.trunc_range > 0.0) {                                        // pf_slam2d.cpp:467-479
            abx = hx - sx; aby = hy - sy; abz = hz - sz;
            ray_length = sqrt((abx * abx + aby * aby) + abz * abz);
            if (prm.trunc_range < ray_length) {
                hx = sx + abx / ray_length * prm.trunc_range;
                hy = sy + aby / ray_length * prm.trunc_range;
                hz = sz + abz / ray_length * prm.trunc_range;
                mark_hit = false;
            }
        }
        if (mark_hit && prm.trunc_ray > 0.0) {                              // :481-491
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
        const uint32_t mhx = w2m(prm, hx), mhy = w2m(prm, hy), mhz = w2m(prm, hz);
        const uint32_t msx = w2m(prm, sx), msy = w2m(prm, sy), msz = w2m(prm, sz);

        // ---- hit cell: setOccupied -> addObstacle (lane 0) ------------------------------- :493-498
        {
            const uint32_t rx = mhx - prm.wx0, ry = mhy - prm.wy0;
            const bool inwin = rx < prm.WC && ry < prm.WC;
            if (mark_hit && !inwin && lane == 0) atomicOr(prm.err, ERR_WINDOW);
            const bool want = mark_hit && inwin && lane == 0;
            const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5);
            const uint32_t ci = (rx & 31u) | ((ry & 31u) << 5);
            const int slot = coop_slot(occ_dir, occ_count, (int)prm.occ_cap, want, pidx, ERR_OCC_CAP, prm.err);
            bool changed = false;
            if (want && slot >= 0) {
                uint32_t* cell = occ + (size_t)slot * 1024 + ci;
                const uint32_t v = *cell;
                uint32_t o = v & 0xFFFFu, vis = v >> 16;
                const bool occupied = vis != 0 && 4u * o > vis;             // prob > 0.25
                o = (o + 1) & 0xFFFFu; vis = (vis + 1) & 0xFFFFu;
                *cell = o | (vis << 16);
                changed = !occupied && (vis != 0 && 4u * o > vis);
                uint64_t* w = occ_mask + (size_t)slot * 16 + (ci >> 6);
                const uint64_t bit = 1ull << (ci & 63);
                if (!(*w & bit)) *w |= bit;
            }
            const int dslot = coop_slot(dm_dir, dm_count, (int)prm.dm_cap, changed, pidx, ERR_DM_CAP, prm.err);
            bool push = false;
            if (changed && dslot >= 0) {                                    // addObstacle :212-226
                uint64_t* w = dm_mask + (size_t)dslot * 16 + (ci >> 6);
                const uint64_t bit = 1ull << (ci & 63);
                if (!(*w & bit)) *w |= bit;
                const uint32_t di = (uint32_t)dslot * 1024u + ci;
                const uint16_t s = dm_sv[di];
                if (!((s & SV_VALID) && (s & SV_SQMASK) == 0)) {
                    dm_sv[di] = (uint16_t)(SV_VALID | SV_QUEUED);
                    dm_obs[di] = 0;
                    push = true;
                }
            }
            const unsigned long long pm = __ballot(push);
            if (pm) {
                if (nl >= prm.qcap) { if (lane == 0) atomicOr(prm.err, ERR_QUEUE); }
                else { if (push) lower.set(nl, q_entry(0, (int)rx, (int)ry)); nl += 1; }
            }
        }

        // ---- free cells: computeRay(w2m(start), mhit, setFree -> removeObstacle) --------- :500-504
        const int64_t d0 = (int64_t)mhx - (int64_t)msx, d1 = (int64_t)mhy - (int64_t)msy, d2 = (int64_t)mhz - (int64_t)msz;
        const int64_t a0 = d0 < 0 ? -d0 : d0, a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
        const int64_t nn = a0 > a1 ? (a0 > a2 ? a0 : a2) : (a1 > a2 ? a1 : a2);
        const int s0 = d0 < 0 ? -1 : 1, s1 = d1 < 0 ? -1 : 1;
        const int steps = (int)nn - 1;                       // cells t = 1 .. n-1
        if (steps > 0) ray_cells += (uint64_t)steps;
        for (int base = 0; base < steps; base += 64) {
            const int t = base + lane + 1;
            const bool act = t <= steps;
            const int64_t st0 = (2 * (int64_t)t * a0 + nn) / (2 * nn);
            const int64_t st1 = (2 * (int64_t)t * a1 + nn) / (2 * nn);
            const uint32_t cx = (uint32_t)((int64_t)msx + s0 * st0), cy = (uint32_t)((int64_t)msy + s1 * st1);
            const uint32_t rx = cx - prm.wx0, ry = cy - prm.wy0;
            const bool inwin = rx < prm.WC && ry < prm.WC;
            if (act && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool want = act && inwin;
            const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5);
            const uint32_t ci = (rx & 31u) | ((ry & 31u) << 5);
            const int slot = coop_slot(occ_dir, occ_count, (int)prm.occ_cap, want, pidx, ERR_OCC_CAP, prm.err);
            bool changed = false;
            if (want && slot >= 0) {                                        // setFree :65-74
                uint32_t* cell = occ + (size_t)slot * 1024 + ci;
                const uint32_t v = *cell;
                const uint32_t o = v & 0xFFFFu;
                uint32_t vis = v >> 16;
                const bool was_free = vis != 0 && 4u * o < vis;             // prob < 0.25
                vis = (vis + 1) & 0xFFFFu;
                *cell = o | (vis << 16);
                changed = !was_free && (vis != 0 && 4u * o < vis);
                const uint64_t bit = 1ull << (ci & 63);
                uint64_t* w = occ_mask + (size_t)slot * 16 + (ci >> 6);
                if (!(*w & bit)) atomicOr((unsigned long long*)w, (unsigned long long)bit);
            }
            const int dslot = coop_slot(dm_dir, dm_count, (int)prm.dm_cap, changed, pidx, ERR_DM_CAP, prm.err);
            bool push = false;
            if (changed && dslot >= 0) {                                    // removeObstacle :228-242
                const uint64_t bit = 1ull << (ci & 63);
                uint64_t* w = dm_mask + (size_t)dslot * 16 + (ci >> 6);
                if (!(*w & bit)) atomicOr((unsigned long long*)w, (unsigned long long)bit);
                const uint32_t di = (uint32_t)dslot * 1024u + ci;
                const uint16_t s = dm_sv[di];
                if ((s & SV_VALID) && (s & SV_SQMASK) == 0) {
                    dm_sv[di] = SV_QUEUED;
                    dm_obs[di] = 0;
                    push = true;
                }
            }
            const unsigned long long pm = __ballot(push);
            if (pm) {
                const int cnt = __popcll(pm);
                if (nr + (uint32_t)cnt > prm.qcap) { if (lane == 0) atomicOr(prm.err, ERR_QUEUE); }
                else {
                    if (push) {
                        const int rank = __popcll(pm & ((1ull << lane) - 1ull));
                        raise.set(nr + (uint32_t)rank, q_entry(0, (int)rx, (int)ry));
                    }
                    nr += (uint32_t)cnt;
                }
            }
        }
        // make this beam's stores visible to the next beam's loads (same wave: ordering only)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    __syncthreads();

    // ---- dm->update() : exact sequential brushfire on lane 0 -------------------------------- :508
    uint64_t processed = 0;
    if (lane == 0) {
        BfCtx c{prm, dm_dir, dm_sv, dm_obs, dm_mask, dm_count, lower, raise, nl, nr, 0};
        bf_update(c);
        processed = c.processed;
        prm.counts[2 * p] = dm_count;
        prm.counts[2 * p + 1] = occ_count;
        prm.stats[4 * p + 2] = ray_cells;
        prm.stats[4 * p + 3] = processed;
    }
}

// ------------------------------------------------------------------------------------------------
// k_copy_particles -- dst particle i := src particle idx[i] (directories, counts, used slots of every
// plane); slots of dst beyond the source's count that the old dst owner had used are re-zeroed so the
// "unused slot == calloc'd" invariant holds.  grid = (P, 7 planes), 256 threads, 16-byte vectors.
// ------------------------------------------------------------------------------------------------
struct SetPtrs {
    int16_t* dm_dir; int16_t* occ_dir; uint16_t* dm_sv; uint32_t* dm_obs; uint64_t* dm_mask; uint32_t* occ; uint64_t* occ_mask; int32_t* counts;
};

__global__ __launch_bounds__(256) void k_copy_particles(SetPtrs dst, SetPtrs src, const int32_t* __restrict__ idx,
                                                         const int32_t* __restrict__ old_dst_counts, uint32_t W,
                                                         uint32_t dm_cap, uint32_t occ_cap, int same_set)
{
    const int i = blockIdx.x;
    const int j = idx[i];
    if (same_set && i == j) return;
    const int plane = blockIdx.y;
    const int sdm = src.counts[2 * j], socc = src.counts[2 * j + 1];
    const int odm = old_dst_counts[2 * i], oocc = old_dst_counts[2 * i + 1];
    const size_t WW = (size_t)W * W;
    const uint4* s; uint4* d; size_t ncopy, nzero;   // in 16-byte units
    switch (plane) {
    case 0: s = (const uint4*)(src.dm_dir + j * WW); d = (uint4*)(dst.dm_dir + i * WW); ncopy = WW * 2 / 16; nzero = 0; break;
    case 1: s = (const uint4*)(src.occ_dir + j * WW); d = (uint4*)(dst.occ_dir + i * WW); ncopy = WW * 2 / 16; nzero = 0; break;
    case 2: s = (const uint4*)(src.dm_sv + (size_t)j * dm_cap * 1024); d = (uint4*)(dst.dm_sv + (size_t)i * dm_cap * 1024);
            ncopy = (size_t)sdm * 2048 / 16; nzero = odm > sdm ? (size_t)(odm - sdm) * 2048 / 16 : 0; break;
    case 3: s = (const uint4*)(src.dm_obs + (size_t)j * dm_cap * 1024); d = (uint4*)(dst.dm_obs + (size_t)i * dm_cap * 1024);
            ncopy = (size_t)sdm * 4096 / 16; nzero = odm > sdm ? (size_t)(odm - sdm) * 4096 / 16 : 0; break;
    case 4: s = (const uint4*)(src.dm_mask + (size_t)j * dm_cap * 16); d = (uint4*)(dst.dm_mask + (size_t)i * dm_cap * 16);
            ncopy = (size_t)sdm * 128 / 16; nzero = odm > sdm ? (size_t)(odm - sdm) * 128 / 16 : 0; break;
    case 5: s = (const uint4*)(src.occ + (size_t)j * occ_cap * 1024); d = (uint4*)(dst.occ + (size_t)i * occ_cap * 1024);
            ncopy = (size_t)socc * 4096 / 16; nzero = oocc > socc ? (size_t)(oocc - socc) * 4096 / 16 : 0; break;
    default: s = (const uint4*)(src.occ_mask + (size_t)j * occ_cap * 16); d = (uint4*)(dst.occ_mask + (size_t)i * occ_cap * 16);
            ncopy = (size_t)socc * 128 / 16; nzero = oocc > socc ? (size_t)(oocc - socc) * 128 / 16 : 0; break;
    }
    for (size_t k = threadIdx.x; k < ncopy; k += 256) d[k] = s[k];
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t k = threadIdx.x; k < nzero; k += 256) d[ncopy + k] = z;
    if (plane == 0 && threadIdx.x == 0) { dst.counts[2 * i] = sdm; dst.counts[2 * i + 1] = socc; }
}

} // namespace lama_dev
