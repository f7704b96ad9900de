// lama_kernels.h -- hand-written HIP kernels (gfx950, wave64) of the particle-filter scan-matching path.
//
//   k_scan_match   : region 1 of PFSlam2D::update (src/pf_slam2d.cpp:254-266): one workgroup per particle,
//                    whole Gauss-Newton loop on the device, wave-shuffle reduction of the 10 normal-
//                    equation scalars, fp64 (map coordinates are ~4.2e7 cells: fp32 is not an option).
//   k_raycast      : region 2a (src/pf_slam2d.cpp:458-505): ray-cast + occupancy counters + obstacle events,
//   k_brushfire    : region 2b (:508): exact dynamic brushfire; both one wave per particle.
//   k_copy_particles: resample() / first-scan cloning (src/pf_slam2d.cpp:204-216, 558-574).
//   k_loglik_batch : calculateLikelihood for B poses on one map (src/pf_slam2d.cpp:393-414).
//
// All of this is HBM/latency-bound gather/scatter + integer RMW work: no MFMA anywhere.
#pragma once
#include "lama_dev.h"

namespace lama_dev {

constexpr int SM_BLOCK = 256;       // scan-match workgroup: 4 waves
constexpr int UM_BLOCK = 64;        // update-maps workgroup: 1 wave
// brushfire queue windows in LDS (entries): a small one for throughput (many waves per CU) and a big resume stage
#ifdef LAMA_TEST_SMALL_QUEUES       // tests/sim only: tiny first-stage queues, so that the hand-over to the resume stage happens mid-update
constexpr int LQ_SMALL = LAMA_TEST_SMALL_QUEUES, RQ_SMALL = LAMA_TEST_SMALL_QUEUES / 4;
#else
constexpr int LQ_SMALL = 1024, RQ_SMALL = 256;     // 8 + 2 KiB
#endif
constexpr int LQ_BIG = 8192, RQ_BIG = 2048;        // 64 + 16 KiB
constexpr uint32_t BF_TW_MAX_PARTICLES = 1u << 30;     // the helper-wave form wins at every measured particle count (30 .. 3000); cfg.brushfire_waves = 1 forces one wave

// ------------------------------------------------------------------------------------------------
// wave / block reductions (fixed shape => results do not depend on how particles are sharded)
// ------------------------------------------------------------------------------------------------
// One DPP step of a wave reduction on a double: v + (v of the lane the DPP control selects); rows not in `row_mask` add 0.0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const uint64_t u = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)u, CTRL, ROW_MASK, 0xF, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(u >> 32), CTRL, ROW_MASK, 0xF, false);
    return v + __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}

// Block sum of NV per-thread values with a FIXED tree (=> the result does not depend on how particles are sharded):
// within a wave pairs, quads, half rows, rows (quad_perm / row_half_mirror / row_mirror: both partners add the same two
// operands), then (row3 + row2) + (row1 + row0) through row_bcast15 / row_bcast31 into lane 63; the four waves are added in
// order by thread k.  All register-to-register (DPP), no LDS round trips: a __shfl_xor butterfly costs two ds_bpermute
// per step and value -- 3,800 cycles for the 11 sums of an evaluation pass.
template <int NV>
__device__ inline void block_sum(double (&acc)[NV], double* sh /*[SM_BLOCK/64][NV]*/, double* out /*[NV]*/)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = dpp_add<0xB1, 0xF>(acc[k]);        // quad_perm [1,0,3,2]
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = dpp_add<0x4E, 0xF>(acc[k]);        // quad_perm [2,3,0,1]
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = dpp_add<0x141, 0xF>(acc[k]);       // row_half_mirror
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = dpp_add<0x140, 0xF>(acc[k]);       // row_mirror: every lane holds its row's sum
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = dpp_add<0x142, 0xA>(acc[k]);       // row_bcast15 into rows 1, 3
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = dpp_add<0x143, 0xC>(acc[k]);       // row_bcast31 into rows 2, 3: lane 63 = total
    if (lane == 63) {
#pragma unroll
        for (int k = 0; k < NV; ++k) sh[wave * NV + k] = acc[k];
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
//   acc[0..5] = lower triangle of A (00,10,11,20,21,22), acc[6..8] = g, acc[9] = chi2,
//   acc[10]   = sum -(d*d)/meas_sigma of the UNWEIGHTED distances (calculateLikelihood, pf_slam2d.cpp:393-414)
// The same pass therefore serves as the validation of a step (chi2), as the linearisation of the next iteration (the
// reference evaluates the problem twice at that state, solver.cpp:90 and :71) and as the likelihood at the solution.
// ------------------------------------------------------------------------------------------------
constexpr int NJ = 11;

// The gathers of an evaluation pass, batched: a thread owns the beams t, t + 256, ... (SM_NB of them per batch).  Written
// beam by beam, every bilinear corner is a chain of two dependent loads (directory entry, then the cell) behind early-out
// branches -- with one wave per SIMD that is 8 round trips per beam.  Here the directory entries of ALL corners of ALL the
// thread's beams are requested first, then all cells, then the arithmetic runs in the original beam order (same
// operations, same summation order => bit-identical sums).  sqrt(sqdist) * resolution of the few hundred possible sqdist
// values comes from an LDS table built with the same expression (DynamicDistanceMap::distance(Vector3ui),
// src/sdm/dynamic_distance_map.cpp:140-147).
constexpr int SM_NB = 5;            // beams per thread and batch (1080 beams / 256 threads = 4.2)
constexpr int SM_LUT = 512;         // sqdist values with a table entry (max_sqdist is 100 / 400 in the reference's configurations;
                                    // contexts with a larger max_sqdist run the BIGSQ instantiations, which take the square root)

__device__ inline void sm_build_lut(const DevParams& prm, double* lut)
{
    for (int k = threadIdx.x; k < SM_LUT; k += SM_BLOCK) lut[k] = sqrt((double)k) * prm.resolution;
}

struct BeamCorners {
    double hx, hy, mu0, mu1;
    uint32_t off[4];                // directory index, then cell offset (slot * 1024 + cell), of the four corners
    uint32_t okm;                   // bit c: corner c is inside the window (stage 1) / has a patch (stage 2)
    uint32_t ci[4];
};

__device__ inline void sm_corners_begin(const DevParams& prm, const Affine& tf, double px, double py, double pz, BeamCorners& b)
{
    b.hx = ((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0];
    b.hy = ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1];
    const double mx = w2m_nocast(prm, b.hx), my = w2m_nocast(prm, b.hy);
    const uint32_t dx = (uint32_t)mx, dy = (uint32_t)my;
    b.mu0 = mx - (double)dx; b.mu1 = my - (double)dy;
    b.okm = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t rx = dx + (uint32_t)(c & 1) - prm.wx0, ry = dy + (uint32_t)(c >> 1) - prm.wy0;
        const bool in = rx < prm.WC && ry < prm.WC;
        b.off[c] = in ? (ry >> 5) * prm.W + (rx >> 5) : 0u;
        b.ci[c] = (rx & 31u) | ((ry & 31u) << 5);
        b.okm |= in ? (1u << c) : 0u;
    }
}

// value + gradient of DynamicDistanceMap::distance(Vector3d, Vector3d*) (src/sdm/dynamic_distance_map.cpp:66-91) from the
// four fetched cells
template <bool BIGSQ>
__device__ inline double sm_corners_finish(const DevParams& prm, const double* lut, const BeamCorners& b, const sv_t (&v)[4],
                                           double* gx, double* gy)
{
    double d[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t sq = v[c] & SV_SQMASK;
        const bool valid = ((b.okm >> c) & 1u) && (v[c] & SV_VALID);
        const double t = BIGSQ ? sqrt((double)sq) * prm.resolution : lut[sq & (uint32_t)(SM_LUT - 1)];   // valid => sq < max_sqdist
        d[c] = valid ? t : prm.maxdist;
    }
    const double mu0 = b.mu0, mu1 = b.mu1, muinv0 = 1.0 - mu0, muinv1 = 1.0 - mu1;
    const double dist = d[0] * muinv0 * muinv1 + d[1] * muinv1 * mu0 + d[2] * muinv0 * mu1 + d[3] * mu0 * mu1;
    if (gx) {
        *gx = -((d[0] - d[1]) * muinv1 + (d[2] - d[3]) * mu1) * prm.scale;
        *gy = -((d[0] - d[2]) * muinv0 + (d[1] - d[3]) * mu0) * prm.scale;
    }
    return dist;
}

// stages 1 + 2 for the thread's beams of the batch starting at `base`; returns the number of beams it owns there
__device__ inline int sm_gather(const DevParams& prm, const int16_t* __restrict__ dir, const sv_t* __restrict__ sv,
                                const double* __restrict__ pts, int n, int base, const Affine& tf, BeamCorners (&bc)[SM_NB],
                                sv_t (&cv)[SM_NB][4])
{
    int nb = 0;
#pragma unroll
    for (int b = 0; b < SM_NB; ++b) {
        const int i = base + b * SM_BLOCK + (int)threadIdx.x;
        const int ic = i < n ? i : n - 1;
        if (i < n) nb = b + 1;
        sm_corners_begin(prm, tf, pts[3 * ic], pts[3 * ic + 1], pts[3 * ic + 2], bc[b]);
    }
    int16_t sl[SM_NB][4];
#pragma unroll
    for (int b = 0; b < SM_NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) sl[b][c] = dir[bc[b].off[c]];
#pragma unroll
    for (int b = 0; b < SM_NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool have = ((bc[b].okm >> c) & 1u) && sl[b][c] >= 0;
            bc[b].off[c] = have ? (uint32_t)sl[b][c] * 1024u + bc[b].ci[c] : 0u;
            if (!have) bc[b].okm &= ~(1u << c);
        }
#pragma unroll
    for (int b = 0; b < SM_NB; ++b)
#pragma unroll
        for (int c = 0; c < 4; ++c) cv[b][c] = sv[bc[b].off[c]];
    return nb;
}

template <bool BIGSQ>
__device__ inline void eval_beams_jac(const DevParams& prm, const int16_t* dir, const sv_t* sv,
                                      const double* __restrict__ pts, int n, const Affine& tf, double (&acc)[NJ], const double* lut)
{
#pragma unroll
    for (int k = 0; k < NJ; ++k) acc[k] = 0.0;
    for (int base = 0; base < n; base += SM_NB * SM_BLOCK) {
        BeamCorners bc[SM_NB];
        sv_t cv[SM_NB][4];
        const int nb = sm_gather(prm, dir, sv, pts, n, base, tf, bc, cv);
#pragma unroll
        for (int b = 0; b < SM_NB; ++b) {
        {
        // no branch around a beam the thread does not own (its inputs are those of beam n - 1): the five chains can then be
        // interleaved by the scheduler; what is added for it is +0.0, which leaves every sum (never -0.0) unchanged
        const bool own = b < nb;
        const double hx = bc[b].hx, hy = bc[b].hy;
        double gx, gy;
        double r = sm_corners_finish<BIGSQ>(prm, lut, bc[b], cv[b], &gx, &gy);
        acc[10] += own ? -(r * r) / prm.meas_sigma : 0.0;
        const double w = sqrt(cauchy015(r));
        r *= w;
        const double j0 = gx * w, j1 = gy * w, j2 = (gy * hx - gx * hy) * w;
        acc[0] += own ? j0 * j0 : 0.0; acc[1] += own ? j1 * j0 : 0.0; acc[2] += own ? j1 * j1 : 0.0;
        acc[3] += own ? j2 * j0 : 0.0; acc[4] += own ? j2 * j1 : 0.0; acc[5] += own ? j2 * j2 : 0.0;
        acc[6] += own ? j0 * r : 0.0;  acc[7] += own ? j1 * r : 0.0;  acc[8] += own ? j2 * r : 0.0;
        acc[9] += own ? r * r : 0.0;
        }
        }
    }
}

// residual-only evaluation: acc[0] = sum (w r)^2 (validation, solver.cpp:90-96),
//                           acc[1] = sum -(d*d)/meas_sigma (calculateLikelihood, pf_slam2d.cpp:393-414)
template <bool BIGSQ>
__device__ inline void eval_beams_res(const DevParams& prm, const int16_t* dir, const sv_t* sv,
                                      const double* __restrict__ pts, int n, const Affine& tf, double (&acc)[2], const double* lut)
{
    acc[0] = 0.0; acc[1] = 0.0;
    for (int base = 0; base < n; base += SM_NB * SM_BLOCK) {
        BeamCorners bc[SM_NB];
        sv_t cv[SM_NB][4];
        const int nb = sm_gather(prm, dir, sv, pts, n, base, tf, bc, cv);
#pragma unroll
        for (int b = 0; b < SM_NB; ++b) {
            const bool own = b < nb;               // see eval_beams_jac
            const double d = sm_corners_finish<BIGSQ>(prm, lut, bc[b], cv[b], nullptr, nullptr);
            const double wr = d * sqrt(cauchy015(d));
            acc[0] += own ? wr * wr : 0.0;
            acc[1] += own ? -(d * d) / prm.meas_sigma : 0.0;
        }
    }
}

struct SMShared {
    double lut[SM_LUT];    // sqrt(sqdist) * resolution (sm_build_lut)
    double red[(SM_BLOCK / 64) * NJ];
    double tot[NJ];        // sums of the linearisation at the current state
    double tot2[NJ];       // sums at the trial state of the step being validated
    int have_lin;          // tot already holds the linearisation at the current state (taken over from the validation)
    int lin_is_final;      // on return: sh.tot was evaluated at exactly the returned state
    Affine tf;
    SE2 state;
    double h[3];
    int ctl;      // 0 = continue, 1 = stop
    // LevenbergMarquard state (src/nlls/levenberg_marquardt.cpp:49-54) and the sums of the last accepted linearisation
    double mu, v, keep[NJ];
    int reuse;    // 1 = the last step was rejected: step again from `keep` without re-evaluating (solver.cpp:69)
};

// ------------------------------------------------------------------------------------------------
// Solver::solve loop (src/nlls/solver.cpp:67-107) with GaussNewton (src/nlls/gauss_newton.cpp:53-91) and
// CauchyWeight(0.15), executed by one workgroup on the problem (dir, sv, pts, state in sh.state).
// On return sh.state / sh.tf hold the solution; returns the iteration count (applied + reverted steps).
// ------------------------------------------------------------------------------------------------
#ifdef LAMA_PROFILE_SM                 // developer build: cycles per phase of the solver loop -> prm.dbg (tools/prof_sm.py)
#define SMT(k) do { const uint64_t t_ = __builtin_readcyclecounter(); smp[k] += t_ - smt; smt = t_; } while (0)
#else
#define SMT(k) do {} while (0)
#endif
template <bool BIGSQ>
__device__ inline uint32_t gn_solve(const DevParams& prm, const int16_t* dir, const sv_t* sv, const double* __restrict__ pts, int n,
                                    const Affine& mtf, SMShared& sh, uint32_t& evals)
{
#ifdef LAMA_PROFILE_SM
    uint64_t smp[4] = {0, 0, 0, 0};
    uint64_t smt = __builtin_readcyclecounter();
#endif
    const double eps1 = 1e-4, eps2 = 1e-4, tau = 1e-4;
    const bool lm = prm.strategy == 1;
    uint32_t iter = 0;
    if (threadIdx.x == 0) { sh.mu = -1.0; sh.v = 2.0; sh.reuse = 0; sh.have_lin = 0; sh.lin_is_final = 0; }      // strategy->reset()
    __syncthreads();
    while (iter < prm.max_iter) {
        // 1. residuals + Jacobian at the current state, weighted, reduced: taken over from the validation pass of the
        //    previous (accepted) step, or restored after a rejected LM step, or evaluated
        if (sh.have_lin) {
            ++evals;
        } else if (!sh.reuse) {
            double acc[NJ];
            {
                const Affine tf = sh.tf;
                eval_beams_jac<BIGSQ>(prm, dir, sv, pts, n, tf, acc, sh.lut);
            }
            SMT(0);
            block_sum<NJ>(acc, sh.red, sh.tot);
            SMT(1);
            ++evals;
        } else {
            __syncthreads();
            if (threadIdx.x < NJ) sh.tot[threadIdx.x] = sh.keep[threadIdx.x];
            __syncthreads();
        }
        // 2. Gauss-Newton / Levenberg-Marquardt step (one thread; 3x3)
        if (threadIdx.x == 0) {
            const double* t = sh.tot;
            const double g[3] = {t[6], t[7], t[8]};
            const double max_abs_g = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
            int stop = 0;
            sh.lin_is_final = 1;                            // unless a step is applied below
            if (lm) for (int k = 0; k < NJ; ++k) sh.keep[k] = t[k];
            if (max_abs_g < eps1) {
                stop = 1;                                   // h = 0, not applied
            } else {
                const double A[3][3] = {{t[0], 0, 0}, {t[1], t[2], 0}, {t[3], t[4], t[5]}};
                const double mg[3] = {-g[0], -g[1], -g[2]};
                double h[3];
                if (lm) {                                   // levenberg_marquardt.cpp:69-77
                    if (sh.mu < 0) sh.mu = tau * fmax(t[0], fmax(t[2], t[5]));
                    const double Au[3][3] = {{t[0] + sh.mu, t[1], t[3]}, {0, t[2] + sh.mu, t[4]}, {0, 0, t[5] + sh.mu}};
                    llt3_solve(Au, mg, h);
                } else
                ldlt3_solve(A, mg, h);
                const double max_abs_h = fmax(fabs(h[0]), fmax(fabs(h[1]), fabs(h[2])));
                if (max_abs_h < eps2) stop = 1;             // returned but NOT applied (solver.cpp:84-86)
                sh.h[0] = h[0]; sh.h[1] = h[1]; sh.h[2] = h[2];
                if (!stop) {
                    bool ok = true;
                    sh.state = se2_exp_mul(h, sh.state, ok);    // problem.update(h)
                    if (!ok) atomicOr(prm.err, ERR_NUMERIC);
                    sh.tf = scan_tf(sh.state, mtf);
                    sh.lin_is_final = 0;
                }
            }
            sh.ctl = stop;
        }
        __syncthreads();
        SMT(2);
        if (sh.ctl) break;
        // 3. validation: the problem at the updated state (chi2 decides; the other sums are next iteration's linearisation)
        double acc2[NJ];
        {
            const Affine tf = sh.tf;
            eval_beams_jac<BIGSQ>(prm, dir, sv, pts, n, tf, acc2, sh.lut);
        }
        SMT(0);
        block_sum<NJ>(acc2, sh.red, sh.tot2);
        SMT(1);
        ++evals;
        if (threadIdx.x == 0) {
            const double dF = sh.tot[9] - sh.tot2[9];
            int stop = 0;
            bool invalid = !(dF > 0);
            if (lm) {                                       // LevenbergMarquard::valid, levenberg_marquardt.cpp:86-101
                const double* h = sh.h;
                const double* gk = sh.keep + 6;
                const double dL = 0.5 * ((h[0] * (sh.mu * h[0] - gk[0]) + h[1] * (sh.mu * h[1] - gk[1])) + h[2] * (sh.mu * h[2] - gk[2]));
                if (dL > 0.0 && dF > 0.0) {
                    const double q = 2 * (dF / dL) - 1;
                    sh.mu = sh.mu * fmax(1.0 / 3.0, 1 - q * q * q);
                    sh.v = 2.0;
                    invalid = false;
                } else {
                    sh.mu = sh.mu * sh.v; sh.v = 2 * sh.v;
                    invalid = true;
                }
                sh.reuse = invalid ? 1 : 0;
            }
            if (invalid) {
                stop = lm ? 0 : 1;                          // invalid: revert (solver.cpp:99-102); GaussNewton also stops
                const double mh[3] = {-sh.h[0], -sh.h[1], -sh.h[2]};
                bool ok = true;
                sh.state = se2_exp_mul(mh, sh.state, ok);   // NOT bitwise the state of sh.tot: lin_is_final stays 0
                if (!ok) atomicOr(prm.err, ERR_NUMERIC);
                sh.tf = scan_tf(sh.state, mtf);
                sh.have_lin = 0;
            } else {
                for (int k = 0; k < NJ; ++k) sh.tot[k] = sh.tot2[k];
                sh.have_lin = 1;
                sh.lin_is_final = 1;                        // if the loop ends here (max_iter), tot is at the returned state
            }
            sh.ctl = stop;
        }
        ++iter;
        __syncthreads();
        SMT(3);
        if (sh.ctl) break;
    }
#ifdef LAMA_PROFILE_SM
    if (threadIdx.x == 0) for (int k = 0; k < 4; ++k) prm.dbg[8 * blockIdx.x + k] = smp[k];
#endif
    return iter;
}
#undef SMT

// ------------------------------------------------------------------------------------------------
// k_scan_match: PFSlam2D::scanMatch (src/pf_slam2d.cpp:416-437) for every particle of the shard.
// ------------------------------------------------------------------------------------------------
// (128 VGPRs: four waves per SIMD.  Left alone the allocator takes 136 since the single-lane step calls sincos -- three waves per SIMD,
// +50 % at 3000 particles -- for 2 us at 30.)
template <bool BIGSQ>
__global__ __launch_bounds__(SM_BLOCK) __attribute__((amdgpu_num_vgpr(128))) void k_scan_match(DevParams prm, const double* __restrict__ pts, int n, Affine mtf,
                                                          double* __restrict__ loglik_out, int32_t* __restrict__ iters_out)
{
    __shared__ SMShared sh;
    const int p = blockIdx.x;
    const PV pv_ = pview_w(prm, p);
    const int16_t* dir = pv_.dm_dir;
    const sv_t* sv = pv_.dm_sv;
    if (threadIdx.x == 0) {
        const double* q = prm.poses + 4 * p;
        sh.state = SE2{cload_f64(q), cload_f64(q + 1), cload_f64(q + 2), cload_f64(q + 3)};
        sh.tf = scan_tf(sh.state, mtf);
        sh.ctl = 0;
    }
    sm_build_lut(prm, sh.lut);
    __syncthreads();
    uint32_t evals = 0;
    const uint32_t iter = gn_solve<BIGSQ>(prm, dir, sv, pts, n, mtf, sh, evals);
    // likelihood at the final state (pf_slam2d.cpp:433-436): already summed by the last linearisation when that was taken
    // at exactly the returned state; after a reverted step the state differs in the last bits, so it is evaluated
    double loglik;
    if (sh.lin_is_final) {
        loglik = sh.tot[10];
    } else {
        double a2[2];
        {
            const Affine tf = sh.tf;
            eval_beams_res<BIGSQ>(prm, dir, sv, pts, n, tf, a2, sh.lut);
        }
        block_sum<2>(a2, sh.red, sh.tot2);
        loglik = sh.tot2[1];
    }
    ++evals;
    if (threadIdx.x == 0) {
        double* q = prm.poses + 4 * p;
        q[0] = sh.state.c; q[1] = sh.state.s; q[2] = sh.state.tx; q[3] = sh.state.ty;
        loglik_out[p] = loglik;
        iters_out[p] = (int32_t)iter;
        prm.stats[4 * p + 0] = iter;
        prm.stats[4 * p + 1] = evals;
    }
}

// ------------------------------------------------------------------------------------------------
// k_match_solve: Solve(options, MatchSurface2D(dm, scan, pose), &cov) as Loc2D::update uses it
// (src/loc2d.cpp:168-180): pose in/out, plus what the covariance and the RMSE need at the solution:
//   out[0..5] lower triangle of J^T J with J weighted (Solver::solve cov branch, src/nlls/solver.cpp:109-116)
//   out[6]    sum of squared UNWEIGHTED residuals (RMSE, src/loc2d.cpp:178-180)
// ------------------------------------------------------------------------------------------------
template <bool BIGSQ>
__global__ __launch_bounds__(SM_BLOCK) void k_match_solve(DevParams prm, int particle, const double* __restrict__ pts, int n, Affine mtf,
                                                           double* __restrict__ pose_io, double* __restrict__ out7, int32_t* __restrict__ iters_out,
                                                           int do_solve)
{
    __shared__ SMShared sh;
    const PV pv_ = pview_w(prm, particle);
    const int16_t* dir = pv_.dm_dir;
    const sv_t* sv = pv_.dm_sv;
    if (threadIdx.x == 0) {
        sh.state = SE2{cload_f64(pose_io), cload_f64(pose_io + 1), cload_f64(pose_io + 2), cload_f64(pose_io + 3)};
        sh.tf = scan_tf(sh.state, mtf);
        sh.ctl = 0;
    }
    sm_build_lut(prm, sh.lut);
    __syncthreads();
    uint32_t evals = 0;
    const uint32_t iter = do_solve ? gn_solve<BIGSQ>(prm, dir, sv, pts, n, mtf, sh, evals) : 0u;
    double acc[10];
    const Affine tf = sh.tf;
#pragma unroll
    for (int k = 0; k < 10; ++k) acc[k] = 0.0;
    for (int i = threadIdx.x; i < n; i += SM_BLOCK) {
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const double hx = ((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0];
        const double hy = ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1];
        double gx, gy;
        const double r = dm_distance(prm, dir, sv, hx, hy, &gx, &gy);
        const double w = sqrt(cauchy015(r));
        const double j0 = gx * w, j1 = gy * w, j2 = (gy * hx - gx * hy) * w;
        acc[0] += j0 * j0; acc[1] += j1 * j0; acc[2] += j1 * j1;
        acc[3] += j2 * j0; acc[4] += j2 * j1; acc[5] += j2 * j2;
        acc[6] += r * r;
    }
    block_sum<10>(acc, sh.red, sh.tot);
    if (threadIdx.x == 0) {
        pose_io[0] = sh.state.c; pose_io[1] = sh.state.s; pose_io[2] = sh.state.tx; pose_io[3] = sh.state.ty;
        for (int k = 0; k < 7; ++k) out7[k] = sh.tot[k];
        iters_out[0] = (int32_t)iter;
    }
}

// MatchSurface2D::eval (src/match_surface_2d.cpp:42-90) as the reference's Problem interface exposes it: the UNWEIGHTED
// residual of every beam and, optionally, its Jacobian row [gx, gy, gy*hx - gx*hy] (column-major n x 3 like Eigen's MatrixXd).
__global__ __launch_bounds__(256) void k_match_eval(DevParams prm, int particle, const double* __restrict__ pts, int n, Affine mtf,
                                                     const double* __restrict__ pose, double* __restrict__ r_out, double* __restrict__ j_out,
                                                     int cell_mode /* 1: DynamicDistanceMap::distance(w2m(hit)), no interpolation (MatchSurface2D::error) */)
{
    __shared__ Affine tfs;
    const PV pv_ = pview_w(prm, particle);
    const int16_t* dir = pv_.dm_dir;
    const sv_t* sv = pv_.dm_sv;
    if (threadIdx.x == 0) tfs = scan_tf(SE2{cload_f64(pose), cload_f64(pose + 1), cload_f64(pose + 2), cload_f64(pose + 3)}, mtf);     // (a pose the host uploaded)
    __syncthreads();
    const Affine tf = tfs;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const double hx = ((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0];
    const double hy = ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1];
    if (cell_mode) { r_out[i] = dm_distance_cell(prm, dir, sv, w2m(prm, hx), w2m(prm, hy)); return; }
    double gx, gy;
    r_out[i] = dm_distance(prm, dir, sv, hx, hy, &gx, &gy);
    if (j_out) { j_out[i] = gx; j_out[(size_t)n + i] = gy; j_out[2 * (size_t)n + i] = gy * hx - gx * hy; }
}

// calculateLikelihood for B poses against particle `particle`'s distance map
template <bool BIGSQ>
__global__ __launch_bounds__(SM_BLOCK) void k_loglik_batch(DevParams prm, int particle, const double* __restrict__ pts, int n,
                                                            Affine mtf, const double* __restrict__ poses, double* __restrict__ out)
{
    __shared__ double red[(SM_BLOCK / 64) * 2];
    __shared__ double tot[2];
    __shared__ double lut[SM_LUT];
    __shared__ Affine tfs;
    const int b = blockIdx.x;
    const PV pv_ = pview_w(prm, particle);
    const int16_t* dir = pv_.dm_dir;
    const sv_t* sv = pv_.dm_sv;
    if (threadIdx.x == 0) {
        const double* q = poses + 4 * b;
        tfs = scan_tf(SE2{cload_f64(q), cload_f64(q + 1), cload_f64(q + 2), cload_f64(q + 3)}, mtf);
    }
    sm_build_lut(prm, lut);
    __syncthreads();
    double a2[2];
    const Affine tf = tfs;
    eval_beams_res<BIGSQ>(prm, dir, sv, pts, n, tf, a2, lut);
    block_sum<2>(a2, red, tot);
    if (threadIdx.x == 0) out[b] = tot[1];
}

// Loc2D::globalLocalization's inner evaluation (src/loc2d.cpp:275-280): B candidate poses against particle
// `particle`'s distance map -> squared norm of MatchSurface2D's residuals (no robust weight) and, for free, the
// particle filter's log-likelihood.  One workgroup per pose; the map is shared by all B workgroups (L2 resident).
__global__ __launch_bounds__(SM_BLOCK) void k_eval_batch(DevParams prm, int particle, const double* __restrict__ pts, int n,
                                                          Affine mtf, const double* __restrict__ poses,
                                                          double* __restrict__ sqnorm_out, double* __restrict__ loglik_out)
{
    __shared__ double red[(SM_BLOCK / 64) * 2];
    __shared__ double tot[2];
    __shared__ Affine tfs;
    const int b = blockIdx.x;
    const PV pv_ = pview_w(prm, particle);
    const int16_t* dir = pv_.dm_dir;
    const sv_t* sv = pv_.dm_sv;
    if (threadIdx.x == 0) {
        const double* q = poses + 4 * b;
        tfs = scan_tf(SE2{cload_f64(q), cload_f64(q + 1), cload_f64(q + 2), cload_f64(q + 3)}, mtf);
    }
    __syncthreads();
    const Affine tf = tfs;
    double a2[2] = {0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += SM_BLOCK) {
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const double hx = ((tf.R[0][0] * px + tf.R[0][1] * py) + tf.R[0][2] * pz) + tf.t[0];
        const double hy = ((tf.R[1][0] * px + tf.R[1][1] * py) + tf.R[1][2] * pz) + tf.t[1];
        const double d = dm_distance(prm, dir, sv, hx, hy, nullptr, nullptr);
        a2[0] += d * d;
        a2[1] += -(d * d) / prm.meas_sigma;
    }
    block_sum<2>(a2, red, tot);
    if (threadIdx.x == 0) {
        if (sqnorm_out) sqnorm_out[b] = tot[0];
        if (loglik_out) loglik_out[b] = tot[1];
    }
}

// Loc2D::addSamplingCovariance's inner loop (src/loc2d.cpp:217-234): for sample translation k the scan is placed at
// `base` (rotation and sensor offset, computed on the host) shifted by xy[k]; every `step`-th point looks up the
// NON-interpolated distance and contributes exp(-d^2/0.01)^3.  One wave per sample; the <= 128 terms are summed in
// point order by one lane (the reference's sequential sum).
constexpr int SL_MAX_TERMS = 128;
__global__ __launch_bounds__(64) void k_sample_likelihood(DevParams prm, int particle, const double* __restrict__ pts, int n, int step,
                                                          Affine base, const double* __restrict__ xy, double* __restrict__ l_out)
{
    __shared__ double terms[SL_MAX_TERMS];
    const int k = blockIdx.x;
    const PV pv_ = pview_w(prm, particle);
    const int16_t* dir = pv_.dm_dir;
    const sv_t* sv = pv_.dm_sv;
    // (xy is rewritten by the host before every call: a coherent load, not the s_load a uniform read of it would become)
    const double tx = base.t[0] + cload_f64(xy + 2 * k), ty = base.t[1] + cload_f64(xy + 2 * k + 1);
    const int nterms = (n + step - 1) / step;
    for (int j = threadIdx.x; j < nterms; j += 64) {
        const int i = j * step;
        const double px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
        const double hx = ((base.R[0][0] * px + base.R[0][1] * py) + base.R[0][2] * pz) + tx;
        const double hy = ((base.R[1][0] * px + base.R[1][1] * py) + base.R[1][2] * pz) + ty;
        const double dist = dm_distance_cell(prm, dir, sv, w2m(prm, hx), w2m(prm, hy));
        const double e = exp(-(dist * dist) / 0.01);
        terms[j] = e * e * e;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double l = 0.0;
        for (int j = 0; j < nterms; ++j) l += terms[j];
        l_out[k] = l;
    }
}

// ------------------------------------------------------------------------------------------------
// Map update = PFSlam2D::updateParticleMaps (src/pf_slam2d.cpp:439-509), two kernels, one wave per particle:
//   k_raycast   : beams in order; hit cell + Bresenham cells 64 at a time -> uint16 counters, add/remove
//                 obstacle events applied to the DM cells and appended to the lower/raise queues in HBM;
//   k_brushfire : DynamicDistanceMap::update (src/sdm/dynamic_distance_map.cpp:160-197), exact.
//
// Exactness contract: the counters AND the order of add/remove obstacle events AND the order in which the
// brushfire pops equal-priority cells are those of the reference's sequential code, so the resulting maps are
// bit-identical (cells, obstacle offsets, flags, masks, patch sets).
//   * cells of one ray are distinct => plain RMW, no atomics; the closed form
//     steps_j(t) = floor((2 t |d_j| + n) / (2 n)) replays Map::computeRay (src/sdm/map.cpp:198-227);
//   * the beam's hit (-> lower queue) and its ray cells (-> raise queue) touch distinct cells and distinct
//     queues, so they share one 64-lane section (lane 0 of the first section is the hit);
//   * events are appended in (beam, step) order via ballot ranks (priority 0 => libstdc++ push_heap leaves
//     them at the end of the heap array);
//   * the brushfire pops one cell at a time through the libstdc++-exact heap of lama_heap.h (LDS resident);
//     the work of one pop is spread over 5 lanes (cell + 4 neighbours) with all loads of a round issued
//     together; only the queue pushes are serialised, in the reference's neighbour order.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t DC_EMPTY = 0xFFFFFFFFu;
constexpr int DC_SIZE = 512;

// direct-mapped LDS cache of window-directory entries, XOR-folded index.  A word holds ((pidx >> 3) << 15) | (slot + 1): the low
// three bits of pidx are implied by the index (the fold XORs them with higher bits of pidx, all of which the tag holds), 17 tag bits
// cover windows up to 1016 x 1016 patches (round 4: the window grows with the map), slot + 1 in 15 bits (0 = no such patch).  An
// empty word carries a tag no window position has.
struct DirCache {
    uint32_t* e;
    const int16_t* dir;
    uint32_t W;
    // pidx = wy * W + wx with W a multiple of 8: (pidx & 7) = wx & 7; fold the rest without a division
    __device__ static inline uint32_t index(uint32_t pidx) { return (pidx ^ (pidx >> 9) ^ (pidx >> 5)) & (DC_SIZE - 1); }
    __device__ static inline uint32_t pack(uint32_t pidx, int slot) { return ((pidx >> 3) << 15) | (uint32_t)(slot < 0 ? 0 : slot + 1); }
    __device__ static inline bool hit(uint32_t v, uint32_t pidx) { return (v >> 15) == (pidx >> 3); }
    __device__ static inline int slot_of(uint32_t v) { return (int)(v & 0x7FFFu) - 1; }
    __device__ inline int lookup(uint32_t pidx) const
    {
        const uint32_t k = index(pidx);
        const uint32_t v = e[k];
        if (hit(v, pidx)) return slot_of(v);
        const int slot = dir[pidx];
        e[k] = pack(pidx, slot);
        return slot < 0 ? -1 : slot;
    }
    __device__ inline void update(uint32_t pidx, int slot) const { e[index(pidx)] = pack(pidx, slot); }
};

// Wave-cooperative "non-const Map::get" patch lookup (src/sdm/map.cpp:371-412): every lane with `want`
// gets the slot of window patch `pidx`, allocating missing patches once per distinct patch.
// Must be called by all 64 lanes.  Returns -1 on capacity overflow.
__device__ inline int coop_slot(const DirCache& dc, int16_t* dir, int& count, int cap, bool want, uint32_t pidx, int errbit, int32_t* err)
{
    int slot = want ? dc.lookup(pidx) : 0;
    bool need = want && slot < 0;
    unsigned long long m = __ballot(need);
    while (m) {
        const int leader = __ffsll((long long)m) - 1;
        const uint32_t lp = __shfl(pidx, leader, 64);
        int ns;
        if (count < cap) { ns = count; ++count; } else { ns = -1; }
        if ((int)(threadIdx.x & 63) == leader) {
            if (ns >= 0) { dir[lp] = (int16_t)ns; dc.update(lp, ns); } else atomicOr(err, errbit);
        }
        if (need && pidx == lp) { slot = ns; need = false; }
        m = __ballot(need);
    }
    return slot;
}

__global__ __launch_bounds__(UM_BLOCK) void k_raycast(DevParams prm, const double* __restrict__ pts, int n,
                                                       const double* __restrict__ tfs /*[P][12]*/, int first_particle)
{
    // (when the allocation phase that precedes this kernel failed, nothing is touched: the host grows the arenas and retries)
    if (map_update_aborted(prm)) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(prm.err, ERR_CLEAN_ABORT); return; }
    __shared__ uint32_t lds_dc[2 * DC_SIZE];
    const int p = first_particle + blockIdx.x;
    const int lane = threadIdx.x;
    const PV pv = pview(prm, p);
    int16_t* dm_dir = pv.dm_dir;
    int16_t* occ_dir = pv.occ_dir;
    sv_t* dm_sv = pv.dm_sv;
    uint32_t* dm_obs = pv.dm_obs;
    uint64_t* dm_mask = pv.dm_mask;
    uint32_t* occ = pv.occ;
    uint64_t* occ_mask = pv.occ_mask;
    uint64_t* q_lower = prm.q_lower + (size_t)p * prm.qcap;
    uint64_t* q_raise = prm.q_raise + (size_t)p * prm.qcap;
    int dm_count = prm.counts[2 * p], occ_count = prm.counts[2 * p + 1];
    for (int k = lane; k < 2 * DC_SIZE; k += UM_BLOCK) lds_dc[k] = DC_EMPTY;
    __syncthreads();
    const DirCache occ_dc{lds_dc, occ_dir, prm.W}, dm_dc{lds_dc + DC_SIZE, dm_dir, prm.W};

    uint32_t nl = 0, nr = 0;            // wave-uniform queue sizes
    uint64_t ray_cells = 0;

    // tf = fixed_tf * moving_tf of this particle (12 doubles, computed on the host with libm exactly
    // like the reference does on the CPU, so the integer cell coordinates below are reproducible)
    double T[12];
    uload_f64_w<12>(tfs + 12 * (size_t)p, T);            // (rewritten by the host before every update: coherent uniform loads, lama_dev.h)
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
        if (prm.ray_rule == 1) {                                            // LidarOdometry2D::updateMaps, lidar_odometry_2d.cpp:108-114
            abx = hx - sx; aby = hy - sy; abz = hz - sz;
            ray_length = sqrt((abx * abx + aby * aby) + abz * abz);
            if (ray_length >= 1.0) { sx = hx - abx / ray_length; sy = hy - aby / ray_length; sz = hz - abz / ray_length; }
        } else if (mark_hit && prm.trunc_ray > 0.0) {                       // :481-491
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

        const int64_t d0 = (int64_t)mhx - (int64_t)msx, d1 = (int64_t)mhy - (int64_t)msy, d2 = (int64_t)mhz - (int64_t)msz;
        const int64_t a0 = d0 < 0 ? -d0 : d0, a1 = d1 < 0 ? -d1 : d1, a2 = d2 < 0 ? -d2 : d2;
        const int64_t nn = a0 > a1 ? (a0 > a2 ? a0 : a2) : (a1 > a2 ? a1 : a2);
        const int s0 = d0 < 0 ? -1 : 1, s1 = d1 < 0 ? -1 : 1;
        int steps = nn > 0 ? (int)nn - 1 : 0;                // free cells t = 1 .. n-1 (src/sdm/map.cpp:198-227)
        if (nn >= 8192) { if (lane == 0) atomicOr(prm.err, ERR_WINDOW); steps = 0; }   // longer than any window: cannot fit
        ray_cells += (uint64_t)steps;
        // steps_j(t) = floor((2 t a_j + n) / (2 n)) without a per-lane division: n < 2^13 so the numerator is < 2^27 and
        // q = (num * M) >> 42 with M = floor(2^42 / (2n)) + 1 is exact (num * 2n < 2^42); one division per beam.
        const uint64_t magic = (1ull << 42) / (uint64_t)(2 * (nn > 0 ? nn : 1)) + 1ull;
        const uint32_t ua0 = (uint32_t)a0, ua1 = (uint32_t)a1, un = (uint32_t)nn;

        // sections of 64 lanes over t = 0 (the hit cell, :493-498) , 1 .. steps (the free cells, :500-504)
        for (int base = 0; base <= steps; base += 64) {
            const int t = base + lane;
            const bool is_hit = t == 0;
            const bool act = is_hit ? mark_hit : (t <= steps);
            uint32_t cx, cy;
            if (is_hit) { cx = mhx; cy = mhy; }
            else {
                const uint32_t st0 = (uint32_t)(((uint64_t)(2u * (uint32_t)t * ua0 + un) * magic) >> 42);
                const uint32_t st1 = (uint32_t)(((uint64_t)(2u * (uint32_t)t * ua1 + un) * magic) >> 42);
                cx = msx + (uint32_t)(s0 * (int)st0); cy = msy + (uint32_t)(s1 * (int)st1);
            }
            const uint32_t rx = cx - prm.wx0, ry = cy - prm.wy0;
            const bool inwin = rx < prm.WC && ry < prm.WC;
            if (act && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool want = act && inwin;
            const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5);
            const uint32_t ci = (rx & 31u) | ((ry & 31u) << 5);
            const int slot = coop_slot(occ_dc, occ_dir, occ_count, (int)pv.occ_cap, want, pidx, ERR_OCC_CAP, prm.err);
            bool changed = false;
            if (want && slot >= 0) {
                uint32_t* cell = occ + (size_t)slot * 1024 + ci;
                const uint32_t v = *cell;
                uint32_t o = v & 0xFFFFu, vis = v >> 16;
                if (prm.occ_policy == 1) {
                    // ProbabilisticOccupancyMap (probabilistic_occupancy_map.cpp:82-107): float log-odds cell, double parameters,
                    // occ_thresh = 0; every get() turns the mask bit on
                    const float lp = __uint_as_float(v);
                    float np_;
                    if (is_hit) {
                        const bool occupied = (double)lp > 0.0;
                        np_ = (float)fmin((double)lp + prm.lo_hit, prm.lo_max);
                        changed = !occupied && ((double)np_ > 0.0);
                    } else {
                        const bool was_free = (double)lp < 0.0;
                        np_ = (float)fmax((double)lp + prm.lo_miss, prm.lo_min);
                        changed = !was_free && ((double)np_ < 0.0);
                    }
                    *cell = __float_as_uint(np_);
                    atomicOr((unsigned long long*)(occ_mask + (size_t)slot * 16 + (ci >> 6)), 1ull << (ci & 63));
                } else {
                if (is_hit) {                                               // setOccupied (frequency_occupancy_map.cpp:81-91)
                    const bool occupied = vis != 0 && 4u * o > vis;         // prob > 0.25
                    o = (o + 1) & 0xFFFFu; vis = (vis + 1) & 0xFFFFu;
                    changed = !occupied && (vis != 0 && 4u * o > vis);
                } else {                                                    // setFree (:65-74)
                    const bool was_free = vis != 0 && 4u * o < vis;         // prob < 0.25
                    vis = (vis + 1) & 0xFFFFu;
                    changed = !was_free && (vis != 0 && 4u * o < vis);
                }
                *cell = o | (vis << 16);
                // Container mask bit of an occupancy cell == "visited != 0"; only a uint16 wrap needs the plane
                if (vis == 0) atomicOr((unsigned long long*)(occ_mask + (size_t)slot * 16 + (ci >> 6)), 1ull << (ci & 63));
                }
            }
            const int dslot = coop_slot(dm_dc, dm_dir, dm_count, (int)pv.dm_cap, changed, pidx, ERR_DM_CAP, prm.err);
            bool push = false;
            if (changed && dslot >= 0) {
                const uint64_t bit = 1ull << (ci & 63);
                uint64_t* w = dm_mask + (size_t)dslot * 16 + (ci >> 6);
                if (!(*w & bit)) atomicOr((unsigned long long*)w, (unsigned long long)bit);
                const uint32_t di = (uint32_t)dslot * 1024u + ci;
                const sv_t s = dm_sv[di];
                const bool is_obstacle = (s & SV_VALID) && (s & SV_SQMASK) == 0;
                if (is_hit) {                                               // addObstacle (dynamic_distance_map.cpp:212-226)
                    if (!is_obstacle) { dm_sv[di] = (sv_t)(SV_VALID | SV_QUEUED); dm_obs[di] = 0; push = true; }
                } else {                                                    // removeObstacle (:228-242)
                    if (is_obstacle) { dm_sv[di] = SV_QUEUED; dm_obs[di] = 0; push = true; }
                }
            }
            const unsigned long long pm = __ballot(push);
            if (pm) {
                const unsigned long long pm_hit = (base == 0) ? (pm & 1ull) : 0ull;
                const unsigned long long pm_ray = pm & ~pm_hit;
                if (pm_hit) {
                    if (nl >= prm.qcap) { if (lane == 0) atomicOr(prm.err, ERR_QUEUE); }
                    else { if (push && is_hit) q_lower[nl] = q_entry(0, (int)rx, (int)ry); nl += 1; }
                }
                if (pm_ray) {
                    const int cnt = __popcll(pm_ray);
                    if (nr + (uint32_t)cnt > prm.qcap) { if (lane == 0) atomicOr(prm.err, ERR_QUEUE); }
                    else {
                        if (push && !is_hit) {
                            const int rank = __popcll(pm_ray & ((1ull << lane) - 1ull));
                            q_raise[nr + (uint32_t)rank] = q_entry(0, (int)rx, (int)ry);
                        }
                        nr += (uint32_t)cnt;
                    }
                }
            }
        }
        // make this beam's stores visible to the next beam's loads (same wave: ordering only)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (lane == 0) {
        prm.counts[2 * p] = dm_count;
        prm.counts[2 * p + 1] = occ_count;
        prm.qsizes[2 * p] = nl;
        prm.qsizes[2 * p + 1] = nr;
        prm.stats[4 * p + 2] += ray_cells;
    }
}

// ------------------------------------------------------------------------------------------------
// k_brushfire -- DynamicDistanceMap::update / raise / lower (src/sdm/dynamic_distance_map.cpp:160-197,
// 244-279, 281-330).  One wave per particle; lane 4 owns the popped cell, lanes 0..3 its 4-neighbours in the
// reference's order (+x, +y, -x, -y).
//
// Which get() calls of the reference have side effects (patch allocation + mask bit, src/sdm/map.cpp:371-412)?
//   - the popped cell: none (it was get()-ed when it was queued);
//   - obstacle cells reached through a valid cell's offset: none (an obstacle was get()-ed when it was added;
//     a cleared cell has offset 0 = itself);
//   - the neighbours: YES -- all four in raise(), and in lower() those that pass the "away from the obstacle"
//     test, and only when lower() really runs.  They are prefetched side-effect free and the allocation /
//     mask update is applied once the decision is known.
// ------------------------------------------------------------------------------------------------
#ifdef LAMA_PROFILE_BF
#ifdef LAMA_PROFILE_BF_COUNT           // event counts instead of cycles (tools/prof_bf.py, LAMA_PROF_COUNT=1)
#define BFT(k) do { (void)tprev; } while (0)
#else
#define BFT(k) do { const uint64_t t_ = __builtin_readcyclecounter(); prof[k] += t_ - tprev; tprev = t_; } while (0)
#endif
#else
#define BFT(k) do {} while (0)
#endif
#ifdef LAMA_PROFILE_BF_MAIN
#define BFT_MAIN(k) BFT(k)
#else
#define BFT_MAIN(k) do {} while (0)
#endif
#ifdef LAMA_PROFILE_BF_FINE            // developer build: eight consecutive sections of the straight-line lower pop (replaces the buckets above)
#undef BFT
#undef BFT_MAIN
#define BFT(k) do {} while (0)
#define BFT_MAIN(k) do {} while (0)
#define BFF(k) do { const uint64_t t_ = __builtin_readcyclecounter(); prof[k] += t_ - tprev; tprev = t_; } while (0)
#else
#define BFF(k) do {} while (0)
#endif

constexpr int BF_DUMMY = 8;   // dummy slots per wave (lane & 7): 64 of them cost the 12th workgroup of a CU (LDS is granted in 1,280-byte blocks)
template <int LQ, int RQ>
struct BfLds {
    // the small, hot fields first: a DS instruction's immediate offset reaches 64 KB, and the resume stage's queues alone are 80 KB --
    // behind them every mailbox / cache access needed its address computed in a register
    // TW mailboxes (double buffered by iteration parity)
    uint64_t pl_e[2][4];   // main -> helper: the entries to push into the LOWER queue, in neighbour order
    uint64_t pr_e[2][4];   // main -> helper: the entries raise() pushes into the RAISE queue
    uint64_t topq[2];      // helper -> main: the heap's root after pop() (before the pushes)
    uint64_t dummy[2][BF_DUMMY]; // per wave: absorbs the LDS stores of lanes without work (an address select instead of an exec mask)
    uint32_t pl_n[2];      //                 and how many
    uint32_t pr_n[2];
    uint32_t cmd_r;        // raise-queue length at the hand-over
    uint32_t cmd;          // lower-queue length at the hand-over, or BF_CMD_EXIT
    uint32_t dc[DC_SIZE];
    uint64_t lower[LQ];
    uint64_t raise[RQ];    // directly behind `lower`: once the raise queue is empty (lower phase) the lower heap of the wave pair may grow into it
};
constexpr uint32_t BF_CMD_EXIT = 0xFFFFFFFFu;

// ---- LDS-resident libstdc++ heap (same algorithm as lama_heap.h).  All heap state is wave-uniform: every lane
// executes the code, lane 0 stores.
// pop() looks at the whole 6-level subtree below the hole at once instead of chasing one dependent LDS read per
// level: lane L holds the node at relative heap position L (children of L are 2L+1 and 2L+2; absolute index
// hole * 2^depth(L) + L) and its sibling.  Every lane decides locally whether __adjust_heap, standing on its parent,
// would step to it (`comp(right, left)` -> left, else right; the parent must have both children); a ballot turns that
// into a mask and a lane is on the hole's path iff the bits of ALL its ancestors are set -- one AND/compare against a
// per-lane constant (`anc`), no serial chase.  The moved entries are written with a single ds_write: each node on the
// path stores itself into its parent's slot.
__device__ __forceinline__ uint64_t lds_pop_ancestors(int lane)
{
    uint64_t anc = 0;
    if (lane < 63) for (int a = lane; a > 0; a = (a - 1) >> 1) anc |= 1ull << a;
    return anc;
}
// The pop itself (run by the helper wave of the two-wave brushfire, or by the only wave): the same final heap array as
// __adjust_heap + __push_heap, computed top-down.  __adjust_heap moves every entry n_1 .. n_k of the hole's path (the preferred
// children down to a leaf) up one slot, then __push_heap walks the re-inserted last entry v back up, moving entries
// with prio(n_i) > prio(v) down again -- into the very slots they came from.  Priorities are non-decreasing along
// the path, so the net effect is: the entries with prio(n_i) <= prio(v), a prefix of the path, move up one slot, v
// takes the slot of the last of them, everything below stays.  The descent can therefore stop at the first entry
// that stays; nothing is read back, and the entry that ends up at the root is known from registers (returned;
// meaningful when size > 0 afterwards).
// Sift `value` down from the hole at slot H0 of the heap h[0, len) (the loop of std::__adjust_heap followed by std::__push_heap,
// top-down: see above).  Returns the entry that ends up at slot H0 if H0 == 0 (the new root; meaningful when len > 0).
__device__ __forceinline__ uint64_t lds_sift_topdown(uint64_t* h, const uint32_t len, const uint32_t H0, const uint64_t value, int lane, uint64_t anc)
{
    const uint32_t vprio = heap_prio(value);
    const uint32_t lim = (len - 1) / 2;                              // nodes below `lim` have both children
    const int d = 31 - __clz(lane + 1);
    const bool is_left = (lane & 1) != 0;
    uint32_t H = H0;
    uint32_t root_lo = (uint32_t)value, root_hi = (uint32_t)(value >> 32);
    bool stopped = false;
    while (H < lim) {
        const uint32_t idx = (H << d) + (uint32_t)lane;              // lane L: the node at relative position L below H
        const uint32_t left = is_left ? idx : idx - 1;
        const uint32_t parent = (left - 1) >> 1;
        const bool cand = lane >= 1 && lane < 63 && parent < lim;
        const uint32_t la = cand ? left : 1u;
        const uint64_t vl = h[la], vr = h[la + 1];                   // one ds_read2_b64
        const bool take_left = heap_prio(vr) > heap_prio(vl);        // __adjust_heap: right unless comp(right, left)
        const bool step_to_me = cand && (is_left == take_left);
        const unsigned long long okm = __ballot(step_to_me);
        const bool onpath = cand && (okm & anc) == anc;
        const uint64_t mine = is_left ? vl : vr;
        const bool moves = onpath && !(heap_prio(mine) > vprio);
        const unsigned long long pathm = __ballot(onpath), mvm = __ballot(moves);
        if (moves) h[parent] = mine;
        if (mvm) {
            if (H == 0) {
                const int r1 = (mvm & 2ull) ? 1 : 2;
                root_hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), r1);
                root_lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, r1);
            }
            const int rel = 63 - __clzll((long long)mvm);            // deepest moved entry: its old slot is the new hole
            H = (H << (31 - __clz(rel + 1))) + (uint32_t)rel;
        }
        if (mvm != pathm) { stopped = true; break; }
    }
    LAMA_LOCKSTEP();                                                  // every lane has read what it needs of the old array
    if (!stopped && (len & 1) == 0 && H == (len - 2) / 2) {          // the hole has a lone left child, the array's last entry
        const uint64_t c = h[len - 1];
        const bool up = (uint32_t)__builtin_amdgcn_readfirstlane((int)heap_prio(c)) <= (uint32_t)__builtin_amdgcn_readfirstlane((int)vprio);
        if (up) {
            if (lane == 0) h[H] = c;
            if (H == 0) { root_lo = (uint32_t)c; root_hi = (uint32_t)(c >> 32); }
            H = len - 1;
        }
    }
    if (lane == 0) h[H] = value;
    LAMA_LOCKSTEP();                                                  // the next reader of the heap (any lane) sees lane 0's store
    return ((uint64_t)root_hi << 32) | root_lo;
}
__device__ __forceinline__ uint64_t lds_pop_topdown(uint64_t* h, uint32_t& size, int lane, uint64_t anc)
{
    --size;
    const uint32_t len = size;
    if (len == 0) return 0;
    return lds_sift_topdown(h, len, 0u, h[len], lane, anc);
}
__device__ __forceinline__ void lds_push(uint64_t* h, uint32_t& size, uint64_t value, bool writer)
{
    uint32_t hole = size++;
    while (hole > 0) {
        const uint32_t parent = (hole - 1) / 2;
        const uint64_t pv = h[parent];
        if (!heap_comp(pv, value)) break;
        if (writer) h[hole] = pv;
        hole = parent;
    }
    if (writer) h[hole] = value;
    LAMA_LOCKSTEP();
}

// ------------------------------------------------------------------------------------------------
// Round 4: the wave-pair brushfire as (nearly) straight-line code.
//
// Measured on the MI355X (tools/ubench/issue_costs.hip, profiles/r04_issue_costs_single_wave.txt), one wave alone on a SIMD:
// a dependent VALU or SALU instruction issues every 4-5 cycles, but every hand-over from the vector to the scalar side costs
// ~18 cycles on top (v_cmp -> s_cbranch_vccz 28 cycles NOT taken, 44 taken; v_readfirstlane -> s_add -> v_mov 28), an
// exec-masked region `s_and_saveexec / s_cbranch_execz / s_or exec` 46-52 cycles whether it is entered or skipped, against 13
// for v_cmp + v_add + v_cndmask.  The round-3 kernel executed 52 branches per pop (SQ_INSTS_BRANCH, profiles/r04_sq_brushfire_
// before.txt) -- the compiler's lowering of `if (lane has work) load / store` -- which is where its "12 cycles per
// instruction" came from.  This form keeps control flow wave-uniform and rare:
//   * loads and stores of lanes without work are not masked off, they go through BUFFER instructions with an offset beyond the
//     buffer's extent: the hardware drops such a store and returns 0 for such a load -- exactly the "absent cell reads as zero"
//     semantics of the map (a raw buffer resource per plane of the particle's arena);
//   * LDS stores of lanes without work go to a per-lane dummy slot (an address select instead of an exec mask);
//   * everything a pop can need only rarely -- a directory-cache miss, a patch allocation, the second load round of a tie, a
//     stale queue entry, a cell outside the window -- is detected by ONE ballot and handled by the general code (the round-3
//     loop body, kept as it was); the speculative path stores nothing before that test;
//   * the heap pop of the helper wave is a fixed number of predicated rounds instead of a data-dependent loop.
// The sequence of heap operations and of map stores is the sequential one, as before: results stay bit-identical.
// ------------------------------------------------------------------------------------------------
#ifdef LAMA_WAVE_SIM       // tests/sim: host stand-ins with the same out-of-range semantics
struct BufRsrc { uint8_t* base; uint32_t bytes; };
__device__ __forceinline__ BufRsrc buf_make(void* p, uint32_t bytes) { return BufRsrc{(uint8_t*)p, bytes}; }
__device__ __forceinline__ uint32_t buf_load_u16(const BufRsrc& r, uint32_t off) { uint16_t v = 0; if (off <= r.bytes - 2u && off < r.bytes) std::memcpy(&v, r.base + off, 2); return v; }
__device__ __forceinline__ uint32_t buf_load_u32(const BufRsrc& r, uint32_t off) { uint32_t v = 0; if (off <= r.bytes - 4u && off < r.bytes) std::memcpy(&v, r.base + off, 4); return v; }
__device__ __forceinline__ void buf_store_u16(const BufRsrc& r, uint32_t off, uint32_t v) { if (off <= r.bytes - 2u && off < r.bytes) { const uint16_t w = (uint16_t)v; std::memcpy(r.base + off, &w, 2); } }
__device__ __forceinline__ void buf_store_u32(const BufRsrc& r, uint32_t off, uint32_t v) { if (off <= r.bytes - 4u && off < r.bytes) std::memcpy(r.base + off, &v, 4); }
__device__ __forceinline__ void buf_or_u64(const BufRsrc& r, uint32_t off, uint64_t v) { if (off <= r.bytes - 8u && off < r.bytes) { uint64_t w; std::memcpy(&w, r.base + off, 8); w |= v; std::memcpy(r.base + off, &w, 8); } }
__device__ __forceinline__ uint32_t lane_rank(unsigned long long m, int lane) { return (uint32_t)__popcll(m & ((1ull << lane) - 1ull)); }
#else
using BufRsrc = __amdgpu_buffer_rsrc_t;
// raw buffer (stride 0, offsets checked against `bytes`); 0x00020000 = the gfx9 / CDNA default word 3 (32-bit untyped data)
__device__ __forceinline__ BufRsrc buf_make(void* p, uint32_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }
__device__ __forceinline__ uint32_t buf_load_u16(BufRsrc r, uint32_t off) { return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0); }
__device__ __forceinline__ uint32_t buf_load_u32(BufRsrc r, uint32_t off) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0); }
__device__ __forceinline__ void buf_store_u16(BufRsrc r, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, (int)off, 0, 0); }
__device__ __forceinline__ void buf_store_u32(BufRsrc r, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)off, 0, 0); }
// (no builtin for the 64-bit OR.  The s_nop covers "VALU writes SGPR -> VMEM reads that SGPR: 5 wait states": the compiler's hazard
// recogniser does not look inside inline assembly, and when it has spilled the resource to VGPR lanes it restores it with
// v_readlane right in front of this instruction -- found on the device: the atomic went elsewhere, mask words were lost)
__device__ __forceinline__ void buf_or_u64(BufRsrc r, uint32_t off, uint64_t v) { asm volatile("s_nop 4\n\tbuffer_atomic_or_x2 %0, %1, %2, 0 offen" :: "v"(v), "v"(off), "s"(r) : "memory"); }
__device__ __forceinline__ uint32_t lane_rank(unsigned long long m, int) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); }
#endif
// a cell of plane A of the distance map (sv_t: 2 bytes, 4 in the wide build)
#ifdef LAMA_WIDE_DM
#define buf_load_sv buf_load_u32
#define buf_store_sv buf_store_u32
#else
#define buf_load_sv buf_load_u16
#define buf_store_sv buf_store_u16
#endif
// A value the optimiser must take as it is, in a vector register: lane predicates kept as 0 / ~0 words stay on the VALU (v_and /
// v_or) instead of becoming wave masks that are combined on the scalar unit -- a vector -> scalar hand-over per operation.
#ifdef LAMA_WAVE_SIM
__device__ __forceinline__ uint32_t opq(uint32_t x) { return x; }
#else
__device__ __forceinline__ uint32_t opq(uint32_t x) { asm("" : "+v"(x)); return x; }
#endif
// The same, but the value is also kept WHERE it is computed: the optimiser sinks a computation into the only (conditional) block that
// uses it, i.e. behind the loads' `s_waitcnt`; a volatile empty asm is neither sunk nor hoisted.
#ifdef LAMA_WAVE_SIM
__device__ __forceinline__ uint32_t pin(uint32_t x) { return x; }
#else
__device__ __forceinline__ uint32_t pin(uint32_t x) { asm volatile("" : "+v"(x)); return x; }
#endif
// 0 / ~0 words of the unsigned comparisons a < b and a <= b, for operands below 2^31
__device__ __forceinline__ uint32_t m_lt(uint32_t a, uint32_t b) { return opq((uint32_t)((int32_t)(opq(a) - b) >> 31)); }
__device__ __forceinline__ uint32_t m_le(uint32_t a, uint32_t b) { return ~m_lt(b, a); }
// ... of v != 0 for any word, of a > 0 for a signed word (|a| < 2^31), and the select (m ? a : b) -- one v_bfi_b32
__device__ __forceinline__ uint32_t m_nz(uint32_t v) { return opq((uint32_t)((int32_t)(opq(v) | (0u - v)) >> 31)); }
__device__ __forceinline__ uint32_t m_pos(int32_t a) { return opq((uint32_t)((int32_t)(0u - (uint32_t)opq((uint32_t)a)) >> 31)); }
__device__ __forceinline__ uint32_t m_sel(uint32_t m, uint32_t a, uint32_t b) { return (a & m) | (b & ~m); }
#ifdef LAMA_WAVE_SIM
__device__ __forceinline__ uint32_t bf_mul24(uint32_t a, uint32_t b) { return a * b; }
#else
__device__ __forceinline__ uint32_t bf_mul24(uint32_t a, uint32_t b) { return __umul24(a, b); }       // v_mul_u32_u24 / v_mad_u32_u24: full rate
#endif
constexpr uint32_t BUF_OOB = 0x7FFFFFF0u;      // a byte offset beyond any plane of a particle's arena (<= 134 MB)

// pop() of the LDS heap h[0, size) by the helper wave, NR predicated rounds of the 5-level subtree walk of lds_sift_topdown (two
// cover a heap of 1024 entries, three one of 8192) and the lone-left-child step, no data-dependent branch: same final array as
// std::pop_heap.  The entry that ends up at the root is stored to *root_out (LDS) by whichever lane holds it -- nothing is read
// back across lanes.  `dummy`: this lane's LDS word that absorbs its stores when it has no work.
template <int NR>
__device__ __forceinline__ void lds_pop_flat(uint64_t* h, uint32_t& size, const int lane, const uint64_t anc, uint64_t* root_out, uint64_t* dummy)
{
    // Lane predicates are 0 / ~0 words combined with integer arithmetic: as `bool`s the compiler keeps them as wave masks in scalar
    // registers and every v_cmp -> s_and -> v_cndmask chain is a vector -> scalar hand-over (~18 cycles each, a dozen per round).
    --size;
    const uint32_t len = size;
    if (len == 0) return;
    const uint64_t value = h[len];                                   // the re-inserted last entry
    const uint64_t tailc = h[len - 1];                               // the lone left child, if the hole ends above it
    const uint32_t vprio = heap_prio(value);
    const uint32_t lim = (len - 1) >> 1;                             // nodes below `lim` have both children
    const int d = 31 - __clz(lane + 1);
    const uint32_t leftm = (lane & 1) ? 0xFFFFFFFFu : 0u;            // odd relative positions are left children
    const uint32_t innerm = (lane >= 1 && lane < 63) ? 0xFFFFFFFFu : 0u;
    const uint32_t anc_lo = (uint32_t)anc, anc_hi = (uint32_t)(anc >> 32);
    uint32_t H = 0;
    uint32_t go = 0xFFFFFFFFu;                                       // wave-uniform: the descent has not stopped
    uint64_t mine0 = 0;                                              // first round: what lanes 1 / 2 moved to the root ...
    uint32_t root_child = 0;                                         // ... if they did (per lane) ...
    bool root_value = true;                                          // ... else the re-inserted entry is the new root (wave-uniform)
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        // (a later round is skipped -- one scalar branch -- when the descent has stopped or the heap ends above its levels)
        if (r > 0 && (go == 0u || ((H << 1) + 1u) >= len)) break;
        const uint32_t idx = (H << d) + (uint32_t)lane;              // lane L: the node at relative position L below H
        const uint32_t lm1 = idx - 2u - leftm;                       // (left child of my parent) - 1: idx - 1 for a left child, idx - 2 for a right one
        const uint32_t parent = lm1 >> 1;
        const uint32_t cand = opq(innerm & go & m_lt(parent, lim));  // my parent has both children
        const uint32_t la = 1u + (lm1 & cand);
        const uint64_t vl = h[la], vr = h[la + 1];
        const uint32_t pl = (uint32_t)(vl >> 32) >> 16, pr = (uint32_t)(vr >> 32) >> 16;
        const uint32_t take_left = m_lt(pl, pr);                     // __adjust_heap: right unless prio(right) > prio(left)
        const uint32_t step = opq(cand & ~(take_left ^ leftm));      // it steps to me
        const unsigned long long okm = __ballot(step != 0u);
        const uint32_t offpath = opq((((uint32_t)okm & anc_lo) ^ anc_lo) | (((uint32_t)(okm >> 32) & anc_hi) ^ anc_hi));   // 0: all my ancestors were stepped to
        const uint32_t onpath = opq(offpath == 0u ? cand : 0u);
        const uint64_t mine = leftm ? vl : vr;
        const uint32_t pm = leftm ? pl : pr;
        const uint32_t moves = opq(onpath & m_le(pm, vprio));        // prio(mine) <= prio(value)
        const unsigned long long pathm = __ballot(onpath != 0u), mvm = __ballot(moves != 0u);
        LAMA_LOCKSTEP();
        uint64_t* dst = moves ? h + parent : dummy;
        *dst = mine;
        if (r == 0) { mine0 = mine; root_child = opq(((lane == 1 || lane == 2) ? 0xFFFFFFFFu : 0u) & moves); root_value = (mvm & 6ull) == 0ull; }
        const int rel = 63 - __clzll((long long)(mvm | 1ull));       // deepest moved entry: its old slot is the new hole
        const uint32_t Hn = (H << (31 - __clz(rel + 1))) + (uint32_t)rel;
        H = mvm ? Hn : H;
        go = (mvm != pathm) ? 0u : go;
        LAMA_LOCKSTEP();
    }
    // the hole has a lone left child, the array's last entry (rare: only when the descent ran to the very end of the array)
    const bool up = go != 0u && (len & 1u) == 0u && H == (len - 2) / 2 && heap_prio(tailc) <= vprio;
    if (__builtin_expect(up, 0)) {                                   // (rare: laid out of line, the common path falls through)
        if (lane == 0) { h[H] = tailc; if (H == 0) *root_out = tailc; }
        root_value = root_value && H != 0;
        H = len - 1;
        LAMA_LOCKSTEP();
    }
    // ONE store for what is left: the re-inserted entry into the hole (lane 0), and the new root for the main wave -- the child
    // that moved up in the first round (lane 1 or 2 still holds it) or the re-inserted entry itself (lane 3)
    {
        const uint32_t l0 = lane == 0 ? 0xFFFFFFFFu : 0u, l3 = (lane == 3 && root_value) ? 0xFFFFFFFFu : 0u;
        uint64_t* dst = l0 ? h + H : ((l3 | root_child) ? root_out : dummy);
        const uint64_t v = root_child ? mine0 : value;
        *dst = v;
    }
    LAMA_LOCKSTEP();
}

// push_heap of the mailbox entries ent[0 .. cnt) (cnt <= 4, in order) by the helper wave: one gather of all would-be parents; when
// no new entry has to move up (the normal case in a Dijkstra wave) they are appended, which is what the sequential push_heap calls
// would have done; else those calls are replayed one by one.
__device__ __forceinline__ void lds_push_flat(uint64_t* heap, uint32_t& n, const uint64_t* ent, const uint32_t* cnt_p, const int lane, uint64_t* dummy)
{
    const uint32_t l4 = (uint32_t)lane & 3u;
    const uint64_t entry = ent[l4];                                  // count, entries and would-be parents: ONE LDS round trip
    const uint32_t pos = n + l4;
    const uint32_t pprio = heap_prio(heap[n >= 4 ? (pos - 1) / 2 : 0]);
    const uint32_t cnt_v = *cnt_p;
    const bool mine = (uint32_t)lane < cnt_v;
    const bool up = mine && pprio > heap_prio(entry);
    const unsigned long long upm = __ballot(up);
    const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_v);
    if (cnt == 0) return;
    if (__builtin_expect(n >= 4 && upm == 0ull, 1)) {
        uint64_t* dst = mine ? heap + pos : dummy;
        *dst = entry;
        n += cnt;
        LAMA_LOCKSTEP();
        return;
    }
    #pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) lds_push(heap, n, ent[i], lane == 0);
}

// the same when the caller holds the (wave-uniform) count in a register: the one-wave form of the straight-line pop pushes its own
// mailbox without the round trip through the count word
__device__ __forceinline__ void lds_push_flat_n(uint64_t* heap, uint32_t& n, const uint64_t* ent, const uint32_t cnt, const int lane, uint64_t* dummy)
{
    if (cnt == 0) return;
    const uint32_t l4 = (uint32_t)lane & 3u;
    const uint64_t entry = ent[l4];
    const uint32_t pos = n + l4;
    const uint32_t pprio = heap_prio(heap[n >= 4 ? (pos - 1) / 2 : 0]);
    const bool mine = (uint32_t)lane < cnt;
    const bool up = mine && pprio > heap_prio(entry);
    const unsigned long long upm = __ballot(up);
    if (n >= 4 && upm == 0ull) {
        uint64_t* dst = mine ? heap + pos : dummy;
        *dst = entry;
        n += cnt;
        LAMA_LOCKSTEP();
        return;
    }
    #pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) lds_push(heap, n, ent[i], lane == 0);
}

// push_heap of `cnt` (<= 4) entries ent[0 .. cnt), in order.  Fast path: one gather of all would-be parents; if none of the new
// entries has to move up (parent priority <= new priority: the normal case in a Dijkstra wave) they are simply appended, exactly
// what the sequential push_heap calls would have done.  Returns true if the entries were appended unmoved.
__device__ __forceinline__ bool lds_push_list(uint64_t* heap, uint32_t& n, const uint64_t* ent, const uint32_t cnt, int lane)
{
    if (!cnt) return false;
    const uint32_t l4 = (uint32_t)lane & 3u;
    const uint64_t entry = ent[l4];
    const uint32_t pos = n + l4;
    if (n >= 4) {
        const uint32_t pprio = heap_prio(heap[(pos - 1) / 2]);
        const bool mine = (uint32_t)lane < cnt;
        const bool up = mine && pprio > heap_prio(entry);
        if (__ballot(up) == 0) {
            if (mine) heap[pos] = entry;
            n += cnt;
            LAMA_LOCKSTEP();
            return true;
        }
    }
    #pragma unroll 1
    for (uint32_t i = 0; i < cnt; ++i) lds_push(heap, n, ent[i], lane == 0);
    return false;
}

// ------------------------------------------------------------------------------------------------
// k_brushfire -- DynamicDistanceMap::update / raise / lower (src/sdm/dynamic_distance_map.cpp:160-197,
// 244-279, 281-330).  One wave per particle; lane 4 owns the popped cell, lanes 0..3 its 4-neighbours in the
// reference's order (+x, +y, -x, -y).  Queues live in LDS; a particle whose queue would not fit is handed,
// state intact, to k_brushfire_slow (generic code, queues in HBM) through prm.slow[p].
//
// Which get() calls of the reference have side effects (patch allocation + mask bit, src/sdm/map.cpp:371-412)?
//   - the popped cell: none (it was get()-ed when it was queued);
//   - obstacle cells reached through a valid cell's offset: none (an obstacle was get()-ed when it was added;
//     a cleared cell has offset 0 = itself);
//   - the neighbours: YES -- all four in raise(), and in lower() those that pass the "away from the obstacle"
//     test, and only when lower() really runs.  They are prefetched side-effect free and the allocation /
//     mask update is applied once the decision is known.
// ------------------------------------------------------------------------------------------------
//
// TW ("two waves"): the kernel is bound by the instruction issue rate of its single wave, so ALL heap work of both
// queues -- pop()'s sift-down and the push_heap calls, pure LDS code, about half of the instructions of one
// iteration -- is done by a helper wave on another SIMD of the CU, concurrently with the main wave's cell loads and
// raise() / lower() decision.  One workgroup barrier per pop:
//   main  : [knows top e_k] loads, decision, map stores, writes the entries to push into a mailbox | barrier D_k |
//           reads the root the helper saw after pop_k and derives e_{k+1} from it and its own pushes:
//           push_heap moves a new entry above its parent only if the parent's priority is strictly greater, so
//           after the pushes the root is the first pushed entry of the smallest priority if that priority is
//           smaller than the old root's, else the old root -- no need to wait for the pushes themselves;
//   helper: pop_k, publishes the new root | barrier D_k | pushes_k, then straight on to pop_{k+1}.
// The sequence of heap operations is exactly the sequential one, so the result stays bit-identical.
// Workgroup barrier that orders LDS traffic only: the main wave's global stores / atomics of the iteration need not be
// acknowledged before it meets the helper wave (which never touches global memory); __syncthreads() would drain vmcnt.
#ifdef LAMA_WAVE_SIM       // tests/sim: this source compiled for the host under the lane-level simulator (test infrastructure only)
__device__ __forceinline__ void lds_barrier() { __syncthreads(); }
#else
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// a 64-bit value that is the same in every lane, said so to the compiler (two v_readfirstlane; folded away when it already knows)
#ifdef LAMA_WAVE_SIM
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) { return v; }
#else
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v)
{
    return ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
}
#endif

// hand particle p to the next stage: the first stage lists it for the resume stage, the resume stage flags it for k_brushfire_slow
__device__ __forceinline__ void bf_hand_over(const DevParams& prm, int p, bool from_resume)
{
    prm.slow[p] = 1;
    if (!from_resume) prm.slow_list[atomicAdd(prm.slow_n, 1u)] = (uint32_t)p;
}

// the brushfire of ONE particle by the calling workgroup (body of k_brushfire)
template <int LQ_LDS, int RQ_LDS, bool RESUME, bool TW>
__device__ __forceinline__ void bf_particle(const DevParams& prm, const int p_any, BfLds<LQ_LDS, RQ_LDS>& sh)
{
    const int p = __builtin_amdgcn_readfirstlane(p_any);      // wave-uniform (the resume stage reads it from the hand-over list): scalar base addresses
    // routed to the big-queue stage, which runs beside this one (k_bf_route), or a particle of the early lane
    if (!RESUME && prm.heavy != nullptr && (prm.heavy[p] != 0 || (prm.early != nullptr && prm.early[p] != 0))) return;
    const uint32_t handed = RESUME ? prm.slow[p] : 1u;   // resume stage: only particles an earlier stage handed over
    const int lane = threadIdx.x & 63;
    const int tid = threadIdx.x, nthreads = TW ? 2 * UM_BLOCK : UM_BLOCK;
    const PV pv = pview_w(prm, p);                             // (p is wave-uniform, both waves are complete here)
    int16_t* dir = pv.dm_dir;
    sv_t* sv = pv.dm_sv;
    uint32_t* obs = pv.dm_obs;
    uint64_t* mask = pv.dm_mask;
    const uint32_t dm_cap = pv.dm_cap;
    uint64_t* g_lower = prm.q_lower + (size_t)p * prm.qcap;
    uint64_t* g_raise = prm.q_raise + (size_t)p * prm.qcap;
    // (the counts and queue sizes are the ray-cast kernels' -- of another stream of the context when this stage was routed or belongs to
    // the early lane: coherent uniform loads, lama_dev.h)
    int count = uload_i32(prm.counts + 2 * p);
    const uint64_t nlr = uload_u64(prm.qsizes + 2 * p);
    uint32_t nl = (uint32_t)nlr, nr = (uint32_t)(nlr >> 32);
    // BOTH waves must have read the hand-over flag before thread 0 clears it (the helper wave may start later than the main wave)
    if (RESUME && TW) __syncthreads(); else LAMA_LOCKSTEP();
    if (RESUME && handed == 0) return;
    if (tid == 0) prm.slow[p] = 0;
    if (nl == 0 && nr == 0) return;
    if (nl + 4 > (uint32_t)LQ_LDS || nr + 4 > (uint32_t)RQ_LDS) { if (tid == 0) bf_hand_over(prm, p, RESUME); return; }

    for (int k = tid; k < DC_SIZE; k += nthreads) sh.dc[k] = DC_EMPTY;
    for (uint32_t k = tid; k < nl; k += nthreads) sh.lower[k] = g_lower[k];
    for (uint32_t k = tid; k < nr; k += nthreads) sh.raise[k] = g_raise[k];
    __syncthreads();
    const uint64_t anc = lds_pop_ancestors(lane);
    // lower phase of the wave pair: the raise queue is empty for good (lower() never raises) and lies directly behind the lower heap in
    // LDS -- the heap may grow into it (first stage: 1,280 entries instead of 1,024 before the particle is handed over; the big stage
    // keeps its own limit, which is what cfg.queue_capacity is checked against)
    constexpr uint32_t LOWER_CAP = RESUME ? (uint32_t)LQ_LDS : (uint32_t)(LQ_LDS + RQ_LDS);
#ifndef LAMA_WAVE_SIM
    // Chip full (12 workgroups per CU: three main and three helper waves per SIMD): the issue slots of a SIMD are the scarce
    // resource and the main wave is the longer chain of the pair (the helper waits for it), so it gets the arbiter's preference.
    if (TW) { if (tid >= UM_BLOCK) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3); }
#endif
    if (TW && tid >= UM_BLOCK) {
        // helper wave: owns both queues from the hand-over on.  While the raise queue is not empty it is the one popped
        // (dynamic_distance_map.cpp:162-173), then the lower queue (:175-194).
        __syncthreads();                                       // S0: heaps consistent
        uint32_t hnl = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.cmd), hnr = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.cmd_r);
        if (hnl == BF_CMD_EXIT) return;
#ifdef LAMA_PROFILE_BF
        uint64_t hp[3] = {0, 0, 0};
        uint64_t ht = __builtin_readcyclecounter();
        #define HFT(k) do { const uint64_t t_ = __builtin_readcyclecounter(); hp[k] += t_ - ht; ht = t_; } while (0)
#else
        #define HFT(k) do {} while (0)
#endif
        // rounds of lds_pop_flat, 5 heap levels each: two reach the bottom of a heap of up to 2,047 entries (levels 0 .. 10), three
        // that of 8,192.  The resume stage's big queues are mostly far from full: it takes the third round only when the heap needs it.
        constexpr int NR = LQ_LDS > 2047 ? 3 : 2;
        #define BF_HELPER_POP(H, N, B) do { if (NR == 3 && (N) > 2047u) lds_pop_flat<3>(H, N, lane, anc, &sh.topq[B], dmy); else lds_pop_flat<2>(H, N, lane, anc, &sh.topq[B], dmy); } while (0)
        uint64_t* const dmy = sh.dummy[1] + (lane & (BF_DUMMY - 1));
        uint32_t it = 0;
        bool hstop = false;
        while (hnr > 0) {                                      // raise phase: pops of the raise queue, pushes into both
            const uint32_t b = it & 1u;
            BF_HELPER_POP(sh.raise, hnr, b);
            lds_barrier();                                     // D
            lds_push_flat(sh.raise, hnr, sh.pr_e[b], &sh.pr_n[b], lane, dmy);   // raise() is the only producer of raise entries
            lds_push_flat(sh.lower, hnl, sh.pl_e[b], &sh.pl_n[b], lane, dmy);
            ++it;
            if (hnl + 4 > (uint32_t)LQ_LDS || (hnr > 0 && hnr + 4 > (uint32_t)RQ_LDS) || (hnr == 0 && hnl == 0)) { hstop = true; break; }   // the main wave takes the same decision
            if (hnr == 0) lds_barrier();                       // X: phase switch, the main wave reads lower[0] after the pushes
        }
        if (!hstop && hnl > 0) for (;;) {                      // lower phase
            const uint32_t b = it & 1u;
            BF_HELPER_POP(sh.lower, hnl, b);
            HFT(0);
            lds_barrier();                                     // D
            HFT(1);
            lds_push_flat(sh.lower, hnl, sh.pl_e[b], &sh.pl_n[b], lane, dmy);
            HFT(2);
            ++it;
            if (hnl - 1u >= LOWER_CAP - 4u) break;   // empty, or about to outgrow the LDS window (in the lower phase the idle raise queue behind it is part of it)
        }
#ifdef LAMA_PROFILE_BF
#ifndef LAMA_PROFILE_BF_MAIN
        if (lane == 0) for (int k = 0; k < 3; ++k) prm.dbg[8 * p + 5 + k] = hp[k];
#endif
#endif
        #undef HFT
        #undef BF_HELPER_POP
        __syncthreads();                                       // F: last pushes applied
        return;
    }
    const DirCache dc{sh.dc, dir, prm.W};

    const int ddx = lane == 0 ? 1 : (lane == 2 ? -1 : 0), ddy = lane == 1 ? 1 : (lane == 3 ? -1 : 0);
    const bool is_cur = lane == 4, is_nb = lane < 4;
    uint64_t processed = 0;
    bool spill = false;
#ifdef LAMA_PROFILE_BF
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = __builtin_readcyclecounter();
#endif

    // round A: every role lane loads its own cell (side-effect free)
    #define BF_LOAD_A()                                                                                     \
        const int x = rx + ddx, y = ry + ddy;                                                               \
        const bool role = lane < 5;                                                                         \
        const bool inwin = (uint32_t)x < prm.WC && (uint32_t)y < prm.WC;                                    \
        const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);                              \
        const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);                               \
        int slot = (role && inwin) ? dc.lookup(pidx) : -1;                                                  \
        sv_t s = 0; uint32_t ob = 0;                                                                    \
        if (slot >= 0) { s = sv[slot * 1024 + (int)ci]; ob = obs[slot * 1024 + (int)ci]; }
    // round B: the cell my offset points to (obstacle cell); offset 0 -> myself
    #define BF_LOAD_B()                                                                                     \
        const int ox = x + obs_x(ob), oy = y + obs_y(ob);                                                   \
        sv_t os = 0;                                                                                    \
        {                                                                                                   \
            const bool oin = role && slot >= 0 && (uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC;           \
            const uint32_t opidx = ((uint32_t)oy >> 5) * prm.W + ((uint32_t)ox >> 5);                       \
            const int oslot = oin ? dc.lookup(opidx) : -1;                                                  \
            if (oslot >= 0) os = sv[oslot * 1024 + (int)(((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5))]; \
        }
    // pop() of the LDS heap H with both load rounds issued underneath the sift-down
    #define BF_POP_WITH_LOADS(H, N)                                                                         \
        BF_LOAD_A()                                                                                         \
        BFT(1);                                                                                             \
        if (TW) { --N; /* the helper wave pops */ } else { (void)lds_pop_topdown(H, N, lane, anc); }        \
        BF_LOAD_B()                                                                                         \
        BFT(2);

    // TW: hand both queues to the helper wave; from here on the main wave derives every next top itself
    uint64_t e_next = 0;
    uint32_t tw_it = 0;
    bool tw_running = false, tw_go = false;
    if (TW) {
        spill = nl + 4 > (uint32_t)LQ_LDS || (nr > 0 && nr + 4 > (uint32_t)RQ_LDS);
        tw_running = tw_go = !spill && (nr > 0 || nl > 0);
        if (lane == 0) { sh.cmd = tw_go ? nl : BF_CMD_EXIT; sh.cmd_r = nr; }
        if (tw_go) e_next = uniform_u64(nr > 0 ? sh.raise[0] : sh.lower[0]);
        __syncthreads();                                       // S0
    }
    // end of a TW iteration: deliver the push lists, meet the helper (its pop is done), derive the next top from the root it
    // published and my own pushes into the queue that was popped (push_heap lifts an entry above its parent only if the
    // parent's priority is strictly greater; a pushed entry therefore becomes the root iff its priority is smaller than the
    // root's, the first of the smallest ones).  On the raise -> lower switch the lower queue's root is read after the pushes.
    #define BF_TW_TAIL(WAS_RAISE, NPOP, CNT_R, CNT_L, OWN_ENTRY, OWN_MASK)                                  \
        {                                                                                                   \
            const uint32_t b_ = tw_it & 1u;                                                                 \
            if (lane == 0) { sh.pr_n[b_] = (CNT_R); sh.pl_n[b_] = (CNT_L); }                                \
            /* my own best push: smallest (priority, neighbour index); scalar min over the 4 neighbour lanes */ \
            const uint32_t key_ = (((OWN_MASK) >> (lane & 3)) & 1u) ? ((heap_prio(OWN_ENTRY) << 2) | (uint32_t)(lane & 3)) : 0xFFFFFFFFu; \
            const uint32_t k0_ = (uint32_t)__builtin_amdgcn_readlane((int)key_, 0), k1_ = (uint32_t)__builtin_amdgcn_readlane((int)key_, 1); \
            const uint32_t k2_ = (uint32_t)__builtin_amdgcn_readlane((int)key_, 2), k3_ = (uint32_t)__builtin_amdgcn_readlane((int)key_, 3); \
            const uint32_t k01_ = k0_ < k1_ ? k0_ : k1_, k23_ = k2_ < k3_ ? k2_ : k3_;                      \
            const uint32_t kb_ = k01_ < k23_ ? k01_ : k23_;                                                 \
            const int bl_ = (int)(kb_ & 3u);                                                                \
            const uint32_t olo_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(OWN_ENTRY), bl_);     \
            const uint32_t ohi_ = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((OWN_ENTRY) >> 32), bl_); \
            const bool have_own_ = kb_ != 0xFFFFFFFFu;                                                      \
            BFT_MAIN(5);                                                                                    \
            lds_barrier();                                     /* D */                                      \
            BFT_MAIN(7);                                                                                    \
            const bool have_root_ = (NPOP) > 0;                                                             \
            const uint64_t root_ = sh.topq[b_];                                                             \
            const bool own_wins_ = have_own_ && (!have_root_ || (kb_ >> 2) < heap_prio(root_));             \
            const uint64_t cand_ = own_wins_ ? (((uint64_t)ohi_ << 32) | olo_) : root_;                     \
            nr += (CNT_R); nl += (CNT_L);                                                                   \
            spill = nl + 4 > (uint32_t)LQ_LDS || (nr > 0 && nr + 4 > (uint32_t)RQ_LDS);                      \
            tw_running = !spill && (nr > 0 || nl > 0);                                                     \
            if (tw_running) {                                                                               \
                if ((WAS_RAISE) && nr == 0) { lds_barrier(); /* X */ e_next = sh.lower[0]; }                \
                else e_next = cand_;                                                                        \
            }                                                                                               \
            ++tw_it;                                                                                        \
        }

    // ---- raise wave of the wave pair (round 4: the same straight-line form as the lower wave below) ---------- :162-173, raise() :244-279
    // Lanes 0..3: the four neighbours, lane 4: the popped cell, lane 5: the obstacle cell the popped cell pointed to when it was queued
    // (raise entries carry that offset since round 4; the reference's entries have none and the heap only orders by distance) --
    // inside a raise wave most neighbours point to that very cell, so its state answers "is my obstacle still one?" without the second,
    // dependent load round; a neighbour that points elsewhere gets it (one ballot).  Rare cases -- a cell that is not cached, a
    // patch to allocate, a neighbour whose obstacle cell is another neighbour (raise() handles them in order) -- take the general code.
    if (TW) {
        uint64_t* const dmy = sh.dummy[0] + (lane & (BF_DUMMY - 1));
        const BufRsrc rsv = buf_make(sv, dm_cap * SV_PATCH_BYTES), robs = buf_make(obs, dm_cap * 4096u), rmask = buf_make(mask, dm_cap * 128u);
        const bool is_oc = lane == 5;
        const uint32_t rolem = lane < 6 ? 0xFFFFFFFFu : 0u, nbm = is_nb ? 0xFFFFFFFFu : 0u, curm = is_cur ? 0xFFFFFFFFu : 0u;

        // the general raise pop: the reference's statements with every rare case; leaves the entries to push in the two mailboxes
        auto general_raise = [&](const uint64_t e, uint32_t& cnt_r, uint32_t& cnt_l, uint64_t& entry_r, uint32_t& rm_out) {
            const int rx = q_rx(e), ry = q_ry(e);
            BF_LOAD_A()
            BF_LOAD_B()
            if (is_nb && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool nb = is_nb && inwin;
            const bool fresh = nb && slot < 0;
            if (__ballot(fresh)) { const int ns_ = coop_slot(dc, dir, count, (int)dm_cap, fresh, pidx, ERR_DM_CAP, prm.err); if (fresh) slot = ns_; }
            const bool nbok = nb && slot >= 0;
            if (nbok) {
                const uint64_t bit = 1ull << (ci & 63);
                if (fresh || !(s & (SV_VALID | SV_QUEUED))) atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);
            }
            const bool cand = nbok && !(s & SV_QUEUED) && (s & SV_VALID);        // :253
            bool ovalid = (os & SV_VALID) != 0;
            bool clear = cand && !ovalid;
            #pragma unroll 1
            for (int i = 0; i < 3; ++i) {                                        // neighbour i is handled before j > i
                const int cxi = __builtin_amdgcn_readlane(x, i), cyi = __builtin_amdgcn_readlane(y, i);
                const bool ci_clear = __builtin_amdgcn_readlane((int)clear, i) != 0;
                if (lane > i && lane < 4 && cand && ci_clear && ox == cxi && oy == cyi) { ovalid = false; clear = true; }
            }
            const bool to_raise = cand && !ovalid, to_lower = cand && ovalid;    // :262-272
            const uint32_t rm = (uint32_t)__ballot(to_raise) & 15u, lm = (uint32_t)__ballot(to_lower) & 15u;
            const uint32_t below = (1u << lane) - 1u;
            entry_r = q_entry((uint32_t)(s & SV_SQMASK), x, y, obs_x(ob), obs_y(ob));
            if (to_raise) sh.pr_e[tw_it & 1u][__popc(rm & below)] = entry_r;
            if (to_lower) sh.pl_e[tw_it & 1u][__popc(lm & below)] = entry_r;
            cnt_r = (uint32_t)__popc(rm); cnt_l = (uint32_t)__popc(lm); rm_out = rm;
            if (to_raise) { sv[slot * 1024 + (int)ci] = SV_QUEUED; obs[slot * 1024 + (int)ci] = 0u; }
            if (to_lower) sv[slot * 1024 + (int)ci] = (sv_t)(s | SV_QUEUED);
            if (is_cur && slot >= 0) sv[slot * 1024 + (int)ci] = (sv_t)(s & ~SV_QUEUED);      // :278
        };

        while (tw_running && nr > 0) {
            const uint64_t e = uniform_u64(e_next);                       // (see the lower wave)
            ++processed;
#ifdef LAMA_PROFILE_BF_COUNT
            prof[7] += 1;
#endif
            --nr;                                                          // the helper wave pops
            uint32_t cnt_r = 0, cnt_l = 0, rm = 0;
            uint64_t entry = 0;
            bool general;
            {
#ifdef LAMA_WIDE_DM
                const int rx = q_rx(e), ry = q_ry(e), eox = q_ox(e), eoy = q_oy(e);
#else
                const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
                const int rx = (int)(elo & 0xFFFFu), ry = (int)(elo >> 16);
                const int eox = (int)(ehi & 0xFFu) - 128, eoy = (int)((ehi >> 8) & 0xFFu) - 128;
#endif
                const int obx = rx + eox, oby = ry + eoy;                  // the obstacle cell the popped cell pointed to
                const int x = is_oc ? obx : rx + ddx, y = is_oc ? oby : ry + ddy;
                const uint32_t mxy = (uint32_t)x > (uint32_t)y ? (uint32_t)x : (uint32_t)y;
                const uint32_t inwin = opq(mxy < prm.WC ? 0xFFFFFFFFu : 0u);
                const uint32_t pidx = bf_mul24((uint32_t)y >> 5, prm.W) + ((uint32_t)x >> 5);
                const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
                const uint32_t dv = sh.dc[dc.index(pidx & inwin)];
                const uint32_t hit = opq(inwin & ~m_nz((dv >> 15) ^ (pidx >> 3)));
                const uint32_t slotw = (dv & 0x7FFFu) - 1u;
                const uint32_t absent = opq((uint32_t)((int32_t)slotw >> 31));
                const uint32_t have = opq(rolem & hit & ~absent);
                const uint32_t coff = (slotw << 10) | ci;
                const uint32_t s = buf_load_sv(rsv, m_sel(have, coff * SV_BYTES, BUF_OOB));
                const uint32_t ob = buf_load_u32(robs, m_sel(have, coff * 4u, BUF_OOB));
                // general code: a role cell that is not cached or outside the window, a neighbour (or the popped cell) without a patch
                const uint32_t rare0 = (rolem & ~hit) | ((nbm | curm) & hit & absent);
                // :253  neighbours that are valid and not queued
                const uint32_t svalid = opq((uint32_t)((int32_t)(s << (31 - SV_VALID_BIT)) >> 31)), squeued = opq((uint32_t)((int32_t)(s << (32 - SV_VALID_BIT)) >> 31));
                const uint32_t cand = opq(nbm & svalid & ~squeued);
                // the neighbour's obstacle cell: the one lane 5 holds, or another one (second round)
                const int ox = x + obs_x(ob), oy = y + obs_y(ob);
                const uint32_t same = ~(m_nz((uint32_t)(ox ^ obx)) | m_nz((uint32_t)(oy ^ oby)));
                const uint32_t other = opq(cand & ~same);
                // raise() handles the neighbours in order: one whose obstacle cell is an EARLIER neighbour sees what that one became --
                // only possible when the offset points at a cell next to the popped one: general code
                const int ax = ox - rx, ay = oy - ry;
                const uint32_t adj = cand & ~m_nz((uint32_t)(ax * ax + ay * ay) ^ 1u);
                const unsigned long long rarem = __ballot((rare0 | adj) != 0u);
                const uint32_t cos_ = (uint32_t)__builtin_amdgcn_readlane((int)s, 5);
                uint32_t ovalid = (cos_ & SV_VALID) ? 0xFFFFFFFFu : 0u;   // (lanes whose obstacle cell is lane 5's)
                general = rarem != 0ull;
                if (!general && __builtin_expect(__ballot(other != 0u) != 0ull, 0)) {
                    const uint32_t oin = opq((other != 0u && (uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC) ? 0xFFFFFFFFu : 0u);
                    const uint32_t opidx = bf_mul24((uint32_t)oy >> 5, prm.W) + ((uint32_t)ox >> 5);
                    const uint32_t odv = sh.dc[dc.index(opidx & oin)];
                    const uint32_t ohit = opq((odv >> 15) == (opidx >> 3) ? oin : 0u);
                    const uint32_t oslot = (odv & 0x7FFFu) - 1u;
                    const uint32_t ooff = ((oslot << 10) | ((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5)) * SV_BYTES;
                    const uint32_t os2 = buf_load_sv(rsv, m_sel(ohit & ~(uint32_t)((int32_t)oslot >> 31), ooff, BUF_OOB));   // outside the window / absent: reads as 0
                    ovalid = m_sel(other, (uint32_t)((int32_t)(os2 << (31 - SV_VALID_BIT)) >> 31), ovalid);
                    general = __ballot((oin & ~ohit) != 0u) != 0ull;       // its directory entry is not cached
                }
                if (__builtin_expect(!general, 1)) {
                    const uint32_t to_raise = opq(cand & ~ovalid), to_lower = opq(cand & ovalid);       // :262-272
                    // get() of the four neighbours: the Container mask bit of a flag-less cell
                    const uint32_t need_bit = nbm & ~(svalid | squeued);
                    buf_or_u64(rmask, m_sel(need_bit, ((slotw << 4) + (ci >> 6)) * 8u, BUF_OOB), 1ull << (ci & 63u));
                    // raised: is_queued only, obstacle cleared; lowered: is_queued added; the popped cell: is_queued removed (:278)
                    const uint32_t nsv = m_sel(to_raise, (uint32_t)SV_QUEUED, m_sel(curm, s & ~(uint32_t)SV_QUEUED, s | (uint32_t)SV_QUEUED));
                    buf_store_sv(rsv, m_sel(to_raise | to_lower | curm, coff * SV_BYTES, BUF_OOB), nsv);
                    buf_store_u32(robs, m_sel(to_raise, coff * 4u, BUF_OOB), 0u);
                    const unsigned long long rmm = __ballot(to_raise != 0u), lmm = __ballot(to_lower != 0u);
                    entry = q_entry(s & SV_SQMASK, x, y, obs_x(ob), obs_y(ob));
                    const uint32_t rank_r = opq(lane_rank(rmm, lane) & 3u), rank_l = opq(lane_rank(lmm, lane) & 3u);
                    uint64_t* dst = to_raise ? &sh.pr_e[tw_it & 1u][rank_r] : (to_lower ? &sh.pl_e[tw_it & 1u][rank_l] : dmy);
                    *dst = entry;
                    cnt_r = (uint32_t)__popcll(rmm); cnt_l = (uint32_t)__popcll(lmm); rm = (uint32_t)rmm & 15u;
                }
            }
            if (__builtin_expect(general, 0)) general_raise(e, cnt_r, cnt_l, entry, rm);
#ifndef LAMA_WAVE_SIM
            cnt_r = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_r); cnt_l = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt_l);
            rm = (uint32_t)__builtin_amdgcn_readfirstlane((int)rm);
#endif
            // ---- hand-over (see the lower wave): both counts in one LDS store, meet the helper, next top = the raise heap's root after
            // pop() unless one of my own raise pushes beats it -- push_heap lifts an entry above its parent only if the parent's
            // priority is strictly greater, so the first of my smallest pushes becomes the root iff its priority is smaller than the
            // root's.  (A raise wave's pushes are not ordered against the popped cell: no shortcut as in the lower wave.)
            {
                const uint32_t b_ = tw_it & 1u;
                uint32_t* np = lane == 0 ? &sh.pr_n[b_] : (lane == 1 ? &sh.pl_n[b_] : (uint32_t*)dmy);
                *np = lane == 0 ? cnt_r : cnt_l;
                lds_barrier();                                     // D
                const uint64_t root_ = sh.topq[b_];
                const uint32_t rlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)root_), rhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(root_ >> 32));
                const bool have_root = nr > 0;
                uint64_t cand_ = ((uint64_t)rhi << 32) | rlo;
                if (cnt_r > 0) {
                    uint32_t key = (lane < 4 && ((rm >> lane) & 1u)) ? ((heap_prio(entry) << 2) | (uint32_t)lane) : 0xFFFFFFFFu;
                    uint32_t blo = (uint32_t)entry, bhi = (uint32_t)(entry >> 32);
#define BF_QUAD_MIN(CTRL)                                                                                              \
                    {                                                                                                  \
                        const uint32_t k2 = (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, CTRL, 0xF, 0xF, false); \
                        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp((int)blo, (int)blo, CTRL, 0xF, 0xF, false); \
                        const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp((int)bhi, (int)bhi, CTRL, 0xF, 0xF, false); \
                        const bool t_ = k2 < key;                                                                      \
                        key = t_ ? k2 : key; blo = t_ ? l2 : blo; bhi = t_ ? h2 : bhi;                                 \
                    }
                    BF_QUAD_MIN(0xB1)                              // quad_perm [1,0,3,2]
                    BF_QUAD_MIN(0x4E)                              // quad_perm [2,3,0,1]
#undef BF_QUAD_MIN
                    const uint32_t okey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
                    const uint32_t olo = (uint32_t)__builtin_amdgcn_readfirstlane((int)blo), ohi = (uint32_t)__builtin_amdgcn_readfirstlane((int)bhi);
                    if (!have_root || (okey >> 2) < (rhi >> 16)) cand_ = ((uint64_t)ohi << 32) | olo;
                }
                nr += cnt_r; nl += cnt_l;
                spill = nl + 4 > (uint32_t)LQ_LDS || (nr > 0 && nr + 4 > (uint32_t)RQ_LDS);
                tw_running = !spill && (nr > 0 || nl > 0);
                if (tw_running) {
                    if (nr == 0) { lds_barrier(); /* X: the helper has applied the pushes */ e_next = uniform_u64(sh.lower[0]); }
                    else e_next = cand_;
                }
                ++tw_it;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        }
    }
    // ---- raise wave, one-wave form ------------------------------------------------------------ :162-173
    while (!TW && nr > 0) {
        if (!TW && (nl + 4 > (uint32_t)LQ_LDS || nr + 4 > (uint32_t)RQ_LDS)) { spill = true; break; }
        const uint64_t e = TW ? e_next : sh.raise[0];    // priority_queue::top(); the cell loads below are in
        const int rx = q_rx(e), ry = q_ry(e);            // flight while pop() sifts the heap in LDS
        ++processed;
#ifdef LAMA_PROFILE_BF_COUNT
        prof[7] += 1;
#endif
        BF_POP_WITH_LOADS(sh.raise, nr)
        // neighbours: get() = allocate + mask bit (all four)
        if (is_nb && !inwin) atomicOr(prm.err, ERR_WINDOW);
        const bool nb = is_nb && inwin;
        const bool fresh = nb && slot < 0;
        if (__ballot(fresh)) { const int ns_ = coop_slot(dc, dir, count, (int)dm_cap, fresh, pidx, ERR_DM_CAP, prm.err); if (fresh) slot = ns_; }
        const bool nbok = nb && slot >= 0;
        if (nbok) {
            const uint64_t bit = 1ull << (ci & 63);
            // the Container mask bit of a cell with a flag set is on already (every writer get()s the cell first): only
            // flag-less cells need the OR (idempotent for those that were touched before) -- no mask word is loaded
            if (fresh || !(s & (SV_VALID | SV_QUEUED))) atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);
        }
        // :253  skip queued or invalid neighbours
        const bool cand = nbok && !(s & SV_QUEUED) && (s & SV_VALID);
        bool ovalid = (os & SV_VALID) != 0;
        // sequential semantics: neighbour i is handled before j > i; if i gets cleared and j's offset points
        // at i, j must see i as invalid (only possible through stale offsets; replayed here to stay exact)
        bool clear = cand && !ovalid;
        #pragma unroll 1
        for (int i = 0; i < 3; ++i) {
            const int cxi = __builtin_amdgcn_readlane(x, i), cyi = __builtin_amdgcn_readlane(y, i);
            const bool ci_clear = __builtin_amdgcn_readlane((int)clear, i) != 0;
            if (lane > i && lane < 4 && cand && ci_clear && ox == cxi && oy == cyi) { ovalid = false; clear = true; }
        }
        const bool to_raise = cand && !ovalid;           // :262-268
        const bool to_lower = cand && ovalid;            // :269-272
        uint32_t rw_cnt_r = 0, rw_cnt_l = 0, rw_rm = 0;
        uint64_t rw_entry_r = 0;
        if (TW) {                                        // pushes in neighbour order: the helper wave applies them
            const uint32_t rm = (uint32_t)__ballot(to_raise) & 15u, lm = (uint32_t)__ballot(to_lower) & 15u;
            const uint32_t below = (1u << lane) - 1u;
            rw_entry_r = q_entry((uint32_t)(s & SV_SQMASK), x, y);
            if (to_raise) sh.pr_e[tw_it & 1u][__popc(rm & below)] = rw_entry_r;
            if (to_lower) sh.pl_e[tw_it & 1u][__popc(lm & below)] = q_entry((uint32_t)(s & SV_SQMASK), x, y, obs_x(ob), obs_y(ob));
            rw_cnt_r = (uint32_t)__popc(rm); rw_cnt_l = (uint32_t)__popc(lm); rw_rm = rm;
        }
        #pragma unroll 1
        for (int i = 0; !TW && i < 4; ++i) {
            const bool r_i = __builtin_amdgcn_readlane((int)to_raise, i) != 0, l_i = __builtin_amdgcn_readlane((int)to_lower, i) != 0;
            if (r_i || l_i) {
                const uint32_t prio = (uint32_t)__builtin_amdgcn_readlane((int)(s & SV_SQMASK), i);
                const int ex = __builtin_amdgcn_readlane(x, i), ey = __builtin_amdgcn_readlane(y, i);
                const uint32_t eo = (uint32_t)__builtin_amdgcn_readlane((int)ob, i);
                if (r_i) lds_push(sh.raise, nr, q_entry(prio, ex, ey), lane == 0);
                else lds_push(sh.lower, nl, q_entry(prio, ex, ey, obs_x(eo), obs_y(eo)), lane == 0);
            }
        }
        if (to_raise) { sv[slot * 1024 + (int)ci] = SV_QUEUED; obs[slot * 1024 + (int)ci] = 0u; }
        if (to_lower) sv[slot * 1024 + (int)ci] = (sv_t)(s | SV_QUEUED);
        if (is_cur && slot >= 0) sv[slot * 1024 + (int)ci] = (sv_t)(s & ~SV_QUEUED);      // :278
        if (TW) BF_TW_TAIL(true, nr, rw_cnt_r, rw_cnt_l, rw_entry_r, rw_rm)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }

    // ---- lower wave of the wave pair (round 4: speculative straight-line pop, general code for the rare cases) ---- :175-194
    // Round 6: the same straight-line pop as a ONE-WAVE form (TW == false): the wave pops its heap itself (lds_pop_flat, issued under
    // the cell loads) and pushes its own mailbox -- no helper wave, no barrier, no second copy of the loop control.  Fewer
    // instructions per pop in total than the pair, a longer chain per pop: the form for a chip that is full of particles, where the
    // instruction streams of the pairs compete for the same issue slots (section 8 of DESIGN.md).
    constexpr bool PAIR = TW;
    if (TW || !spill) {
        if (!TW) { tw_running = nl > 0; if (tw_running) e_next = uniform_u64(sh.lower[0]); }
        BFT(7);                                                    // (profiling build: everything before the lower wave)
        uint64_t* const dmy = sh.dummy[0] + (lane & (BF_DUMMY - 1));
        const BufRsrc rsv = buf_make(sv, dm_cap * SV_PATCH_BYTES), robs = buf_make(obs, dm_cap * 4096u), rmask = buf_make(mask, dm_cap * 128u);
        const bool is_oc = lane == 5, role = lane < 6;
        const uint32_t rolem = role ? 0xFFFFFFFFu : 0u, nbm = is_nb ? 0xFFFFFFFFu : 0u, curm = is_cur ? 0xFFFFFFFFu : 0u;

        // The general pop: the reference's statements with every rare case (cache miss, patch allocation, stale entry, second
        // load round of a tie, window error).  Leaves the entries to push in the mailbox and returns how many.
        auto general_pop = [&](const uint64_t e, bool& over_out, uint64_t& entry_out) -> uint32_t {
            over_out = false; entry_out = 0;
            const int rx = q_rx(e), ry = q_ry(e);
            // ONE load round: lanes 0..4 their own cell, lane 5 the obstacle cell the entry says the popped cell points to
            const int x = rx + (is_oc ? q_ox(e) : ddx), y = ry + (is_oc ? q_oy(e) : ddy);
            const bool inwin = (uint32_t)x < prm.WC && (uint32_t)y < prm.WC;
            const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);
            const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
            int slot = (role && inwin) ? dc.lookup(pidx) : -1;
            sv_t s = 0; uint32_t ob = 0;
            if (slot >= 0) { s = sv[slot * 1024 + (int)ci]; if (!is_oc) ob = obs[slot * 1024 + (int)ci]; }
            const sv_t cs = (sv_t)__builtin_amdgcn_readlane((int)s, 4);
            const uint32_t cob = (uint32_t)__builtin_amdgcn_readlane((int)ob, 4);
            sv_t cos_ = (sv_t)__builtin_amdgcn_readlane((int)s, 5);
            if (obs_x(cob) != q_ox(e) || obs_y(cob) != q_oy(e)) {
                // stale entry (the cell was overwritten after it was queued): fetch the obstacle cell it points to now
                const int ox2 = rx + obs_x(cob), oy2 = ry + obs_y(cob);
                sv_t t = 0;
                if ((uint32_t)ox2 < prm.WC && (uint32_t)oy2 < prm.WC) {
                    const int os2 = dc.lookup(((uint32_t)oy2 >> 5) * prm.W + ((uint32_t)ox2 >> 5));
                    if (os2 >= 0) t = sv[os2 * 1024 + (int)(((uint32_t)ox2 & 31u) | (((uint32_t)oy2 & 31u) << 5))];
                }
                cos_ = t;
            }
            // :183-192  valid, its obstacle still has sqdist 0 (valid NOT tested), and lower() :283 still queued
            const bool fire = (cs & SV_VALID) && (cos_ & SV_SQMASK) == 0 && (cs & SV_QUEUED);
            if (!fire) return 0u;
            const int cox = obs_x(cob), coy = obs_y(cob);
            const int obx = rx + cox, oby = ry + coy;
            const bool away = is_nb && !(ddx * cox > 0 || ddy * coy > 0);             // :296
            if (away && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool nb = away && inwin;
            const bool fresh = nb && slot < 0;
            if (__ballot(fresh)) { const int ns_ = coop_slot(dc, dir, count, (int)dm_cap, fresh, pidx, ERR_DM_CAP, prm.err); if (fresh) slot = ns_; }
            const bool nbok = nb && slot >= 0;
            if (nbok) {
                const uint64_t bit = 1ull << (ci & 63);
                if (fresh || !(s & (SV_VALID | SV_QUEUED))) atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);   // see raise()
            }
            const int qx = x - obx, qy = y - oby;
            const uint32_t new_sq = (uint32_t)(qx * qx + qy * qy);
            const uint32_t cmp = (s & SV_VALID) ? (uint32_t)(s & SV_SQMASK) : prm.max_sqdist;
            bool over = nbok && new_sq < cmp;
            const bool tie = nbok && !over && new_sq == (uint32_t)(s & SV_SQMASK);     // :311-317
            if (__ballot(tie)) {
                // the neighbour's own obstacle cell: usually the very cell the popped cell points to, whose state lane 5 holds
                const int ox = x + obs_x(ob), oy = y + obs_y(ob);
                const bool same = ox == obx && oy == oby;
                sv_t os = cos_;
                if (__ballot(tie && !same)) {
                    if (tie && !same) {
                        os = 0;
                        if ((uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC) {
                            const int oslot = dc.lookup(((uint32_t)oy >> 5) * prm.W + ((uint32_t)ox >> 5));
                            if (oslot >= 0) os = sv[oslot * 1024 + (int)(((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5))];
                        }
                    }
                }
                if (tie && (!(s & SV_VALID) || !((os & SV_VALID) && (os & SV_SQMASK) == 0))) over = true;
            }
            if (over) {
                sv[slot * 1024 + (int)ci] = (sv_t)(SV_VALID | SV_QUEUED | (new_sq & SV_SQMASK));
                obs[slot * 1024 + (int)ci] = pack_obs(obx - x, oby - y);
            }
            if (is_cur && slot >= 0) sv[slot * 1024 + (int)ci] = (sv_t)(cs & ~SV_QUEUED);   // :329
            const unsigned long long om = __ballot(over);
            over_out = over; entry_out = q_entry(new_sq, x, y, obx - x, oby - y);
            if (over) sh.pl_e[tw_it & 1u][__popcll(om & ((1ull << lane) - 1ull))] = entry_out;
            return (uint32_t)__popcll(om);
        };

        if (tw_running && nl > 0) for (;;) {
            // the same value in every lane -- SAID so (round 6): as a vector value the entry makes `stale`, hence `general`, hence every
            // branch of the pop divergent to the compiler, and everything defined under them (the push count, the queue length, the
            // next-top test, the loop test) a vector phi with exec-mask control flow: ~12 VALU instructions and half a dozen
            // vector -> scalar hand-overs behind the barrier, on the chain of every pop
            const uint64_t e = uniform_u64(e_next);
            ++processed;
            if (PAIR) --nl;                                                // the helper wave pops
            uint32_t cnt = 0;
            bool general;
            uint32_t floor_sq = 0;                                         // every push of this pop has a priority above this
            bool over = false;                                             // lanes 0..3: my neighbour is lowered and pushed ...
            uint64_t entry = 0;                                            // ... as this queue entry
            {
                // ---- speculative loads: every lane its role's cell through the directory cache, absent / foreign lanes out of range.
                // Lane predicates are 0 / ~0 words (see lds_pop_flat); what is wave-uniform stays in scalar registers.
#ifdef LAMA_WIDE_DM
                const int rx = q_rx(e), ry = q_ry(e), eox = q_ox(e), eoy = q_oy(e);
#else
                const uint32_t elo = (uint32_t)e, ehi = (uint32_t)(e >> 32);
                const int rx = (int)(elo & 0xFFFFu), ry = (int)(elo >> 16);
                const int eox = (int)(ehi & 0xFFu) - 128, eoy = (int)((ehi >> 8) & 0xFFu) - 128;
#endif
                const int x = rx + (is_oc ? eox : ddx), y = ry + (is_oc ? eoy : ddy);
                const uint32_t mxy = (uint32_t)x > (uint32_t)y ? (uint32_t)x : (uint32_t)y;
                const uint32_t inwin = opq(mxy < prm.WC ? 0xFFFFFFFFu : 0u);
                const uint32_t pidx = bf_mul24((uint32_t)y >> 5, prm.W) + ((uint32_t)x >> 5);     // (24-bit multiply: both factors < 2^11)
                const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
                const uint32_t dv = sh.dc[dc.index(pidx & inwin)];
                // (an empty cache word has a tag no window position has)
                const uint32_t hit = opq(inwin & ~m_nz((dv >> 15) ^ (pidx >> 3)));
                const uint32_t slotw = (dv & 0x7FFFu) - 1u;
                const uint32_t absent = opq((uint32_t)((int32_t)slotw >> 31));                // cached "no such patch"
                const uint32_t have = opq(rolem & hit & ~absent);
                const uint32_t unknown = opq(rolem & ~hit);                                   // not cached / outside the window: general code
                const uint32_t coff = (slotw << 10) | ci;
                const uint32_t s = buf_load_sv(rsv, m_sel(have, coff * SV_BYTES, BUF_OOB));
                const uint32_t ob = buf_load_u32(robs, m_sel(have, coff * 4u, BUF_OOB));
                const unsigned long long unk = __ballot(unknown != 0u);
                // Round 5: what lower() computes from the popped cell's obstacle offset alone -- the "away" test, every neighbour's new
                // squared distance, the offset it would store and the entry it would push (:296-305, :319-326) -- needs no loaded
                // value on this path: the entry carries the offset the cell had when it was queued, and an entry whose cell says
                // otherwise (`stale`) goes to the general code.  Written HERE, between the loads and their first use, these ~30
                // instructions run in the shadow of the load round instead of behind it (profiles/r05_pop_floor_table.md).
                const int obx = rx + eox, oby = ry + eoy;
                const uint32_t away = opq(nbm & ~m_pos(ddx * eox + ddy * eoy));                     // :296 (one of ddx, ddy is 0)
                const int qx = x - obx, qy = y - oby;
                const uint32_t new_sq = pin((uint32_t)(qx * qx + qy * qy));
                const uint32_t my_obs = pin(pack_obs(obx - x, oby - y));                           // what I would store: my obstacle seen from the neighbour
                const uint64_t entry_raw = q_entry(new_sq, x, y, obx - x, oby - y);
                const uint64_t entry_spec = ((uint64_t)pin((uint32_t)(entry_raw >> 32)) << 32) | pin((uint32_t)entry_raw);
                const uint32_t nsv_new = pin((uint32_t)(SV_VALID | SV_QUEUED) | (new_sq & SV_SQMASK));
                const uint32_t mask_off = pin(((slotw << 4) + (ci >> 6)) * 8u);
                const uint32_t mask_bit_lo = pin((ci & 32u) ? 0u : (1u << (ci & 31u))), mask_bit_hi = pin((ci & 32u) ? (1u << (ci & 31u)) : 0u);
                const uint64_t mask_bit = ((uint64_t)mask_bit_hi << 32) | mask_bit_lo;
                BFT(0); BFF(0);
                if (!PAIR) {                                               // one-wave form: pop() of the heap while the cell loads are in flight
                    if (LQ_LDS > 2047 && nl > 2047u) lds_pop_flat<3>(sh.lower, nl, lane, anc, &sh.topq[tw_it & 1u], dmy);
                    else lds_pop_flat<2>(sh.lower, nl, lane, anc, &sh.topq[tw_it & 1u], dmy);
                }
                // the popped cell, and the obstacle cell it pointed to when it was queued
                const uint32_t cs = (uint32_t)__builtin_amdgcn_readlane((int)s, 4);
                const uint32_t cob = (uint32_t)__builtin_amdgcn_readlane((int)ob, 4);
                const uint32_t cos_ = (uint32_t)__builtin_amdgcn_readlane((int)s, 5);
                // (one compare of the packed offsets, one of both flags: this test is scalar code on the chain of every pop)
                const bool stale = cob != pack_obs(eox, eoy);
                const bool fire0 = (cs & (uint32_t)(SV_VALID | SV_QUEUED)) == (uint32_t)(SV_VALID | SV_QUEUED);      // :183, lower() :283
                general = (unk != 0ull) | (fire0 & stale);
                BFT(1); BFF(1);
#ifdef LAMA_PROFILE_BF_COUNT
                prof[2] += unk != 0ull ? 1 : 0; prof[3] += (fire0 & stale) ? 1 : 0; prof[6] += (fire0 & ((cos_ & SV_SQMASK) == 0u)) ? 1 : 0;
#endif
                if (__builtin_expect(!general && fire0 && (cos_ & SV_SQMASK) == 0u, 1)) {   // :191 (valid NOT tested); the likely path, so that it is laid out in line: two taken branches less per firing pop
                    floor_sq = cs & SV_SQMASK;                                           // lower() :303: candidates of cells further from the obstacle
                    const uint32_t nbok = away & ~absent;                                // (cox == eox, coy == eoy here: not stale)
                    const uint32_t ssq = s & SV_SQMASK;
                    const uint32_t svalid = opq((uint32_t)((int32_t)(s << (31 - SV_VALID_BIT)) >> 31));   // the valid bit
                    const uint32_t cmp = m_sel(svalid, ssq, prm.max_sqdist);
                    const uint32_t lt = opq(nbok & m_lt(new_sq, cmp));
                    const uint32_t tie = opq(nbok & ~lt & ~m_nz(new_sq ^ ssq));                       // :311-317
                    // the neighbour points at another obstacle than mine (lane 5 holds mine): a second load round, one pop in six
                    const uint32_t other = m_nz(ob ^ my_obs);
                    const uint32_t tie_other = opq(tie & other);
                    const uint32_t alloc = away & absent;                                            // a patch to allocate: general code
                    const bool clive = ((cos_ & SV_VALID) != 0u);                                     // my obstacle cell is a live obstacle ((cos_ & SQMASK) == 0 here)
                    uint32_t dead = clive ? 0u : 0xFFFFFFFFu;                                        // per lane: the neighbour's obstacle cell is not one
                    BFF(2);
                    if (__builtin_expect(__ballot((tie_other | alloc) != 0u) != 0ull, 0)) {
                        const int ox = x + obs_x(ob), oy = y + obs_y(ob);
                        const uint32_t oin = opq((tie_other != 0u && (uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC) ? 0xFFFFFFFFu : 0u);
                        const uint32_t opidx = bf_mul24((uint32_t)oy >> 5, prm.W) + ((uint32_t)ox >> 5);
                        const uint32_t odv = sh.dc[dc.index(opidx & oin)];
                        const uint32_t ohit = opq((odv >> 15) == (opidx >> 3) ? oin : 0u);
                        const uint32_t oslot = (odv & 0x7FFFu) - 1u;
                        const uint32_t ooff = ((oslot << 10) | ((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5)) * SV_BYTES;
                        const uint32_t os2 = buf_load_sv(rsv, m_sel(ohit & ~(uint32_t)((int32_t)oslot >> 31), ooff, BUF_OOB));   // outside the window / absent: reads as 0
                        const uint32_t olive = opq((uint32_t)((int32_t)(os2 << (31 - SV_VALID_BIT)) >> 31) & ~m_nz(os2 & SV_SQMASK));
                        dead = m_sel(tie_other, ~olive, dead);
                        // rare among the rare: a directory entry that is not cached, or a patch to allocate: general code
                        general = __ballot(((oin & ~ohit) | alloc) != 0u) != 0ull;
                    }
#ifdef LAMA_PROFILE_BF_COUNT
                    prof[4] += __ballot(alloc != 0u) ? 1 : 0; prof[5] += __ballot(tie_other != 0u) ? 1 : 0;
#endif
                    if (__builtin_expect(!general, 1)) {
                        BFF(3);
                        const uint32_t overm = opq(lt | (tie & (~svalid | dead)));                    // :308-317
                        over = overm != 0u;
                        // get() of the examined neighbours: the Container mask bit of a flag-less cell (raise() has the argument)
                        const uint32_t need_bit = nbok & ~m_nz(s & (uint32_t)(SV_VALID | SV_QUEUED));
                        buf_or_u64(rmask, m_sel(need_bit, mask_off, BUF_OOB), mask_bit);
                        // neighbours that are lowered, and the popped cell's is_queued (:329): one store instruction
                        const uint32_t nsv = m_sel(curm, cs & ~(uint32_t)SV_QUEUED, nsv_new);
                        buf_store_sv(rsv, m_sel(overm | curm, coff * SV_BYTES, BUF_OOB), nsv);
                        buf_store_u32(robs, m_sel(overm, coff * 4u, BUF_OOB), my_obs);
                        BFF(4);
                        const unsigned long long om = __ballot(overm != 0u);
                        entry = entry_spec;
                        const uint32_t rank = opq(lane_rank(om, lane) & 3u);                         // (opaque: computed for every lane, no exec-masked region)
                        uint64_t* dst = overm ? &sh.pl_e[tw_it & 1u][rank] : dmy;
                        *dst = entry;
                        cnt = (uint32_t)__popcll(om);
                        BFF(5);
                    }
                }
            }
            BFT(2);
#ifdef LAMA_PROFILE_BF_COUNT
            prof[0] += 1; prof[1] += general ? 1 : 0;
#endif
            // (the count is wave-uniform by construction -- a ballot's population --, but the general code reaches it through per-lane
            // loads: said where it happens, so that the queue length, the next-top test and the loop test stay on the scalar unit)
#ifdef LAMA_WAVE_SIM
            if (general) cnt = general_pop(e, over, entry);
#else
            if (__builtin_expect(general, 0)) cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)general_pop(e, over, entry));
#endif
            BFT(3);
            // ---- hand-over: the push list is in the mailbox; meet the helper (its pop is done) and derive the next top from the root
            // it saw after pop() and my own pushes: push_heap lifts an entry above its parent only if the parent's priority is
            // strictly greater, so a pushed entry becomes the root iff its priority is smaller than the root's, the first of the
            // smallest ones.  Every push of this pop has a priority above `floor_sq` (lower() only offers distances larger than the
            // popped cell's): while the root's priority is not above floor_sq + 1 -- the normal case inside a priority level -- no
            // push can win and the root is the next top; else my best push (smallest (priority, neighbour index): a minimum over
            // the quad of neighbour lanes, DPP) is compared with it.
            {
                const uint32_t b_ = tw_it & 1u;
                if (PAIR) {
                    uint32_t* np = lane == 0 ? &sh.pl_n[b_] : (uint32_t*)dmy;
                    *np = cnt;
                }
                BFT_MAIN(5); BFF(6);
                if (PAIR) lds_barrier();                           // D
                else LAMA_LOCKSTEP();                              // (one wave: its LDS accesses are in order; the fibers of the simulator meet here)
                BFT_MAIN(6);
                const uint64_t root_ = sh.topq[b_];
                const uint32_t rlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)root_), rhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(root_ >> 32));
                const bool have_root = nl > 0;
                e_next = ((uint64_t)rhi << 32) | rlo;
                // (integer arithmetic on wave-uniform words: as a `&&` / `||` of bools the compiler builds the condition with v_cndmask)
                const uint32_t own_may_win = (cnt != 0u ? 1u : 0u) & ((have_root ? 0u : 1u) | ((rhi >> 16) > floor_sq + 1u ? 1u : 0u));
                if (__builtin_expect(own_may_win != 0u, 0)) {
                    uint32_t key = over ? ((heap_prio(entry) << 2) | (uint32_t)(lane & 3)) : 0xFFFFFFFFu;
                    uint32_t blo = (uint32_t)entry, bhi = (uint32_t)(entry >> 32);
#define BF_QUAD_MIN(CTRL)                                                                                              \
                    {                                                                                                  \
                        const uint32_t k2 = (uint32_t)__builtin_amdgcn_update_dpp((int)key, (int)key, CTRL, 0xF, 0xF, false); \
                        const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp((int)blo, (int)blo, CTRL, 0xF, 0xF, false); \
                        const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp((int)bhi, (int)bhi, CTRL, 0xF, 0xF, false); \
                        const bool t_ = k2 < key;                                                                      \
                        key = t_ ? k2 : key; blo = t_ ? l2 : blo; bhi = t_ ? h2 : bhi;                                 \
                    }
                    BF_QUAD_MIN(0xB1)                              // quad_perm [1,0,3,2]
                    BF_QUAD_MIN(0x4E)                              // quad_perm [2,3,0,1]
#undef BF_QUAD_MIN
                    const uint32_t okey = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);
                    const uint32_t olo = (uint32_t)__builtin_amdgcn_readfirstlane((int)blo), ohi = (uint32_t)__builtin_amdgcn_readfirstlane((int)bhi);
                    if (!have_root || (okey >> 2) < (rhi >> 16)) e_next = ((uint64_t)ohi << 32) | olo;
                }
                if (PAIR) nl += cnt;
                else lds_push_flat_n(sh.lower, nl, sh.pl_e[b_], cnt, lane, dmy);      // (the next top was derived above, as in the pair)
                ++tw_it;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            BFT(4); BFF(7);
            // the queue is empty, or would outgrow its LDS window (the helper wave takes the same decision): one scalar test
            if (nl - 1u >= LOWER_CAP - 4u) { spill = nl != 0u; tw_running = false; break; }
        }
    }
    // ---- lower wave ------------------------------------------------------------------------- :175-194
#ifdef LAMA_PROFILE_BF_MAIN
    BFT(0);
#else
    BFT(7);
#endif
    while (TW ? (tw_running && nl > 0) : (!spill && nl > 0)) {
        if (!TW && nl + 4 > (uint32_t)LQ_LDS) { spill = true; break; }
        const uint64_t e = TW ? e_next : sh.lower[0];
        const int rx = q_rx(e), ry = q_ry(e);
        ++processed;
        BFT(0);
        // ONE load round: lanes 0..4 their own cell, lane 5 the obstacle cell the entry says the popped cell points to
        const bool is_oc = lane == 5;
        const int x = rx + (is_oc ? q_ox(e) : ddx), y = ry + (is_oc ? q_oy(e) : ddy);
        const bool role = lane < 6;
        const bool inwin = (uint32_t)x < prm.WC && (uint32_t)y < prm.WC;
        const uint32_t pidx = ((uint32_t)y >> 5) * prm.W + ((uint32_t)x >> 5);
        const uint32_t ci = ((uint32_t)x & 31u) | (((uint32_t)y & 31u) << 5);
        int slot = (role && inwin) ? dc.lookup(pidx) : -1;
        sv_t s = 0; uint32_t ob = 0;
        if (slot >= 0) { s = sv[slot * 1024 + (int)ci]; if (!is_oc) ob = obs[slot * 1024 + (int)ci]; }
        BFT(1);
        uint32_t tw_cnt = 0, tw_om = 0;
        uint64_t tw_entry = 0;
        if (TW) {
            --nl;                                              // the helper wave pops
        } else {
            (void)lds_pop_topdown(sh.lower, nl, lane, anc);
        }
        BFT(2);
        const sv_t cs = (sv_t)__builtin_amdgcn_readlane((int)s, 4);
        const uint32_t cob = (uint32_t)__builtin_amdgcn_readlane((int)ob, 4);
        sv_t cos_ = (sv_t)__builtin_amdgcn_readlane((int)s, 5);
        if (obs_x(cob) != q_ox(e) || obs_y(cob) != q_oy(e)) {
            // stale entry (the cell was overwritten after it was queued): fetch the obstacle cell it points to now
            const int ox2 = rx + obs_x(cob), oy2 = ry + obs_y(cob);
            sv_t t = 0;
            if ((uint32_t)ox2 < prm.WC && (uint32_t)oy2 < prm.WC) {
                const int os2 = dc.lookup(((uint32_t)oy2 >> 5) * prm.W + ((uint32_t)ox2 >> 5));
                if (os2 >= 0) t = sv[os2 * 1024 + (int)(((uint32_t)ox2 & 31u) | (((uint32_t)oy2 & 31u) << 5))];
            }
            cos_ = t;
        }
        // :183-192  valid, its obstacle still has sqdist 0 (valid NOT tested), and lower() :283 still queued
        const bool fire = (cs & SV_VALID) && (cos_ & SV_SQMASK) == 0 && (cs & SV_QUEUED);
        BFT(3);
#ifdef LAMA_PROFILE_BF_COUNT
        prof[0] += 1; prof[1] += fire ? 1 : 0; prof[2] += (obs_x(cob) != q_ox(e) || obs_y(cob) != q_oy(e)) ? 1 : 0;
#endif
        if (fire) {
            const int cox = obs_x(cob), coy = obs_y(cob);
            const int obx = rx + cox, oby = ry + coy;
            const bool away = is_nb && !(ddx * cox > 0 || ddy * coy > 0);             // :296
            if (away && !inwin) atomicOr(prm.err, ERR_WINDOW);
            const bool nb = away && inwin;
            const bool fresh = nb && slot < 0;
            if (__ballot(fresh)) { const int ns_ = coop_slot(dc, dir, count, (int)dm_cap, fresh, pidx, ERR_DM_CAP, prm.err); if (fresh) slot = ns_; }
            const bool nbok = nb && slot >= 0;
            if (nbok) {
                const uint64_t bit = 1ull << (ci & 63);
                if (fresh || !(s & (SV_VALID | SV_QUEUED))) atomicOr((unsigned long long*)(mask + (size_t)slot * 16 + (ci >> 6)), (unsigned long long)bit);   // see raise()
            }
            const int qx = x - obx, qy = y - oby;
            const uint32_t new_sq = (uint32_t)(qx * qx + qy * qy);
            const uint32_t cmp = (s & SV_VALID) ? (uint32_t)(s & SV_SQMASK) : prm.max_sqdist;
            bool over = nbok && new_sq < cmp;
            const bool tie = nbok && !over && new_sq == (uint32_t)(s & SV_SQMASK);     // :311-317
#ifdef LAMA_PROFILE_BF_COUNT
            prof[3] += __ballot(tie) ? 1 : 0;
#endif
            if (__ballot(tie)) {
                // the neighbour's own obstacle cell: usually the very cell the popped cell points to (both were
                // reached from the same obstacle), whose state lane 5 already holds -- no second load round then
                const int ox = x + obs_x(ob), oy = y + obs_y(ob);
                const bool same = ox == obx && oy == oby;
                sv_t os = cos_;
#ifdef LAMA_PROFILE_BF_COUNT
                prof[4] += __ballot(tie && !same) ? 1 : 0;
#endif
                if (__ballot(tie && !same)) {
                    if (tie && !same) {
                        os = 0;
                        if ((uint32_t)ox < prm.WC && (uint32_t)oy < prm.WC) {
                            const int oslot = dc.lookup(((uint32_t)oy >> 5) * prm.W + ((uint32_t)ox >> 5));
                            if (oslot >= 0) os = sv[oslot * 1024 + (int)(((uint32_t)ox & 31u) | (((uint32_t)oy & 31u) << 5))];
                        }
                    }
                }
                if (tie && (!(s & SV_VALID) || !((os & SV_VALID) && (os & SV_SQMASK) == 0))) over = true;
            }
            if (over) {
                sv[slot * 1024 + (int)ci] = (sv_t)(SV_VALID | SV_QUEUED | (new_sq & SV_SQMASK));
                obs[slot * 1024 + (int)ci] = pack_obs(obx - x, oby - y);
            }
            if (is_cur && slot >= 0) sv[slot * 1024 + (int)ci] = (sv_t)(cs & ~SV_QUEUED);   // :329
            BFT(4);
            // pushes in neighbour order.  Fast path: one gather of all parents; if none of the new entries has
            // to move up (parent priority <= new priority: the normal case in a Dijkstra wave) they are simply
            // appended, exactly what the sequential push_heap calls would have done.
            const unsigned long long om = __ballot(over);
            const int ocnt = __popcll(om);
#ifdef LAMA_PROFILE_BF_COUNT
            prof[5] += ocnt;
#endif
            bool done = ocnt == 0;
            if (TW) {                                          // the helper wave pushes: hand the entries over
                tw_entry = q_entry(new_sq, x, y, obx - x, oby - y);
                if (over) sh.pl_e[tw_it & 1u][__popcll(om & ((1ull << lane) - 1ull))] = tw_entry;
                tw_cnt = (uint32_t)ocnt;
                tw_om = (uint32_t)om & 15u;
                done = true;
            }
            if (!done && nl >= 4) {
                const uint32_t pos = nl + (uint32_t)__popcll(om & ((1ull << lane) - 1ull));
                bool stop = true;
                if (over) stop = !(heap_prio(sh.lower[(pos - 1) / 2]) > new_sq);
                if (__ballot(!stop) == 0) {
                    if (over) sh.lower[pos] = q_entry(new_sq, x, y, obx - x, oby - y);
                    nl += (uint32_t)ocnt;
                    done = true;
                    LAMA_LOCKSTEP();
                }
            }
            if (!done) {
                #pragma unroll 1
                for (int i = 0; i < 4; ++i) {
                    if (__builtin_amdgcn_readlane((int)over, i)) {
                        const uint32_t prio = (uint32_t)__builtin_amdgcn_readlane((int)new_sq, i);
                        const int ex = __builtin_amdgcn_readlane(x, i), ey = __builtin_amdgcn_readlane(y, i);
                        lds_push(sh.lower, nl, q_entry(prio, ex, ey, obx - ex, oby - ey), lane == 0);
                    }
                }
            }
            BFT(5);
        }
        if (TW) BF_TW_TAIL(false, nl, 0u, tw_cnt, tw_entry, tw_om)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        BFT(6);
    }
    if (TW && tw_go) __syncthreads();                  // F: the helper has applied the last pushes
    #undef BF_TW_TAIL
    #undef BF_LOAD_A
    #undef BF_LOAD_B
    #undef BF_POP_WITH_LOADS
    if (spill) {      // hand the particle over, state intact, to the next stage (the helper wave is past its last LDS access: barrier F, or S0 when it never started)
        const uint64_t* lq = sh.lower;                   // (may run on into sh.raise: the lower phase of the wave pair uses both, see BfLds)
        for (uint32_t k = lane; k < nl; k += UM_BLOCK) g_lower[k] = lq[k];
        for (uint32_t k = lane; k < nr; k += UM_BLOCK) g_raise[k] = sh.raise[k];
    }
    if (lane == 0) {
        prm.counts[2 * p] = count;
        prm.stats[4 * p + 3] += processed;
        if (spill) { prm.qsizes[2 * p] = nl; prm.qsizes[2 * p + 1] = nr; bf_hand_over(prm, p, RESUME); }
#ifdef LAMA_PROFILE_BF
#ifdef LAMA_PROFILE_BF_MAIN
        for (int k = 0; k < 8; ++k) prm.dbg[8 * p + k] = prof[k];
#else
        for (int k = 0; k < (TW ? 5 : 8); ++k) prm.dbg[8 * p + k] = prof[k];
#endif
#endif
    }
}

// Stage structure: the first stage (RESUME = false) is a workgroup per particle with small LDS queues.  A particle whose queue does
// not fit (e.g. the first scan) is handed over -- flag prm.slow[p] and an entry in the hand-over list -- to the resume stage with its
// big LDS queues (84 KB: one workgroup per CU).  That stage is launched with a SMALL grid whose workgroups walk the list: with an
// empty list (the normal case) it costs a few microseconds; launched with one workgroup per particle it cost 1.2 ms at 3000
// particles for doing nothing (12 rounds of one 84 KB workgroup per CU, rocprofv3 r03).
// A map update lasts as long as the LONGEST brushfire chain of the pool, and how long a particle's chain will be is known before it
// starts: it is proportional to the number of obstacle events the ray-cast queued (typical particles of the corridor log: 30-55
// events, ~2,000 pops; the ones whose pose has drifted: 70-300 events, 6,000-13,000 pops -- and those are also the ones whose queue
// outgrows the first stage, after which they used to wait for ALL of it to end).  k_bf_route (one workgroup) marks the particles
// with more than 1.5 times the pool's mean number of events (at most `cap`) and lists them; the host launches the big-queue stage
// for that list on a second stream BESIDE the first stage, which skips them: the long chains start at once, one per CU.
__global__ __launch_bounds__(256) void k_bf_route(DevParams prm, int first_particle, int count, uint32_t min_events, uint32_t percent, uint32_t cap)
{
    __shared__ uint32_t part[256];
    const int tid = threadIdx.x;
    if (map_update_aborted(prm)) { for (int i = tid; i < count; i += 256) prm.heavy[first_particle + i] = 0; return; }
    __shared__ uint32_t cnts[256];
    uint32_t sum = 0, cn = 0;                                  // (the early lane's particles are not in this pool: the first stage skips them too)
    for (int i = tid; i < count; i += 256) { const int p = first_particle + i; if (prm.early && prm.early[p]) continue; sum += prm.qsizes[2 * p] + prm.qsizes[2 * p + 1]; ++cn; }
    part[tid] = sum; cnts[tid] = cn;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if (tid < off) { part[tid] += part[tid + off]; cnts[tid] += cnts[tid + off]; } __syncthreads(); }
    const uint32_t mean = part[0] / (cnts[0] ? cnts[0] : 1u);
    const uint32_t scaled = (uint32_t)(((uint64_t)mean * percent) / 100u);
    uint32_t thr = scaled > min_events ? scaled : min_events;
    // more candidates than places: the places go to the LONGEST chains -- raise the threshold while more than `cap` stay above it
    // (never so far that nobody does: then the first `cap` that come get the places)
    auto count_over = [&](uint32_t t) {
        uint32_t over = 0;
        for (int i = tid; i < count; i += 256) { const int p = first_particle + i; if (prm.early && prm.early[p]) continue; over += (prm.qsizes[2 * p] + prm.qsizes[2 * p + 1]) > t ? 1u : 0u; }
        __syncthreads();
        cnts[tid] = over;
        __syncthreads();
        for (int off = 128; off > 0; off >>= 1) { if (tid < off) cnts[tid] += cnts[tid + off]; __syncthreads(); }
        return cnts[0];                                           // (the same value in every thread)
    };
    if (count_over(thr) > cap)
        for (int round = 0; round < 48; ++round) {
            const uint32_t next = thr + thr / 8u + 1u;
            const uint32_t over = count_over(next);
            if (over == 0u) break;
            thr = next;
            if (over <= cap) break;
        }
    for (int i = tid; i < count; i += 256) {
        const int p = first_particle + i;
        const uint32_t q = prm.qsizes[2 * p] + prm.qsizes[2 * p + 1];
        // an early-lane particle is on its way already (its queue sizes are from before the lane emptied them -- or zero, when the lane
        // was faster): it keeps its place in the early lane of the NEXT update only while it is still over the threshold
        if (prm.early && prm.early[p]) { prm.heavy[p] = q > thr ? 1 : 0; continue; }
        uint8_t h = 0;
        if (q > thr) {
            const uint32_t k = atomicAdd(prm.slow_n + 2, 1u);
            if (k < cap) { h = 1; prm.slow[p] = 1; prm.slow_list[2 * (size_t)prm.P + k] = (uint32_t)p; }
        }
        prm.heavy[p] = h;
    }
    __syncthreads();
    if (tid == 0 && prm.slow_n[2] > cap) prm.slow_n[2] = cap;         // (more candidates than places: the others stay in the first stage)
}

// early lane: list + skip flags of this update from (i) the host's list -- the particles whose scan match of THIS step fitted worst:
// the longest chain of an update is the particle with the lowest log-likelihood of the pool (its scan disagrees with its map, so it
// re-draws it), rank 0 or 1 of 3000 in every update with a long chain -- and (ii) the particles the previous update routed
// (`heavy_prev`, nullptr: not valid any more)
__global__ __launch_bounds__(256) void k_early_list(const uint8_t* __restrict__ heavy_prev, const uint32_t* __restrict__ host_list, uint32_t host_n,
                                                     uint8_t* __restrict__ early, uint32_t* __restrict__ elist, uint32_t* __restrict__ elist_n, int P, uint32_t cap)
{
    for (int p = threadIdx.x; p < P; p += 256) early[p] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (uint32_t i = 0; i < host_n && n < cap; ++i) { const uint32_t p = __hip_atomic_load(host_list + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (p < (uint32_t)P && !early[p]) { early[p] = 1; elist[n++] = p; } }
        *elist_n = n;
    }
    __syncthreads();
    if (heavy_prev) {
        for (int p = threadIdx.x; p < P; p += 256)
            if (heavy_prev[p] && !early[p]) { const uint32_t k = atomicAdd(elist_n, 1u); if (k < cap) { early[p] = 1; elist[k] = (uint32_t)p; } }
        __syncthreads();
        if (threadIdx.x == 0 && *elist_n > cap) *elist_n = cap;
    }
}
// the early lane's particles go to the big-queue brushfire stage directly: flag them as handed over
__global__ __launch_bounds__(64) void k_mark_early(DevParams prm)
{
    if (map_update_aborted(prm)) return;                      // nothing was queued for them: the host grows the arenas and repeats the update
    const uint32_t n = uload_u32(prm.elist_n);
    for (uint32_t i = threadIdx.x; i < n; i += 64) prm.slow[prm.elist[i]] = 1;
}

// `routed` (resume stages only): 1 = walk the list of the routed particles (k_bf_route) instead of the first stage's hand-over list,
// 2 = the early lane's list
template <int LQ_LDS, int RQ_LDS, bool RESUME, bool TW>
__global__ __launch_bounds__(TW ? 2 * UM_BLOCK : UM_BLOCK) void k_brushfire(DevParams prm, int first_particle, int routed = 0)
{
    __shared__ BfLds<LQ_LDS, RQ_LDS> sh;
    if (!RESUME) {
        if (map_update_aborted(prm)) return;             // the update's allocation phase failed: nothing was queued, nothing is touched
#ifdef LAMA_PROFILE_BF                                   // where and when this particle ran (constant 100 MHz counter)
        uint64_t rt0; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt0));
#endif
        bf_particle<LQ_LDS, RQ_LDS, RESUME, TW>(prm, first_particle + (int)blockIdx.x, sh);
#ifdef LAMA_PROFILE_BF
        if (threadIdx.x == 0) {
            uint64_t rt1; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(rt1));
            uint32_t hw, xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            uint64_t* o = prm.dbg + 8 * (size_t)prm.P + 8 * (size_t)(first_particle + (int)blockIdx.x);
            o[0] = rt0; o[1] = rt1; o[2] = hw; o[3] = xcc;
        }
#endif
        return;
    }
    if (routed == 2 && map_update_aborted(prm)) return;      // (the other lists are empty after an aborted allocation phase; this one is not)
    // (lists and counts of kernels that may have run on another stream of the context: coherent uniform loads, lama_dev.h)
    const uint32_t n = uload_u32(routed == 2 ? prm.elist_n : (routed ? prm.slow_n + 2 : prm.slow_n));
    const uint32_t* list = routed == 2 ? prm.elist : prm.slow_list + (routed ? 2 * (size_t)prm.P : 0);
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        bf_particle<LQ_LDS, RQ_LDS, RESUME, TW>(prm, (int)uload_u32(list + i), sh);
        __syncthreads();                                 // every wave is done with this particle's LDS before the next one is loaded
    }
}

// ------------------------------------------------------------------------------------------------
// k_map_checksum -- order-independent 64-bit checksum of one particle's map (lama_hip_pf_map_checksums): the sum, modulo
// 2^64, over the allocated patches of mix(patch id) and over their cells of mix(patch id * 1024 + cell, fields), where the
// fields are what the reference stores for the cell (distance_t: obstacle offset, sqdist, valid, queued; frequency: occupied,
// visited) plus the Container mask bit; all-zero cells add nothing.  Lets a caller compare ALL particles' maps with maps
// held elsewhere without downloading them (the same sum is easily computed from the reference's records).
// ------------------------------------------------------------------------------------------------
LAMA_HD uint64_t cks_mix(uint64_t x)                      // splitmix64 finaliser
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
LAMA_HD uint64_t cks_cell(uint64_t patch_id, uint32_t cell, uint64_t fields) { return fields ? cks_mix((patch_id * 1024ull + cell) ^ cks_mix(fields)) : 0ull; }

__global__ __launch_bounds__(256) void k_map_checksum(DevParams prm, int kind /*0 distance, 1 occupancy*/, uint64_t* __restrict__ out)
{
    __shared__ uint64_t part[4];
    const int p = blockIdx.x;
    const size_t WW = (size_t)prm.W * prm.W;
    const PV pv = pview(prm, p);
    const int16_t* dir = kind == 0 ? pv.dm_dir : pv.occ_dir;
    uint64_t acc = 0;
    for (uint32_t w = 0; w < (uint32_t)WW; ++w) {
        const int slot = dir[w];
        if (slot < 0) continue;
        const uint32_t wy = w / prm.W, wx = w % prm.W;
        const uint64_t id = ((uint64_t)(prm.wx0 >> 5) + wx) * 2642244ull + ((uint64_t)(prm.wy0 >> 5) + wy);
        if (threadIdx.x == 0) acc += cks_mix(id ^ 0x5DEECE66Dull);
        for (uint32_t c = threadIdx.x; c < 1024u; c += 256u) {
            uint64_t f;
            if (kind == 0) {
                const uint64_t m = (pv.dm_mask[(size_t)slot * 16 + (c >> 6)] >> (c & 63)) & 1ull;
#ifdef LAMA_WIDE_DM          // the squared distance takes all 16 low bits: the two flags move above the mask bit
                const uint32_t w = pv.dm_sv[(size_t)slot * 1024 + c];
                f = (uint64_t)(w & 0xFFFFu) | ((uint64_t)pv.dm_obs[(size_t)slot * 1024 + c] << 16) | (m << 48) |
                    ((w & SV_VALID) ? 1ull << 49 : 0ull) | ((w & SV_QUEUED) ? 1ull << 50 : 0ull);
#else
                f = (uint64_t)pv.dm_sv[(size_t)slot * 1024 + c] | ((uint64_t)pv.dm_obs[(size_t)slot * 1024 + c] << 16) | (m << 48);
#endif
            } else {
                // Container mask of a frequency cell = "visited != 0", plus the plane bits kept for uint16 wraps (as in the download)
                const uint32_t ov = pv.occ[(size_t)slot * 1024 + c];
                const uint64_t m = ((pv.occ_mask[(size_t)slot * 16 + (c >> 6)] >> (c & 63)) & 1ull) | ((ov >> 16) ? 1ull : 0ull);
                f = (uint64_t)ov | (m << 48);
            }
            acc += cks_cell(id, c, f);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)acc, off, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(acc >> 32), off, 64);
        acc += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[p] = part[0] + part[1] + part[2] + part[3];
}

// ------------------------------------------------------------------------------------------------
// k_brushfire_slow -- generic single-lane DynamicDistanceMap::update with both queues in HBM (libstdc++ heap of
// lama_heap.h).  Runs only for particles k_brushfire flagged (queue larger than its LDS window, e.g. the very
// first scan of a large open space) and resumes exactly where it stopped.
// ------------------------------------------------------------------------------------------------
struct GlobalStore {
    uint64_t* g;
    __device__ inline uint64_t get(uint32_t i) const { return g[i]; }
    __device__ inline void set(uint32_t i, uint64_t v) const { g[i] = v; }
};

struct BfCtx {
    const DevParams& prm;
    int16_t* dir; sv_t* sv; uint32_t* obs; uint64_t* mask;
    int count; int cap;
    GlobalStore lower, raise;
    uint32_t nl, nr;
    uint64_t processed;
};

// serial non-const Map::get on the DM (src/sdm/map.cpp:371-412): allocate + set mask bit; returns slot*1024+cell or -1
__device__ inline int bf_get(BfCtx& c, int rx, int ry)
{
    if ((uint32_t)rx >= c.prm.WC || (uint32_t)ry >= c.prm.WC) { atomicOr(c.prm.err, ERR_WINDOW); return -1; }
    const uint32_t pidx = ((uint32_t)ry >> 5) * c.prm.W + ((uint32_t)rx >> 5);
    int slot = c.dir[pidx];
    if (slot < 0) {
        if (c.count >= c.cap) { atomicOr(c.prm.err, ERR_DM_CAP); return -1; }
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

__device__ __noinline__ void bf_raise(BfCtx& c, int rx, int ry, int cur)            // :244-279
{
    const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
    for (int i = 0; i < 4; ++i) {
        const int nx = rx + DX[i], ny = ry + DY[i];
        const int nc = bf_get(c, nx, ny);
        if (nc < 0) continue;
        const sv_t s = c.sv[nc];
        if ((s & SV_QUEUED) || !(s & SV_VALID)) continue;
        const uint32_t no = c.obs[nc];
        const int oc = bf_get(c, nx + obs_x(no), ny + obs_y(no));
        if (oc < 0) continue;
        if (!(c.sv[oc] & SV_VALID)) {
            bf_push(c, false, s & SV_SQMASK, nx, ny);
            c.sv[nc] = SV_QUEUED;
            c.obs[nc] = 0;
        } else {
            bf_push(c, true, s & SV_SQMASK, nx, ny);
            c.sv[nc] = s | SV_QUEUED;
        }
    }
    c.sv[cur] &= (sv_t)~SV_QUEUED;
}

__device__ __noinline__ void bf_lower(BfCtx& c, int rx, int ry, int cur)            // :281-330
{
    const sv_t s = c.sv[cur];
    if (!(s & SV_QUEUED)) return;
    const uint32_t co = c.obs[cur];
    const int cox = obs_x(co), coy = obs_y(co);
    const int obx = rx + cox, oby = ry + coy;
    const int DX[4] = {1, 0, -1, 0}, DY[4] = {0, 1, 0, -1};
    for (int i = 0; i < 4; ++i) {
        if (DX[i] * cox > 0 || DY[i] * coy > 0) continue;
        const int nx = rx + DX[i], ny = ry + DY[i];
        const int nc = bf_get(c, nx, ny);
        if (nc < 0) continue;
        const sv_t ns = c.sv[nc];
        const int qx = nx - obx, qy = ny - oby;
        const uint32_t new_sq = (uint32_t)(qx * qx + qy * qy);
        const uint32_t cmp = (ns & SV_VALID) ? (uint32_t)(ns & SV_SQMASK) : c.prm.max_sqdist;
        bool over = new_sq < cmp;
        if (!over && new_sq == (uint32_t)(ns & SV_SQMASK)) {
            const uint32_t nobs = c.obs[nc];
            const int oc = bf_get(c, nx + obs_x(nobs), ny + obs_y(nobs));
            if (oc >= 0) {
                const sv_t os = c.sv[oc];
                if (!(ns & SV_VALID) || !((os & SV_VALID) && (os & SV_SQMASK) == 0)) over = true;
            }
        }
        if (over) {
            bf_push(c, true, new_sq, nx, ny);
            c.sv[nc] = (sv_t)(SV_VALID | SV_QUEUED | (new_sq & SV_SQMASK));
            c.obs[nc] = pack_obs(obx - nx, oby - ny);
        }
    }
    c.sv[cur] &= (sv_t)~SV_QUEUED;
}

__global__ __launch_bounds__(UM_BLOCK) void k_brushfire_slow(DevParams prm, int first_particle)
{
    const int p = first_particle + blockIdx.x;
    if (prm.slow[p] == 0 || threadIdx.x != 0) return;
    const PV pv = pview(prm, p);
    BfCtx c{prm, pv.dm_dir, pv.dm_sv, pv.dm_obs, pv.dm_mask, prm.counts[2 * p], (int)pv.dm_cap,
            GlobalStore{prm.q_lower + (size_t)p * prm.qcap}, GlobalStore{prm.q_raise + (size_t)p * prm.qcap},
            prm.qsizes[2 * p], prm.qsizes[2 * p + 1], 0};
    while (c.nr > 0) {                                                              // :162-173
        const uint64_t e = heap_pop(c.raise, c.nr);
        const int rx = q_rx(e), ry = q_ry(e);
        const int cur = bf_get(c, rx, ry);
        ++c.processed;
        if (cur < 0) continue;
        bf_raise(c, rx, ry, cur);
    }
    while (c.nl > 0) {                                                              // :175-194
        const uint64_t e = heap_pop(c.lower, c.nl);
        const int rx = q_rx(e), ry = q_ry(e);
        const int cur = bf_get(c, rx, ry);
        ++c.processed;
        if (cur < 0) continue;
        const sv_t s = c.sv[cur];
        if (s & SV_VALID) {
            const uint32_t o = c.obs[cur];
            const int oc = bf_get(c, rx + obs_x(o), ry + obs_y(o));
            if (oc < 0) continue;
            if ((c.sv[oc] & SV_SQMASK) == 0) bf_lower(c, rx, ry, cur);               // :191 (valid NOT tested)
        }
    }
    prm.counts[2 * p] = c.count;
    prm.stats[4 * p + 3] += c.processed;
    prm.slow[p] = 0;
}

} // namespace lama_dev
#include "lama_raycast_par.h"
#include "lama_raycast_patch.h"
#include "lama_brushfire_canon.h"
namespace lama_dev {

// ------------------------------------------------------------------------------------------------
// k_dm_add_obstacles -- DynamicDistanceMap::addObstacle (src/sdm/dynamic_distance_map.cpp:212-226) for a list of map
// cells, in the given order, on one particle's distance map (Loc2D builds its static map this way); the brushfire
// stages then run dm->update().  One wave; 64 cells per step; pushes appended in list order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(UM_BLOCK) void k_dm_add_obstacles(DevParams prm, int p, const uint32_t* __restrict__ cells_xy, uint32_t n)
{
    __shared__ uint32_t lds_dc[DC_SIZE];
    const int lane = threadIdx.x;
    const PV pv = pview(prm, p);
    int16_t* dm_dir = pv.dm_dir;
    sv_t* dm_sv = pv.dm_sv;
    uint32_t* dm_obs = pv.dm_obs;
    uint64_t* dm_mask = pv.dm_mask;
    uint64_t* q_lower = prm.q_lower + (size_t)p * prm.qcap;
    int dm_count = prm.counts[2 * p];
    for (int k = lane; k < DC_SIZE; k += UM_BLOCK) lds_dc[k] = DC_EMPTY;
    __syncthreads();
    const DirCache dc{lds_dc, dm_dir, prm.W};
    uint32_t nl = 0;
    for (uint32_t base = 0; base < n; base += 64) {
        const uint32_t i = base + lane;
        const bool act = i < n;
        const uint32_t rx = (act ? cells_xy[2 * i] : 0u) - prm.wx0, ry = (act ? cells_xy[2 * i + 1] : 0u) - prm.wy0;
        const bool inwin = rx < prm.WC && ry < prm.WC;
        if (act && !inwin) atomicOr(prm.err, ERR_WINDOW);
        const bool want = act && inwin;
        const uint32_t pidx = (ry >> 5) * prm.W + (rx >> 5), ci = (rx & 31u) | ((ry & 31u) << 5);
        const int slot = coop_slot(dc, dm_dir, dm_count, (int)pv.dm_cap, want, pidx, ERR_DM_CAP, prm.err);
        // duplicates inside one 64-cell step: the first occurrence wins (a later addObstacle of the same cell returns early)
        bool first = true;
        for (int l = 0; l < 63; ++l) {       // executed by all lanes (uniform control flow around the shuffles)
            const uint32_t ox = __shfl(rx, l, 64), oy = __shfl(ry, l, 64);
            const int ow = __shfl((int)want, l, 64);
            if (l < lane && ow && ox == rx && oy == ry) first = false;
        }
        bool push = false;
        if (want && slot >= 0) {
            atomicOr((unsigned long long*)(dm_mask + (size_t)slot * 16 + (ci >> 6)), 1ull << (ci & 63));
            const sv_t s = dm_sv[slot * 1024 + (int)ci];
            if (first && !((s & SV_VALID) && (s & SV_SQMASK) == 0)) {
                dm_sv[slot * 1024 + (int)ci] = (sv_t)(SV_VALID | SV_QUEUED);
                dm_obs[slot * 1024 + (int)ci] = 0;
                push = true;
            }
        }
        const unsigned long long pm = __ballot(push);
        const int cnt = __popcll(pm);
        if (nl + (uint32_t)cnt > prm.qcap) { if (lane == 0) atomicOr(prm.err, ERR_QUEUE); }
        else {
            if (push) q_lower[nl + (uint32_t)__popcll(pm & ((1ull << lane) - 1ull))] = q_entry(0, (int)rx, (int)ry);
            nl += (uint32_t)cnt;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
    if (lane == 0) { prm.counts[2 * p] = dm_count; prm.qsizes[2 * p] = nl; prm.qsizes[2 * p + 1] = 0; }
}

// ------------------------------------------------------------------------------------------------
// k_shift_window -- the window directory of every particle seen through a window whose origin moved by (dx, dy) patches:
// dst[p][wy][wx] = src[p][wy + dy][wx + dx], -1 where that lies outside the old window.  Cells are addressed window-relative
// and patches live in per-particle arenas, so moving the window only permutes directory entries.  A patch that would leave
// the window means the map no longer fits it: ERR_WINDOW (nothing is dropped silently).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_shift_window(const int16_t* __restrict__ src, int16_t* __restrict__ dst, uint32_t Ws, uint32_t Wd, int dx, int dy,
                                                       size_t src_stride, size_t dst_stride, int32_t* err)
{
    // (round 4) source and destination windows may differ in size: the window GROWS when the mapped area outgrows it
    const uint32_t p = blockIdx.x, idx = blockIdx.y * 256u + threadIdx.x;
    if (idx < Wd * Wd) {
        const int wy = (int)(idx / Wd), wx = (int)(idx % Wd);
        const int sx = wx + dx, sy = wy + dy;
        const bool in = sx >= 0 && sy >= 0 && sx < (int)Ws && sy < (int)Ws;
        dst[p * dst_stride + idx] = in ? src[p * src_stride + (size_t)sy * Ws + (size_t)sx] : (int16_t)-1;
    }
    if (idx < Ws * Ws) {
        const int wy = (int)(idx / Ws), wx = (int)(idx % Ws);
        const int tx = wx - dx, ty = wy - dy;                  // where the old entry (wx, wy) ends up
        if (!(tx >= 0 && ty >= 0 && tx < (int)Wd && ty < (int)Wd) && src[p * src_stride + idx] >= 0) atomicOr(err, ERR_WINDOW);
    }
}

// ------------------------------------------------------------------------------------------------
// k_occ_max_visited -- the largest `visited` counter over all frequency cells of all particles (atomicMax into *out).  The
// parallel ray-cast adds the visits of a scan to the uint16 counters in any order, which equals the reference's sequential
// `visited++` (src/sdm/frequency_occupancy_map.cpp:65-91) only while no counter WRAPS inside the scan; the host keeps an upper
// bound of the largest counter and, when bound + beams could reach 65536, refreshes it with this kernel and falls back to the
// beam-sequential ray-cast for scans in which a wrap is possible.
// ------------------------------------------------------------------------------------------------
// After a cleanly aborted update (ERR_CLEAN_ABORT): directory entries the failed allocations left marked (-2 / -3) are absent again.
__global__ __launch_bounds__(256) void k_update_cleanup(DevParams prm)
{
    const size_t WW = (size_t)prm.W * prm.W;
    const size_t k = (size_t)blockIdx.y * 256 + threadIdx.x;
    if (k >= WW) return;
    int16_t* a = prm.occ_dir + (size_t)blockIdx.x * WW + k;
    int16_t* b = prm.dm_dir + (size_t)blockIdx.x * WW + k;
    if (*a < -1) *a = -1;
    if (*b < -1) *b = -1;
}

__global__ __launch_bounds__(256) void k_occ_max_visited(DevParams prm, uint32_t* __restrict__ out)
{
    const int p = blockIdx.x;
    const uint32_t n = (uint32_t)prm.counts[2 * p + 1] * 1024u;
    const uint32_t* cells = pview(prm, p).occ;
    uint32_t m = 0;
    for (uint32_t k = threadIdx.x; k < n; k += 256u) { const uint32_t v = cells[k] >> 16; m = v > m ? v : m; }
    for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// ------------------------------------------------------------------------------------------------
// k_clone_particles -- the particle copies of resample() / of the first scan (src/pf_slam2d.cpp:204-216, 558-574), done IN PLACE:
// the host has turned the sampled indices into a permutation of the PartRec table -- a particle that survives keeps its home and
// its regions, so nothing of it moves -- and lists here only the second and further copies of a multiply drawn particle, each
// with the home / regions of a particle that died as its destination.  The same job form moves ONE particle's used slots to a
// larger region when its maps outgrow their capacity (src_home == dst_home: the directories stay where they are).
// Slots of the destination beyond the source's count that the old owner had used are re-zeroed ("unused slot == calloc'd").
// grid = (jobs, 7 planes, CLONE_SPLIT), 256 threads, 16-byte vectors.  All counts are the host's mirror: no device-side reads.
// ------------------------------------------------------------------------------------------------
struct CloneJob {
    uint32_t src_home, dst_home;
    int32_t sdm, socc;             // used slots of the source
    int32_t odm, oocc;             // slots the destination region's previous owner had used (0: a region fresh from the allocator)
    uint32_t dir_off16, dir_n16;   // the rows of the window directories that can hold a patch (the context's mapped box), 16-byte units
    uint32_t pad0, pad1;
    const void* s[5];              // source region: dm_sv, dm_obs, dm_mask, occ, occ_mask
    void* d[5];                    // destination region, same order
};
constexpr int CLONE_SPLIT = 2;

// (the region addresses come out of the job list: tell the compiler that they are GLOBAL addresses -- a pointer loaded from memory
// is a generic one to it, and generic loads / stores go the slower flat path)
#ifdef LAMA_WAVE_SIM
typedef uint4 vec16;
typedef uint4 g_uint4;
__device__ inline vec16 vec16_zero() { return make_uint4(0, 0, 0, 0); }
#else
typedef unsigned int vec16 __attribute__((ext_vector_type(4)));      // (a plain vector type: HIP's uint4 class has no address-space-qualified members)
typedef __attribute__((address_space(1))) vec16 g_uint4;
__device__ inline vec16 vec16_zero() { return vec16{0u, 0u, 0u, 0u}; }
#endif
__global__ __launch_bounds__(256) void k_clone_particles(DevParams prm, const CloneJob* __restrict__ jobs)
{
    // (the job list is rewritten by the host before every launch: coherent uniform loads, field by field -- no by-value copy, no scratch)
    const CloneJob* J = jobs + blockIdx.x;
    const int plane = blockIdx.y;
    const size_t WW = (size_t)prm.W * prm.W;
    const g_uint4* s; g_uint4* d; size_t ncopy, nzero = 0;         // in 16-byte units
    if (plane < 2) {
        const uint32_t sh = uload_u32(&J->src_home), dh = uload_u32(&J->dst_home);
        if (sh == dh) return;
        int16_t* dir = plane == 0 ? prm.dm_dir : prm.occ_dir;
        // only the rows inside the mapped box: outside it every directory of the context is -1 (at W = 128 the two directories are a
        // tenth of a corridor particle's bytes, at W = 432 more than its maps)
        const uint32_t off16 = uload_u32(&J->dir_off16);
        s = (const g_uint4*)(uintptr_t)(dir + sh * WW) + off16; d = (g_uint4*)(uintptr_t)(dir + dh * WW) + off16; ncopy = uload_u32(&J->dir_n16);
    } else {
        const int k = plane - 2;
        const bool dm = k < 3;
        const size_t bytes = k == 0 ? SV_PATCH_BYTES : (k == 1 || k == 3) ? 4096 : 128;
        const int32_t used = uload_i32(dm ? &J->sdm : &J->socc), old = uload_i32(dm ? &J->odm : &J->oocc);
        s = (const g_uint4*)(uintptr_t)uload_u64(&J->s[k]); d = (g_uint4*)(uintptr_t)uload_u64(&J->d[k]);
        ncopy = (size_t)used * bytes / 16; nzero = old > used ? (size_t)(old - used) * bytes / 16 : 0;
    }
    if (s == d) return;                                             // (this plane of the particle stays where it is)
    const size_t t0 = (size_t)blockIdx.z * 256 + threadIdx.x, step = 256 * (size_t)gridDim.z;
    size_t k = t0;
    for (; k + step < ncopy; k += 2 * step) { const vec16 a = s[k], b2 = s[k + step]; d[k] = a; d[k + step] = b2; }     // two loads in flight per thread
    for (; k < ncopy; k += step) { const vec16 a = s[k]; d[k] = a; }
    const vec16 z = vec16_zero();
    for (size_t q = t0; q < nzero; q += step) d[ncopy + q] = z;
}

// A region that goes back to the allocator is zeroed where it was used: free pool space is all-zero, so a region handed out later
// needs no preparation.  grid = (regions, 5 planes, CLONE_SPLIT); the occupancy hit bits are zero between scans anyway.
struct ZeroJob { void* d[5]; int32_t ndm, nocc; };
__global__ __launch_bounds__(256) void k_zero_regions(const ZeroJob* __restrict__ jobs)
{
    const ZeroJob* J = jobs + blockIdx.x;                          // (host-rewritten list: coherent uniform loads)
    const int k = blockIdx.y;
    const size_t bytes = k == 0 ? SV_PATCH_BYTES : (k == 1 || k == 3) ? 4096 : 128;
    g_uint4* d = (g_uint4*)(uintptr_t)uload_u64(&J->d[k]);
    const size_t n = (size_t)uload_i32(k < 3 ? &J->ndm : &J->nocc) * bytes / 16;
    if (!d) return;
    const vec16 z = vec16_zero();
    for (size_t i = (size_t)blockIdx.z * 256 + threadIdx.x; i < n; i += 256 * (size_t)gridDim.z) d[i] = z;
}


// ------------------------------------------------------------------------------------------------
// Particle shipping (multi-GPU resampling): ALL outgoing / incoming particles of a resample in one launch each.
// Blob layout (16-byte aligned): [pose 4 f64][header 8 i32: dm patches, occ patches, window origin x / y in patches, visited bound, 0, 0, 0]
// [dm_dir W*W i16][occ_dir W*W i16][dm_sv n_dm x 2048 B (wide build: 4096)][dm_obs n_dm x 4096 B][dm_mask n_dm x 128 B][occ n_occ x 4096 B][occ_mask n_occ x 128 B]
// ------------------------------------------------------------------------------------------------
constexpr int BLOB_HEAD = 64;
struct ShipDesc {
    uint8_t* blob;         // device buffer of the particle's blob
    uint32_t particle;     // slot in this context
    int32_t wdx, wdy;      // import: this context's window origin minus the blob's, in patches
    uint32_t pad;
};

__device__ inline const uint4* blob_plane(const uint8_t* blob, int plane, size_t WW, int dmc, int occ, size_t& n16)
{
    size_t off = BLOB_HEAD;
    const size_t sz[7] = {WW * 2, WW * 2, (size_t)dmc * SV_PATCH_BYTES, (size_t)dmc * 4096, (size_t)dmc * 128, (size_t)occ * 4096, (size_t)occ * 128};
    for (int k = 0; k < plane; ++k) off += sz[k];
    n16 = sz[plane] / 16;
    return reinterpret_cast<const uint4*>(blob + off);
}

// grid (n, 7 planes, SHIP_SPLIT): every plane of every outgoing particle in parallel
constexpr int SHIP_SPLIT = 4;
__global__ __launch_bounds__(256) void k_export_particles(DevParams prm, const ShipDesc* __restrict__ desc, const double* __restrict__ poses,
                                                           int32_t wx_patch, int32_t wy_patch, int32_t visit_bound,
                                                           int32_t bbox_x, int32_t bbox_y)
{
    const ShipDesc d = uload_rec(desc + blockIdx.x);               // (host-rewritten descriptors: coherent uniform loads, lama_dev.h)
    const int j = (int)d.particle, plane = blockIdx.y;
    const uint32_t W = prm.W;
    const PV src = pview(prm, j);
    const int dmc = uload_i32(src.counts), occ = uload_i32(src.counts + 1);
    const size_t WW = (size_t)W * W;
    size_t n16;
    uint4* out = const_cast<uint4*>(blob_plane(d.blob, plane, WW, dmc, occ, n16));
    const uint4* in;
    switch (plane) {
    case 0: in = (const uint4*)src.dm_dir; break;
    case 1: in = (const uint4*)src.occ_dir; break;
    case 2: in = (const uint4*)src.dm_sv; break;
    case 3: in = (const uint4*)src.dm_obs; break;
    case 4: in = (const uint4*)src.dm_mask; break;
    case 5: in = (const uint4*)src.occ; break;
    default: in = (const uint4*)src.occ_mask; break;
    }
    for (size_t k = (size_t)blockIdx.z * 256 + threadIdx.x; k < n16; k += 256 * SHIP_SPLIT) out[k] = in[k];
    if (plane == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        double* hp = reinterpret_cast<double*>(d.blob);
        for (int k = 0; k < 4; ++k) hp[k] = uload_f64(poses + 4 * j + k);
        int32_t* hh = reinterpret_cast<int32_t*>(d.blob + 32);
        // [5] the sender's window side, [6] / [7] the extent of its mapped area inside that window (lo | hi << 16, patches)
        hh[0] = dmc; hh[1] = occ; hh[2] = wx_patch; hh[3] = wy_patch; hh[4] = visit_bound; hh[5] = (int32_t)W; hh[6] = bbox_x; hh[7] = bbox_y;
    }
}

// the 64-byte heads of n blobs gathered into one buffer (one device-to-host copy then tells the importer what is coming)
__global__ __launch_bounds__(64) void k_gather_blob_heads(const ShipDesc* __restrict__ desc, uint32_t n, uint8_t* __restrict__ heads)
{
    const uint32_t j = blockIdx.x;
    if (j >= n || threadIdx.x >= 16) return;
    const uint8_t* blob = uload_ptr<const uint8_t>(&desc[j].blob);
    reinterpret_cast<uint32_t*>(heads + (size_t)j * BLOB_HEAD)[threadIdx.x] =
        __hip_atomic_load(reinterpret_cast<const uint32_t*>(blob) + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (the blob came in by a copy the host queued)
}

// The bytes of an incoming blob were written by a copy that somebody else queued -- another stream of this process (torch.distributed's
// gloo backend stages a received message with its own copy stream), a peer device -- into a buffer whose address the caller's
// allocator reuses from one resample to the next: lines of the PREVIOUS blob at that address may still sit in a cache on the way.
// Read with agent-scope loads, as everything else that this library's own stream did not produce (lama_dev.h, uload_*).
#ifdef LAMA_WAVE_SIM
__device__ __forceinline__ uint4 blob_load16(const uint4* q) { return *q; }
__device__ __forceinline__ int16_t blob_load_i16(const int16_t* q) { return *q; }
#else
__device__ __forceinline__ uint4 blob_load16(const uint4* q)
{
    const uint64_t* w = reinterpret_cast<const uint64_t*>(q);
    const uint64_t a = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), b = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
}
__device__ __forceinline__ int16_t blob_load_i16(const int16_t* q)
{
    const uintptr_t u = reinterpret_cast<uintptr_t>(q);
    const uint32_t w = __hip_atomic_load(reinterpret_cast<const uint32_t*>(u & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (int16_t)((u & 2) ? (w >> 16) : (w & 0xFFFFu));
}
#endif

// grid (n, 7 planes, SHIP_SPLIT).  The directories are translated when the sender's window sits elsewhere (see k_shift_window); slots the
// destination used before and the incoming particle does not are zeroed ("unused slot == zero"); old_counts = the destination's counts
// before the import.
__global__ __launch_bounds__(256) void k_import_particles(DevParams prm, const ShipDesc* __restrict__ desc, const int32_t* __restrict__ old_counts, double* __restrict__ poses,
                                                           int32_t* err)
{
    const ShipDesc d = uload_rec(desc + blockIdx.x);              // (host-rewritten descriptors, a blob and counts that copies queued by the
    const int i = (int)d.particle, plane = blockIdx.y;           //  host brought in: coherent uniform loads, lama_dev.h)
    const uint32_t W = prm.W;
    const PV dst = pview(prm, i);                                // (the host has made the regions large enough for what is coming)
    const int32_t* hh = reinterpret_cast<const int32_t*>(d.blob + 32);
    const int dmc = uload_i32(hh), occ = uload_i32(hh + 1);
    const int odm = uload_i32(old_counts + 2 * i), oocc = uload_i32(old_counts + 2 * i + 1);
    const uint32_t Ws = uload_u32(hh + 5);                       // the sender's window side (windows grow independently)
    const size_t WW = (size_t)W * W, WWs = (size_t)Ws * Ws;
    size_t n16;
    const uint4* in = blob_plane(d.blob, plane, WWs, dmc, occ, n16);
    uint4* out; size_t nzero = 0;
    switch (plane) {
    case 0: out = (uint4*)dst.dm_dir; break;
    case 1: out = (uint4*)dst.occ_dir; break;
    case 2: out = (uint4*)dst.dm_sv; nzero = odm > dmc ? (size_t)(odm - dmc) * SV_PATCH_BYTES / 16 : 0; break;
    case 3: out = (uint4*)dst.dm_obs; nzero = odm > dmc ? (size_t)(odm - dmc) * 4096 / 16 : 0; break;
    case 4: out = (uint4*)dst.dm_mask; nzero = odm > dmc ? (size_t)(odm - dmc) * 128 / 16 : 0; break;
    case 5: out = (uint4*)dst.occ; nzero = oocc > occ ? (size_t)(oocc - occ) * 4096 / 16 : 0; break;
    default: out = (uint4*)dst.occ_mask; nzero = oocc > occ ? (size_t)(oocc - occ) * 128 / 16 : 0; break;
    }
    if (plane < 2 && (d.wdx != 0 || d.wdy != 0 || Ws != W)) {
        const int16_t* sdir = reinterpret_cast<const int16_t*>(in);
        int16_t* ddir = reinterpret_cast<int16_t*>(out);
        for (size_t idx = (size_t)blockIdx.z * 256 + threadIdx.x; idx < WW; idx += 256 * SHIP_SPLIT) {
            const int wy = (int)(idx / W), wx = (int)(idx % W);
            const int sx = wx + d.wdx, sy = wy + d.wdy;
            const bool inw = sx >= 0 && sy >= 0 && sx < (int)Ws && sy < (int)Ws;
            ddir[idx] = inw ? blob_load_i16(sdir + (size_t)sy * Ws + (size_t)sx) : (int16_t)-1;
        }
        for (size_t idx = (size_t)blockIdx.z * 256 + threadIdx.x; idx < WWs; idx += 256 * SHIP_SPLIT) {
            const int wy = (int)(idx / Ws), wx = (int)(idx % Ws);
            const int tx = wx - d.wdx, ty = wy - d.wdy;          // where the sender's entry (wx, wy) ends up
            if (!(tx >= 0 && ty >= 0 && tx < (int)W && ty < (int)W) && blob_load_i16(sdir + idx) >= 0) atomicOr(err, ERR_WINDOW);
        }
        n16 = WW * 2 / 16;                                       // (what this plane occupies in the DESTINATION: nothing is zeroed behind it)
    } else {
        for (size_t k = (size_t)blockIdx.z * 256 + threadIdx.x; k < n16; k += 256 * SHIP_SPLIT) out[k] = blob_load16(in + k);
    }
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t k = (size_t)blockIdx.z * 256 + threadIdx.x; k < nzero; k += 256 * SHIP_SPLIT) out[n16 + k] = z;
    if (plane == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
        dst.counts[0] = dmc; dst.counts[1] = occ;
        const double* hp = reinterpret_cast<const double*>(d.blob);
        for (int k = 0; k < 4; ++k) poses[4 * i + k] = uload_f64(hp + k);
    }
}

} // namespace lama_dev
