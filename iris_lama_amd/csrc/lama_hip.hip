// lama_hip.hip -- C-ABI implementation (include/lama_hip.h) of the MI355X scan-matching path.
// gfx950 only.  No CPU fallback: every entry point needs a HIP device and fails loudly without one.
#include "../../include/lama_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "lama_kernels.h"
#include "lama_pgo.h"

using namespace lama_dev;

namespace {

// The maps of all particles of a context: two directories per HOME (P homes), pooled planes in which every particle owns one
// contiguous region per map kind (lama_dev.h).  A pool is a list of CHUNKS: it grows by another chunk (never by copying), a region
// lies inside one chunk.  The reference's patches are heap objects shared through cow_ptr (include/lama/cow_ptr.h:86-118); here
// memory follows use through per-particle region capacities, and a resample moves nothing that survives.
struct RegionAlloc;
struct PoolChunk {
    uint8_t* plane[4] = {nullptr, nullptr, nullptr, nullptr};      // dm: sv, obs, mask, - ; occ: occ, occ_mask, occ_hit, rev
    uint32_t patches = 0;
};
struct MapStore {
    int16_t* dm_dir = nullptr; int16_t* occ_dir = nullptr;          // [P homes][W*W]
    int32_t* counts = nullptr;                                      // [P][2] logical particle
    std::vector<PoolChunk> dm_chunks, occ_chunks;
};
constexpr size_t DM_PLANE_B[3] = {SV_PATCH_BYTES, 4096, 128};                 // dm_sv, dm_obs, dm_mask bytes per patch
constexpr size_t OCC_PLANE_B[4] = {4096, 128, 128, 4};              // occ, occ_mask, occ_hit, rev
// the host's record of a particle: home + region (chunk, offset, capacity) per map kind; the device gets PartRec (addresses)
struct HostPart { uint32_t home, dm_chunk, dm_off, dm_cap, occ_chunk, occ_off, occ_cap; };

// First-fit allocator of contiguous regions (in patches) over a pool; neighbouring free blocks coalesce.  Host side only: the device
// sees the result as the plane addresses in PartRec.  Free space is always all-zero on the device (k_zero_regions).
struct RegionAlloc {
    std::map<uint32_t, uint32_t> free_;        // offset -> length
    uint32_t cap = 0, used = 0;
    void reset(uint32_t c) { free_.clear(); cap = c; used = 0; if (c) free_[0] = c; }
    bool alloc(uint32_t n, uint32_t& off)
    {
        if (n == 0) { off = 0; return true; }
        for (auto it = free_.begin(); it != free_.end(); ++it) {
            if (it->second < n) continue;
            off = it->first;
            const uint32_t rest = it->second - n;
            free_.erase(it);
            if (rest) free_[off + n] = rest;
            used += n;
            return true;
        }
        return false;
    }
    void release(uint32_t off, uint32_t n)
    {
        if (n == 0) return;
        used -= n;
        auto nx = free_.lower_bound(off);
        if (nx != free_.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == off) { off = pv->first; n += pv->second; free_.erase(pv); }
        }
        if (nx != free_.end() && off + n == nx->first) { n += nx->second; free_.erase(nx); }
        free_[off] = n;
    }
    void extend(uint32_t newcap) { if (newcap > cap) { const uint32_t o = cap, n = newcap - cap; cap = newcap; used += n; release(o, n); } }
    uint32_t largest_free() const { uint32_t m = 0; for (auto& kv : free_) m = std::max(m, kv.second); return m; }
};

} // namespace

// Page-locked host array with the part of std::vector's interface the context uses: the per-call transfers (poses, counts,
// statistics, transforms, scan points) then are real DMA transfers that overlap with kernel launches instead of staged,
// blocking pageable copies.
template <class T>
struct PinVec {
    T* p = nullptr; size_t n = 0, cap = 0; bool pinned = false;
    PinVec() {}
    PinVec(const PinVec&) = delete;
    PinVec& operator=(const PinVec&) = delete;
    ~PinVec() { release(); }
    void release() { if (p) { if (pinned) (void)hipHostFree(p); else std::free(p); } p = nullptr; }
    void reserve(size_t m)
    {
        if (m <= cap) return;
        m = std::max(m, 2 * cap);                                   // geometric: a list built element by element re-pins O(log n) times
        T* q = nullptr;
        bool pin = hipHostMalloc((void**)&q, m * sizeof(T), hipHostMallocDefault) == hipSuccess && q;
        if (!pin) { (void)hipGetLastError(); q = (T*)std::malloc(m * sizeof(T)); }       // pageable memory still works, only slower
        if (!q) std::abort();                                                           // out of host memory
        if (p) std::memcpy(q, p, n * sizeof(T));
        release();
        p = q; cap = m; pinned = pin;
    }
    void resize(size_t m) { reserve(m); if (m > n) std::memset(p + n, 0, (m - n) * sizeof(T)); n = m; }
    void assign(size_t m, T v) { reserve(m); n = m; for (size_t i = 0; i < m; ++i) p[i] = v; }
    T* data() { return p; }
    const T* data() const { return p; }
    size_t size() const { return n; }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
};

struct lama_hip_ctx {
    lama_hip_cfg cfg;
    std::string error;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct PendingTimer { hipEvent_t a, b; double* acc; uint64_t* launches; };
    std::vector<hipEvent_t> ev_pool;            // pairs handed out by Timer, resolved after the call's final sync
    std::vector<PendingTimer> pending;
    size_t ev_used = 0;
    bool initialised = false;   // first scan done
    bool poisoned = false;      // a resample ran out of device memory half-way: every later call fails (LAMA_HIP_E_STATE)

    uint32_t P = 0, W = 0, WC = 0, wx0 = 0, wy0 = 0;
    uint32_t max_sqdist = 0;
    double scale = 0, off = 0;

    MapStore ms;
    std::vector<RegionAlloc> ra_dm, ra_occ;     // one allocator per chunk
    std::vector<HostPart> h_part;     // logical particle -> home / regions (host truth) ...
    PartRec* d_part = nullptr;        // ... and its device copy
    PinVec<PartRec> h_part_stage;     // (page-locked staging of the upload)
    PinVec<CloneJob> h_jobs; CloneJob* d_jobs = nullptr; uint32_t jobs_cap = 0;
    PinVec<ZeroJob> h_zjobs; ZeroJob* d_zjobs = nullptr; uint32_t zjobs_cap = 0;
    uint32_t floor_dm = 128, floor_occ = 128;      // smallest region a particle gets (cfg.dm_patch_capacity / occ_patch_capacity)
    PinVec<uint32_t> h_guard;         // the guard's per-particle bound of the last map update
    uint64_t clone_bytes = 0;         // bytes the particle copies of the last resample moved (counters)
    double* d_poses = nullptr;
    uint64_t* d_qlower = nullptr; uint64_t* d_qraise = nullptr;
    uint64_t* d_stats = nullptr;
    uint32_t* d_qsizes = nullptr;
    uint64_t* d_dbg = nullptr;
    uint32_t* d_slow = nullptr;
    uint32_t* d_slow_list = nullptr; uint32_t* d_slow_n = nullptr;      // hand-over list of the brushfire's first stage
    // routing of the long brushfire chains (k_bf_route): second stream + events for the big-queue stage that runs beside the first one
    uint8_t* d_heavy = nullptr;
    // early lane: the particles routed in the previous update (d_heavy still holds them; valid while the particle set is unchanged)
    uint8_t* d_early = nullptr; uint32_t* d_elist = nullptr;
    hipStream_t stream3 = nullptr; hipEvent_t ev_alloc = nullptr, ev_go = nullptr, ev_early = nullptr;
    bool early_ok = false;            // the last map update routed over the whole pool and nothing has permuted the particles since
    uint32_t early_candidates = 0;    // routed + early-lane particles of the last map update (none: the early lane is not even launched)
    std::vector<double> h_ll; bool ll_valid = false;   // log-likelihoods of the last scan match, in the current particle order
    PinVec<uint32_t> h_hlist; uint32_t* d_hlist = nullptr;   // the worst-fitting particles of that scan match (early lane)
    uint32_t early_on = 1;            // LAMA_HIP_BF_ROUTE's fifth number (0: no early lane)
    uint32_t early_min_count = 1024;  // below ~4 workgroups per CU everybody else's ray-cast is too short to be worth overtaking (an override sets 0)
    hipStream_t stream2 = nullptr; hipEvent_t ev_route = nullptr, ev_heavy = nullptr;
    uint32_t route_min_count = 64, route_min_events = 48, route_percent = 150, route_cap = 64;     // LAMA_HIP_BF_ROUTE overrides (tests)
    bool route_forced = false; uint32_t num_cus = 256;
    bool debug_tail = false, debug_window = false;      // LAMA_HIP_DEBUG_TAIL / LAMA_HIP_DEBUG_WINDOW, read at creation
    uint32_t big_stage_pad = 0;                         // dynamic LDS added to the early lane's big-queue workgroups (a FULL chip) so that a long chain has its CU to itself
    uint64_t* d_act = nullptr; uint32_t* d_act_count = nullptr;
    // patch-centric ray-cast (lama_raycast_patch.h): ray records / bounding boxes of the scan's beams, arena slot -> directory position
    lama_dev::RayRec* d_rrec = nullptr; uint64_t* d_rbbox = nullptr; lama_dev::RayChunk* d_rchunk = nullptr; size_t rrec_cap = 0;
    int32_t* d_err = nullptr;
    double* d_pts = nullptr; uint32_t pts_cap = 0; uint32_t last_n = 0;
    PinVec<uint64_t> h_stats;
    PinVec<double> h_tfs, h_pts;
    PinVec<int32_t> h_err;
    PinVec<uint32_t> h_slow_n;        // hand-over counts of the last map update (brushfire, ordered replay)
    double* d_tfs = nullptr;
    double* d_loglik = nullptr; int32_t* d_iters = nullptr;
    // d_poses | d_loglik | d_iters | d_err are carved out of ONE allocation so that a scan match brings all of its results
    // (and the error word) back with a single device-to-host copy
    uint8_t* d_results = nullptr;
    size_t results_bytes = 0, status_off = 0, status_bytes = 0;
    PinVec<uint8_t> h_status;
    PinVec<uint8_t> h_results;
    int32_t* d_oldcounts = nullptr;
    double* d_bposes = nullptr; double* d_bout = nullptr; uint32_t b_cap = 0;

    double scan_reach = 0.0;          // largest point distance of the resident scan (sensor frame, metres)
    uint32_t visit_bound = 0;         // upper bound of the largest `visited` counter of any frequency cell (see k_occ_max_visited)
    uint32_t* d_scalar = nullptr;
    // particle shipping: descriptors / gathered blob heads of one batch
    lama_dev::ShipDesc* d_ship_desc = nullptr; uint8_t* d_ship_heads = nullptr; uint32_t ship_cap = 0;
    PinVec<lama_dev::ShipDesc> h_ship_desc; PinVec<uint8_t> h_ship_heads;
    uint32_t* d_guard = nullptr;      // [P][2] bound on the distance-map patches an update may still allocate + block ticket (k_occ_reverse_dir)
    // what the last run_update_maps was asked to do: a cleanly aborted update (ERR_CLEAN_ABORT) is run again after the arenas grew
    Affine last_mtf; uint32_t last_first = 0, last_count = 0;
    bool last_guarded = false;        // the update went through the parallel ray-cast (its allocation phase precedes every modification)
    int recover_depth = 0;
    // the box of everything that can be mapped so far, absolute patch coordinates (ensure_window)
    int64_t mbx0 = 0, mbx1 = 0, mby0 = 0, mby1 = 0; bool mb_valid = false;
    bool unguarded_retry = false;     // the arenas are at their hard limit and the guard's bound did not fit: run the update without it
    // host-side effects of a map update that a repeated pass (recover_update) must not apply twice
    uint32_t saved_visit_bound = 0; lama_hip_counters saved_ctr;
    bool pending_maps = false;        // lama_hip_pf_update_maps_begin queued work whose status has not been collected yet
    PinVec<double> h_poses;           // host mirror of the particle poses (source of truth between calls)
    PinVec<int32_t> h_counts;         // host mirror of counts of the current set (refreshed after map updates)
    lama_hip_counters ctr;
};

namespace {

#define HIPCHK(ctx, call)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            (ctx)->error = std::string(#call) + ": " + hipGetErrorString(e_);                           \
            return LAMA_HIP_E_HIP;                                                                     \
        }                                                                                              \
    } while (0)

int32_t fail(lama_hip_ctx* ctx, int32_t code, const std::string& msg)
{
    ctx->error = msg;
    return code;
}

// Eigen Quaternion::toRotationMatrix restated (moving_tf = Translation(origin) * q,
// src/match_surface_2d.cpp:49, src/pf_slam2d.cpp:397,444)
Affine moving_tf(const double* origin3, const double* q)
{
    Affine a;
    const double w = q ? q[0] : 1.0, x = q ? q[1] : 0.0, y = q ? q[2] : 0.0, z = q ? q[3] : 0.0;
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    a.R[0][0] = 1.0 - (tyy + tzz); a.R[0][1] = txy - twz;         a.R[0][2] = txz + twy;
    a.R[1][0] = txy + twz;         a.R[1][1] = 1.0 - (txx + tzz); a.R[1][2] = tyz - twx;
    a.R[2][0] = txz - twy;         a.R[2][1] = tyz + twx;         a.R[2][2] = 1.0 - (txx + tyy);
    for (int i = 0; i < 3; ++i) a.t[i] = origin3 ? origin3[i] : 0.0;
    return a;
}

// fixed_tf(pose) * moving_tf on the HOST (libm), 12 doubles [R row-major | t]
// (src/pf_slam2d.cpp:444-451 ; Eigen AngleAxis(theta, UnitZ) restated)
void host_scan_tf(const double* pose4, const Affine& m, double* out12)
{
    const double theta = std::atan2(pose4[1], pose4[0]);
    const double sn = std::sin(theta), cs = std::cos(theta);
    const double F[3][3] = {{cs, 0.0 - sn, 0.0}, {sn, cs, 0.0}, {0.0, 0.0, (1.0 - cs) + cs}};
    const double ft[3] = {pose4[2], pose4[3], 0.0};
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) out12[3 * i + j] = (F[i][0] * m.R[0][j] + F[i][1] * m.R[1][j]) + F[i][2] * m.R[2][j];
        out12[9 + i] = ((F[i][0] * m.t[0] + F[i][1] * m.t[1]) + F[i][2] * m.t[2]) + ft[i];
    }
}

DevParams make_params(const lama_hip_ctx* c)
{
    DevParams p;
    p.P = c->P; p.W = c->W; p.WC = c->WC; p.wx0 = c->wx0; p.wy0 = c->wy0;
    p.qcap = c->cfg.queue_capacity; p.part = c->d_part;
    p.max_sqdist = c->max_sqdist; p.max_iter = c->cfg.max_iter;
    p.scale = c->scale; p.off = c->off; p.resolution = c->cfg.resolution;
    p.maxdist = std::sqrt((double)c->max_sqdist) * c->cfg.resolution;
    p.meas_sigma = c->cfg.meas_sigma;
    p.trunc_ray = c->cfg.truncated_ray; p.trunc_range = c->cfg.truncated_range;
    const MapStore& s = c->ms;
    p.dm_dir = s.dm_dir; p.occ_dir = s.occ_dir; p.counts = s.counts;
    p.guard = c->d_guard;
    p.guard_r = ((uint32_t)std::ceil(std::sqrt((double)c->max_sqdist)) + 1u + 31u) / 32u;
    p.poses = c->d_poses; p.q_lower = c->d_qlower; p.q_raise = c->d_qraise; p.stats = c->d_stats; p.qsizes = c->d_qsizes; p.err = c->d_err; p.dbg = c->d_dbg; p.slow = c->d_slow; p.slow_list = c->d_slow_list; p.slow_n = c->d_slow_n; p.heavy = nullptr; p.elist = nullptr; p.elist_n = nullptr; p.early = nullptr; p.lane = 0;
    p.act = c->d_act; p.act_count = c->d_act_count; p.act_cap = c->cfg.active_capacity;
    p.occ_policy = c->cfg.occupancy_policy; p.ray_rule = c->cfg.ray_rule; p.strategy = c->cfg.solver_strategy;
    // ProbabilisticOccupancyMap's parameters (probabilistic_occupancy_map.cpp:43-59): logods(p) = float(log(p / (1 - p))) of a
    // float argument, stored in double members
    auto logods = [](float prob) { return (double)(float)std::log(prob / (1.0 - prob)); };
    p.lo_miss = logods(0.4f); p.lo_hit = logods(0.7f); p.lo_min = logods(0.12f); p.lo_max = logods(0.97f);
    return p;
}

int32_t upload_scan(lama_hip_ctx* c, const double* pts, uint32_t n)
{
    if (!pts) {      // reuse the scan uploaded by the previous call (same scan for scan_match and update_maps)
        if (c->last_n == 0 || n != c->last_n) return fail(c, LAMA_HIP_E_INVALID, "pts == NULL but no matching scan is resident");
        return LAMA_HIP_OK;
    }
    c->last_n = n;
    if (n > c->pts_cap) {
        if (c->d_pts) HIPCHK(c, hipFree(c->d_pts));
        c->d_pts = nullptr;
        c->pts_cap = std::max<uint32_t>(n, 2048);
        HIPCHK(c, hipMalloc(&c->d_pts, sizeof(double) * 3 * c->pts_cap));
    }
    c->h_pts.resize((size_t)3 * n);                       // the caller's buffer is only borrowed for the call
    std::memcpy(c->h_pts.data(), pts, sizeof(double) * 3 * n);
    double r2 = 0.0;
    for (uint32_t i = 0; i < n; ++i) r2 = std::max(r2, pts[3 * i] * pts[3 * i] + pts[3 * i + 1] * pts[3 * i + 1] + pts[3 * i + 2] * pts[3 * i + 2]);
    // a non-finite coordinate has no cell (the reference's w2m casts it to an integer: undefined behaviour, include/lama/sdm/map.h:125-128)
    if (!std::isfinite(r2)) { c->last_n = 0; return fail(c, LAMA_HIP_E_INVALID, "the scan holds a non-finite point (filter NaN / inf ranges before the update, as iris_lama_ros does)"); }
    c->scan_reach = std::sqrt(r2);
    HIPCHK(c, hipMemcpyAsync(c->d_pts, c->h_pts.data(), sizeof(double) * 3 * n, hipMemcpyHostToDevice, c->stream));
    return LAMA_HIP_OK;
}

void resolve_timers(lama_hip_ctx* c);

// End of an API call: one stream synchronisation that brings back the device error word and, when asked, the
// per-particle patch counts and statistics (a single host round trip per call).
int32_t grow_arenas(lama_hip_ctx* c, uint32_t need_dm, uint32_t need_occ);
int32_t recover_update(lama_hip_ctx* c, int32_t e);
int32_t run_update_maps(lama_hip_ctx* c, uint32_t n, const Affine& mtf, uint32_t first, uint32_t count);
int32_t permute_particles(lama_hip_ctx* c, const int32_t* idx);

int32_t check_device_errors(lama_hip_ctx* c, bool maps = false, bool match = false, bool err_in_results = false)
{
    c->h_err.resize(1);
    int32_t& e = c->h_err[0];
    e = 0;
    const bool stats = maps || match;
    const bool packed = maps && !err_in_results;      // a map update: error word, hand-over counts, statistics, guard and patch counts in ONE copy
    if (stats) c->h_stats.resize((size_t)c->P * 4);
    if (packed) {
        c->h_status.resize(c->status_bytes); c->h_guard.resize(c->P); c->h_slow_n.resize(5);
        HIPCHK(c, hipMemcpyAsync(c->h_status.data(), c->d_results + c->status_off, c->status_bytes, hipMemcpyDeviceToHost, c->stream));
    } else {
        if (stats) {
            HIPCHK(c, hipMemcpyAsync(c->h_counts.data(), c->ms.counts, sizeof(int32_t) * 2 * c->P, hipMemcpyDeviceToHost, c->stream));
            if (maps) { c->h_guard.resize(c->P); HIPCHK(c, hipMemcpyAsync(c->h_guard.data(), c->d_guard, sizeof(uint32_t) * c->P, hipMemcpyDeviceToHost, c->stream)); }
            HIPCHK(c, hipMemcpyAsync(c->h_stats.data(), c->d_stats, sizeof(uint64_t) * c->h_stats.size(), hipMemcpyDeviceToHost, c->stream));
        }
        if (!err_in_results) HIPCHK(c, hipMemcpyAsync(&e, c->d_err, sizeof(e), hipMemcpyDeviceToHost, c->stream));
        if (maps) { c->h_slow_n.resize(5); HIPCHK(c, hipMemcpyAsync(c->h_slow_n.data(), c->d_slow_n, 5 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream)); }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (packed) {
        const uint8_t* b = c->h_status.data();
        const size_t o_slow = (const uint8_t*)c->d_slow_n - (c->d_results + c->status_off);
        std::memcpy(&e, b, sizeof(e));
        std::memcpy(c->h_slow_n.data(), b + o_slow, 5 * sizeof(uint32_t));
        std::memcpy(c->h_stats.data(), b + o_slow + 32, sizeof(uint64_t) * 4 * c->P);
        std::memcpy(c->h_guard.data(), b + o_slow + 32 + sizeof(uint64_t) * 4 * c->P, sizeof(uint32_t) * c->P);
        std::memcpy(c->h_counts.data(), b + o_slow + 32 + sizeof(uint64_t) * 4 * c->P + sizeof(uint32_t) * 2 * c->P, sizeof(int32_t) * 2 * c->P);
    }
    if (maps) { c->ctr.brushfire_handovers += c->h_slow_n[0]; c->ctr.replay_handovers += c->h_slow_n[1] + c->h_slow_n[3]; c->ctr.brushfire_routed += c->h_slow_n[2] + c->h_slow_n[4]; c->ctr.brushfire_early += c->h_slow_n[4]; c->early_candidates = c->h_slow_n[2] + c->h_slow_n[4]; }
    if (err_in_results) std::memcpy(&e, c->h_results.data() + ((const uint8_t*)c->d_err - c->d_results), sizeof(e));
    resolve_timers(c);
    if (e != 0) {
        HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(int32_t), c->stream));
        // The allocation phase of a map update ran out of patches before any cell was modified: grow and run the update again.
        if (maps && (e & ERR_CLEAN_ABORT) && (e & (ERR_DM_CAP | ERR_OCC_CAP)) && !(e & ERR_WINDOW) && c->last_guarded && c->recover_depth < 10) {
            const int32_t rr = recover_update(c, e);
            if (rr != LAMA_HIP_E_CAPACITY) return rr;      // done (or another error); E_CAPACITY: the arenas are at their limit
        }
        if (e & (ERR_DM_CAP | ERR_OCC_CAP)) {
            // A failed allocation keeps counting (dir_alloc_one: the counts then say what a particle WANTED, which the clean-abort path
            // uses).  When the error is final, the counts go back into the regions: every later copy (resample, export, a region move)
            // is sized from them and must not run into a neighbour's region.
            if (!stats) HIPCHK(c, hipMemcpyAsync(c->h_counts.data(), c->ms.counts, sizeof(int32_t) * 2 * c->P, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            bool clamped = false;
            for (uint32_t p = 0; p < c->P; ++p) {
                const HostPart& hp = c->h_part[p];
                if ((uint32_t)c->h_counts[2 * p] > hp.dm_cap) { c->h_counts[2 * p] = (int32_t)hp.dm_cap; clamped = true; }
                if ((uint32_t)c->h_counts[2 * p + 1] > hp.occ_cap) { c->h_counts[2 * p + 1] = (int32_t)hp.occ_cap; clamped = true; }
            }
            if (clamped) {
                HIPCHK(c, hipMemcpyAsync(c->ms.counts, c->h_counts.data(), sizeof(int32_t) * 2 * c->P, hipMemcpyHostToDevice, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
        }
        if (e & ERR_WINDOW) return fail(c, LAMA_HIP_E_WINDOW, "a map cell fell outside the device window (the mapped area is wider than 1016 patches)");
        if (e & ERR_DM_CAP) return fail(c, LAMA_HIP_E_CAPACITY, "a particle's distance map exceeds 32767 patches (or the device is out of memory)");
        if (e & ERR_OCC_CAP) return fail(c, LAMA_HIP_E_CAPACITY, "a particle's occupancy map exceeds 32767 patches (or the device is out of memory)");
        if (e & ERR_QUEUE) return fail(c, LAMA_HIP_E_CAPACITY, "brushfire queue full (raise cfg.queue_capacity)");
        return fail(c, LAMA_HIP_E_NUMERIC, "unit complex number is (near) zero (SophusException in the reference)");
    }
    if (stats) {
        const PinVec<uint64_t>& st = c->h_stats;
        uint64_t dm = 0, oc = 0;
        for (uint32_t p = 0; p < c->P; ++p) {
            if (match) { c->ctr.gn_iterations += st[4 * p]; c->ctr.gn_evals += st[4 * p + 1]; }
            if (maps) { c->ctr.ray_cells += st[4 * p + 2]; c->ctr.bf_cells += st[4 * p + 3]; }
            dm += c->h_counts[2 * p]; oc += c->h_counts[2 * p + 1];
        }
        c->ctr.dm_patches = dm; c->ctr.occ_patches = oc;
        if (maps) {                                                  // the update lasted as long as its longest chain
            uint64_t mx = 0, sum = 0; uint32_t arg = 0;
            for (uint32_t p = 0; p < c->P; ++p) { sum += st[4 * p + 3]; if (st[4 * p + 3] > mx) { mx = st[4 * p + 3]; arg = p; } }
            c->ctr.bf_longest_chain_sum += mx; c->ctr.bf_longest_chain_last = mx;
            if (c->debug_tail) {
                // (how well would the scan match's log-likelihood have predicted the longest chain? its rank among the pool, 0 = lowest)
                uint32_t rank = 0; double ll = 0, mean_ll = 0;
                if (c->h_results.size() >= c->results_bytes && c->P > 1) {
                    const double* L = reinterpret_cast<const double*>(c->h_results.data() + sizeof(double) * 4 * c->P);
                    ll = L[arg];
                    for (uint32_t p = 0; p < c->P; ++p) { rank += L[p] < ll ? 1u : 0u; mean_ll += L[p]; }
                    mean_ll /= c->P;
                }
                std::fprintf(stderr, "brushfire pops: mean %.0f max %llu (particle %u) handovers %u brushfire %.3f ms; log-lik of that particle %.1f (pool mean %.1f), rank %u of %u\n", (double)sum / c->P, (unsigned long long)mx, arg, c->h_slow_n[0], c->ctr.ms_brushfire, ll, mean_ll, rank, c->P);
            }
        }
        if (maps) return grow_arenas(c, 0, 0);
    }
    return LAMA_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// Memory that follows the maps.  The reference's maps allocate patches on demand without bound (src/sdm/map.cpp:400-411) and share
// them between particles copy-on-write (include/lama/cow_ptr.h:86-118).  Here every particle owns one contiguous region of each
// pool; a region has the capacity ITS particle needs (counts are on the host after every map update) and is moved to a larger one
// -- that particle alone -- when it runs short; the pools themselves grow geometrically when the allocator finds no room.
// ------------------------------------------------------------------------------------------------
constexpr uint32_t MAX_PATCHES = 32767u;      // directory entries are int16

static uint32_t round_up32(uint32_t v) { return (v + 31u) / 32u * 32u; }

// the region capacity a particle with `count` used slots should have so that the next update normally fits (`extra`: head room
// asked for explicitly, e.g. the allocation guard's bound)
static uint32_t want_capacity(uint32_t floor_cap, uint32_t count, uint32_t extra)
{
    const uint32_t head = std::max<uint32_t>(std::max<uint32_t>(32u, count / 4u), extra);
    return std::min<uint32_t>(MAX_PATCHES, std::max<uint32_t>(floor_cap, round_up32(count + head)));
}

static PartRec dev_part(const lama_hip_ctx* c, const HostPart& h)
{
    PartRec r{};
    r.home = h.home; r.dm_cap = h.dm_cap; r.occ_cap = h.occ_cap;
    const PoolChunk& d = c->ms.dm_chunks[h.dm_chunk];
    const PoolChunk& o = c->ms.occ_chunks[h.occ_chunk];
    r.dm_sv = (sv_t*)(d.plane[0] + (size_t)h.dm_off * DM_PLANE_B[0]);
    r.dm_obs = (uint32_t*)(d.plane[1] + (size_t)h.dm_off * DM_PLANE_B[1]);
    r.dm_mask = (uint64_t*)(d.plane[2] + (size_t)h.dm_off * DM_PLANE_B[2]);
    r.occ = (uint32_t*)(o.plane[0] + (size_t)h.occ_off * OCC_PLANE_B[0]);
    r.occ_mask = (uint64_t*)(o.plane[1] + (size_t)h.occ_off * OCC_PLANE_B[1]);
    r.occ_hit = (uint64_t*)(o.plane[2] + (size_t)h.occ_off * OCC_PLANE_B[2]);
    r.rev = (int32_t*)(o.plane[3] + (size_t)h.occ_off * OCC_PLANE_B[3]);
    return r;
}

int32_t upload_part(lama_hip_ctx* c)
{
    c->h_part_stage.resize(c->P);
    for (uint32_t p = 0; p < c->P; ++p) c->h_part_stage[p] = dev_part(c, c->h_part[p]);
    HIPCHK(c, hipMemcpyAsync(c->d_part, c->h_part_stage.data(), sizeof(PartRec) * c->P, hipMemcpyHostToDevice, c->stream));
    return LAMA_HIP_OK;
}

// one more chunk for a pool, of at least `min_patches` (free pool space is all-zero: the chunk is zeroed here).  Nothing that
// exists is copied or moved: a pool grows to whatever the device holds without a second copy of itself.
int32_t add_chunk(lama_hip_ctx* c, bool dm, uint32_t min_patches)
{
    std::vector<PoolChunk>& chunks = dm ? c->ms.dm_chunks : c->ms.occ_chunks;
    uint64_t total = 0;
    for (const PoolChunk& k : chunks) total += k.patches;
    uint64_t want = std::max<uint64_t>(std::max<uint64_t>(min_patches, total / 4u), 1024u);      // geometric: the number of chunks stays small, the last one mostly used
    want = (want + 1023u) / 1024u * 1024u;
    const uint64_t per_patch = dm ? (SV_PATCH_BYTES + 4096 + 128) : (4096 + 128 + 128 + 4);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const uint64_t room = free_b > (1ull << 30) ? free_b - (1ull << 30) : 0;       // leave 1 GB for everything else
        if (want * per_patch > room) want = room / per_patch / 32u * 32u;
        if (want < min_patches)
            return fail(c, LAMA_HIP_E_CAPACITY, std::string(dm ? "distance-map" : "occupancy") + " pool: the device has no room for " + std::to_string(min_patches) + " more patches");
    }
    if (want > 0xFFFFFFF0ull) want = 0xFFFFFFF0ull;
    PoolChunk k;
    k.patches = (uint32_t)want;
    const int nplanes = dm ? 3 : 4;
    for (int i = 0; i < nplanes; ++i) {
        const size_t bytes = (size_t)k.patches * (dm ? DM_PLANE_B[i] : OCC_PLANE_B[i]);
        hipError_t e = hipMalloc(&k.plane[i], bytes);
        if (e == hipSuccess && !(!dm && i == 3)) e = hipMemsetAsync(k.plane[i], 0, bytes, c->stream);     // (rev is rebuilt per scan)
        if (e != hipSuccess) {
            for (int j = 0; j <= i; ++j) (void)hipFree(k.plane[j]);
            (void)hipGetLastError();
            return fail(c, LAMA_HIP_E_CAPACITY, std::string(dm ? "distance-map" : "occupancy") + " pool: " + hipGetErrorString(e));
        }
    }
    chunks.push_back(k);
    (dm ? c->ra_dm : c->ra_occ).emplace_back();
    (dm ? c->ra_dm : c->ra_occ).back().reset(k.patches);
    c->ctr.pool_growths += 1;
    return LAMA_HIP_OK;
}

int32_t region_alloc(lama_hip_ctx* c, bool dm, uint32_t n, uint32_t& chunk, uint32_t& off)
{
    std::vector<RegionAlloc>& ras = dm ? c->ra_dm : c->ra_occ;
    for (uint32_t k = 0; k < ras.size(); ++k)
        if (ras[k].alloc(n, off)) { chunk = k; return LAMA_HIP_OK; }
    const int32_t rc = add_chunk(c, dm, n);
    if (rc) return rc;
    chunk = (uint32_t)ras.size() - 1;
    if (!ras[chunk].alloc(n, off)) return fail(c, LAMA_HIP_E_CAPACITY, "patch pool exhausted");
    return LAMA_HIP_OK;
}
static void region_release(lama_hip_ctx* c, bool dm, uint32_t chunk, uint32_t off, uint32_t n) { (dm ? c->ra_dm : c->ra_occ)[chunk].release(off, n); }

// addresses of a region's planes (dm: sv, obs, mask -> [0..2]; occ: occ, occ_mask -> [3..4]) for the clone / zero jobs
static void region_ptrs(const lama_hip_ctx* c, const HostPart& h, void* out[5])
{
    const PartRec r = dev_part(c, h);
    out[0] = r.dm_sv; out[1] = r.dm_obs; out[2] = r.dm_mask; out[3] = r.occ; out[4] = r.occ_mask;
}

static int32_t ensure_job_buffers(lama_hip_ctx* c, uint32_t nj, uint32_t nz)
{
    if (nj > c->jobs_cap) {
        (void)hipFree(c->d_jobs); c->d_jobs = nullptr; c->jobs_cap = 0;
        const uint32_t cap = std::max<uint32_t>(nj, 256u);
        HIPCHK(c, hipMalloc(&c->d_jobs, sizeof(CloneJob) * cap));
        c->jobs_cap = cap;
    }
    if (nz > c->zjobs_cap) {
        (void)hipFree(c->d_zjobs); c->d_zjobs = nullptr; c->zjobs_cap = 0;
        const uint32_t cap = std::max<uint32_t>(nz, 256u);
        HIPCHK(c, hipMalloc(&c->d_zjobs, sizeof(ZeroJob) * cap));
        c->zjobs_cap = cap;
    }
    return LAMA_HIP_OK;
}

// run the jobs in c->h_jobs / c->h_zjobs: `zero_first` = the zero jobs free regions no job reads (resample); else they free the jobs'
// own sources (growth) and run behind the copies
// (j0 / z0: the jobs from these positions of the staging lists on -- set_capacities queues group after group without a host round
// trip in between, so a group's slice of the page-locked lists must stay untouched until the stream has consumed it)
static int32_t run_jobs(lama_hip_ctx* c, bool zero_first, size_t j0 = 0, size_t z0 = 0)
{
    const uint32_t nj = (uint32_t)(c->h_jobs.size() - j0), nz = (uint32_t)(c->h_zjobs.size() - z0);
    if (nj == 0 && nz == 0) return LAMA_HIP_OK;
    const int32_t rb = ensure_job_buffers(c, nj, nz);
    if (rb) return rb;
    const DevParams prm = make_params(c);
    if (nj) HIPCHK(c, hipMemcpyAsync(c->d_jobs, c->h_jobs.data() + j0, sizeof(CloneJob) * nj, hipMemcpyHostToDevice, c->stream));
    if (nz) HIPCHK(c, hipMemcpyAsync(c->d_zjobs, c->h_zjobs.data() + z0, sizeof(ZeroJob) * nz, hipMemcpyHostToDevice, c->stream));
    if (nz && zero_first) hipLaunchKernelGGL(k_zero_regions, dim3(nz, 5, CLONE_SPLIT), dim3(256), 0, c->stream, (const ZeroJob*)c->d_zjobs);
    if (nj) hipLaunchKernelGGL(k_clone_particles, dim3(nj, 7, CLONE_SPLIT), dim3(256), 0, c->stream, prm, (const CloneJob*)c->d_jobs);
    if (nz && !zero_first) hipLaunchKernelGGL(k_zero_regions, dim3(nz, 5, CLONE_SPLIT), dim3(256), 0, c->stream, (const ZeroJob*)c->d_zjobs);
    HIPCHK(c, hipGetLastError());
    return LAMA_HIP_OK;
}

// Give the listed particles regions of (at least) the capacities asked for; a particle whose region is large enough keeps it.  The
// used slots move with one launch for all of them, the old regions go back to the allocator zeroed.  Synchronises the stream.
struct CapRequest { uint32_t p, dm_cap, occ_cap; };
int32_t set_capacities(lama_hip_ctx* c, const std::vector<CapRequest>& reqs)
{
    // The particles of one filter map the same world and outgrow their regions together.  Moved all at once, every new region would
    // have to exist beside every old one (a second copy of the maps, and a pool that never gets its first half back: the freed
    // regions are all smaller than the next request).  So the movers go in groups: a group's old regions are zeroed and released
    // before the next group allocates -- first fit from the lowest address then places the next new regions into the space the
    // previous groups left (neighbouring free regions coalesce), and the pool stays at live data + head room + one group.
    // The groups are ordered by the STREAM, not by the host (one host round trip per group was ~48 of them for one growth
    // event at 3000 particles): group g+1's copies run behind group g's copies and zero jobs, so a region released on the host right
    // after group g was queued may be handed to group g+1.  Every group has its own slice of the page-locked job lists (reserved up
    // front: the lists must not be re-allocated while a transfer is pending) and of nothing else -- the device-side lists are read by
    // a group's kernels before the next group's upload overwrites them, in stream order.
    constexpr size_t GROUP = 64;
    struct Old { bool dm; uint32_t chunk, off, cap; };
    bool any = false;
    int32_t err = LAMA_HIP_OK;                                     // a failed allocation: what was planned so far is still carried out (the table,
                                                                   // the regions and their contents stay consistent), then the error is returned
    c->h_jobs.resize(0); c->h_zjobs.resize(0);
    c->h_jobs.reserve(reqs.size()); c->h_zjobs.reserve(reqs.size());
    for (size_t g0 = 0; g0 < reqs.size() && err == LAMA_HIP_OK; g0 += GROUP) {
        std::vector<Old> released;
        const size_t j0 = c->h_jobs.size(), z0 = c->h_zjobs.size();
        for (size_t q = g0; q < std::min(reqs.size(), g0 + GROUP); ++q) {
            const CapRequest& r = reqs[q];
            HostPart& hp = c->h_part[r.p];
            const bool gd = r.dm_cap > hp.dm_cap, go = r.occ_cap > hp.occ_cap;
            if (!gd && !go) continue;
            if (r.dm_cap > MAX_PATCHES || r.occ_cap > MAX_PATCHES) { err = fail(c, LAMA_HIP_E_CAPACITY, "a particle's map exceeds 32767 patches"); break; }
            const HostPart before = hp;
            uint32_t dch = 0, doff = 0, och = 0, ooff = 0;
            if (gd) { err = region_alloc(c, true, r.dm_cap, dch, doff); if (err) break; }
            if (go) { err = region_alloc(c, false, r.occ_cap, och, ooff); if (err) { if (gd) region_release(c, true, dch, doff, r.dm_cap); break; } }
            if (gd) {
                released.push_back(Old{true, hp.dm_chunk, hp.dm_off, hp.dm_cap});
                hp.dm_chunk = dch; hp.dm_off = doff; hp.dm_cap = r.dm_cap;
            }
            if (go) {
                released.push_back(Old{false, hp.occ_chunk, hp.occ_off, hp.occ_cap});
                hp.occ_chunk = och; hp.occ_off = ooff; hp.occ_cap = r.occ_cap;
            }
            CloneJob j{};
            j.src_home = j.dst_home = hp.home;
            j.sdm = c->h_counts[2 * r.p]; j.socc = c->h_counts[2 * r.p + 1];
            void* sp[5]; void* dp[5];
            region_ptrs(c, before, sp); region_ptrs(c, hp, dp);
            ZeroJob z{};
            for (int k = 0; k < 5; ++k) { j.s[k] = sp[k]; j.d[k] = dp[k]; z.d[k] = (sp[k] != dp[k]) ? sp[k] : nullptr; }
            z.ndm = j.sdm; z.nocc = j.socc;
            c->h_jobs.resize(c->h_jobs.size() + 1); c->h_jobs[c->h_jobs.size() - 1] = j;
            c->h_zjobs.resize(c->h_zjobs.size() + 1); c->h_zjobs[c->h_zjobs.size() - 1] = z;
        }
        if (released.empty()) continue;
        any = true;
        const int32_t rc = run_jobs(c, false, j0, z0);
        if (rc) return rc;
        for (const Old& o : released) region_release(c, o.dm, o.chunk, o.off, o.cap);     // (only now: no job of THIS group may land in a region another one of it still reads)
    }
    if (!any) return err;
    const std::string msg = c->error;
    const int32_t rc = upload_part(c);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->ctr.arena_growths += 1;
    if (err) c->error = msg;
    return err;
}

// after a map update (counts on the host, stream idle): every particle keeps head room for the next one
int32_t grow_arenas(lama_hip_ctx* c, uint32_t /*need_dm*/, uint32_t /*need_occ*/)
{
    std::vector<CapRequest> reqs;
    // the guard's bound of the last update says what the next one will ask for (open space: many occupancy patches without a
    // distance-map patch nearby)
    for (uint32_t p = 0; p < c->P; ++p) {
        const HostPart& pr = c->h_part[p];
        const uint32_t dmc = (uint32_t)c->h_counts[2 * p], occ = (uint32_t)c->h_counts[2 * p + 1];
        const uint32_t gh = p < c->h_guard.size() ? c->h_guard[p] : 0u;
        uint32_t nd = pr.dm_cap, no = pr.occ_cap;
        if (dmc + std::max<uint32_t>(std::max<uint32_t>(16u, dmc / 16u), gh) > pr.dm_cap) nd = want_capacity(c->floor_dm, dmc, gh + gh / 4u);
        if (occ + std::max<uint32_t>(16u, occ / 16u) > pr.occ_cap) no = want_capacity(c->floor_occ, occ, 0u);
        if (nd > pr.dm_cap || no > pr.occ_cap) reqs.push_back(CapRequest{p, std::max(nd, pr.dm_cap), std::max(no, pr.occ_cap)});
    }
    if (reqs.empty()) return LAMA_HIP_OK;
    return set_capacities(c, reqs);
}

// A map update whose allocation phase failed (ERR_CLEAN_ABORT: no cell was modified): clear what the phase left behind, give the
// particles that ran short what they asked for and run the same update again.  The reference's maps simply allocate
// (src/sdm/map.cpp:400-411).  What a particle asked for: the occupancy allocations count on past the capacity (dir_alloc_one), the
// distance-map bound is the guard's (k_occ_reverse_dir -> guard[2p]).
int32_t recover_update(lama_hip_ctx* c, int32_t e)
{
    ++c->recover_depth;
    // the aborted pass changed nothing on the device; undo what it did on the host (wrap-guard bound, scan / launch counters)
    c->visit_bound = c->saved_visit_bound;
    {
        const lama_hip_counters now = c->ctr;
        c->ctr = c->saved_ctr;
        c->ctr.arena_growths = now.arena_growths; c->ctr.pool_growths = now.pool_growths; c->ctr.dm_patches = now.dm_patches; c->ctr.occ_patches = now.occ_patches;
    }
    DevParams prm = make_params(c);
    const size_t WW = (size_t)c->W * c->W;
    hipLaunchKernelGGL(k_update_cleanup, dim3(c->P, (unsigned)((WW + 255) / 256)), dim3(256), 0, c->stream, prm);
    for (const PoolChunk& k : c->ms.occ_chunks) HIPCHK(c, hipMemsetAsync(k.plane[2], 0, (size_t)k.patches * 128, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_act_count, 0, (size_t)c->P * 4, c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_err, 0, sizeof(int32_t), c->stream));
    // the counts hold what every particle WANTED (allocations count on past the capacity); the guard its distance-map bound
    c->h_guard.resize(c->P);
    HIPCHK(c, hipMemcpyAsync(c->h_counts.data(), c->ms.counts, sizeof(int32_t) * 2 * c->P, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_guard.data(), c->d_guard, sizeof(uint32_t) * c->P, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<CapRequest> reqs;
    bool at_limit = false, fixed_counts = false;
    for (uint32_t p = 0; p < c->P; ++p) {
        HostPart& pr = c->h_part[p];
        uint32_t wdm = (uint32_t)c->h_counts[2 * p], wocc = (uint32_t)c->h_counts[2 * p + 1];
        uint32_t nd = pr.dm_cap, no = pr.occ_cap;
        if (wocc > pr.occ_cap) { no = want_capacity(c->floor_occ, wocc, 0u); c->h_counts[2 * p + 1] = (int32_t)pr.occ_cap; fixed_counts = true; }
        if (wdm > pr.dm_cap) { c->h_counts[2 * p] = (int32_t)pr.dm_cap; wdm = pr.dm_cap; fixed_counts = true; nd = want_capacity(c->floor_dm, wdm, 64u); }
        const uint32_t gh = c->h_guard[p];
        if ((e & ERR_DM_CAP) && !c->unguarded_retry && (uint64_t)wdm + gh > pr.dm_cap) {
            if ((uint64_t)wdm + gh > MAX_PATCHES) at_limit = true;      // the guard's bound does not fit the hard limit: run this update unguarded
            nd = std::max(nd, std::min<uint32_t>(MAX_PATCHES, round_up32(wdm + gh + gh / 8u + 16u)));
        }
        if (nd > pr.dm_cap || no > pr.occ_cap) {
            if (no > MAX_PATCHES || (wocc > MAX_PATCHES)) { --c->recover_depth; return LAMA_HIP_E_CAPACITY; }
            reqs.push_back(CapRequest{p, std::max(nd, pr.dm_cap), std::max(no, pr.occ_cap)});
        }
    }
    if (fixed_counts) HIPCHK(c, hipMemcpyAsync(c->ms.counts, c->h_counts.data(), sizeof(int32_t) * 2 * c->P, hipMemcpyHostToDevice, c->stream));
    if (reqs.empty() && !at_limit) { --c->recover_depth; return LAMA_HIP_E_CAPACITY; }     // nothing to grow: a real limit
    if (at_limit && c->unguarded_retry) { --c->recover_depth; return LAMA_HIP_E_CAPACITY; }
    int32_t rc = set_capacities(c, reqs);
    const bool saved_unguarded = c->unguarded_retry;
    c->unguarded_retry = at_limit;
    if (rc == LAMA_HIP_OK) rc = run_update_maps(c, c->last_n, c->last_mtf, c->last_first, c->last_count);
    if (rc == LAMA_HIP_OK) rc = check_device_errors(c, true, false);
    --c->recover_depth;
    c->unguarded_retry = saved_unguarded;
    return rc;
}

// Collects the status of a map update queued by lama_hip_pf_update_maps_begin (one synchronisation); called at the start of
// every other entry point, so a deferred error surfaces in the next call on the context.
int32_t finish_pending(lama_hip_ctx* c)
{
    if (!c->pending_maps) return LAMA_HIP_OK;
    c->pending_maps = false;
    (void)hipSetDevice(c->cfg.device);
    const int32_t rc = check_device_errors(c, true, false);
    if (rc) c->error = "deferred from lama_hip_pf_update_maps_begin: " + c->error;
    return rc;
}
#define ENTER(c) do { if ((c)->poisoned) return fail((c), LAMA_HIP_E_STATE, "the context ran out of device memory in the middle of a resample and cannot continue"); const int32_t rc_enter_ = finish_pending(c); if (rc_enter_) return rc_enter_; } while (0)

// hipEvent bracket around a kernel group; resolved (elapsed time read) by resolve_timers() after the API call's final
// stream synchronisation, so profiling adds no host round trips.
struct Timer {
    lama_hip_ctx* c; double* acc; uint64_t* launches; hipEvent_t a = nullptr, b = nullptr;
    Timer(lama_hip_ctx* c_, double* acc_, uint64_t* l_) : c(c_), acc(acc_), launches(l_)
    {
        if (!c->cfg.profile) return;
        if (c->ev_used + 2 > c->ev_pool.size()) {
            hipEvent_t e0 = nullptr, e1 = nullptr;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            c->ev_pool.push_back(e0); c->ev_pool.push_back(e1);
        }
        a = c->ev_pool[c->ev_used++]; b = c->ev_pool[c->ev_used++];
        (void)hipEventRecord(a, c->stream);
    }
    void stop()
    {
        if (!c->cfg.profile) return;
        (void)hipEventRecord(b, c->stream);
        c->pending.push_back({a, b, acc, launches});
    }
};

void resolve_timers(lama_hip_ctx* c)      // call after the stream has been synchronised
{
    for (auto& t : c->pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { *t.acc += ms; *t.launches += 1; }
    }
    c->pending.clear();
    c->ev_used = 0;
    c->ctr.ms_update_maps = c->ctr.ms_raycast + c->ctr.ms_brushfire;
    c->ctr.launches_update_maps = c->ctr.launches_raycast;
}

// The reference's maps have no extent (src/sdm/map.cpp:400-411); the device window has one, but it MOVES and (round 4) GROWS.  The
// host keeps the box of everything that can be mapped so far -- the union, over all updates, of pose +- scan reach of every
// particle, in absolute patch coordinates (a superset of the mapped area; imported particles bring their sender's).  Before a map
// update (and before an import) the window must hold that box:
//   * it fits the window side: the window is moved by the smallest patch-granular step that holds it (plus a little slack in the
//     direction of the move) -- a permutation of the two directories of every particle (k_shift_window; the other particle set's
//     directories are the scratch destination, then the pointers are swapped);
//   * it does not: the directories are re-allocated with a larger side (at least half as large again, a multiple of 8, at most
//     LAMA_HIP_MAX_WINDOW = 1016 patches = 1.6 km at 0.05 m) and re-tiled by the same kernel; arenas, cells and slots are untouched.
// Only a map wider than 1016 patches remains LAMA_HIP_E_WINDOW (the kernels report what really falls outside).
constexpr int64_t LAMA_HIP_MAX_WINDOW = 1016;

int32_t regrid_window(lama_hip_ctx* c, uint32_t newW, int64_t nox, int64_t noy)
{
    const int64_t ox = c->wx0 >> 5, oy = c->wy0 >> 5;
    const int dx = (int)(nox - ox), dy = (int)(noy - oy);
    const size_t WW = (size_t)c->W * c->W, nWW = (size_t)newW * newW, P = c->P;
    if (!c->initialised && newW == c->W) { c->wx0 = (uint32_t)(nox * 32); c->wy0 = (uint32_t)(noy * 32); return LAMA_HIP_OK; }   // nothing mapped yet
    // both directories of every home are re-tiled into fresh arrays (same side: a permutation; larger side: the window grew), then the
    // old arrays are released.  A failed allocation leaves the window as it was (nothing leaks).
    MapStore& m = c->ms;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && 2 * P * nWW * 2 + (1ull << 28) > free_b)
            return fail(c, LAMA_HIP_E_WINDOW, "not enough device memory to re-tile the window directories (" + std::to_string(2 * P * nWW * 2 >> 20) + " MB for a " + std::to_string(newW) + "-patch window)");
    }
    int16_t* nd[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) {
        hipError_t e = hipMalloc(&nd[k], P * nWW * 2);
        if (e == hipSuccess && newW != c->W) e = hipMemsetAsync(nd[k], 0xFF, P * nWW * 2, c->stream);
        if (e != hipSuccess) { (void)hipFree(nd[0]); (void)hipFree(nd[1]); c->error = std::string("window directories: ") + hipGetErrorString(e); return LAMA_HIP_E_HIP; }
    }
    const dim3 grid((unsigned)P, (unsigned)((std::max(WW, nWW) + 255) / 256));
    hipLaunchKernelGGL(k_shift_window, grid, dim3(256), 0, c->stream, m.dm_dir, nd[0], c->W, newW, dx, dy, WW, nWW, c->d_err);
    hipLaunchKernelGGL(k_shift_window, grid, dim3(256), 0, c->stream, m.occ_dir, nd[1], c->W, newW, dx, dy, WW, nWW, c->d_err);
    {
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) { (void)hipFree(nd[0]); (void)hipFree(nd[1]); c->error = std::string("k_shift_window: ") + hipGetErrorString(e); return LAMA_HIP_E_HIP; }
    }
    (void)hipFree(m.dm_dir); (void)hipFree(m.occ_dir);
    m.dm_dir = nd[0]; m.occ_dir = nd[1];
    if (newW != c->W) {
        c->W = newW; c->WC = newW * 32; c->cfg.window_patches = newW;
        c->ctr.window_growths += 1;
    }
    c->wx0 = (uint32_t)(nox * 32); c->wy0 = (uint32_t)(noy * 32);
    c->ctr.window_shifts += 1;
    c->ctr.window_patches = c->W;
    return LAMA_HIP_OK;
}

// the window must hold the box [x0, x1] x [y0, y1] (absolute patches): union it into the mapped box, then move / grow the window
int32_t ensure_window(lama_hip_ctx* c, int64_t x0, int64_t x1, int64_t y0, int64_t y1)
{
    if (!c->mb_valid) { c->mbx0 = x0; c->mbx1 = x1; c->mby0 = y0; c->mby1 = y1; c->mb_valid = true; }
    else { c->mbx0 = std::min(c->mbx0, x0); c->mbx1 = std::max(c->mbx1, x1); c->mby0 = std::min(c->mby0, y0); c->mby1 = std::max(c->mby1, y1); }
    const int64_t ox = c->wx0 >> 5, oy = c->wy0 >> 5, W = c->W;
    if (c->debug_window) std::fprintf(stderr, "ensure_window: box x [%ld, %ld] y [%ld, %ld] mapped x [%ld, %ld] y [%ld, %ld] window x [%ld, %ld) y [%ld, %ld)\n", (long)x0, (long)x1, (long)y0, (long)y1, (long)c->mbx0, (long)c->mbx1, (long)c->mby0, (long)c->mby1, (long)ox, (long)(ox + W), (long)oy, (long)(oy + W));
    if (c->mbx0 >= ox && c->mbx1 < ox + W && c->mby0 >= oy && c->mby1 < oy + W) return LAMA_HIP_OK;
    const int64_t span = std::max(c->mbx1 - c->mbx0 + 1, c->mby1 - c->mby0 + 1);
    int64_t newW = W;
    if (span > W) newW = std::min<int64_t>(LAMA_HIP_MAX_WINDOW, std::max<int64_t>((span + 8 + 7) / 8 * 8, (W + W / 2 + 7) / 8 * 8));
    // what the (new) window must hold: the whole mapped box when it fits, else at least this call's box (the kernels then report
    // what really falls outside: LAMA_HIP_E_WINDOW)
    int64_t hx0 = c->mbx0, hx1 = c->mbx1, hy0 = c->mby0, hy1 = c->mby1;
    if (span > newW) { hx0 = x0; hx1 = x1; hy0 = y0; hy1 = y1; if (x1 - x0 + 1 > newW || y1 - y0 + 1 > newW) return LAMA_HIP_OK; }
    auto place = [&](int64_t o, int64_t lo, int64_t hi) {
        const int64_t a = hi - newW + 1, b = lo;              // admissible origins: a <= origin <= b
        int64_t no = std::min(std::max(o, a), b);
        if (no > o) no = std::min(no + 4, b); else if (no < o) no = std::max(no - 4, a);
        return no;
    };
    const int64_t nox = c->initialised ? place(ox, hx0, hx1) : (hx0 + hx1 + 1) / 2 - newW / 2;
    const int64_t noy = c->initialised ? place(oy, hy0, hy1) : (hy0 + hy1 + 1) / 2 - newW / 2;
    return regrid_window(c, (uint32_t)newW, nox, noy);
}

int32_t fit_window(lama_hip_ctx* c, const Affine& mtf, uint32_t first, uint32_t count)
{
    // no cell further than truncated_range from the sensor is touched (src/pf_slam2d.cpp:470-476): "no return" readings far
    // beyond it must not push mapped patches out of the window
    double far = c->scan_reach;
    if (c->cfg.ray_rule == 0 && c->cfg.truncated_range > 0.0) far = std::min(far, c->cfg.truncated_range);
    // padding: the brushfire allocates distance-map patches up to guard_r patches beyond a hit (make_params: ceil(l2_max / resolution)
    // + 1 cells), plus one patch for the rounding of the box.  permute_particles copies only the directory rows inside this box, so a
    // patch outside it would be lost by a clone (a fixed 2 patches were too few for l2_max above 64 cells).
    const uint32_t guard_r = ((uint32_t)std::ceil(std::sqrt((double)c->max_sqdist)) + 1u + 31u) / 32u;
    const double reach = far + std::sqrt(mtf.t[0] * mtf.t[0] + mtf.t[1] * mtf.t[1]) + (double)(guard_r + 1u) * 32.0 * c->cfg.resolution;
    double xlo = 1e300, xhi = -1e300, ylo = 1e300, yhi = -1e300;
    for (uint32_t p = first; p < first + count; ++p) {
        const double x = c->h_poses[4 * p + 2], y = c->h_poses[4 * p + 3];
        xlo = std::min(xlo, x - reach); xhi = std::max(xhi, x + reach);
        ylo = std::min(ylo, y - reach); yhi = std::max(yhi, y + reach);
    }
    auto patch = [&](double w) { return (int64_t)std::floor((c->scale * w + c->off) / 32.0); };
    return ensure_window(c, patch(xlo), patch(xhi), patch(ylo), patch(yhi));
}

// The dispatcher hands workgroup number b of a launch to XCD b mod 8, and every XCD has its own L2.  A one-dimensional launch with a
// workgroup per particle therefore keeps particle p on XCD p mod 8 in every kernel of an update; a two-dimensional one (particle,
// something) does so only when its particle dimension is a multiple of 8 -- else the workgroups of one particle are spread over all
// XCDs and the brushfire finds none of its cells in the L2 its ray-cast wrote them through.  Launches over the whole pool round the
// particle dimension up (the kernels skip particles >= P).
static unsigned xcd_grid(const lama_hip_ctx* c, uint32_t first, uint32_t count)
{
    return (first == 0 && count == c->P) ? (count + 7u) / 8u * 8u : count;
}

// allocation phase of a map update (ray records, hit cells' patches, the patches the rays cross, the bound on the distance-map patches
// still to come): afterwards either every patch the update needs exists or an error bit is set and no map cell has been modified
int32_t launch_allocation_phase(lama_hip_ctx* c, const DevParams& prm, uint32_t n, uint32_t first, uint32_t count, int alloc_only)
{
    const size_t need = (size_t)c->P * n;
    if (need > c->rrec_cap) {
        (void)hipFree(c->d_rrec); (void)hipFree(c->d_rbbox); (void)hipFree(c->d_rchunk); c->d_rrec = nullptr; c->d_rbbox = nullptr; c->d_rchunk = nullptr; c->rrec_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_rrec, need * sizeof(lama_dev::RayRec)));
        HIPCHK(c, hipMalloc(&c->d_rbbox, need * sizeof(uint64_t)));
        HIPCHK(c, hipMalloc(&c->d_rchunk, (size_t)c->P * ((n + 63) / 64) * sizeof(lama_dev::RayChunk)));     // per 64 beams of a particle
        c->rrec_cap = need;
    }
    hipLaunchKernelGGL(k_ray_hits, dim3(xcd_grid(c, first, count), (n + 255) / 256), dim3(256), 0, c->stream, prm, c->d_pts, (int)n, c->d_tfs, (int)first,
                       c->d_rrec, c->d_rbbox, alloc_only, c->d_rchunk);
    const int rw_seg = count <= 64 ? 4 : 1;
    hipLaunchKernelGGL(k_ray_alloc_walk, dim3(xcd_grid(c, first, count), (n * rw_seg + 255) / 256), dim3(256), 0, c->stream, prm, (const lama_dev::RayRec*)c->d_rrec, (int)n, (int)first, rw_seg);
    double far = c->scan_reach;                                   // no cell further than truncated_range from the sensor is touched
    if (c->cfg.truncated_range > 0.0) far = std::min(far, c->cfg.truncated_range);
    const int reach_cells = (int)std::ceil(far * c->scale) + 2;
    hipLaunchKernelGGL(k_occ_reverse_dir, dim3(count), dim3(256), 0, c->stream, prm, (int)first, (const double*)c->d_tfs, reach_cells,
                       c->unguarded_retry ? 1 : 0);
    return LAMA_HIP_OK;
}

// how many particles may go to the big-queue stage beside the first one: an 84 KB workgroup takes the LDS of seven first-stage
// workgroups, and a first stage that no longer fits the chip in one round (12 per CU) costs far more than the routing saves
static uint32_t route_places(const lama_hip_ctx* c, uint32_t count)
{
    const uint32_t slots = c->num_cus * 12u;
    if (c->route_forced || count >= slots) return c->route_cap;                 // (several rounds anyway)
    return std::min<uint32_t>(c->route_cap, std::max<uint32_t>(8u, (slots - count) / 7u));
}

int32_t run_update_maps(lama_hip_ctx* c, uint32_t n, const Affine& mtf, uint32_t first, uint32_t count)
{
    {
        const int32_t rcw = fit_window(c, mtf, first, count);
        if (rcw) return rcw;
    }
    c->last_mtf = mtf; c->last_first = first; c->last_count = count;
    if (c->recover_depth == 0) { c->saved_visit_bound = c->visit_bound; c->saved_ctr = c->ctr; }
    PinVec<double>& tfs = c->h_tfs;
    tfs.resize((size_t)c->P * 12);
    for (uint32_t p = 0; p < c->P; ++p) host_scan_tf(&c->h_poses[4 * p], mtf, &tfs[12 * (size_t)p]);
    HIPCHK(c, hipMemcpyAsync(c->d_tfs, tfs.data(), sizeof(double) * tfs.size(), hipMemcpyHostToDevice, c->stream));
    DevParams prm = make_params(c);
    bool early_lane = false;
    {
        Timer t(c, &c->ctr.ms_raycast, &c->ctr.launches_raycast);
        // both ray-casts are bit-exact; the parallel one wins while the chip is not yet full of particles
        (void)hipMemsetAsync(c->d_slow_n, 0, 32 + sizeof(uint64_t) * 4 * c->P, c->stream);      // hand-over counts and statistics: adjacent (ctx_create)
        bool sequential = c->cfg.sequential_raycast == 1 || c->cfg.occupancy_policy == 1 ||
                          c->cfg.ray_rule == 1;      // the parallel kernels implement the frequency counters and the PF ray rule only
        // every hit is an order-sensitive visit (the list holds active_capacity of them) and the visit key carries the beam index
        // in 13 bits (act_key) and k_ray_patches keeps one 64-beam chunk per lane: scans with more than 4096 points go beam by beam
        if (!sequential && (n > 4096u || (uint64_t)n + 4096u > c->cfg.active_capacity)) sequential = true;
        // a uint16 `visited` counter that wraps INSIDE a scan makes the order of the visits matter (k_occ_max_visited)
        if (!sequential && (uint64_t)c->visit_bound + n >= 65536u) {
            uint32_t m = 0;
            if (c->initialised) {
                (void)hipMemsetAsync(c->d_scalar, 0, sizeof(uint32_t), c->stream);
                hipLaunchKernelGGL(k_occ_max_visited, dim3(c->P), dim3(256), 0, c->stream, prm, c->d_scalar);
                HIPCHK(c, hipMemcpyAsync(&m, c->d_scalar, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
                HIPCHK(c, hipStreamSynchronize(c->stream));
            }
            c->visit_bound = m;
            if ((uint64_t)m + n >= 65536u) { sequential = true; c->ctr.wrap_guard_scans += 1; }
        }
        c->visit_bound = (uint32_t)std::min<uint64_t>((uint64_t)c->visit_bound + n, 65535u);
        if (sequential) c->ctr.sequential_raycast_scans += 1; else c->ctr.parallel_raycast_scans += 1;
        // The patches an update needs are allocated BEFORE any cell is modified, so that running out of patches is recoverable
        // (grow + run again, recover_update).  The allocation walk uses the PF ray rule; LidarOdometry's rule / log-odds cells
        // (one-particle contexts) get a conservative pre-grown arena instead: every patch a scan of this reach can touch.
        c->last_guarded = c->cfg.ray_rule == 0;
        if (!c->last_guarded) {
            const double side = 32.0 * c->cfg.resolution;
            const double rp = c->scan_reach / side + 2.5;
            const uint32_t bound = (uint32_t)std::min(32767.0, std::ceil(3.1416 * rp * rp));
            std::vector<CapRequest> reqs;
            for (uint32_t p = first; p < first + count; ++p) {
                const HostPart& pr = c->h_part[p];
                const uint32_t nd = std::min<uint32_t>(MAX_PATCHES, round_up32((uint32_t)c->h_counts[2 * p] + bound)), no = std::min<uint32_t>(MAX_PATCHES, round_up32((uint32_t)c->h_counts[2 * p + 1] + bound));
                if (nd > pr.dm_cap || no > pr.occ_cap) reqs.push_back(CapRequest{p, std::max(nd, pr.dm_cap), std::max(no, pr.occ_cap)});
            }
            if (!reqs.empty()) {
                const int32_t rg = set_capacities(c, reqs);
                if (rg) return rg;
                prm = make_params(c);
            }
        }
        if (sequential) {
            if (c->last_guarded) { const int32_t ra = launch_allocation_phase(c, prm, n, first, count, 1); if (ra) return ra; }
            hipLaunchKernelGGL(k_raycast, dim3(count), dim3(UM_BLOCK), 0, c->stream, prm, c->d_pts, (int)n, c->d_tfs, (int)first);
        } else {
            // patch-centric visits: a workgroup owns one occupancy patch of one particle, no global atomics on the counters
            // Early lane: the particles the previous update routed (the long brushfire chains; usually long again) get their modifying
            // ray-cast kernels and their brushfire on a stream of their own right after the allocation phase -- their chain, which is
            // what the update waits for, no longer starts behind everybody else's ray-cast.
            const bool two_waves_e = c->cfg.brushfire_waves == 2 || (c->cfg.brushfire_waves == 0 && count <= BF_TW_MAX_PARTICLES);
            early_lane = c->early_on && two_waves_e && c->cfg.brushfire_mode == 0 && first == 0 && count == c->P &&
                         count >= std::max<uint32_t>(c->route_min_count, c->early_min_count) && c->route_cap > 0;
            uint32_t host_n = 0;
            if (early_lane && c->ll_valid) {
                // the particles whose scan match of this step fitted far worse than the pool's (log-likelihood below 2.5 times the
                // mean, both negative): theirs are the long chains of this update -- known before anything is queued
                const uint32_t places = route_places(c, count);
                double mean = 0; for (uint32_t p = 0; p < c->P; ++p) mean += c->h_ll[p]; mean /= c->P;
                const double lim = 2.5 * std::min(mean, 0.0) - 5.0;
                std::vector<std::pair<double, uint32_t>> cand;
                for (uint32_t p = 0; p < c->P; ++p) if (c->h_ll[p] < lim) cand.emplace_back(c->h_ll[p], p);
                if (cand.size() > places) { std::partial_sort(cand.begin(), cand.begin() + places, cand.end()); cand.resize(places); }
                c->h_hlist.resize(256);
                for (auto& e : cand) c->h_hlist[host_n++] = e.second;
                if (host_n) HIPCHK(c, hipMemcpyAsync(c->d_hlist, c->h_hlist.data(), sizeof(uint32_t) * host_n, hipMemcpyHostToDevice, c->stream));
            }
            const bool prev_ok = c->early_ok && c->early_candidates > 0;
            early_lane = early_lane && (host_n > 0 || prev_ok);
            if (early_lane)
                hipLaunchKernelGGL(k_early_list, dim3(1), dim3(256), 0, c->stream, prev_ok ? (const uint8_t*)c->d_heavy : (const uint8_t*)nullptr,
                                   (const uint32_t*)c->d_hlist, host_n, c->d_early, c->d_elist, c->d_slow_n + 4, (int)c->P, route_places(c, count));
            { const int32_t ra = launch_allocation_phase(c, prm, n, first, count, 0); if (ra) return ra; }
            // workgroups per particle = patches of a particle in flight at once.  A chip full of particles needs no more parallelism
            // inside one: 8 instead of 32 workgroups per particle (each then walks ~8 of its ~65 patches behind ONE prologue) took the
            // ray-cast of the 3000-particle pool from 1.31 to 1.24 ms (round 6); 4 and 16 are within noise of it, 64 is slower
#ifdef LAMA_WAVE_SIM
            const unsigned gy = 4u;                              // (tests/sim runs every thread as a fiber: 128 mostly idle workgroups per particle cost it minutes)
#else
            const unsigned gy = count <= 64 ? 128u : (count >= 1024 ? 8u : 32u);
#endif
            if (early_lane) {
                // The early lane's own ray-cast kernels run HERE, on an otherwise idle chip (~0.1 ms for a few particles); its
                // brushfire goes to a stream of its own.  The main lane's 96 k-workgroup launches must not reach the dispatcher before
                // the early lane's big-queue workgroups (84 KB of LDS each) are placed -- a saturated chip refills every freed slot
                // with the saturating kernel's next workgroup and starves them until it has drained -- so the main lane waits for an
                // event that the early stream records directly in front of its brushfire kernel.
                DevParams pe = prm;
                pe.elist = c->d_elist; pe.elist_n = c->d_slow_n + 4; pe.lane = 1;
                const unsigned eg = route_places(c, count);
                hipLaunchKernelGGL(k_ray_patches, dim3(eg, 128u), dim3(256), 0, c->stream, pe, (const lama_dev::RayRec*)c->d_rrec,
                                   (const uint64_t*)c->d_rbbox, (const lama_dev::RayChunk*)c->d_rchunk, (int)n, 0);
                hipLaunchKernelGGL((k_ray_replay<RP_SORT_SMALL, RP_SORT_SMALL, false, RP_BLOCK_LARGE>), dim3(eg), dim3(RP_BLOCK_LARGE), 0, c->stream, pe, 0);
                hipLaunchKernelGGL((k_ray_replay<8192, 8192, true, RP_BLOCK_LARGE>), dim3(eg), dim3(RP_BLOCK_LARGE), 0, c->stream, pe, 0);
                HIPCHK(c, hipEventRecord(c->ev_alloc, c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream3, c->ev_alloc, 0));
                hipLaunchKernelGGL(k_mark_early, dim3(1), dim3(64), 0, c->stream3, pe);
                HIPCHK(c, hipEventRecord(c->ev_go, c->stream3));
                hipLaunchKernelGGL((k_brushfire<LQ_BIG, RQ_BIG, true, true>), dim3(eg), dim3(2 * UM_BLOCK), c->big_stage_pad, c->stream3, pe, 0, 2);
                HIPCHK(c, hipEventRecord(c->ev_early, c->stream3));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_go, 0));
                prm.early = c->d_early;                          // everybody else: the main lane skips them
            }
            hipLaunchKernelGGL(k_ray_patches, dim3(xcd_grid(c, first, count), gy), dim3(256), 0, c->stream, prm, (const lama_dev::RayRec*)c->d_rrec,
                               (const uint64_t*)c->d_rbbox, (const lama_dev::RayChunk*)c->d_rchunk, (int)n, (int)first);
            const unsigned resume_grid = std::min<unsigned>(count, 256u);       // walks the (usually empty) hand-over list
            if (count <= 512) {
                hipLaunchKernelGGL((k_ray_replay<RP_SORT_SMALL, RP_SORT_SMALL, false, RP_BLOCK_LARGE>), dim3(count), dim3(RP_BLOCK_LARGE), 0, c->stream, prm, (int)first);
                hipLaunchKernelGGL((k_ray_replay<8192, 8192, true, RP_BLOCK_LARGE>), dim3(resume_grid), dim3(RP_BLOCK_LARGE), 0, c->stream, prm, (int)first);
            } else {
                hipLaunchKernelGGL((k_ray_replay<RP_SORT_SMALL, RP_SORT_SMALL, false, RP_BLOCK_SMALL>), dim3(count), dim3(RP_BLOCK_SMALL), 0, c->stream, prm, (int)first);
                hipLaunchKernelGGL((k_ray_replay<8192, 8192, true, RP_BLOCK_SMALL>), dim3(resume_grid), dim3(RP_BLOCK_SMALL), 0, c->stream, prm, (int)first);
            }
        }
        t.stop();
    }
    HIPCHK(c, hipGetLastError());
    {
        Timer t(c, &c->ctr.ms_brushfire, &c->ctr.launches_brushfire);
        if (c->cfg.brushfire_mode == 1)     // opt-in parallel variant; whatever it leaves (qsizes != 0) falls through to the exact stages
            hipLaunchKernelGGL(k_brushfire_canon, dim3(count), dim3(CN_BLOCK), 0, c->stream, prm, (int)first);
        // stage 1: small LDS window, every particle; stage 2: big window, resumes particles whose queue outgrew stage 1
        // (e.g. the first scan); stage 3: generic HBM-queue kernel for anything larger still
        // few particles: CUs are idle, spend a helper wave per particle on the heap (see k_brushfire, TW)
        const bool two_waves = c->cfg.brushfire_waves == 2 || (c->cfg.brushfire_waves == 0 && count <= BF_TW_MAX_PARTICLES);
        c->ctr.brushfire_mode = c->cfg.brushfire_mode;
        c->ctr.brushfire_waves = two_waves ? 2u : 1u;
        const unsigned resume_grid = std::min<unsigned>(count, 256u);      // workgroups that walk the hand-over list (usually empty)
        // the long chains are known before they start (k_bf_route): they go to the big-queue stage on a second stream, beside the
        // first stage (which skips them), instead of outgrowing it half-way and waiting for it to end
        const bool route = two_waves && count >= c->route_min_count && c->route_cap > 0;
        if (two_waves) {
            if (route) {
                // The big-queue workgroups (84 KB of LDS each) must be placed BEFORE the first stage fills every CU's LDS, or they wait
                // for a third of it to finish: their kernel follows k_bf_route directly in this stream, the first stage goes to the
                // second stream behind an event -- the cross-queue signal is what makes it the later dispatch.
                prm.heavy = c->d_heavy;
                hipLaunchKernelGGL(k_bf_route, dim3(1), dim3(256), 0, c->stream, prm, (int)first, (int)count, c->route_min_events, c->route_percent, route_places(c, count));
                HIPCHK(c, hipEventRecord(c->ev_route, c->stream));
                hipLaunchKernelGGL((k_brushfire<LQ_BIG, RQ_BIG, true, true>), dim3(route_places(c, count)), dim3(2 * UM_BLOCK), 0, c->stream, prm, (int)first, 1);
                HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_route, 0));
                // (round 6 measured the one-wave straight-line form here for a chip full of particles -- fewer instructions per pop in
                // total, no barrier: brushfire 4.90 ms against 4.08 ms for the pair at 3000 particles, 3.0 against 1.86 ms at 300.  The
                // pair stays; the one-wave form is what cfg.brushfire_waves = 1 selects.)
                hipLaunchKernelGGL((k_brushfire<LQ_SMALL, RQ_SMALL, false, true>), dim3(count), dim3(2 * UM_BLOCK), 0, c->stream2, prm, (int)first, 0);
                HIPCHK(c, hipEventRecord(c->ev_heavy, c->stream2));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_heavy, 0));
            } else {
                hipLaunchKernelGGL((k_brushfire<LQ_SMALL, RQ_SMALL, false, true>), dim3(count), dim3(2 * UM_BLOCK), 0, c->stream, prm, (int)first, 0);
            }
            hipLaunchKernelGGL((k_brushfire<LQ_BIG, RQ_BIG, true, true>), dim3(resume_grid), dim3(2 * UM_BLOCK), 0, c->stream, prm, (int)first, 0);
        } else {
            hipLaunchKernelGGL((k_brushfire<LQ_SMALL, RQ_SMALL, false, false>), dim3(count), dim3(UM_BLOCK), 0, c->stream, prm, (int)first, 0);
            hipLaunchKernelGGL((k_brushfire<LQ_BIG, RQ_BIG, true, false>), dim3(resume_grid), dim3(UM_BLOCK), 0, c->stream, prm, (int)first, 0);
        }
        if (early_lane) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_early, 0));
        hipLaunchKernelGGL(k_brushfire_slow, dim3(count), dim3(UM_BLOCK), 0, c->stream, prm, (int)first);
        c->early_ok = route && first == 0 && count == c->P;      // d_heavy now holds this update's routed particles (early lane included)
        t.stop();
    }
    HIPCHK(c, hipGetLastError());
    return LAMA_HIP_OK;
}


// new particle i := old particle idx[i] (pose + both maps), IN PLACE (resample(), src/pf_slam2d.cpp:558-574; the first scan's
// copies, :204-216).  The first new particle that draws an old one takes over its home and regions -- nothing moves --, every further
// copy of it goes to the home of a particle nobody drew; its regions are reused when they have the source's capacities, else they
// go back to the allocator (zeroed where used) and fresh ones are taken.  What the device copies is exactly the clones.
int32_t permute_particles(lama_hip_ctx* c, const int32_t* idx)
{
    const uint32_t P = c->P;
    std::vector<uint8_t> taken(P, 0);
    std::vector<HostPart> np(P);
    std::vector<int32_t> nc(2 * (size_t)P);
    std::vector<double> npose(4 * (size_t)P);
    std::vector<uint32_t> clones;                                   // new particles that need a copy
    for (uint32_t i = 0; i < P; ++i) {
        const uint32_t j = (uint32_t)idx[i];
        std::memcpy(&npose[4 * i], &c->h_poses[4 * j], sizeof(double) * 4);
        nc[2 * i] = c->h_counts[2 * j]; nc[2 * i + 1] = c->h_counts[2 * j + 1];
        if (!taken[j]) { taken[j] = 1; np[i] = c->h_part[j]; } else clones.push_back(i);
    }
    std::vector<uint32_t> dead;                                     // old particles nobody drew: their homes / regions are free
    for (uint32_t j = 0; j < P; ++j) if (!taken[j]) dead.push_back(j);
    c->h_jobs.resize(0); c->h_zjobs.resize(0);
    c->clone_bytes = 0;
    const size_t WW = (size_t)c->W * c->W;
    // the directory rows that can hold a patch at all: the mapped box of the context (a superset of every particle's map, ensure_window);
    // W is a multiple of 8, so a row is a whole number of 16-byte units
    uint32_t dir_off16 = 0, dir_n16 = (uint32_t)(WW * 2 / 16);
    {
        const int64_t oy = c->wy0 >> 5, W = c->W;                   // (the rows of mapped_box_rel(c, 1))
        if (c->mb_valid) {
            const int64_t lo = std::min(std::max(c->mby0 - oy, (int64_t)0), W - 1), hi = std::min(std::max(c->mby1 - oy, (int64_t)0), W - 1);
            if (hi >= lo) { dir_off16 = (uint32_t)(lo * W * 2 / 16); dir_n16 = (uint32_t)((hi - lo + 1) * W * 2 / 16); }
        }
    }
    // first pass: dead particles whose regions cannot be reused as they are go back to the allocator (so that the clones can have them)
    // -- a dead particle is matched with the clone of the same rank; capacities are mostly equal (particles of one filter map the
    // same world), so the common case is a plain overwrite
    struct Slot { bool keep_dm, keep_occ; HostPart rec; int32_t odm, oocc; };
    std::vector<Slot> slots(dead.size());
    for (size_t k = 0; k < dead.size(); ++k) {
        const uint32_t d = dead[k], i = clones[k], j = (uint32_t)idx[i];
        const HostPart& dr = c->h_part[d];
        const HostPart& sr = c->h_part[j];
        Slot sl{dr.dm_cap == sr.dm_cap, dr.occ_cap == sr.occ_cap, dr, c->h_counts[2 * d], c->h_counts[2 * d + 1]};
        void* dp[5];
        region_ptrs(c, dr, dp);
        ZeroJob z{};
        if (!sl.keep_dm) { z.ndm = sl.odm; z.d[0] = dp[0]; z.d[1] = dp[1]; z.d[2] = dp[2]; region_release(c, true, dr.dm_chunk, dr.dm_off, dr.dm_cap); }
        if (!sl.keep_occ) { z.nocc = sl.oocc; z.d[3] = dp[3]; z.d[4] = dp[4]; region_release(c, false, dr.occ_chunk, dr.occ_off, dr.occ_cap); }
        if (z.ndm || z.nocc) { c->h_zjobs.resize(c->h_zjobs.size() + 1); c->h_zjobs[c->h_zjobs.size() - 1] = z; }
        slots[k] = sl;
    }
    for (size_t k = 0; k < dead.size(); ++k) {
        const uint32_t i = clones[k], j = (uint32_t)idx[i];
        const HostPart sr = c->h_part[j];
        Slot& sl = slots[k];
        HostPart r = sl.rec;                                        // the dead particle's home; its regions where they are kept
        r.dm_cap = sr.dm_cap; r.occ_cap = sr.occ_cap;
        CloneJob job{};
        job.src_home = sr.home; job.dst_home = r.home;
        job.sdm = c->h_counts[2 * j]; job.socc = c->h_counts[2 * j + 1];
        // (a region of the capacity a dead particle of another size has just given back: this can only fail when the device is out of
        // memory for another chunk.  The particle table is then half permuted: the context refuses further work.)
        if (sl.keep_dm) job.odm = sl.odm;
        else { const int32_t rc = region_alloc(c, true, sr.dm_cap, r.dm_chunk, r.dm_off); if (rc) { c->poisoned = true; return rc; } job.odm = 0; }
        if (sl.keep_occ) job.oocc = sl.oocc;
        else { const int32_t rc = region_alloc(c, false, sr.occ_cap, r.occ_chunk, r.occ_off); if (rc) { c->poisoned = true; return rc; } job.oocc = 0; }
        void* sp[5]; void* dp[5];
        region_ptrs(c, sr, sp); region_ptrs(c, r, dp);
        for (int q = 0; q < 5; ++q) { job.s[q] = sp[q]; job.d[q] = dp[q]; }
        job.dir_off16 = dir_off16; job.dir_n16 = dir_n16;
        np[i] = r;
        c->h_jobs.resize(c->h_jobs.size() + 1); c->h_jobs[c->h_jobs.size() - 1] = job;
        c->clone_bytes += 2 * (uint64_t)dir_n16 * 16 + (uint64_t)job.sdm * (SV_PATCH_BYTES + 4096 + 128) + (uint64_t)job.socc * (4096 + 128);
    }
    c->h_part = np;
    std::memcpy(c->h_poses.data(), npose.data(), npose.size() * sizeof(double));
    std::memcpy(c->h_counts.data(), nc.data(), nc.size() * sizeof(int32_t));
    int32_t rc = upload_part(c);                                    // (the jobs carry absolute bases; the table is for everybody after them)
    if (rc) return rc;
    {
        Timer t(c, &c->ctr.ms_resample, &c->ctr.launches_resample);
        rc = run_jobs(c, true);
        t.stop();
    }
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->ms.counts, c->h_counts.data(), sizeof(int32_t) * 2 * P, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_poses, c->h_poses.data(), sizeof(double) * 4 * P, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_timers(c);
    c->ctr.resample_clones += clones.size();
    c->ctr.resample_bytes += c->clone_bytes;
    return LAMA_HIP_OK;
}

} // namespace

extern "C" {

void lama_hip_default_cfg(lama_hip_cfg* cfg)
{
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->particles = 30;
    cfg->resolution = 0.05;
    cfg->patch_size = 32;
    cfg->l2_max = 0.5;
    cfg->meas_sigma = 0.05;
    cfg->max_iter = 100;
    cfg->device = 0;
    cfg->window_patches = 128;
    cfg->dm_patch_capacity = 128;
    cfg->occ_patch_capacity = 128;
    cfg->queue_capacity = 32768;
    cfg->active_capacity = 8192;
}

int32_t lama_hip_device_count(int32_t* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (count) *count = (e == hipSuccess) ? n : 0;
    return e == hipSuccess ? LAMA_HIP_OK : LAMA_HIP_E_HIP;
}

const char* lama_hip_last_error(const lama_hip_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }

int32_t lama_hip_ctx_create(const lama_hip_cfg* cfg_in, lama_hip_ctx** out)
{
    if (!cfg_in || !out) return LAMA_HIP_E_INVALID;
    *out = nullptr;
    lama_hip_cfg cfg = *cfg_in;
    if (cfg.window_patches == 0) cfg.window_patches = 128;
    if (cfg.dm_patch_capacity == 0) cfg.dm_patch_capacity = 128;
    if (cfg.occ_patch_capacity == 0) cfg.occ_patch_capacity = 128;
    if (cfg.queue_capacity == 0) cfg.queue_capacity = 32768;
    if (cfg.active_capacity == 0) cfg.active_capacity = 8192;
    if (cfg.active_capacity > 8192) cfg.active_capacity = 8192;      // largest k_ray_replay stage
    // No environment variable changes what a context computes or which kernels it runs: the variants are chosen through the
    // configuration only (cfg.brushfire_mode = 1, the one variant that is NOT bit-identical to the reference, must be asked for
    // by the caller), and lama_hip_get_counters reports what actually ran.
    if (cfg.brushfire_mode > 1 || cfg.brushfire_waves > 2) return LAMA_HIP_E_INVALID;
    if (cfg.particles == 0 || cfg.patch_size != 32 || !(cfg.resolution > 0) || cfg.window_patches > (uint32_t)LAMA_HIP_MAX_WINDOW ||
        (cfg.window_patches & 7) || cfg.dm_patch_capacity > 32767 || cfg.occ_patch_capacity > 32767 ||
        cfg.queue_capacity < (uint32_t)LQ_BIG)
        return LAMA_HIP_E_INVALID;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || cfg.device < 0 || cfg.device >= ndev) return LAMA_HIP_E_HIP;

    lama_hip_ctx* c = new lama_hip_ctx();
    c->cfg = cfg;
    std::memset(&c->ctr, 0, sizeof(c->ctr));
    c->P = cfg.particles; c->W = cfg.window_patches; c->WC = c->W * 32;
    c->ctr.window_patches = c->W;
    // developer switches, read ONCE here (never on the path of an update): LAMA_HIP_DEBUG_TAIL / LAMA_HIP_DEBUG_WINDOW print what the
    // longest chain / the window did; LAMA_HIP_BF_ROUTE overrides the routing thresholds (tests, experiments)
    c->debug_tail = std::getenv("LAMA_HIP_DEBUG_TAIL") != nullptr;
    c->debug_window = std::getenv("LAMA_HIP_DEBUG_WINDOW") != nullptr;
    {   // A long chain that shares its CU with six first-stage workgroups runs at 0.79 us per pop, alone at 0.75 (and 0.65 on an idle
        // chip): the routed / early big-queue workgroups of a full chip ask for the REST of the CU's LDS as dynamic shared memory, so
        // nothing else is placed beside them (workgroups of that launch without a list entry exit at once and free their CU).
        // 3000 particles: brushfire 4.31 -> 4.10 ms, step 6.36 -> 6.17 ms (round 6); 120 KB instead of the whole CU: no gain.  Only the
        // EARLY lane's launch is padded -- the chains predicted to be the longest; padding the routed launch as well (up to 64 places
        // when many particles re-draw their walls, e.g. after every resample) takes more LDS than the first stage can spare and
        // measured 0.5 - 1 % slower.
        constexpr uint32_t CU_LDS = 160u * 1024u, GRAN = 1280u;
        const uint32_t used = (uint32_t)((sizeof(lama_dev::BfLds<LQ_BIG, RQ_BIG>) + GRAN - 1) / GRAN * GRAN);
        c->big_stage_pad = CU_LDS > used + 2u * GRAN ? CU_LDS - used - 2u * GRAN : 0u;
    }
    if (const char* rr = std::getenv("LAMA_HIP_BF_ROUTE")) {      // "min particles,min events,percent of the mean,places" (tests, experiments; places 0: off)
        unsigned a = 0, b = 0, pc = 0, d = 0, e = 1;
        if (std::sscanf(rr, "%u,%u,%u,%u,%u", &a, &b, &pc, &d, &e) >= 4) { c->route_min_count = a; c->route_min_events = b; c->route_percent = pc; c->route_cap = std::min(d, 256u); c->early_on = e; c->early_min_count = 0; c->route_forced = true; }
    }
    c->scale = 1.0 / cfg.resolution;
    c->off = double(2642244ull >> 1) * 32.0;                      // src/sdm/map.cpp:55-58
    // DynamicDistanceMap::setMaxDistance (src/sdm/dynamic_distance_map.cpp:149-153)
    uint32_t md = (uint32_t)std::ceil(cfg.l2_max * c->scale);
    c->max_sqdist = md * md;
    // this build's queue entries carry obstacle offsets up to MAX_OFFSET_CELLS and its distance plane that range squared: 127 cells
    // in liblama_hip.so, 255 -- all that the reference's uint16_t sqdist can hold -- in liblama_hip_wide.so (lama_dev.h)
    if (c->max_sqdist == 0 || md > MAX_OFFSET_CELLS) { delete c; return LAMA_HIP_E_INVALID; }

#define CHK(call) do { if ((call) != hipSuccess) { lama_hip_ctx_destroy(c); return LAMA_HIP_E_HIP; } } while (0)
    CHK(hipSetDevice(cfg.device));
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, cfg.device) == hipSuccess && prop.multiProcessorCount > 0) c->num_cus = (uint32_t)prop.multiProcessorCount; }
    CHK(hipStreamCreate(&c->stream));
    CHK(hipEventCreate(&c->ev0));
    CHK(hipEventCreate(&c->ev1));
    const size_t P = c->P, WW = (size_t)c->W * c->W, dc = cfg.dm_patch_capacity, oc = cfg.occ_patch_capacity;
    {
        // one particle set: directories per home, pooled planes with one region per particle and map kind (lama_dev.h).  The pools
        // start at P x the configured capacities and grow with the maps.
        MapStore& m = c->ms;
        c->floor_dm = (uint32_t)dc; c->floor_occ = (uint32_t)oc;
        if (P * dc > 0xFFFFFFF0ull || P * oc > 0xFFFFFFF0ull) { lama_hip_ctx_destroy(c); return LAMA_HIP_E_INVALID; }
        CHK(hipMalloc(&m.dm_dir, P * WW * 2));                 CHK(hipMemset(m.dm_dir, 0xFF, P * WW * 2));
        CHK(hipMalloc(&m.occ_dir, P * WW * 2));                CHK(hipMemset(m.occ_dir, 0xFF, P * WW * 2));
        // (m.counts lives in the results / status block below: one copy brings a map update's counters back)
        // first chunk of each pool: P regions of the configured capacity
        if (add_chunk(c, true, (uint32_t)(P * dc)) != LAMA_HIP_OK || add_chunk(c, false, (uint32_t)(P * oc)) != LAMA_HIP_OK) { lama_hip_ctx_destroy(c); return LAMA_HIP_E_HIP; }
        c->ctr.pool_growths = 0;
        c->h_part.resize(P);
        for (size_t p = 0; p < P; ++p) {
            HostPart r{};
            r.home = (uint32_t)p; r.dm_cap = (uint32_t)dc; r.occ_cap = (uint32_t)oc;
            if (region_alloc(c, true, (uint32_t)dc, r.dm_chunk, r.dm_off) != LAMA_HIP_OK || region_alloc(c, false, (uint32_t)oc, r.occ_chunk, r.occ_off) != LAMA_HIP_OK) { lama_hip_ctx_destroy(c); return LAMA_HIP_E_HIP; }
            c->h_part[p] = r;
        }
        CHK(hipMalloc(&c->d_part, P * sizeof(PartRec)));
        if (upload_part(c) != LAMA_HIP_OK) { lama_hip_ctx_destroy(c); return LAMA_HIP_E_HIP; }
        CHK(hipStreamSynchronize(c->stream));
    }
    // ONE allocation: [poses | loglik | iters | err]   what a scan match brings back (results_bytes, one copy) ...
    //                 [err | slow_n | stats | guard | counts]   ... and what a map update brings back (status_bytes from status_off, one
    // copy: round 6 -- the five copies it used to take were 23 us at the end of every 1.8 ms update); slow_n and stats are adjacent:
    // one memset clears both when an update starts.
    c->results_bytes = P * 4 * 8 + P * 8 + P * 4 + 8;
    c->status_off = P * 4 * 8 + P * 8 + P * 4;                                  // the error word: last of the results, first of the status
    const size_t slow_off = (c->results_bytes + 7) / 8 * 8;
    c->status_bytes = (slow_off - c->status_off) + 32 + P * 4 * 8 + P * 2 * 4 + P * 2 * 4;
    CHK(hipMalloc(&c->d_results, c->status_off + c->status_bytes));   CHK(hipMemset(c->d_results, 0, c->status_off + c->status_bytes));
    c->d_poses = reinterpret_cast<double*>(c->d_results);
    c->d_loglik = c->d_poses + P * 4;
    c->d_iters = reinterpret_cast<int32_t*>(c->d_loglik + P);
    c->d_err = c->d_iters + P;
    c->d_slow_n = reinterpret_cast<uint32_t*>(c->d_results + slow_off);
    c->d_stats = reinterpret_cast<uint64_t*>(c->d_results + slow_off + 32);
    c->d_guard = reinterpret_cast<uint32_t*>(c->d_stats + P * 4);
    c->ms.counts = reinterpret_cast<int32_t*>(c->d_guard + P * 2);
    CHK(hipMalloc(&c->d_qlower, P * (size_t)cfg.queue_capacity * 8));
    CHK(hipMalloc(&c->d_qraise, P * (size_t)cfg.queue_capacity * 8));
    CHK(hipMalloc(&c->d_qsizes, P * 2 * 4));         CHK(hipMemset(c->d_qsizes, 0, P * 2 * 4));
    CHK(hipMalloc(&c->d_dbg, P * 16 * 8 + (1u << 20)));   CHK(hipMemset(c->d_dbg, 0, P * 16 * 8 + (1u << 20)));   // + 1 MiB developer event log
    CHK(hipMalloc(&c->d_slow, P * 4));               CHK(hipMemset(c->d_slow, 0, P * 4));
    CHK(hipMalloc(&c->d_slow_list, 4 * P * 4));      CHK(hipMemset(c->d_slow_list, 0, 4 * P * 4));
    CHK(hipMalloc(&c->d_heavy, P));                  CHK(hipMemset(c->d_heavy, 0, P));
    CHK(hipMalloc(&c->d_early, P));                  CHK(hipMemset(c->d_early, 0, P));
    CHK(hipMalloc(&c->d_elist, 256 * 4));            CHK(hipMemset(c->d_elist, 0, 256 * 4));
    CHK(hipMalloc(&c->d_hlist, 256 * 4));            CHK(hipMemset(c->d_hlist, 0, 256 * 4));
    CHK(hipStreamCreate(&c->stream2));
    CHK(hipStreamCreate(&c->stream3));
    CHK(hipEventCreateWithFlags(&c->ev_alloc, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&c->ev_early, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&c->ev_go, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&c->ev_route, hipEventDisableTiming)); CHK(hipEventCreateWithFlags(&c->ev_heavy, hipEventDisableTiming));
    CHK(hipMalloc(&c->d_scalar, 16));                CHK(hipMemset(c->d_scalar, 0, 16));
    CHK(hipMalloc(&c->d_act, P * (size_t)cfg.active_capacity * 8));
    CHK(hipMalloc(&c->d_act_count, P * 4));          CHK(hipMemset(c->d_act_count, 0, P * 4));
    CHK(hipMalloc(&c->d_tfs, P * 12 * 8));
    CHK(hipMalloc(&c->d_oldcounts, P * 2 * 4));
    CHK(hipDeviceSynchronize());
#undef CHK
    c->h_poses.assign(P * 4, 0.0);
    c->h_counts.assign(P * 2, 0);
    *out = c;
    return LAMA_HIP_OK;
}

void lama_hip_ctx_destroy(lama_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    {
        MapStore& m = c->ms;
        (void)hipFree(m.dm_dir); (void)hipFree(m.occ_dir);
        for (PoolChunk& k : m.dm_chunks) for (int i = 0; i < 4; ++i) (void)hipFree(k.plane[i]);
        for (PoolChunk& k : m.occ_chunks) for (int i = 0; i < 4; ++i) (void)hipFree(k.plane[i]);
        (void)hipFree(c->d_part); (void)hipFree(c->d_jobs); (void)hipFree(c->d_zjobs);
    }
    (void)hipFree(c->d_results); (void)hipFree(c->d_qlower); (void)hipFree(c->d_qraise); (void)hipFree(c->d_qsizes); (void)hipFree(c->d_dbg); (void)hipFree(c->d_slow); (void)hipFree(c->d_slow_list); (void)hipFree(c->d_heavy); (void)hipFree(c->d_early); (void)hipFree(c->d_elist); (void)hipFree(c->d_hlist); (void)hipFree(c->d_scalar);  (void)hipFree(c->d_ship_desc); (void)hipFree(c->d_ship_heads); (void)hipFree(c->d_act); (void)hipFree(c->d_act_count); (void)hipFree(c->d_rrec); (void)hipFree(c->d_rbbox); (void)hipFree(c->d_rchunk);
    (void)hipFree(c->d_pts); (void)hipFree(c->d_tfs);
    (void)hipFree(c->d_oldcounts);
    (void)hipFree(c->d_bposes); (void)hipFree(c->d_bout);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream3) (void)hipStreamDestroy(c->stream3);
    if (c->ev_alloc) (void)hipEventDestroy(c->ev_alloc);
    if (c->ev_early) (void)hipEventDestroy(c->ev_early);
    if (c->ev_go) (void)hipEventDestroy(c->ev_go);
    if (c->ev_route) (void)hipEventDestroy(c->ev_route);
    if (c->ev_heavy) (void)hipEventDestroy(c->ev_heavy);
    delete c;
}

int32_t lama_hip_pf_set_poses(lama_hip_ctx* c, const double* poses)
{
    if (!c || !poses) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    std::memcpy(c->h_poses.data(), poses, sizeof(double) * 4 * c->P);
    // asynchronous: the source is the context's own host mirror, which stays valid; later calls are stream ordered
    HIPCHK(c, hipMemcpyAsync(c->d_poses, c->h_poses.data(), sizeof(double) * 4 * c->P, hipMemcpyHostToDevice, c->stream));
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_get_poses(lama_hip_ctx* c, double* poses)
{
    if (!c || !poses) return LAMA_HIP_E_INVALID;
    ENTER(c);
    std::memcpy(poses, c->h_poses.data(), sizeof(double) * 4 * c->P);
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_init(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin3, const double* quat,
                         const double* pose0)
{
    if (!c || !pts || !pose0 || n == 0) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_pf_init called twice");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    // centre the map window on the first pose (patch aligned)
    const double mx = c->scale * pose0[2] + c->off, my = c->scale * pose0[3] + c->off;
    const uint32_t px = ((uint32_t)mx) >> 5, py = ((uint32_t)my) >> 5;
    c->wx0 = (px - c->W / 2) * 32;
    c->wy0 = (py - c->W / 2) * 32;
    for (uint32_t p = 0; p < c->P; ++p) std::memcpy(&c->h_poses[4 * p], pose0, sizeof(double) * 4);
    HIPCHK(c, hipMemcpyAsync(c->d_poses, c->h_poses.data(), sizeof(double) * 4 * c->P, hipMemcpyHostToDevice, c->stream));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    const Affine mtf = moving_tf(origin3, quat);
    rc = run_update_maps(c, n, mtf, 0, 1);                          // particle 0 only (pf_slam2d.cpp:204)
    if (rc) return rc;
    rc = check_device_errors(c, true, false);                        // (grows the arenas and repeats the update if it ran out of patches)
    if (rc) return rc;
    if (c->P > 1) {                                                 // copy-construct the others (:206-216)
        std::vector<int32_t> idx(c->P, 0);
        rc = permute_particles(c, idx.data());
        if (rc) return rc;
    }
    rc = check_device_errors(c, true, false);
    if (rc) return rc;
    c->initialised = true;
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_scan_match(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin3, const double* quat,
                               double* poses_out, double* loglik_out, int32_t* iters_out)
{
    if (!c || !pts || n == 0) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_pf_scan_match before lama_hip_pf_init");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    const Affine mtf = moving_tf(origin3, quat);
    DevParams prm = make_params(c);
    {
        Timer t(c, &c->ctr.ms_scan_match, &c->ctr.launches_scan_match);
        if (c->max_sqdist > (uint32_t)SM_LUT) hipLaunchKernelGGL(k_scan_match<true>, dim3(c->P), dim3(SM_BLOCK), 0, c->stream, prm, c->d_pts, (int)n, mtf, c->d_loglik, c->d_iters);
        else hipLaunchKernelGGL(k_scan_match<false>, dim3(c->P), dim3(SM_BLOCK), 0, c->stream, prm, c->d_pts, (int)n, mtf, c->d_loglik, c->d_iters);
        t.stop();
    }
    HIPCHK(c, hipGetLastError());
    // poses, log-likelihoods, iteration counts and the error word: one copy (they share an allocation)
    c->h_results.resize(c->results_bytes);
    HIPCHK(c, hipMemcpyAsync(c->h_results.data(), c->d_results, c->results_bytes, hipMemcpyDeviceToHost, c->stream));
    rc = check_device_errors(c, false, c->cfg.profile != 0, true);   // synchronises
    if (rc) return rc;
    const uint8_t* r = c->h_results.data();
    std::memcpy(c->h_poses.data(), r, sizeof(double) * 4 * c->P);
    if (poses_out) std::memcpy(poses_out, r, sizeof(double) * 4 * c->P);
    if (loglik_out) std::memcpy(loglik_out, r + sizeof(double) * 4 * c->P, sizeof(double) * c->P);
    if (iters_out) std::memcpy(iters_out, r + sizeof(double) * 5 * c->P, sizeof(int32_t) * c->P);
    c->h_ll.resize(c->P); std::memcpy(c->h_ll.data(), r + sizeof(double) * 4 * c->P, sizeof(double) * c->P); c->ll_valid = true;
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_resample(lama_hip_ctx* c, const int32_t* sample_idx)
{
    if (!c || !sample_idx) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_pf_resample before lama_hip_pf_init");
    for (uint32_t i = 0; i < c->P; ++i)
        if (sample_idx[i] < 0 || (uint32_t)sample_idx[i] >= c->P) return fail(c, LAMA_HIP_E_INVALID, "sample_idx out of range");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->early_ok = false;                                            // particle indices change meaning
    if (c->ll_valid) { std::vector<double> nl(c->P); for (uint32_t i = 0; i < c->P; ++i) nl[i] = c->h_ll[sample_idx[i]]; c->h_ll.swap(nl); }
    return permute_particles(c, sample_idx);
}

int32_t lama_hip_pf_update_maps_begin(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin3, const double* quat)
{
    if (!c || n == 0) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_pf_update_maps before lama_hip_pf_init");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    const Affine mtf = moving_tf(origin3, quat);
    rc = run_update_maps(c, n, mtf, 0, c->P);
    if (rc) return rc;
    c->pending_maps = true;
    return LAMA_HIP_OK;
}

int32_t lama_hip_ctx_device(const lama_hip_ctx* c) { return c ? c->cfg.device : -1; }

int32_t lama_hip_sync(lama_hip_ctx* c)
{
    if (!c) return LAMA_HIP_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    return finish_pending(c);
}

int32_t lama_hip_pf_update_maps(lama_hip_ctx* c, const double* pts, uint32_t n, const double* origin3, const double* quat)
{
    const int32_t rc = lama_hip_pf_update_maps_begin(c, pts, n, origin3, quat);
    if (rc) return rc;
    return lama_hip_sync(c);
}

int32_t lama_hip_pf_map_patches(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t* num)
{
    if (!c || !num || particle >= c->P || (kind != LAMA_HIP_MAP_DISTANCE && kind != LAMA_HIP_MAP_OCCUPANCY)) return LAMA_HIP_E_INVALID;
    ENTER(c);
    *num = (uint32_t)c->h_counts[2 * particle + (kind == LAMA_HIP_MAP_OCCUPANCY ? 1 : 0)];
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_patch_ids(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t cap, uint64_t* patch_ids, uint32_t* num_patches)
{
    if (!c || particle >= c->P || (kind != LAMA_HIP_MAP_DISTANCE && kind != LAMA_HIP_MAP_OCCUPANCY)) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const MapStore& s = c->ms;
    const size_t WW = (size_t)c->W * c->W;
    const bool dm = kind == LAMA_HIP_MAP_DISTANCE;
    std::vector<int16_t> dir(WW);
    HIPCHK(c, hipMemcpy(dir.data(), (dm ? s.dm_dir : s.occ_dir) + c->h_part[particle].home * WW, WW * 2, hipMemcpyDeviceToHost));
    std::vector<uint64_t> ids;
    for (uint32_t wy = 0; wy < c->W; ++wy)
        for (uint32_t wx = 0; wx < c->W; ++wx)
            if (dir[wy * c->W + wx] >= 0) ids.push_back(((uint64_t)(c->wx0 >> 5) + wx) * 2642244ull + ((uint64_t)(c->wy0 >> 5) + wy));
    std::sort(ids.begin(), ids.end());
    if (num_patches) *num_patches = (uint32_t)ids.size();
    if (patch_ids) for (size_t k = 0; k < ids.size() && k < cap; ++k) patch_ids[k] = ids[k];
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_map_checksums(lama_hip_ctx* c, int32_t kind, uint64_t* out)
{
    if (!c || !out || (kind != LAMA_HIP_MAP_DISTANCE && kind != LAMA_HIP_MAP_OCCUPANCY)) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (c->cfg.occupancy_policy != 0 && kind == LAMA_HIP_MAP_OCCUPANCY) return fail(c, LAMA_HIP_E_INVALID, "map checksums cover the frequency occupancy policy");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    uint64_t* d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_out, sizeof(uint64_t) * c->P));
    DevParams prm = make_params(c);
    hipLaunchKernelGGL(k_map_checksum, dim3(c->P), dim3(256), 0, c->stream, prm, kind == LAMA_HIP_MAP_DISTANCE ? 0 : 1, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof(uint64_t) * c->P, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    HIPCHK(c, e);
    return LAMA_HIP_OK;
}

// Map::deletePatchAt on both maps.  The arenas are bump allocated (slot = count++), so a deleted slot is refilled with the
// last used slot and the directory entry of that patch is redirected: the arena stays dense.  Host driven (a handful of
// patches per scan at most): device-to-device copies + memsets on the context's stream.
int32_t lama_hip_pf_delete_patches(lama_hip_ctx* c, uint32_t particle, const uint64_t* patch_ids, uint32_t n, uint32_t* deleted)
{
    if (!c || particle >= c->P || (!patch_ids && n)) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (deleted) *deleted = 0;
    if (n == 0) return LAMA_HIP_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    MapStore& s = c->ms;
    const size_t WW = (size_t)c->W * c->W;
    const HostPart hp = c->h_part[particle];
    const PartRec pr = dev_part(c, hp);                             // the particle's regions (plane addresses)
    for (int kind = 0; kind < 2; ++kind) {
        const bool dm = kind == 0;
        int16_t* d_dir = (dm ? s.dm_dir : s.occ_dir) + hp.home * WW;
        std::vector<int16_t> dir(WW);
        HIPCHK(c, hipMemcpy(dir.data(), d_dir, WW * 2, hipMemcpyDeviceToHost));
        int count = c->h_counts[2 * particle + (dm ? 0 : 1)];
        bool touched = false;
        for (uint32_t k = 0; k < n; ++k) {
            const uint64_t px = patch_ids[k] / 2642244ull, py = patch_ids[k] % 2642244ull;
            const int64_t wx = (int64_t)px - (int64_t)(c->wx0 >> 5), wy = (int64_t)py - (int64_t)(c->wy0 >> 5);
            if (wx < 0 || wy < 0 || wx >= (int64_t)c->W || wy >= (int64_t)c->W) continue;
            const size_t pidx = (size_t)wy * c->W + (size_t)wx;
            const int slot = dir[pidx];
            if (slot < 0) continue;
            const int last = count - 1;
            auto plane_move = [&](void* plane, size_t bytes_per_slot) -> hipError_t {
                char* b = (char*)plane;
                if (slot != last) {
                    hipError_t e = hipMemcpyAsync(b + (size_t)slot * bytes_per_slot, b + (size_t)last * bytes_per_slot, bytes_per_slot, hipMemcpyDeviceToDevice, c->stream);
                    if (e != hipSuccess) return e;
                }
                return hipMemsetAsync(b + (size_t)last * bytes_per_slot, 0, bytes_per_slot, c->stream);
            };
            if (dm) {
                HIPCHK(c, plane_move(pr.dm_sv, SV_PATCH_BYTES)); HIPCHK(c, plane_move(pr.dm_obs, 4096)); HIPCHK(c, plane_move(pr.dm_mask, 128));
            } else {
                HIPCHK(c, plane_move(pr.occ, 4096)); HIPCHK(c, plane_move(pr.occ_mask, 128));
                HIPCHK(c, plane_move(pr.occ_hit, 128));
            }
            if (slot != last)
                for (size_t q = 0; q < WW; ++q) if (dir[q] == last) { dir[q] = (int16_t)slot; break; }
            dir[pidx] = -1;
            --count;
            touched = true;
            if (dm && deleted) ++*deleted;
        }
        if (touched) {
            HIPCHK(c, hipMemcpyAsync(d_dir, dir.data(), WW * 2, hipMemcpyHostToDevice, c->stream));
            c->h_counts[2 * particle + (dm ? 0 : 1)] = count;
            HIPCHK(c, hipMemcpyAsync(s.counts + 2 * particle + (dm ? 0 : 1), &c->h_counts[2 * particle + (dm ? 0 : 1)], sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));          // `dir` and h_counts are read by the copies above
        }
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_download_map(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t cap,
                                 uint64_t* patch_ids, uint8_t* cells, uint64_t* masks, uint32_t* num_patches)
{
    if (!c || particle >= c->P || (kind != LAMA_HIP_MAP_DISTANCE && kind != LAMA_HIP_MAP_OCCUPANCY)) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const MapStore& s = c->ms;
    const size_t WW = (size_t)c->W * c->W;
    const bool dm = kind == LAMA_HIP_MAP_DISTANCE;
    const HostPart hp = c->h_part[particle];
    const PartRec pr = dev_part(c, hp);
    const uint32_t count = (uint32_t)c->h_counts[2 * particle + (dm ? 0 : 1)];
    if (num_patches) *num_patches = count;
    if (count == 0 || cap == 0) return LAMA_HIP_OK;
    std::vector<int16_t> dir(WW);
    HIPCHK(c, hipMemcpy(dir.data(), (dm ? s.dm_dir : s.occ_dir) + hp.home * WW, WW * 2, hipMemcpyDeviceToHost));
    // (reference patch id, slot), ascending id
    std::vector<std::pair<uint64_t, int>> order;
    for (uint32_t wy = 0; wy < c->W; ++wy)
        for (uint32_t wx = 0; wx < c->W; ++wx) {
            int slot = dir[wy * c->W + wx];
            if (slot < 0) continue;
            const uint64_t px = (c->wx0 >> 5) + wx, py = (c->wy0 >> 5) + wy;
            order.emplace_back(px * 2642244ull + py, slot);                   // Map::m2p
        }
    std::sort(order.begin(), order.end());
    if (order.size() != count) return fail(c, LAMA_HIP_E_STATE, "directory / count mismatch");
    std::vector<uint64_t> hmask((size_t)count * 16);
    HIPCHK(c, hipMemcpy(hmask.data(), dm ? pr.dm_mask : pr.occ_mask,
                        hmask.size() * 8, hipMemcpyDeviceToHost));
    std::vector<sv_t> hsv; std::vector<uint32_t> hobs, hocc;
    if (dm) {
        hsv.resize((size_t)count * 1024); hobs.resize((size_t)count * 1024);
        HIPCHK(c, hipMemcpy(hsv.data(), pr.dm_sv, hsv.size() * SV_BYTES, hipMemcpyDeviceToHost));
        HIPCHK(c, hipMemcpy(hobs.data(), pr.dm_obs, hobs.size() * 4, hipMemcpyDeviceToHost));
    } else {
        hocc.resize((size_t)count * 1024);
        HIPCHK(c, hipMemcpy(hocc.data(), pr.occ, hocc.size() * 4, hipMemcpyDeviceToHost));
    }
    const uint32_t nout = std::min<uint32_t>(cap, count);
    for (uint32_t k = 0; k < nout; ++k) {
        const int slot = order[k].second;
        if (patch_ids) patch_ids[k] = order[k].first;
        if (masks) std::memcpy(masks + (size_t)k * 16, &hmask[(size_t)slot * 16], 128);
        if (!cells) continue;
        if (dm) {
            uint8_t* o = cells + (size_t)k * 10240;
            for (int ci = 0; ci < 1024; ++ci) {
                const sv_t sv = hsv[(size_t)slot * 1024 + ci];
                const uint32_t ob = hobs[(size_t)slot * 1024 + ci];
                const int16_t ox = (int16_t)(ob & 0xFFFF), oy = (int16_t)(ob >> 16), oz = 0;
                const uint16_t sq = (uint16_t)(sv & SV_SQMASK);
                std::memcpy(o + 10 * ci + 0, &ox, 2); std::memcpy(o + 10 * ci + 2, &oy, 2); std::memcpy(o + 10 * ci + 4, &oz, 2);
                std::memcpy(o + 10 * ci + 6, &sq, 2);
                o[10 * ci + 8] = (sv & SV_VALID) ? 1 : 0;
                o[10 * ci + 9] = (sv & SV_QUEUED) ? 1 : 0;
            }
        } else {
            std::memcpy(cells + (size_t)k * 4096, &hocc[(size_t)slot * 1024], 4096);   // {u16 occupied, u16 visited} little endian
            if (masks && c->cfg.occupancy_policy == 0) {   // Container mask of a frequency cell = "visited != 0" (plus the plane bits set on uint16 wrap)
                uint64_t* mk = masks + (size_t)k * 16;
                for (int ci = 0; ci < 1024; ++ci)
                    if (hocc[(size_t)slot * 1024 + ci] >> 16) mk[ci >> 6] |= 1ull << (ci & 63);
            }
        }
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_upload_map(lama_hip_ctx* c, uint32_t particle, int32_t kind, uint32_t n, const uint64_t* patch_ids, const uint8_t* cells,
                               const uint64_t* masks)
{
    if (!c || particle >= c->P || (kind != LAMA_HIP_MAP_DISTANCE && kind != LAMA_HIP_MAP_OCCUPANCY) || (n && (!patch_ids || !cells || !masks))) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (n > MAX_PATCHES) return fail(c, LAMA_HIP_E_CAPACITY, "a particle's map exceeds 32767 patches");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    const bool dm = kind == LAMA_HIP_MAP_DISTANCE;
    // Convert and validate everything FIRST: a rejected upload must leave the particle's previous map as it was (the old
    // slots were zeroed and the window / region possibly moved before a squared distance beyond the plane's 14 bits was noticed).
    std::vector<sv_t> hsv; std::vector<uint32_t> hobs;
    if (n && dm) {     // distance_t (10 B: int16 obstacle[3], uint16 sqdist, bool valid_obstacle, bool is_queued) -> the two planes
        hsv.resize((size_t)n * 1024); hobs.resize((size_t)n * 1024);
        for (size_t k = 0; k < n; ++k)
            for (int ci = 0; ci < 1024; ++ci) {
                const uint8_t* o = cells + k * 10240 + 10 * (size_t)ci;
                int16_t ox, oy; uint16_t sq;
                std::memcpy(&ox, o, 2); std::memcpy(&oy, o + 2, 2); std::memcpy(&sq, o + 6, 2);
                if ((uint32_t)sq > (uint32_t)SV_SQMASK) return fail(c, LAMA_HIP_E_INVALID, "the uploaded distance map holds a squared distance beyond 16383 cells^2 (l2_max above 127 cells)");
                hsv[k * 1024 + ci] = (sv_t)((sv_t)sq | (o[8] ? SV_VALID : (sv_t)0) | (o[9] ? SV_QUEUED : (sv_t)0));
                hobs[k * 1024 + ci] = ((uint32_t)(uint16_t)ox) | (((uint32_t)(uint16_t)oy) << 16);      // (pack_obs)
            }
    }
    if (n) {           // a patch given twice is an error (checked on the ids: the window may still move below)
        std::vector<uint64_t> ids(patch_ids, patch_ids + n);
        std::sort(ids.begin(), ids.end());
        if (std::adjacent_find(ids.begin(), ids.end()) != ids.end()) return fail(c, LAMA_HIP_E_INVALID, "a patch index appears twice in the uploaded map");
    }
    if (n) {   // the window must hold the patches (it is centred / moved / grown for them)
        int64_t x0 = INT64_MAX, x1 = INT64_MIN, y0 = INT64_MAX, y1 = INT64_MIN;
        for (uint32_t k = 0; k < n; ++k) {
            const int64_t px = (int64_t)(patch_ids[k] / 2642244ull), py = (int64_t)(patch_ids[k] % 2642244ull);
            x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
        }
        if (x1 - x0 + 1 > LAMA_HIP_MAX_WINDOW || y1 - y0 + 1 > LAMA_HIP_MAX_WINDOW) return fail(c, LAMA_HIP_E_WINDOW, "the map is wider than the largest device window (1016 patches)");
        const int32_t rw = ensure_window(c, x0, x1, y0, y1);
        if (rw) return rw;
    }
    if (!c->initialised) {
        for (uint32_t p = 0; p < c->P; ++p) { c->h_poses[4 * p] = 1.0; c->h_poses[4 * p + 1] = 0.0; c->h_poses[4 * p + 2] = 0.0; c->h_poses[4 * p + 3] = 0.0; }
        HIPCHK(c, hipMemcpyAsync(c->d_poses, c->h_poses.data(), sizeof(double) * 4 * c->P, hipMemcpyHostToDevice, c->stream));
        c->initialised = true;
    }
    {   // room for the patches (the particle's region moves if it must; what it held is replaced below)
        const HostPart& hp = c->h_part[particle];
        const uint32_t cap = dm ? hp.dm_cap : hp.occ_cap;
        if (n > cap) {
            const uint32_t want = want_capacity(dm ? c->floor_dm : c->floor_occ, n, 0u);
            const int32_t rg = set_capacities(c, {CapRequest{particle, dm ? want : hp.dm_cap, dm ? hp.occ_cap : want}});
            if (rg) return rg;
        }
    }
    const HostPart hp = c->h_part[particle];
    const PartRec pr = dev_part(c, hp);
    const size_t WW = (size_t)c->W * c->W;
    const int32_t old = c->h_counts[2 * particle + (dm ? 0 : 1)];
    // directory: slot k = the k-th given patch; a patch given twice is an error
    std::vector<int16_t> dir(WW, (int16_t)-1);
    for (uint32_t k = 0; k < n; ++k) {
        const int64_t wx = (int64_t)(patch_ids[k] / 2642244ull) - (int64_t)(c->wx0 >> 5), wy = (int64_t)(patch_ids[k] % 2642244ull) - (int64_t)(c->wy0 >> 5);
        if (wx < 0 || wy < 0 || wx >= (int64_t)c->W || wy >= (int64_t)c->W) return fail(c, LAMA_HIP_E_WINDOW, "a patch of the uploaded map fell outside the device window");
        int16_t& e = dir[(size_t)wy * c->W + (size_t)wx];
        if (e >= 0) return fail(c, LAMA_HIP_E_INVALID, "a patch index appears twice in the uploaded map");
        e = (int16_t)k;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // what the particle held before: zero the slots the new map does not overwrite ("unused slot == calloc'd")
    if (old > (int32_t)n) {
        const size_t keep = n, gone = (size_t)old - n;
        if (dm) {
            HIPCHK(c, hipMemsetAsync((char*)pr.dm_sv + keep * SV_PATCH_BYTES, 0, gone * SV_PATCH_BYTES, c->stream));
            HIPCHK(c, hipMemsetAsync((char*)pr.dm_obs + keep * 4096, 0, gone * 4096, c->stream));
            HIPCHK(c, hipMemsetAsync((char*)pr.dm_mask + keep * 128, 0, gone * 128, c->stream));
        } else {
            HIPCHK(c, hipMemsetAsync((char*)pr.occ + keep * 4096, 0, gone * 4096, c->stream));
            HIPCHK(c, hipMemsetAsync((char*)pr.occ_mask + keep * 128, 0, gone * 128, c->stream));
        }
    }
    // The copies go through the context's OWN stream, the one whose kernels read these planes afterwards with plain loads: a copy that
    // another stream carries out (a synchronous hipMemcpy runs on the null stream) into memory this context's kernels have read before
    // can leave them looking at cached lines of the old content -- the import of particle blobs taught that.
    if (n) {
        if (dm) {
            HIPCHK(c, hipMemcpyAsync(pr.dm_sv, hsv.data(), hsv.size() * SV_BYTES, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(pr.dm_obs, hobs.data(), hobs.size() * 4, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(pr.dm_mask, masks, (size_t)n * 128, hipMemcpyHostToDevice, c->stream));
        } else {       // frequency {u16 occupied, u16 visited} / float log-odds: 4 B cells as they are
            HIPCHK(c, hipMemcpyAsync(pr.occ, cells, (size_t)n * 4096, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(pr.occ_mask, masks, (size_t)n * 128, hipMemcpyHostToDevice, c->stream));
        }
    }
    HIPCHK(c, hipMemcpyAsync((dm ? c->ms.dm_dir : c->ms.occ_dir) + hp.home * WW, dir.data(), WW * 2, hipMemcpyHostToDevice, c->stream));
    c->h_counts[2 * particle + (dm ? 0 : 1)] = (int32_t)n;
    HIPCHK(c, hipMemcpyAsync(c->ms.counts + 2 * particle + (dm ? 0 : 1), &c->h_counts[2 * particle + (dm ? 0 : 1)], sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));                   // (the caller's buffers and the staging vectors are free again)
    if (!dm) c->visit_bound = 65535u;                             // unknown counters: let the wrap guard look before the next parallel ray-cast
    c->early_ok = false;
    return LAMA_HIP_OK;
}

int32_t lama_hip_match_batch(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3,
                             const double* quat, const double* poses, uint32_t B, double* out)
{
    if (!c || !pts || !poses || !out || n == 0 || B == 0 || particle >= c->P) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_match_batch before lama_hip_pf_init");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    if (B > c->b_cap) {
        (void)hipFree(c->d_bposes); (void)hipFree(c->d_bout);
        c->d_bposes = nullptr; c->d_bout = nullptr; c->b_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_bposes, sizeof(double) * 4 * B));
        HIPCHK(c, hipMalloc(&c->d_bout, sizeof(double) * B));
        c->b_cap = B;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_bposes, poses, sizeof(double) * 4 * B, hipMemcpyHostToDevice, c->stream));
    const Affine mtf = moving_tf(origin3, quat);
    DevParams prm = make_params(c);
    if (c->max_sqdist > (uint32_t)SM_LUT) hipLaunchKernelGGL(k_loglik_batch<true>, dim3(B), dim3(SM_BLOCK), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, mtf, c->d_bposes, c->d_bout);
    else hipLaunchKernelGGL(k_loglik_batch<false>, dim3(B), dim3(SM_BLOCK), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, mtf, c->d_bposes, c->d_bout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(out, c->d_bout, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LAMA_HIP_OK;
}

int32_t lama_hip_eval_batch(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3,
                            const double* quat, const double* poses, uint32_t B, double* sqnorm_out, double* loglik_out)
{
    if (!c || !poses || (!sqnorm_out && !loglik_out) || n == 0 || B == 0 || particle >= c->P) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_eval_batch before the map exists");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    if (2 * B > c->b_cap) {
        (void)hipFree(c->d_bposes); (void)hipFree(c->d_bout);
        c->d_bposes = nullptr; c->d_bout = nullptr; c->b_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_bposes, sizeof(double) * 4 * 2 * B));
        HIPCHK(c, hipMalloc(&c->d_bout, sizeof(double) * 2 * B));
        c->b_cap = 2 * B;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_bposes, poses, sizeof(double) * 4 * B, hipMemcpyHostToDevice, c->stream));
    const Affine mtf = moving_tf(origin3, quat);
    DevParams prm = make_params(c);
    {
        Timer t(c, &c->ctr.ms_eval_batch, &c->ctr.launches_eval_batch);
        hipLaunchKernelGGL(k_eval_batch, dim3(B), dim3(SM_BLOCK), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, mtf, c->d_bposes,
                           c->d_bout, c->d_bout + B);
        t.stop();
    }
    HIPCHK(c, hipGetLastError());
    if (sqnorm_out) HIPCHK(c, hipMemcpyAsync(sqnorm_out, c->d_bout, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    if (loglik_out) HIPCHK(c, hipMemcpyAsync(loglik_out, c->d_bout + B, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    resolve_timers(c);
    return LAMA_HIP_OK;
}

int32_t lama_hip_map_sample_likelihood(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3,
                                       const double* quat, double yaw, const double* xy, uint32_t K, uint32_t point_step, double* l_out)
{
    if (!c || !xy || !l_out || n == 0 || K == 0 || point_step == 0 || particle >= c->P) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_map_sample_likelihood before the map exists");
    if ((n + point_step - 1) / point_step > (uint32_t)SL_MAX_TERMS) return fail(c, LAMA_HIP_E_INVALID, "more than 128 sampled points per pose");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    if (2 * K > c->b_cap) {
        (void)hipFree(c->d_bposes); (void)hipFree(c->d_bout);
        c->d_bposes = nullptr; c->d_bout = nullptr; c->b_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_bposes, sizeof(double) * 4 * 2 * K));
        HIPCHK(c, hipMalloc(&c->d_bout, sizeof(double) * 2 * K));
        c->b_cap = 2 * K;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_bposes, xy, sizeof(double) * 2 * K, hipMemcpyHostToDevice, c->stream));
    // (Translation(x, y, 0) * AngleAxis(yaw, Z)) * moving_tf without the per-sample translation (src/loc2d.cpp:211-224)
    const Affine mtf = moving_tf(origin3, quat);
    const double sn = std::sin(yaw), cs = std::cos(yaw);          // Eigen AngleAxis(yaw, UnitZ), as in host_scan_tf
    Affine fixed;
    const double F[3][3] = {{cs, 0.0 - sn, 0.0}, {sn, cs, 0.0}, {0.0, 0.0, (1.0 - cs) + cs}};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) fixed.R[i][j] = F[i][j]; fixed.t[i] = 0.0; }
    Affine base;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) base.R[i][j] = (fixed.R[i][0] * mtf.R[0][j] + fixed.R[i][1] * mtf.R[1][j]) + fixed.R[i][2] * mtf.R[2][j];
        base.t[i] = (fixed.R[i][0] * mtf.t[0] + fixed.R[i][1] * mtf.t[1]) + fixed.R[i][2] * mtf.t[2];
    }
    DevParams prm = make_params(c);
    hipLaunchKernelGGL(k_sample_likelihood, dim3(K), dim3(64), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, (int)point_step, base,
                       c->d_bposes, c->d_bout);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(l_out, c->d_bout, sizeof(double) * K, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LAMA_HIP_OK;
}

// ------------------------------------------------------------------------------------------------
// SE2 pose-graph linearisation (lama_pgo.h)
// ------------------------------------------------------------------------------------------------
struct lama_hip_pgo {
    int32_t device = 0;
    uint32_t N = 0, F = 0, blocksF = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *d_poses = nullptr, *d_meas = nullptr, *d_sqrt = nullptr, *d_err = nullptr, *d_hoff = nullptr, *d_fdi = nullptr, *d_fdj = nullptr,
           *d_fg = nullptr, *d_hdiag = nullptr, *d_b = nullptr, *d_chi = nullptr;
    int32_t *d_fi = nullptr, *d_fj = nullptr, *d_incptr = nullptr, *d_inc = nullptr;
    std::vector<double> h_chi;
    std::string error;
};

#define PGOCHK(g, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { (g)->error = std::string(#call) + ": " + hipGetErrorString(e_); return LAMA_HIP_E_HIP; } } while (0)

void lama_hip_pgo_destroy(lama_hip_pgo* g);
#define PGOCHK_C(call) do { if ((call) != hipSuccess) { lama_hip_pgo_destroy(p); return LAMA_HIP_E_HIP; } } while (0)
int32_t lama_hip_pgo_create(int32_t device, uint32_t N, const int32_t* fi, const int32_t* fj, const double* meas4, const double* sqrt_info3,
                            uint32_t F, lama_hip_pgo** out)
{
    if (!out || !fi || !fj || !meas4 || !sqrt_info3 || N == 0 || F == 0) return LAMA_HIP_E_INVALID;
    *out = nullptr;
    for (uint32_t k = 0; k < F; ++k)
        if (fi[k] < 0 || (uint32_t)fi[k] >= N || fj[k] >= (int32_t)N || fj[k] == fi[k]) return LAMA_HIP_E_INVALID;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= device || device < 0) return LAMA_HIP_E_HIP;
    if (hipSetDevice(device) != hipSuccess) return LAMA_HIP_E_HIP;
    lama_hip_pgo* p = new lama_hip_pgo;
    p->device = device; p->N = N; p->F = F; p->blocksF = (F + PGO_BLOCK - 1) / PGO_BLOCK;
    // incidence lists in factor order (counting sort): what the reference's sequential loop over the factors adds to a variable
    std::vector<int32_t> ptr(N + 1, 0), inc;
    for (uint32_t k = 0; k < F; ++k) { ++ptr[fi[k] + 1]; if (fj[k] >= 0) ++ptr[fj[k] + 1]; }
    for (uint32_t v = 0; v < N; ++v) ptr[v + 1] += ptr[v];
    inc.resize(ptr[N]);
    std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
    for (uint32_t k = 0; k < F; ++k) { inc[fill[fi[k]]++] = (int32_t)(2 * k); if (fj[k] >= 0) inc[fill[fj[k]]++] = (int32_t)(2 * k + 1); }
    PGOCHK_C(hipStreamCreate(&p->stream));
    PGOCHK_C(hipEventCreate(&p->ev0)); PGOCHK_C(hipEventCreate(&p->ev1));
    PGOCHK_C(hipMalloc(&p->d_poses, sizeof(double) * 4 * N)); PGOCHK_C(hipMalloc(&p->d_meas, sizeof(double) * 4 * F));
    PGOCHK_C(hipMalloc(&p->d_sqrt, sizeof(double) * 3 * F)); PGOCHK_C(hipMalloc(&p->d_err, sizeof(double) * 3 * F));
    PGOCHK_C(hipMalloc(&p->d_hoff, sizeof(double) * 9 * F)); PGOCHK_C(hipMalloc(&p->d_fdi, sizeof(double) * 9 * F));
    PGOCHK_C(hipMalloc(&p->d_fdj, sizeof(double) * 9 * F)); PGOCHK_C(hipMalloc(&p->d_fg, sizeof(double) * 6 * F));
    PGOCHK_C(hipMalloc(&p->d_hdiag, sizeof(double) * 9 * N)); PGOCHK_C(hipMalloc(&p->d_b, sizeof(double) * 3 * N));
    PGOCHK_C(hipMalloc(&p->d_chi, sizeof(double) * p->blocksF));
    PGOCHK_C(hipMalloc(&p->d_fi, sizeof(int32_t) * F)); PGOCHK_C(hipMalloc(&p->d_fj, sizeof(int32_t) * F));
    PGOCHK_C(hipMalloc(&p->d_incptr, sizeof(int32_t) * (N + 1))); PGOCHK_C(hipMalloc(&p->d_inc, sizeof(int32_t) * std::max<size_t>(inc.size(), 1)));
    PGOCHK_C(hipMemcpy(p->d_fi, fi, sizeof(int32_t) * F, hipMemcpyHostToDevice)); PGOCHK_C(hipMemcpy(p->d_fj, fj, sizeof(int32_t) * F, hipMemcpyHostToDevice));
    PGOCHK_C(hipMemcpy(p->d_meas, meas4, sizeof(double) * 4 * F, hipMemcpyHostToDevice));
    PGOCHK_C(hipMemcpy(p->d_sqrt, sqrt_info3, sizeof(double) * 3 * F, hipMemcpyHostToDevice));
    PGOCHK_C(hipMemcpy(p->d_incptr, ptr.data(), sizeof(int32_t) * (N + 1), hipMemcpyHostToDevice));
    PGOCHK_C(hipMemcpy(p->d_inc, inc.data(), sizeof(int32_t) * inc.size(), hipMemcpyHostToDevice));
    p->h_chi.resize(p->blocksF);
    *out = p;
    return LAMA_HIP_OK;
}

void lama_hip_pgo_destroy(lama_hip_pgo* g)
{
    if (!g) return;
    (void)hipSetDevice(g->device);
    void* ptrs[] = {g->d_poses, g->d_meas, g->d_sqrt, g->d_err, g->d_hoff, g->d_fdi, g->d_fdj, g->d_fg, g->d_hdiag, g->d_b, g->d_chi,
                    g->d_fi, g->d_fj, g->d_incptr, g->d_inc};
    for (void* q : ptrs) if (q) (void)hipFree(q);
    if (g->ev0) (void)hipEventDestroy(g->ev0);
    if (g->ev1) (void)hipEventDestroy(g->ev1);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
}

const char* lama_hip_pgo_last_error(const lama_hip_pgo* g) { return g ? g->error.c_str() : "null handle"; }

int32_t lama_hip_pgo_linearize(lama_hip_pgo* g, const double* poses4, double* err, double* Hdiag, double* Hoff, double* b, double* chi2,
                               double* kernel_ms)
{
    if (!g || !poses4) return LAMA_HIP_E_INVALID;
    PGOCHK(g, hipSetDevice(g->device));
    PGOCHK(g, hipMemcpyAsync(g->d_poses, poses4, sizeof(double) * 4 * g->N, hipMemcpyHostToDevice, g->stream));
    PgoPtrs p{g->d_poses, g->d_fi, g->d_fj, g->d_meas, g->d_sqrt, g->d_err, g->d_hoff, g->d_fdi, g->d_fdj, g->d_fg, g->d_incptr, g->d_inc,
              g->d_hdiag, g->d_b, g->d_chi};
    PGOCHK(g, hipEventRecord(g->ev0, g->stream));
    hipLaunchKernelGGL(k_pgo_factors, dim3(g->blocksF), dim3(PGO_BLOCK), 0, g->stream, p, g->F);
    hipLaunchKernelGGL(k_pgo_reduce, dim3((g->N + PGO_BLOCK - 1) / PGO_BLOCK), dim3(PGO_BLOCK), 0, g->stream, p, g->N);
    PGOCHK(g, hipEventRecord(g->ev1, g->stream));
    PGOCHK(g, hipGetLastError());
    if (err) PGOCHK(g, hipMemcpyAsync(err, g->d_err, sizeof(double) * 3 * g->F, hipMemcpyDeviceToHost, g->stream));
    if (Hoff) PGOCHK(g, hipMemcpyAsync(Hoff, g->d_hoff, sizeof(double) * 9 * g->F, hipMemcpyDeviceToHost, g->stream));
    if (Hdiag) PGOCHK(g, hipMemcpyAsync(Hdiag, g->d_hdiag, sizeof(double) * 9 * g->N, hipMemcpyDeviceToHost, g->stream));
    if (b) PGOCHK(g, hipMemcpyAsync(b, g->d_b, sizeof(double) * 3 * g->N, hipMemcpyDeviceToHost, g->stream));
    PGOCHK(g, hipMemcpyAsync(g->h_chi.data(), g->d_chi, sizeof(double) * g->blocksF, hipMemcpyDeviceToHost, g->stream));
    PGOCHK(g, hipStreamSynchronize(g->stream));
    if (chi2) { double t = 0; for (double x : g->h_chi) t += x; *chi2 = t; }
    if (kernel_ms) { float ms = 0; PGOCHK(g, hipEventElapsedTime(&ms, g->ev0, g->ev1)); *kernel_ms = ms; }
    return LAMA_HIP_OK;
}

int32_t lama_hip_map_add_obstacles(lama_hip_ctx* c, uint32_t particle, const uint32_t* cells_xy, uint32_t n)
{
    if (!c || !cells_xy || n == 0 || particle >= c->P) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipSetDevice(c->cfg.device));
    {   // the window must hold the listed cells and what the brushfire reaches from them (it is centred / moved / grown for that)
        const int64_t r = (int64_t)(((uint32_t)std::ceil(std::sqrt((double)c->max_sqdist)) + 1u + 31u) / 32u);
        int64_t x0 = INT64_MAX, x1 = INT64_MIN, y0 = INT64_MAX, y1 = INT64_MIN;
        for (uint32_t i = 0; i < n; ++i) {
            const int64_t px = cells_xy[2 * i] >> 5, py = cells_xy[2 * i + 1] >> 5;
            x0 = std::min(x0, px); x1 = std::max(x1, px); y0 = std::min(y0, py); y1 = std::max(y1, py);
        }
        const int32_t rw = ensure_window(c, x0 - r, x1 + r, y0 - r, y1 + r);
        if (rw) return rw;
    }
    if (!c->initialised) {
        for (uint32_t p = 0; p < c->P; ++p) { c->h_poses[4 * p] = 1.0; c->h_poses[4 * p + 1] = 0.0; c->h_poses[4 * p + 2] = 0.0; c->h_poses[4 * p + 3] = 0.0; }
        HIPCHK(c, hipMemcpyAsync(c->d_poses, c->h_poses.data(), sizeof(double) * 4 * c->P, hipMemcpyHostToDevice, c->stream));
        c->initialised = true;
    }
    {   // every distance-map patch this call can allocate lies within guard_r patches of a listed cell's patch: make room first
        const int r = (int)(((uint32_t)std::ceil(std::sqrt((double)c->max_sqdist)) + 1u + 31u) / 32u);
        std::vector<uint64_t> keys;
        keys.reserve((size_t)n);
        uint64_t last = ~0ull;
        for (uint32_t i = 0; i < n; ++i) {
            const uint64_t k = ((uint64_t)(cells_xy[2 * i + 1] >> 5) << 32) | (uint64_t)(cells_xy[2 * i] >> 5);
            if (k != last) { keys.push_back(k); last = k; }
        }
        std::sort(keys.begin(), keys.end());
        keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
        std::vector<uint64_t> dil;
        dil.reserve(keys.size() * (size_t)((2 * r + 1) * (2 * r + 1)));
        for (uint64_t k : keys)
            for (int dy = -r; dy <= r; ++dy)
                for (int dx = -r; dx <= r; ++dx) dil.push_back((((k >> 32) + (uint64_t)(int64_t)dy) << 32) | (uint32_t)((uint32_t)k + (uint32_t)dx));
        std::sort(dil.begin(), dil.end());
        dil.erase(std::unique(dil.begin(), dil.end()), dil.end());
        const uint64_t want = (uint64_t)c->h_counts[2 * particle] + dil.size();
        if (want > c->h_part[particle].dm_cap) {
            const int32_t rg = set_capacities(c, {CapRequest{particle, (uint32_t)std::min<uint64_t>(MAX_PATCHES, (want + 31u) / 32u * 32u), c->h_part[particle].occ_cap}});
            if (rg) return rg;
        }
        c->last_guarded = false;
    }
    uint32_t* d_cells = nullptr;
    HIPCHK(c, hipMalloc(&d_cells, sizeof(uint32_t) * 2 * (size_t)n));
    HIPCHK(c, hipMemcpyAsync(d_cells, cells_xy, sizeof(uint32_t) * 2 * (size_t)n, hipMemcpyHostToDevice, c->stream));
    DevParams prm = make_params(c);
    // queue capacity bounds one batch of additions; larger lists go in slices (each followed by dm->update(): adding
    // obstacles then updating in slices yields the same distance map only if no slice boundary matters -- so a list
    // longer than the queue is rejected instead)
    if (n > c->cfg.queue_capacity) { (void)hipFree(d_cells); return fail(c, LAMA_HIP_E_CAPACITY, "more obstacle cells than cfg.queue_capacity"); }
    HIPCHK(c, hipMemsetAsync(c->d_slow_n, 0, 5 * sizeof(uint32_t), c->stream));    // all five: check_device_errors(maps) adds every word to the counters
    hipLaunchKernelGGL(k_dm_add_obstacles, dim3(1), dim3(UM_BLOCK), 0, c->stream, prm, (int)particle, d_cells, n);
    hipLaunchKernelGGL((k_brushfire<LQ_SMALL, RQ_SMALL, false, true>), dim3(1), dim3(2 * UM_BLOCK), 0, c->stream, prm, (int)particle, 0);
    hipLaunchKernelGGL((k_brushfire<LQ_BIG, RQ_BIG, true, true>), dim3(1), dim3(2 * UM_BLOCK), 0, c->stream, prm, (int)particle, 0);
    hipLaunchKernelGGL(k_brushfire_slow, dim3(1), dim3(UM_BLOCK), 0, c->stream, prm, (int)particle);
    HIPCHK(c, hipGetLastError());
    int32_t rc = check_device_errors(c, true, false);
    (void)hipFree(d_cells);
    return rc;
}

static int32_t match_solve_impl(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                                double* pose_inout, double* out7, int32_t* iters_out, int32_t do_solve, int32_t strategy, uint32_t max_iterations);

int32_t lama_hip_match_solve(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                             double* pose_inout, double* out7, int32_t* iters_out, int32_t do_solve)
{
    return match_solve_impl(c, particle, pts, n, origin3, quat, pose_inout, out7, iters_out, do_solve, -1, 0);
}

int32_t lama_hip_match_solve_with(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                                  double* pose_inout, double* out7, int32_t* iters_out, int32_t strategy, uint32_t max_iterations)
{
    if (strategy != 0 && strategy != 1) return LAMA_HIP_E_INVALID;
    return match_solve_impl(c, particle, pts, n, origin3, quat, pose_inout, out7, iters_out, 1, strategy, max_iterations);
}

static int32_t match_eval_impl(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                               const double* pose, double* residuals, double* jacobian, int cell_mode);
int32_t lama_hip_match_eval(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                            const double* pose, double* residuals, double* jacobian)
{
    return match_eval_impl(c, particle, pts, n, origin3, quat, pose, residuals, jacobian, 0);
}
int32_t lama_hip_match_cell_distances(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                                      const double* pose, double* distances)
{
    return match_eval_impl(c, particle, pts, n, origin3, quat, pose, distances, nullptr, 1);
}
static int32_t match_eval_impl(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                               const double* pose, double* residuals, double* jacobian, int cell_mode)
{
    if (!c || !pts || !pose || !residuals || n == 0 || particle >= c->P) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_match_eval before a map exists");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    double *d_pose = nullptr, *d_out = nullptr;
    HIPCHK(c, hipMalloc(&d_pose, sizeof(double) * 4));
    hipError_t e = hipMalloc(&d_out, sizeof(double) * 4 * (size_t)n);
    if (e == hipSuccess) e = hipMemcpyAsync(d_pose, pose, sizeof(double) * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const Affine mtf = moving_tf(origin3, quat);
        DevParams prm = make_params(c);
        hipLaunchKernelGGL(k_match_eval, dim3((n + 255) / 256), dim3(256), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, mtf, d_pose, d_out,
                           jacobian ? d_out + n : nullptr, cell_mode);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(residuals, d_out, sizeof(double) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && jacobian) e = hipMemcpyAsync(jacobian, d_out + n, sizeof(double) * 3 * (size_t)n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_pose); (void)hipFree(d_out);
    HIPCHK(c, e);
    return LAMA_HIP_OK;
}

static int32_t match_solve_impl(lama_hip_ctx* c, uint32_t particle, const double* pts, uint32_t n, const double* origin3, const double* quat,
                                double* pose_inout, double* out7, int32_t* iters_out, int32_t do_solve, int32_t strategy, uint32_t max_iterations)
{
    if (!c || !pts || !pose_inout || n == 0 || particle >= c->P) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "lama_hip_match_solve before a map exists");
    HIPCHK(c, hipSetDevice(c->cfg.device));
    int32_t rc = upload_scan(c, pts, n);
    if (rc) return rc;
    if (c->b_cap < 4) {
        (void)hipFree(c->d_bposes); (void)hipFree(c->d_bout);
        c->d_bposes = nullptr; c->d_bout = nullptr; c->b_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_bposes, sizeof(double) * 4 * 16));
        HIPCHK(c, hipMalloc(&c->d_bout, sizeof(double) * 16));
        c->b_cap = 16;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_bposes, pose_inout, sizeof(double) * 4, hipMemcpyHostToDevice, c->stream));
    const Affine mtf = moving_tf(origin3, quat);
    DevParams prm = make_params(c);
    if (strategy >= 0) prm.strategy = strategy;
    if (max_iterations) prm.max_iter = max_iterations;
    if (c->max_sqdist > (uint32_t)SM_LUT)
        hipLaunchKernelGGL(k_match_solve<true>, dim3(1), dim3(SM_BLOCK), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, mtf, c->d_bposes, c->d_bout,
                           c->d_iters, (int)do_solve);
    else
        hipLaunchKernelGGL(k_match_solve<false>, dim3(1), dim3(SM_BLOCK), 0, c->stream, prm, (int)particle, c->d_pts, (int)n, mtf, c->d_bposes, c->d_bout,
                           c->d_iters, (int)do_solve);
    HIPCHK(c, hipGetLastError());
    double o7[7]; int32_t it = 0;
    HIPCHK(c, hipMemcpyAsync(pose_inout, c->d_bposes, sizeof(double) * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(o7, c->d_bout, sizeof(double) * 7, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&it, c->d_iters, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    rc = check_device_errors(c);
    if (rc) return rc;
    if (out7) std::memcpy(out7, o7, sizeof(o7));
    if (iters_out) *iters_out = it;
    return LAMA_HIP_OK;
}

// Particle blob layout: [pose 4 f64][header 8 i32][dm_dir][occ_dir][dm_sv used][dm_obs used][dm_mask used][occ used][occ_mask used]
// (the directories in the SENDER's window side W: windows grow independently, the header carries the side)
static uint64_t blob_bytes_w(uint64_t W, int dmc, int occ)
{
    return 32 + 32 + 2 * W * W * 2 + (uint64_t)dmc * (SV_PATCH_BYTES + 4096 + 128) + (uint64_t)occ * (4096 + 128);
}
static uint64_t blob_bytes(const lama_hip_ctx* c, int dmc, int occ) { return blob_bytes_w(c->W, dmc, occ); }

// the extent of this context's mapped box inside its window (lo | hi << 16 in patches; the whole window when nothing is known)
static int32_t mapped_box_rel(const lama_hip_ctx* c, int axis)
{
    const int64_t o = (axis == 0 ? c->wx0 : c->wy0) >> 5, W = c->W;
    int64_t lo = 0, hi = W - 1;
    if (c->mb_valid) { lo = std::min(std::max((axis == 0 ? c->mbx0 : c->mby0) - o, (int64_t)0), W - 1); hi = std::min(std::max((axis == 0 ? c->mbx1 : c->mby1) - o, (int64_t)0), W - 1); }
    return (int32_t)((uint32_t)lo | ((uint32_t)hi << 16));
}

// ---- particle shipping: all outgoing / incoming particles of a resample in one launch each (k_export_particles / k_import_particles)
static int32_t upload_ship_desc(lama_hip_ctx* c, uint32_t n)
{
    if (n > c->ship_cap) {
        (void)hipFree(c->d_ship_desc); (void)hipFree(c->d_ship_heads);
        c->d_ship_desc = nullptr; c->d_ship_heads = nullptr; c->ship_cap = 0;
        const uint32_t cap = std::max<uint32_t>(n, 64u);
        HIPCHK(c, hipMalloc(&c->d_ship_desc, sizeof(ShipDesc) * cap));
        HIPCHK(c, hipMalloc(&c->d_ship_heads, (size_t)BLOB_HEAD * cap));
        c->ship_cap = cap;
    }
    HIPCHK(c, hipMemcpyAsync(c->d_ship_desc, c->h_ship_desc.data(), sizeof(ShipDesc) * n, hipMemcpyHostToDevice, c->stream));
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_export_particles(lama_hip_ctx* c, uint32_t n, const uint32_t* particles, void* const* bufs, const uint64_t* caps, uint64_t* bytes)
{
    if (!c || (n && !particles)) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "export before init");
    bool sizes_only = bufs == nullptr;
    for (uint32_t k = 0; k < n; ++k) {
        if (particles[k] >= c->P) return LAMA_HIP_E_INVALID;
        const uint64_t need = blob_bytes(c, c->h_counts[2 * particles[k]], c->h_counts[2 * particles[k] + 1]);
        if (bytes) bytes[k] = need;
        if (!sizes_only && (!bufs[k] || !caps || caps[k] < need)) return fail(c, LAMA_HIP_E_INVALID, "export buffer missing or too small");
    }
    if (sizes_only || n == 0) return LAMA_HIP_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->h_ship_desc.resize(n);
    for (uint32_t k = 0; k < n; ++k) c->h_ship_desc[k] = ShipDesc{(uint8_t*)bufs[k], particles[k], 0, 0, 0};
    const int32_t rc = upload_ship_desc(c, n);
    if (rc) return rc;
    hipLaunchKernelGGL(k_export_particles, dim3(n, 7, SHIP_SPLIT), dim3(256), 0, c->stream, make_params(c), (const ShipDesc*)c->d_ship_desc,
                       (const double*)c->d_poses, (int32_t)(c->wx0 >> 5), (int32_t)(c->wy0 >> 5),
                       (int32_t)c->visit_bound, mapped_box_rel(c, 0), mapped_box_rel(c, 1));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_export_particle(lama_hip_ctx* c, uint32_t particle, void* buf, uint64_t cap, uint64_t* bytes)
{
    if (!buf) return lama_hip_pf_export_particles(c, 1, &particle, nullptr, nullptr, bytes);
    void* b1 = buf;
    return lama_hip_pf_export_particles(c, 1, &particle, &b1, &cap, bytes);
}

int32_t lama_hip_pf_import_particles(lama_hip_ctx* c, uint32_t n, const uint32_t* particles, const void* const* bufs, const uint64_t* bytes)
{
    if (!c || (n && (!particles || !bufs || !bytes))) return LAMA_HIP_E_INVALID;
    ENTER(c);
    if (!c->initialised) return fail(c, LAMA_HIP_E_STATE, "import before init");
    if (n == 0) return LAMA_HIP_OK;
    for (uint32_t k = 0; k < n; ++k) if (particles[k] >= c->P || !bufs[k] || bytes[k] < (uint64_t)BLOB_HEAD) return LAMA_HIP_E_INVALID;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    c->early_ok = false; c->ll_valid = false;                       // other particles sit in these slots now
    // 1. what is coming: the blobs' heads in one copy
    c->h_ship_desc.resize(n);
    for (uint32_t k = 0; k < n; ++k) c->h_ship_desc[k] = ShipDesc{(uint8_t*)const_cast<void*>(bufs[k]), particles[k], 0, 0, 0};
    int32_t rc = upload_ship_desc(c, n);
    if (rc) return rc;
    hipLaunchKernelGGL(k_gather_blob_heads, dim3(n), dim3(64), 0, c->stream, (const ShipDesc*)c->d_ship_desc, n, c->d_ship_heads);
    c->h_ship_heads.resize((size_t)BLOB_HEAD * n);
    HIPCHK(c, hipMemcpyAsync(c->h_ship_heads.data(), c->d_ship_heads, (size_t)BLOB_HEAD * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    uint32_t max_dm = 0, max_occ = 0;
    for (uint32_t k = 0; k < n; ++k) {
        int32_t hdr[8];
        std::memcpy(hdr, c->h_ship_heads.data() + (size_t)BLOB_HEAD * k + 32, 32);
        if (hdr[0] < 0 || hdr[1] < 0 || hdr[0] > 32767 || hdr[1] > 32767 || hdr[5] < 8 || hdr[5] > (int32_t)LAMA_HIP_MAX_WINDOW || (hdr[5] & 7) ||
            blob_bytes_w((uint64_t)hdr[5], hdr[0], hdr[1]) != bytes[k])
            return fail(c, LAMA_HIP_E_INVALID, "particle blob does not match this context's geometry");
        max_dm = std::max<uint32_t>(max_dm, (uint32_t)hdr[0]); max_occ = std::max<uint32_t>(max_occ, (uint32_t)hdr[1]);
        // every shard's window follows (and grows with) its own particles: this one must hold what the sender has mapped
        const int64_t sx = hdr[2], sy = hdr[3];
        rc = ensure_window(c, sx + (hdr[6] & 0xFFFF), sx + ((uint32_t)hdr[6] >> 16), sy + (hdr[7] & 0xFFFF), sy + ((uint32_t)hdr[7] >> 16));
        if (rc) return rc;
    }
    for (uint32_t k = 0; k < n; ++k) {
        int32_t hdr[8];
        std::memcpy(hdr, c->h_ship_heads.data() + (size_t)BLOB_HEAD * k + 32, 32);
        // the sender's window sits elsewhere: the kernel translates the directories
        c->h_ship_desc[k].wdx = (int32_t)((int64_t)(c->wx0 >> 5) - (int64_t)hdr[2]);
        c->h_ship_desc[k].wdy = (int32_t)((int64_t)(c->wy0 >> 5) - (int64_t)hdr[3]);
    }
    {   // every destination region must hold what its incoming particle brings (the sender's maps may have grown before ours)
        std::vector<CapRequest> reqs;
        for (uint32_t k = 0; k < n; ++k) {
            int32_t hdr[8];
            std::memcpy(hdr, c->h_ship_heads.data() + (size_t)BLOB_HEAD * k + 32, 32);
            const HostPart& pr = c->h_part[particles[k]];
            const uint32_t nd = want_capacity(c->floor_dm, (uint32_t)hdr[0], 0u), no = want_capacity(c->floor_occ, (uint32_t)hdr[1], 0u);
            if ((uint32_t)hdr[0] > pr.dm_cap || (uint32_t)hdr[1] > pr.occ_cap) reqs.push_back(CapRequest{particles[k], std::max(nd, pr.dm_cap), std::max(no, pr.occ_cap)});
        }
        (void)max_dm; (void)max_occ;
        if (!reqs.empty()) { rc = set_capacities(c, reqs); if (rc) return rc; }
    }
    // 2. the copies
    rc = upload_ship_desc(c, n);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(c->d_oldcounts, c->ms.counts, sizeof(int32_t) * 2 * c->P, hipMemcpyDeviceToDevice, c->stream));
    hipLaunchKernelGGL(k_import_particles, dim3(n, 7, SHIP_SPLIT), dim3(256), 0, c->stream, make_params(c), (const ShipDesc*)c->d_ship_desc, (const int32_t*)c->d_oldcounts,
                       c->d_poses, c->d_err);
    HIPCHK(c, hipGetLastError());
    // the import itself reports what it could not place (a patch of the sender's window outside this one: ERR_WINDOW) -- not the
    // next, unrelated call on the context
    { const int32_t ri = check_device_errors(c); if (ri) return ri; }
    // 3. host mirrors
    for (uint32_t k = 0; k < n; ++k) {
        const uint8_t* head = c->h_ship_heads.data() + (size_t)BLOB_HEAD * k;
        int32_t hdr[8];
        std::memcpy(hdr, head + 32, 32);
        const uint32_t i = particles[k];
        std::memcpy(&c->h_poses[4 * i], head, 32);
        c->h_counts[2 * i] = hdr[0]; c->h_counts[2 * i + 1] = hdr[1];
        // the wrap guard's bound must hold for the counters this particle brings (they may be far above this context's own)
        c->visit_bound = std::max<uint32_t>(c->visit_bound, (uint32_t)std::min<int32_t>(std::max<int32_t>(hdr[4], 0), 65535));
    }
    return LAMA_HIP_OK;
}

int32_t lama_hip_pf_import_particle(lama_hip_ctx* c, uint32_t particle, const void* buf, uint64_t bytes)
{
    const void* b1 = buf;
    return lama_hip_pf_import_particles(c, 1, &particle, &b1, &bytes);
}

int32_t lama_hip_blob_alloc(lama_hip_ctx* c, uint64_t bytes, void** buf)
{
    if (!c || !buf || bytes == 0) return LAMA_HIP_E_INVALID;
    *buf = nullptr;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipMalloc(buf, bytes));
    return LAMA_HIP_OK;
}

int32_t lama_hip_blob_free(lama_hip_ctx* c, void* buf)
{
    if (!c) return LAMA_HIP_E_INVALID;
    if (!buf) return LAMA_HIP_OK;
    HIPCHK(c, hipSetDevice(c->cfg.device));
    HIPCHK(c, hipFree(buf));
    return LAMA_HIP_OK;
}

// Peer access between two devices, enabled once per ordered pair and process: without it hipMemcpyPeerAsync is
// staged through host memory by the runtime instead of going GPU to GPU over xGMI.  Returns 1 when BOTH directions are enabled.
static int peer_access_between(int a, int b)
{
    // (the shards of a multi-GPU object copy concurrently, a host thread each: the table is shared by all of them)
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> state;
    std::lock_guard<std::mutex> lock(mu);
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);
    auto one_way = [&](int from, int to) -> int {
        const auto key = std::make_pair(from, to);
        auto it = state.find(key);
        if (it != state.end()) return it->second;
        int can = 0, ok = 0;
        if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can) {
            if (hipSetDevice(from) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
                ok = (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) ? 1 : 0;
                if (!ok) (void)hipGetLastError();
            }
        } else (void)hipGetLastError();
        state[key] = ok;
        return ok;
    };
    const int ab = one_way(a, b), ba = one_way(b, a);
    if (prev_dev >= 0) (void)hipSetDevice(prev_dev);              // enabling access switches the calling thread's device
    return ab && ba;
}

int32_t lama_hip_blob_copy(lama_hip_ctx* dc, void* dst, lama_hip_ctx* sc, const void* src, uint64_t bytes)
{
    if (!dc || !sc || !dst || !src) return LAMA_HIP_E_INVALID;
    if (bytes == 0) return LAMA_HIP_OK;
    if (dc->cfg.device == sc->cfg.device) {
        HIPCHK(dc, hipSetDevice(dc->cfg.device));
        HIPCHK(dc, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, dc->stream));
        HIPCHK(dc, hipStreamSynchronize(dc->stream));
        return LAMA_HIP_OK;
    }
    // two devices: peer access first (queried and enabled once per pair), then the copy on the destination's stream; without peer
    // access the same call still works -- the runtime stages it -- and the counters say so (peer_access = 0)
    const int peer = peer_access_between(dc->cfg.device, sc->cfg.device);
    dc->ctr.peer_access = (uint32_t)peer;      // (the destination's own counter only: its thread is the caller; the source learns the value when it is a destination)
    HIPCHK(dc, hipSetDevice(dc->cfg.device));
    hipEvent_t e0 = dc->ev0, e1 = dc->ev1;
    HIPCHK(dc, hipEventRecord(e0, dc->stream));
    HIPCHK(dc, hipMemcpyPeerAsync(dst, dc->cfg.device, src, sc->cfg.device, bytes, dc->stream));      // GPU to GPU over xGMI
    HIPCHK(dc, hipEventRecord(e1, dc->stream));
    HIPCHK(dc, hipStreamSynchronize(dc->stream));
    float ms = 0;
    if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess) { dc->ctr.peer_copy_ms += ms; dc->ctr.peer_copy_bytes += bytes; }
    return LAMA_HIP_OK;
}

// profiling builds only (-DLAMA_PROFILE_BF): per-particle cycle counters of the last k_brushfire launch
int32_t lama_hip_debug_cycles(lama_hip_ctx* c, uint64_t* out /* P x 8 */)
{
    if (!c || !out) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipMemcpy(out, c->d_dbg, sizeof(uint64_t) * 8 * c->P, hipMemcpyDeviceToHost));
    return LAMA_HIP_OK;
}

// second block of developer counters (-DLAMA_PROFILE_LSE): kernel-level timestamps of k_brushfire per particle
int32_t lama_hip_debug_cycles2(lama_hip_ctx* c, uint64_t* out /* P x 8 */)
{
    if (!c || !out) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipMemcpy(out, c->d_dbg + 8 * (size_t)c->P, sizeof(uint64_t) * 8 * c->P, hipMemcpyDeviceToHost));
    return LAMA_HIP_OK;
}

// developer event log of particle 0 (-DLAMA_PROFILE_LSE): out[0] = number of words, then (event << 56 | timestamp) words
int32_t lama_hip_debug_log(lama_hip_ctx* c, uint64_t* out /* 131072 words */)
{
    if (!c || !out) return LAMA_HIP_E_INVALID;
    ENTER(c);
    HIPCHK(c, hipMemcpy(out, c->d_dbg + 16 * (size_t)c->P, 1u << 20, hipMemcpyDeviceToHost));
    return LAMA_HIP_OK;
}

#if defined(LAMA_KC_PROBE) || (defined(LAMA_KC_OLD) && LAMA_KC_OLD == 4)
// experiment build only (tools/kc_probe.py): out[0] = number of mismatches logged by kc_probe (lama_dev.h), then 64 x 8 words; clears the log
int32_t lama_hip_debug_kc_probe(uint64_t* out /* 1 + 512 words */)
{
    uint32_t n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(lama_dev::g_kc_n), sizeof(n)) != hipSuccess) return LAMA_HIP_E_HIP;
    out[0] = n;
    if (hipMemcpyFromSymbol(out + 1, HIP_SYMBOL(lama_dev::g_kc_ev), sizeof(uint64_t) * 512) != hipSuccess) return LAMA_HIP_E_HIP;
    n = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(lama_dev::g_kc_n), &n, sizeof(n)) != hipSuccess) return LAMA_HIP_E_HIP;
    return LAMA_HIP_OK;
}
#endif

// device memory the context holds: the maps (pools + directories), of which used, and everything
static void memory_figures(const lama_hip_ctx* c, lama_hip_counters* o)
{
    const MapStore& m = c->ms;
    const uint64_t P = c->P, WW = (uint64_t)c->W * c->W;
    const uint64_t dirs = 2 * P * WW * 2;
    uint64_t dm_used = 0, occ_used = 0;
    for (uint32_t p = 0; p < c->P; ++p) { dm_used += (uint64_t)c->h_counts[2 * p]; occ_used += (uint64_t)c->h_counts[2 * p + 1]; }
    uint64_t dm_pool = 0, occ_pool = 0;
    for (const PoolChunk& k : m.dm_chunks) dm_pool += k.patches;
    for (const PoolChunk& k : m.occ_chunks) occ_pool += k.patches;
    o->hbm_bytes_allocated = dirs + dm_pool * (SV_PATCH_BYTES + 4096 + 128) + occ_pool * (4096 + 128 + 128 + 4);
    o->hbm_bytes_used = dirs + dm_used * (SV_PATCH_BYTES + 4096 + 128) + occ_used * (4096 + 128 + 128 + 4);
    const uint64_t other = 2 * P * (uint64_t)c->cfg.queue_capacity * 8 + P * (uint64_t)c->cfg.active_capacity * 8 + (uint64_t)c->rrec_cap * (sizeof(lama_dev::RayRec) + 8) +
                           P * (16 * 8 + 12 * 8 + 4 * 8 + 64) + (1u << 20) + (uint64_t)c->pts_cap * 24;
    o->hbm_bytes_total = o->hbm_bytes_allocated + other;
}

int32_t lama_hip_get_counters(lama_hip_ctx* c, lama_hip_counters* out)
{
    if (!c || !out) return LAMA_HIP_E_INVALID;
    ENTER(c);
    *out = c->ctr;
    memory_figures(c, out);
    out->struct_bytes = (uint32_t)sizeof(lama_hip_counters);
    return LAMA_HIP_OK;
}

uint32_t lama_hip_counters_bytes(void) { return (uint32_t)sizeof(lama_hip_counters); }

int32_t lama_hip_get_counters_sized(lama_hip_ctx* c, void* out, uint32_t bytes)
{
    if (!c || !out) return LAMA_HIP_E_INVALID;
    lama_hip_counters full;
    const int32_t rc = lama_hip_get_counters(c, &full);
    if (rc) return rc;
    std::memcpy(out, &full, std::min<size_t>(bytes, sizeof(full)));
    return LAMA_HIP_OK;
}

int32_t lama_hip_reset_counters(lama_hip_ctx* c)
{
    if (!c) return LAMA_HIP_E_INVALID;
    ENTER(c);
    const lama_hip_counters old = c->ctr;
    std::memset(&c->ctr, 0, sizeof(c->ctr));
    c->ctr.dm_patches = old.dm_patches; c->ctr.occ_patches = old.occ_patches;
    c->ctr.brushfire_mode = old.brushfire_mode; c->ctr.brushfire_waves = old.brushfire_waves;
    c->ctr.window_patches = c->W; c->ctr.peer_access = old.peer_access;
    return LAMA_HIP_OK;
}

} // extern "C"
