// pf_slam2d.cpp -- host-side lama::PFSlam2D (include/lama/pf_slam2d.h).  The orchestration follows the
// reference's PFSlam2D::update (src/pf_slam2d.cpp:178-312) step by step; per-particle work is delegated to the
// device C-ABI.  Citations are relative to /root/reference.
#include "lama/pf_slam2d.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <tuple>

#include "hip_engine.hpp"

namespace lama {

namespace {
double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
} // namespace

// ---- Options::gpus > 1: one object, several devices -------------------------------------------------------------------------
// The shards are ordinary PFSlam2D objects (shard_rank = r, shard_world = gpus), each bound to its own device context and driven
// by its own persistent host thread: run(f) executes f(r) on every shard's thread and returns when all are done -- the two
// parallel regions of the reference's update() (src/pf_slam2d.cpp:254-266, 292-302) become `gpus` device streams fed concurrently.
struct PFSlam2D::Group {
    std::vector<std::unique_ptr<PFSlam2D>> shards;
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    const std::function<void(uint32_t)>* job = nullptr;
    uint64_t epoch = 0;
    uint32_t pending = 0;
    bool quit = false;
    std::vector<std::exception_ptr> errors;
    std::vector<void*> stage;               // per shard: device staging buffer for particle blobs (outgoing, then incoming)
    std::vector<uint64_t> stage_cap;

    void start()
    {
        const uint32_t G = (uint32_t)shards.size();
        errors.assign(G, nullptr);
        stage.assign(G, nullptr); stage_cap.assign(G, 0);
        for (uint32_t r = 0; r < G; ++r)
            threads.emplace_back([this, r]() {
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void(uint32_t)>* f = nullptr;
                    {
                        std::unique_lock<std::mutex> lk(m);
                        cv_go.wait(lk, [&] { return quit || epoch != seen; });
                        if (quit) return;
                        seen = epoch; f = job;
                    }
                    try { (*f)(r); } catch (...) { errors[r] = std::current_exception(); }
                    {
                        std::lock_guard<std::mutex> lk(m);
                        if (--pending == 0) cv_done.notify_all();
                    }
                }
            });
    }
    void run(const std::function<void(uint32_t)>& f)
    {
        {
            std::lock_guard<std::mutex> lk(m);
            job = &f; pending = (uint32_t)shards.size(); ++epoch;
        }
        cv_go.notify_all();
        std::unique_lock<std::mutex> lk(m);
        cv_done.wait(lk, [&] { return pending == 0; });
        for (auto& e : errors)
            if (e) { std::exception_ptr x = e; for (auto& y : errors) y = nullptr; std::rethrow_exception(x); }
    }
    void reserve_stage(uint32_t r, uint64_t bytes)
    {
        if (bytes <= stage_cap[r]) return;
        PFSlam2D& s = *shards[r];
        if (stage[r]) (void)s.eng_->blob_free(s.ctx_, stage[r]);
        stage[r] = nullptr; stage_cap[r] = 0;
        const uint64_t want = bytes + bytes / 2;
        const int32_t rc = s.eng_->blob_alloc(s.ctx_, want, &stage[r]);
        if (rc) s.fail(rc, "lama_hip_blob_alloc");
        stage_cap[r] = want;
    }
    ~Group()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv_go.notify_all();
        for (auto& t : threads) t.join();
        for (size_t r = 0; r < shards.size(); ++r)
            if (stage[r]) (void)shards[r]->eng_->blob_free(shards[r]->ctx_, stage[r]);
    }
};

PFSlam2D::PFSlam2D(const Options& options) : options_(options)
{
    if (options_.particles == 0) throw std::runtime_error("lama::PFSlam2D: Options::particles must be > 0");
    if (options_.use_compression) throw std::runtime_error("lama::PFSlam2D: use_compression is not supported on the device path");
    if (options_.shard_world == 0 || options_.shard_rank >= options_.shard_world)
        throw std::runtime_error("lama::PFSlam2D: invalid shard_rank / shard_world");
    if (options_.gpus < 1) options_.gpus = 1;
    if (options_.gpus > 1) {
        if (options_.shard_world != 1) throw std::runtime_error("lama::PFSlam2D: Options::gpus > 1 needs shard_world == 1 (one process drives all shards)");
        if ((uint32_t)options_.gpus > options_.particles) throw std::runtime_error("lama::PFSlam2D: more GPUs than particles");
        // all shards replay the same host random stream: settle the seed first (0 = random_device, src/pf_slam2d.cpp:131-134)
        if (options_.seed == 0) options_.seed = std::random_device()() | 1u;
        eng_ = defaultEngine(options_.l2_max, options_.resolution);
        int32_t ndev = 0;
        if (eng_->device_count(&ndev) != 0 || ndev <= 0)
            throw std::runtime_error("lama::PFSlam2D: no usable MI355X / HIP device; there is no CPU fallback");
        group_.reset(new Group());
        for (int32_t r = 0; r < options_.gpus; ++r) {
            Options so = options_;
            so.gpus = 1; so.shard_rank = (uint32_t)r; so.shard_world = (uint32_t)options_.gpus;
            so.gpu_device = (options_.gpu_device + r) % ndev;
            so.create_summary = false;              // the Summary is the whole object's: kept by this object
            group_->shards.emplace_back(new PFSlam2D(so));
        }
        group_->start();
        lo_ = 0; hi_ = options_.particles;
        particles_.assign(options_.particles, Particle());
        if (options_.create_summary) summary = new Summary();
        return;
    }
    // contiguous blocks: particle i lives on shard floor(i * G / P)  
    const uint64_t P = options_.particles, G = options_.shard_world, r = options_.shard_rank;
    lo_ = (uint32_t)((r * P + G - 1) / G);
    hi_ = (uint32_t)(((r + 1) * P + G - 1) / G);
    if (hi_ <= lo_) throw std::runtime_error("lama::PFSlam2D: more shards than particles");

    eng_ = defaultEngine(options_.l2_max, options_.resolution);
    lama_hip_cfg cfg;
    eng_->default_cfg(&cfg);
    cfg.particles = hi_ - lo_;
    cfg.resolution = options_.resolution;
    cfg.patch_size = options_.patch_size;
    cfg.l2_max = options_.l2_max;
    cfg.meas_sigma = options_.meas_sigma;
    cfg.max_iter = options_.max_iter;
    cfg.truncated_ray = options_.truncated_ray;
    cfg.truncated_range = options_.truncated_range;
    cfg.device = options_.gpu_device;
    cfg.profile = options_.profile ? 1 : 0;
    cfg.brushfire_mode = options_.brushfire_mode;
    if (options_.window_patches) cfg.window_patches = options_.window_patches;
    if (options_.dm_patch_capacity) cfg.dm_patch_capacity = options_.dm_patch_capacity;
    if (options_.occ_patch_capacity) cfg.occ_patch_capacity = options_.occ_patch_capacity;
    if (options_.queue_capacity) cfg.queue_capacity = options_.queue_capacity;
    const int32_t rc = eng_->ctx_create(&cfg, &ctx_);
    if (rc != 0 || !ctx_) {
        char msg[256];
        std::snprintf(msg, sizeof(msg), "lama::PFSlam2D: lama_hip_ctx_create failed (status %d): no usable MI355X / HIP device "
                                        "or unsupported options; there is no CPU fallback", rc);
        throw std::runtime_error(msg);
    }
    particles_.assign(options_.particles, Particle());
    local_loglik_.assign(hi_ - lo_, 0.0);

    // rng seed: 0 -> random_device (src/pf_slam2d.cpp:131-134, src/random.cpp:41-49).  Every shard of a sharded pool must replay
    // the SAME host random stream (motion noise of all P particles, the resampling draw): a per-process random seed would make
    // the shards disagree on the resampling indices, so it is refused there -- broadcast one seed instead (distributed.py does).
    if (options_.seed == 0 && options_.shard_world > 1)
        throw std::runtime_error("lama::PFSlam2D: Options::seed = 0 (random_device) is not allowed with shard_world > 1: all shards need the same seed");
    if (options_.seed == 0) options_.seed = std::random_device()();
    gen_.seed(options_.seed);
    if (options_.create_summary) summary = new Summary();
}

PFSlam2D::~PFSlam2D()
{
    group_.reset();                         // joins the shard threads, destroys the shard objects (and their contexts)
    if (ctx_) eng_->ctx_destroy(ctx_);
    delete summary;
}

lama_hip_ctx* PFSlam2D::deviceContext() const { return group_ ? group_->shards[0]->ctx_ : ctx_; }
const HipEngine* PFSlam2D::engine() const { return eng_.get(); }
size_t PFSlam2D::numShards() const { return group_ ? group_->shards.size() : 0; }
const PFSlam2D* PFSlam2D::shard(size_t r) const { return group_ && r < group_->shards.size() ? group_->shards[r].get() : nullptr; }
const PFSlam2D* PFSlam2D::ownerOf(uint32_t i) const
{
    if (!group_) return this;
    const uint64_t G = group_->shards.size(), P = options_.particles;
    return group_->shards[(size_t)(((uint64_t)i * G) / P)].get();
}
void PFSlam2D::syncDevice()
{
    const int32_t rs = eng_->sync(ctx_);
    if (rs) fail(rs, "lama_hip_sync");
}

// Host mirror of a multi-GPU object: weights are replicated on every shard (shard 0's are taken), a particle's pose is its owner's.
void PFSlam2D::mirrorShards(bool poses_changed)
{
    const PFSlam2D& s0 = *group_->shards[0];
    const uint32_t P = options_.particles;
    for (uint32_t i = 0; i < P; ++i) {
        particles_[i].weight = s0.particles_[i].weight;
        particles_[i].normalized_weight = s0.particles_[i].normalized_weight;
        particles_[i].weight_sum = s0.particles_[i].weight_sum;
        if (poses_changed) particles_[i].pose = ownerOf(i)->particles_[i].pose;
    }
    neff_ = s0.neff_;
    num_resamples_ = s0.num_resamples_;
}

// PFSlam2D::update (src/pf_slam2d.cpp:178-312) over `gpus` shards in one process.
bool PFSlam2D::updateGroup(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp)
{
    Group& g = *group_;
    const uint32_t G = (uint32_t)g.shards.size(), P = options_.particles;
    const double t_begin = now_s();
    dropMapViews();
    xt_ = ExchangeTimes();
    const bool first = !has_first_scan;
    for (auto& sh : g.shards) sh->pose_ = pose_;                 // setPrior() of this object reaches the shards
    std::vector<Phase> ph(G, kNoUpdate);
    g.run([&](uint32_t r) { ph[r] = g.shards[r]->updateBegin(surface, odometry, timestamp); });
    const double t_solved = now_s();
    xt_.phase_begin = t_solved - t_begin;
    if (first) {                                                 // :185-228
        has_first_scan = true;
        timestamps_.push_back(timestamp);
        mirrorShards(true);
        for (uint32_t i = 0; i < P; ++i) { particles_[i].pose = pose_; particles_[i].poses.push_back(pose_); }
        if (summary) {
            g.run([&](uint32_t r) { g.shards[r]->syncDevice(); });
            const double el = now_s() - t_begin;
            summary->timestamp.push_back(timestamp); summary->time.push_back(el); summary->time_mapping.push_back(el);
            summary->memory.push_back((double)getMemoryUsage());
        }
        return true;
    }
    if (ph[0] == kNoUpdate) { mirrorShards(true); return false; }   // motion gate closed on every shard alike
    // the poses of this update (history like the single-shard object: pushed before resampling reorders the particles)
    mirrorShards(true);
    for (uint32_t i = 0; i < P; ++i) particles_[i].poses.push_back(particles_[i].pose);
    if (summary) summary->time_solving.push_back(t_solved - t_begin);

    // ---- exchange 1: the P log-likelihoods.  They are host results of the scan match already (include/lama_hip.h,
    // lama_hip_pf_scan_match): the gather is a concatenation in host memory, no device collective could be cheaper.
    double t0 = now_s();
    std::vector<double> all_ll(P);
    for (uint32_t r = 0; r < G; ++r) {
        const PFSlam2D& sh = *g.shards[r];
        std::copy(sh.local_loglik_.begin(), sh.local_loglik_.end(), all_ll.begin() + sh.lo_);
    }
    xt_.gather = now_s() - t0;
    // normalise / Neff / resampling indices: every shard from its own (identical) copy of the state and random stream
    t0 = now_s();
    std::vector<std::vector<int32_t>> idxs(G);
    std::vector<char> due(G, 0);
    g.run([&](uint32_t r) { due[r] = g.shards[r]->planResample(all_ll.data(), idxs[r]) ? 1 : 0; });
    if (summary) summary->time_normalizing.push_back(now_s() - t0);
    if (due[0]) {                                                // :280-287
        const double tr0 = now_s();
        const std::vector<int32_t>& idx = idxs[0];
        auto owner = [&](uint32_t i) { return (uint32_t)(((uint64_t)i * G) / P); };
        // ---- exchange 2: clones whose source lives on another shard.  One blob per (source particle, destination shard).
        struct Xfer { uint32_t src_shard, dst_shard, particle; uint64_t bytes, src_off, dst_off; };
        std::vector<Xfer> xf;
        {
            std::map<std::pair<uint32_t, uint32_t>, bool> seen;      // (destination shard, source particle)
            for (uint32_t i = 0; i < P; ++i) {
                const uint32_t sp = (uint32_t)idx[i], ss = owner(sp), ds = owner(i);
                if (ss == ds || seen.count({ds, sp})) continue;
                seen[{ds, sp}] = true;
                xf.push_back(Xfer{ss, ds, sp, 0, 0, 0});
            }
            // (destination, source, particle) order: the blobs of one (source, destination) pair lie back to back in the source's
            // staging buffer AND in the destination's, so a pair moves with one copy
            std::sort(xf.begin(), xf.end(), [](const Xfer& a, const Xfer& b) {
                return std::tie(a.dst_shard, a.src_shard, a.particle) < std::tie(b.dst_shard, b.src_shard, b.particle); });
        }
        double ts0 = now_s();
        if (!xf.empty()) {
            // sizes and offsets first: a shard's staging buffer holds its outgoing blobs, then (behind them) the incoming ones
            std::vector<uint64_t> out_bytes(G, 0), in_bytes(G, 0);
            for (auto& x : xf) {
                const PFSlam2D& sh = *g.shards[x.src_shard];
                const int32_t rc = sh.eng_->pf_export_particle(sh.ctx_, x.particle - sh.lo_, nullptr, 0, &x.bytes);
                if (rc) sh.fail(rc, "lama_hip_pf_export_particle (size)");
                x.src_off = out_bytes[x.src_shard]; out_bytes[x.src_shard] += (x.bytes + 255) & ~255ull;
            }
            for (auto& x : xf) { x.dst_off = out_bytes[x.dst_shard] + in_bytes[x.dst_shard]; in_bytes[x.dst_shard] += (x.bytes + 255) & ~255ull; }
            // every source shard exports its outgoing particles into its buffer: ONE launch per shard, all shards concurrently ...
            g.run([&](uint32_t r) {
                if (out_bytes[r] + in_bytes[r] == 0) return;
                g.reserve_stage(r, out_bytes[r] + in_bytes[r]);
                PFSlam2D& sh = *g.shards[r];
                std::vector<uint32_t> parts; std::vector<void*> bufs; std::vector<uint64_t> caps;
                for (auto& x : xf)
                    if (x.src_shard == r) { parts.push_back(x.particle - sh.lo_); bufs.push_back((uint8_t*)g.stage[r] + x.src_off); caps.push_back(x.bytes); }
                if (parts.empty()) return;
                const int32_t rc = sh.eng_->pf_export_particles(sh.ctx_, (uint32_t)parts.size(), parts.data(), bufs.data(), caps.data(), nullptr);
                if (rc) sh.fail(rc, "lama_hip_pf_export_particles");
            });
            // ... and every destination shard pulls each source's run of blobs with ONE GPU-to-GPU copy (hipMemcpyPeerAsync over
            // xGMI between different devices)
            g.run([&](uint32_t r) {
                PFSlam2D& dsh = *g.shards[r];
                for (size_t a = 0; a < xf.size();) {
                    if (xf[a].dst_shard != r) { ++a; continue; }
                    size_t b = a;
                    uint64_t run = 0;
                    while (b < xf.size() && xf[b].dst_shard == r && xf[b].src_shard == xf[a].src_shard) { run += (xf[b].bytes + 255) & ~255ull; ++b; }
                    PFSlam2D& ssh = *g.shards[xf[a].src_shard];
                    const int32_t rc = dsh.eng_->blob_copy(dsh.ctx_, (uint8_t*)g.stage[r] + xf[a].dst_off, ssh.ctx_,
                                                           (const uint8_t*)g.stage[xf[a].src_shard] + xf[a].src_off, run);
                    if (rc) dsh.fail(rc, "lama_hip_blob_copy");
                    a = b;
                }
            });
            for (auto& x : xf) { xt_.shipped_particles += 1; xt_.shipped_bytes += x.bytes; }
        }
        xt_.ship = now_s() - ts0;
        // local copies on every shard, then the imports into the slots whose source was remote
        ts0 = now_s();
        std::vector<double> local_s(G, 0.0);
        g.run([&](uint32_t r) {
            PFSlam2D& sh = *g.shards[r];
            const double tl0 = now_s();
            sh.applyResample(idx);
            local_s[r] = now_s() - tl0;
            // one import launch for all the slots of this shard whose source was remote (several slots may take the same blob)
            std::vector<uint32_t> slots; std::vector<const void*> bufs; std::vector<uint64_t> nbytes;
            for (uint32_t i = sh.lo_; i < sh.hi_; ++i) {
                const uint32_t sp = (uint32_t)idx[i];
                if (owner(sp) == r) continue;
                for (auto& x : xf)
                    if (x.dst_shard == r && x.particle == sp) { slots.push_back(i - sh.lo_); bufs.push_back((const uint8_t*)g.stage[r] + x.dst_off); nbytes.push_back(x.bytes); break; }
            }
            const bool any = !slots.empty();
            if (any) {
                const int32_t rc = sh.eng_->pf_import_particles(sh.ctx_, (uint32_t)slots.size(), slots.data(), bufs.data(), nbytes.data());
                if (rc) sh.fail(rc, "lama_hip_pf_import_particles");
            }
            if (any) {                                           // the pose travels inside the blob: refresh the host mirror
                std::vector<double> poses((size_t)(sh.hi_ - sh.lo_) * 4);
                const int32_t rc = sh.eng_->pf_get_poses(sh.ctx_, poses.data());
                if (rc) sh.fail(rc, "lama_hip_pf_get_poses");
                for (uint32_t i = sh.lo_; i < sh.hi_; ++i)
                    if (owner((uint32_t)idx[i]) != r) sh.particles_[i].pose.state = SE2d::fromArray(&poses[4 * (size_t)(i - sh.lo_)]);
            }
        });
        xt_.local_copies = *std::max_element(local_s.begin(), local_s.end());   // resample copies inside a shard (any pool has them)
        xt_.import_ = now_s() - ts0 - xt_.local_copies;
        // this object's particle list follows the same permutation (histories included), :561-570
        std::vector<Particle> next(P);
        for (uint32_t i = 0; i < P; ++i) { next[i] = particles_[(uint32_t)idx[i]]; }
        particles_.swap(next);
        if (summary) summary->time_resampling.push_back(now_s() - tr0);
    }
    t0 = now_s();
    g.run([&](uint32_t r) { g.shards[r]->updateMaps(); if (summary) g.shards[r]->syncDevice(); });    // :289-302
    mirrorShards(due[0] != 0);
    xt_.phase_maps = now_s() - t0;
    if (summary) {
        summary->time_mapping.push_back(now_s() - t0);
        summary->time.push_back(now_s() - t_begin);
        summary->timestamp.push_back(timestamp);
        summary->memory.push_back((double)getMemoryUsage());
    }
    return true;
}

void PFSlam2D::fail(int32_t rc, const char* what) const
{
    char msg[512];
    std::snprintf(msg, sizeof(msg), "lama::PFSlam2D: %s failed (status %d): %s", what, rc, eng_->last_error(ctx_));
    throw std::runtime_error(msg);
}

double PFSlam2D::normal(double stddev)      // src/random.cpp:69-73: a fresh distribution per call
{
    std::normal_distribution<double> distribution(0.0, stddev);
    return distribution(gen_);
}

// src/pf_slam2d.cpp:365-391
void PFSlam2D::drawFromMotion(const Pose2D& delta, Pose2D& pose)
{
    double sigma, x, y, yaw;
    const double sxy = 0.3 * options_.stt;
    sigma = options_.stt * std::fabs(delta.x()) + options_.str * std::fabs(delta.rotation()) + sxy * std::fabs(delta.y());
    x = delta.x() + normal(sigma);
    sigma = options_.stt * std::fabs(delta.y()) + options_.str * std::fabs(delta.rotation()) + sxy * std::fabs(delta.x());
    y = delta.y() + normal(sigma);
    sigma = options_.srr * std::fabs(delta.rotation()) + options_.srt * delta.xy().norm();
    yaw = delta.rotation() + normal(sigma);
    yaw = std::fmod(yaw, 2 * M_PI);
    if (yaw > M_PI) yaw -= 2 * M_PI;
    pose += Pose2D(x, y, yaw);
}

// src/pf_slam2d.cpp:511-535
void PFSlam2D::normalize()
{
    const uint32_t P = options_.particles;
    const double gain = 1.0 / (options_.meas_sigma_gain * P);
    double max_l = particles_[0].weight;
    for (uint32_t i = 1; i < P; ++i)
        if (max_l < particles_[i].weight) max_l = particles_[i].weight;
    double sum = 0;
    for (uint32_t i = 0; i < P; ++i) {
        particles_[i].normalized_weight = std::exp(gain * (particles_[i].weight - max_l));
        sum += particles_[i].normalized_weight;
    }
    neff_ = 0;
    for (uint32_t i = 0; i < P; ++i) {
        particles_[i].normalized_weight /= sum;
        neff_ += particles_[i].normalized_weight * particles_[i].normalized_weight;
    }
    neff_ = 1.0 / neff_;
}

// src/pf_slam2d.cpp:537-556 (systematic resampling; `u01` is the single random::uniform() draw)
std::vector<int32_t> PFSlam2D::resampleIndices(double u01) const
{
    const uint32_t P = options_.particles;
    std::vector<int32_t> sample_idx(P);
    const double interval = 1.0 / (double)P;
    double target = interval * u01;
    double cw = 0.0;
    uint32_t n = 0;
    for (size_t i = 0; i < P; ++i) {
        cw += particles_[i].normalized_weight;
        // n < P: the reference would write past the array.  Slots the loop does not reach (rounding of the cumulative weight)
        // keep the vector's zero initialisation exactly as the reference's `std::vector<int32_t> sample_idx(num_particles)` does.
        while (cw > target && n < P) {
            sample_idx[n++] = (int32_t)i;
            target += interval;
        }
    }
    return sample_idx;
}

void PFSlam2D::scanToArrays(const PointCloudXYZ& s)
{
    pts_.resize(s.points.size() * 3);
    for (size_t i = 0; i < s.points.size(); ++i) {
        pts_[3 * i] = s.points[i].x(); pts_[3 * i + 1] = s.points[i].y(); pts_[3 * i + 2] = s.points[i].z();
    }
    origin_[0] = s.sensor_origin_.x(); origin_[1] = s.sensor_origin_.y(); origin_[2] = s.sensor_origin_.z();
    quat_[0] = s.sensor_orientation_.w(); quat_[1] = s.sensor_orientation_.x();
    quat_[2] = s.sensor_orientation_.y(); quat_[3] = s.sensor_orientation_.z();
}

void PFSlam2D::uploadLocalPoses()
{
    std::vector<double> buf((size_t)(hi_ - lo_) * 4);
    for (uint32_t i = lo_; i < hi_; ++i) particles_[i].pose.state.toArray(&buf[4 * (size_t)(i - lo_)]);
    const int32_t rc = eng_->pf_set_poses(ctx_, buf.data());
    if (rc) fail(rc, "lama_hip_pf_set_poses");
}

PFSlam2D::Phase PFSlam2D::updateBegin(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp)
{
    if (group_) throw std::runtime_error("lama::PFSlam2D: the step-wise API belongs to single shards; an object with Options::gpus > 1 runs the whole step in update()");
    if (!surface || surface->points.empty()) throw std::runtime_error("lama::PFSlam2D::update: empty scan");
    t_begin_ = now_s();
    last_timestamp_ = timestamp;
    scan_resident_ = false;
    dropMapViews();
    current_surface_ = surface;
    scanToArrays(*surface);
    const uint32_t n = (uint32_t)surface->points.size();
    const uint32_t P = options_.particles;

    if (!has_first_scan) {                                                  // src/pf_slam2d.cpp:185-228
        odom_ = odometry;
        timestamps_.push_back(timestamp);
        for (uint32_t i = 0; i < P; ++i) {
            particles_[i].poses.push_back(pose_);
            particles_[i].pose = pose_;
            particles_[i].weight = 0.0;
            particles_[i].weight_sum = 0.0;
        }
        double p0[4];
        pose_.state.toArray(p0);
        const int32_t rc = eng_->pf_init(ctx_, pts_.data(), n, origin_, quat_, p0);
        if (rc) fail(rc, "lama_hip_pf_init");
        has_first_scan = true;
        if (summary) {
            const double el = now_s() - t_begin_;
            summary->timestamp.push_back(timestamp);
            summary->time.push_back(el);
            summary->time_mapping.push_back(el);
            summary->memory.push_back((double)getMemoryUsage());
        }
        return kFirstScan;
    }

    // 1. predict from odometry: ALL particles in order, so the RNG stream does not depend on the sharding
    Pose2D odelta = odom_ - odometry;                                       // :231
    odom_ = odometry;
    for (uint32_t i = 0; i < P; ++i) drawFromMotion(odelta, particles_[i].pose);

    acc_trans_ += odelta.xy().norm();                                       // :239-243
    acc_rot_ += std::fabs(odelta.rotation());
    if (acc_trans_ <= options_.trans_thresh && acc_rot_ <= options_.rot_thresh) return kNoUpdate;
    acc_trans_ = 0;
    acc_rot_ = 0;

    // 2. scan matching of the local shard on the device                    :252-266, :416-437
    const double t0 = now_s();
    uploadLocalPoses();
    std::vector<double> poses((size_t)(hi_ - lo_) * 4);
    const int32_t rc = eng_->pf_scan_match(ctx_, pts_.data(), n, origin_, quat_, poses.data(), local_loglik_.data(), nullptr);
    if (rc) fail(rc, "lama_hip_pf_scan_match");
    scan_resident_ = true;
    for (uint32_t i = lo_; i < hi_; ++i) {
        particles_[i].pose.state = SE2d::fromArray(&poses[4 * (size_t)(i - lo_)]);
        if (options_.shard_world == 1) particles_[i].poses.push_back(particles_[i].pose);
    }
    t_solve_ = now_s() - t0;
    if (summary) summary->time_solving.push_back(t_solve_);
    return kMatched;
}

bool PFSlam2D::planResample(const double* all_loglik, std::vector<int32_t>& sample_idx)
{
    if (group_) throw std::runtime_error("lama::PFSlam2D: the step-wise API belongs to single shards; an object with Options::gpus > 1 runs the whole step in update()");
    const uint32_t P = options_.particles;
    const double t0 = now_s();
    for (uint32_t i = 0; i < P; ++i) {                                      // :434-436
        particles_[i].weight_sum += all_loglik[i];
        particles_[i].weight += all_loglik[i];
    }
    normalize();                                                            // :274
    if (summary) summary->time_normalizing.push_back(now_s() - t0);
    if (!(neff_ < (P * 0.5))) return false;                                 // :280
    const double u = std::uniform_real_distribution<double>(0.0, 1.0)(gen_);   // src/random.cpp:51-55
    sample_idx = resampleIndices(u);
    return true;
}

void PFSlam2D::applyResample(const std::vector<int32_t>& sample_idx)
{
    if (group_) throw std::runtime_error("lama::PFSlam2D: the step-wise API belongs to single shards; an object with Options::gpus > 1 runs the whole step in update()");
    const uint32_t P = options_.particles;
    const double t0 = now_s();
    std::vector<Particle> next(P);
    for (uint32_t i = 0; i < P; ++i) {                                      // :561-570
        const uint32_t idx = (uint32_t)sample_idx[i];
        next[i] = particles_[idx];
        next[i].weight = 0.0;
        next[i].weight_sum = particles_[idx].weight_sum;
    }
    // device copies for the local block; remote sources get a placeholder (self) and are imported afterwards
    std::vector<int32_t> local(hi_ - lo_);
    for (uint32_t i = lo_; i < hi_; ++i) {
        const uint32_t src = (uint32_t)sample_idx[i];
        local[i - lo_] = ownsParticle(src) ? (int32_t)(src - lo_) : (int32_t)(i - lo_);
    }
    const int32_t rc = eng_->pf_resample(ctx_, local.data());
    if (rc) fail(rc, "lama_hip_pf_resample");
    particles_.swap(next);
    ++num_resamples_;
    if (summary) summary->time_resampling.push_back(now_s() - t0);
}

void PFSlam2D::updateMaps()
{
    if (group_) throw std::runtime_error("lama::PFSlam2D: the step-wise API belongs to single shards; an object with Options::gpus > 1 runs the whole step in update()");
    const double t0 = now_s();
    dropMapViews();
    const uint32_t n = (uint32_t)(pts_.size() / 3);
    // the scan is already resident on the device (uploaded by scan_match of this update)
    // queued, not awaited: the host part of the next scan overlaps with the kernels; the status is collected by the next call
    // on the context (a deferred device error makes THAT call fail)
    const int32_t rc = eng_->pf_update_maps_begin(ctx_, scan_resident_ ? nullptr : pts_.data(), n, origin_, quat_);      // :289-302
    if (rc) fail(rc, "lama_hip_pf_update_maps");
    if (summary) {
        // the Summary's per-update buckets need the real duration: wait for the kernels here
        const int32_t rs = eng_->sync(ctx_);
        if (rs) fail(rs, "lama_hip_sync");
        summary->time_mapping.push_back(now_s() - t0);
        summary->time.push_back(now_s() - t_begin_);
        summary->timestamp.push_back(last_timestamp_);              // probeStamp(timestamp), src/pf_slam2d.cpp:306
        summary->memory.push_back((double)getMemoryUsage());
    }
}

bool PFSlam2D::update(const PointCloudXYZ::Ptr& surface, const Pose2D& odometry, double timestamp)
{
    if (group_) return updateGroup(surface, odometry, timestamp);
    if (options_.shard_world != 1)
        throw std::runtime_error("lama::PFSlam2D::update: with shard_world > 1 drive the step-wise API (all-gather needed)");
    const Phase ph = updateBegin(surface, odometry, timestamp);
    if (ph == kNoUpdate) return false;
    if (ph == kFirstScan) return true;
    std::vector<int32_t> idx;
    if (planResample(local_loglik_.data(), idx)) applyResample(idx);
    updateMaps();
    return true;
}

size_t PFSlam2D::getBestParticleIdx() const                                 // :314-330
{
    size_t best_idx = 0;
    double best_ws = particles_[0].weight_sum;
    for (uint32_t i = 1; i < options_.particles; ++i)
        if (best_ws < particles_[i].weight_sum) { best_ws = particles_[i].weight_sum; best_idx = i; }
    return best_idx;
}

Pose2D PFSlam2D::getPose() const { return particles_[getBestParticleIdx()].pose; }

uint64_t PFSlam2D::getMemoryUsage() const
{
    if (group_) { uint64_t t = 0; for (auto& sh : group_->shards) t += sh->getMemoryUsage(); return t; }
    lama_hip_counters c;
    if (eng_->get_counters(ctx_, &c) != 0) return 0;
    // patch payloads in the reference's record sizes (Container::memory, src/sdm/container.cpp:92-95)
    return c.dm_patches * 10240ull + c.occ_patches * 4096ull;
}

uint64_t PFSlam2D::getMemoryUsage(uint64_t& occmem, uint64_t& dmmem) const
{
    occmem = 0; dmmem = 0;
    if (!has_first_scan) return 0;
    if (group_) return group_->shards[0]->getMemoryUsage(occmem, dmmem);
    uint32_t nd = 0, no = 0;
    if (eng_->pf_map_patches(ctx_, 0, LAMA_HIP_MAP_DISTANCE, &nd) != 0 || eng_->pf_map_patches(ctx_, 0, LAMA_HIP_MAP_OCCUPANCY, &no) != 0) return 0;
    occmem = (uint64_t)options_.particles * no * 4096ull;
    dmmem = (uint64_t)options_.particles * nd * 10240ull;
    return occmem + dmmem;
}

static bool download(const HipEngine* e, lama_hip_ctx* ctx, uint32_t particle, int kind, size_t cell_bytes,
                     std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks)
{
    uint32_t n = 0;
    if (e->pf_map_patches(ctx, particle, kind, &n) != 0) return false;
    ids.assign(n, 0);
    cells.assign((size_t)n * cell_bytes * 1024, 0);
    masks.assign((size_t)n * 16, 0);
    uint32_t got = 0;
    return e->pf_download_map(ctx, particle, kind, n, ids.data(), cells.data(), masks.data(), &got) == 0 && got == n;
}

// Any particle's maps (the reference's public Particle::dm / Particle::occ, include/lama/pf_slam2d.h:83-84) in the reference's
// record formats.  False before the first scan, for i >= P, or when particle i lives on another process' shard.
bool PFSlam2D::downloadParticleDistanceMap(size_t i, std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const
{
    if (!has_first_scan || i >= particles_.size()) return false;
    if (group_) return ownerOf((uint32_t)i)->downloadParticleDistanceMap(i, ids, cells, masks);
    if (!ownsParticle((uint32_t)i)) return false;
    return download(eng_.get(), ctx_, (uint32_t)i - lo_, LAMA_HIP_MAP_DISTANCE, 10, ids, cells, masks);
}

bool PFSlam2D::downloadParticleOccupancyMap(size_t i, std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const
{
    if (!has_first_scan || i >= particles_.size()) return false;
    if (group_) return ownerOf((uint32_t)i)->downloadParticleOccupancyMap(i, ids, cells, masks);
    if (!ownsParticle((uint32_t)i)) return false;
    return download(eng_.get(), ctx_, (uint32_t)i - lo_, LAMA_HIP_MAP_OCCUPANCY, 4, ids, cells, masks);
}

// the best particle's (its replicated weights name the same best particle on every shard)
bool PFSlam2D::downloadDistanceMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const
{
    return has_first_scan && downloadParticleDistanceMap(getBestParticleIdx(), ids, cells, masks);
}

bool PFSlam2D::downloadOccupancyMap(std::vector<uint64_t>& ids, std::vector<uint8_t>& cells, std::vector<uint64_t>& masks) const
{
    return has_first_scan && downloadParticleOccupancyMap(getBestParticleIdx(), ids, cells, masks);
}

std::shared_ptr<const FrequencyOccupancyMap> PFSlam2D::getParticleOccupancyMap(size_t i) const
{
    sdm::HostMap m;
    m.kind = sdm::kFrequencyOccupancyMap; m.resolution = options_.resolution;
    if (!downloadParticleOccupancyMap(i, m.ids, m.cells, m.masks)) return nullptr;
    return std::make_shared<const FrequencyOccupancyMap>(std::move(m));
}

std::shared_ptr<const DynamicDistanceMap> PFSlam2D::getParticleDistanceMap(size_t i) const
{
    sdm::HostMap m;
    m.kind = sdm::kDistanceMap; m.resolution = options_.resolution;
    const uint32_t r = (uint32_t)std::ceil(options_.l2_max * (1.0 / options_.resolution));    // DynamicDistanceMap::setMaxDistance :149-153
    m.max_sqdist = r * r;
    if (!downloadParticleDistanceMap(i, m.ids, m.cells, m.masks)) return nullptr;
    // A snapshot the CALLER owns is never bound to the device: the slot (ctx, i - lo_) holds another particle's map after the next
    // update() / resample, and the context dies with this object -- a lama::MatchSurface2D built on the snapshot would then
    // evaluate against the wrong map or a freed context.  Such a problem is refused by MatchSurface2D ("the
    // distance map does not live on the device"); getDistanceMap()'s view, which update() drops, is the one that is bound.
    return std::make_shared<DynamicDistanceMap>(std::move(m));
}

// src/pf_slam2d.cpp:49-104 (same buckets; plain loops instead of Eigen::Map)
std::string PFSlam2D::Summary::report() const
{
    auto stats = [](const DynamicArray<double>& v, double out[4]) {
        out[0] = out[1] = out[2] = out[3] = 0;
        if (v.empty()) return;
        double s = 0, mn = v[0], mx = v[0];
        for (double x : v) { s += x; mn = std::min(mn, x); mx = std::max(mx, x); }
        const double mean = s / v.size();
        double ss = 0;
        for (double x : v) ss += (x - mean) * (x - mean);
        out[0] = mean * 1e3; out[1] = (v.size() > 1 ? std::sqrt(ss / (v.size() - 1)) : 0.0) * 1e3; out[2] = mn * 1e3; out[3] = mx * 1e3;
    };
    double t[4], ts[4], tn[4], tr[4], tm[4];
    stats(time, t); stats(time_solving, ts); stats(time_normalizing, tn); stats(time_resampling, tr); stats(time_mapping, tm);
    double span = 0;
    for (double x : time) span += x;
    const double stampdiff = timestamp.empty() ? 0.0 : timestamp.back() - timestamp.front();
    double maxmem = 0;
    for (double m : memory) maxmem = std::max(maxmem, m);
    char buf[2048];
    std::snprintf(buf, sizeof(buf),
                  "\n LaMa PF Slam2D (MI355X) - Report\n ================================\n"
                  " Number of updates     %zu\n Number of resamples   %zu\n Max memory usage      %.2f MiB\n"
                  " Execution time span   %.3f s\n Execution frequency   %.2f Hz\n Realtime factor       %.2fx\n"
                  "\n Execution time (mean +- std [min, max]) in milliseconds\n"
                  " --------------------------------------------------------\n"
                  " Update          %f +- %f [%f, %f]\n   Optimization  %f +- %f [%f, %f]\n   Normalizing   %f +- %f [%f, %f]\n"
                  "   Resampling    %f +- %f [%f, %f]\n   Mapping       %f +- %f [%f, %f]\n",
                  time.size(), time_resampling.size(), maxmem / 1024.0 / 1024.0, span,
                  time.empty() ? 0.0 : 1.0 / (span / time.size()), span > 0 ? stampdiff / span : 0.0,
                  t[0], t[1], t[2], t[3], ts[0], ts[1], ts[2], ts[3], tn[0], tn[1], tn[2], tn[3],
                  tr[0], tr[1], tr[2], tr[3], tm[0], tm[1], tm[2], tm[3]);
    return std::string(buf);
}

} // namespace lama
