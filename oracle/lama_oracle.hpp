// =====================================================================================
// lama_oracle.hpp -- CPU restatement of the iris_lama (LaMa v1.3.1) particle-filter
// scan-matching path.  TEST INFRASTRUCTURE ONLY.
//
//   * This file is the parity CHECKER.  Only tests/, __graft_entry__.smoke() and
//     bench.py's `cpu_baseline` leg may build, link, load or call it.  Nothing under
//     iris_lama_amd/ (the product) includes or links anything from oracle/.
//   * PARITY UNPINNED: the reference ships no tests / golden vectors / fixtures
//      and cannot be compiled in this image because Eigen3 is absent
//     (SURVEY.md F4).  The restatement below follows the cited reference lines
//     statement by statement and is pinned only against the source-derived known
//     answers of SURVEY.md Appendix A.9 (tests/test_oracle_kat.py) and against
//     brute-force self-consistency checks.
//   * Third-party arithmetic the reference uses on this path and that is absent here:
//     Eigen3 (>=3.3, CMakeLists.txt:15; CI used Ubuntu 20.04 libeigen3-dev = 3.3.7).
//     Its published algorithms are restated where used (AngleAxis/Quaternion ->
//     rotation matrix, Affine composition, pivoted LDLT and its solve).  Summation
//     order inside Eigen's dynamic-size products (J^T r, J^T J over 1080 rows) is
//     implementation-defined (packet width / unrolling); this file sums sequentially.
//     Expected oracle-vs-reference differences: ~1e-13 relative in g/A, none in
//     integer map contents for identical poses.
//   * libstdc++ pieces used by the reference are used here directly, so they behave
//     identically: std::mt19937, std::normal_distribution, std::uniform_real_distribution
//     (src/random.cpp:38-73) and std::priority_queue tie order
//     (include/lama/sdm/dynamic_distance_map.h:90-98).
//
// All `file:line` citations are relative to /root/reference.
// Plain C++14, no dependencies.
// =====================================================================================
#pragma once

#include <array>
#include <limits>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <functional>
#include <memory>
#include <mutex>
#include <queue>
#include <random>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace orc {

// -------------------------------------------------------------------------------------
// small PODs
// -------------------------------------------------------------------------------------
struct V3d { double x, y, z; };
struct V3u { uint32_t x, y, z; };
struct V3l { int64_t x, y, z; };

inline bool operator==(const V3u& a, const V3u& b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// -------------------------------------------------------------------------------------
// SE2 = unit complex (c,s) + translation.  include/lama/sophus/so2.hpp, se2.hpp,
// src/pose2d.cpp.  
// -------------------------------------------------------------------------------------
struct SE2 {
    double c = 1.0, s = 0.0;   // unit complex (so2.hpp:474 default = identity)
    double tx = 0.0, ty = 0.0; // se2.hpp:611 default translation zero
};

// so2.hpp:205-214  normalize(); throws SophusException on (near) zero -- we abort.
inline void so2_normalize(double& c, double& s)
{
    double length = std::sqrt(c * c + s * s);
    if (length < 1e-10) { std::abort(); }          // sophus.hpp:37-39 epsilon
    c /= length;
    s /= length;
}

// so2.hpp:322-324 exp(theta) = SO2Group(cos, sin) ; ctor so2.hpp:506-509 normalizes
inline void so2_exp(double theta, double& c, double& s)
{
    c = std::cos(theta);
    s = std::sin(theta);
    so2_normalize(c, s);
}

// se2.hpp:648-651  SE2Group(theta, translation) -> so2_(theta) (so2.hpp:537-539)
inline SE2 se2_from_xyr(double x, double y, double theta)
{
    SE2 r;
    so2_exp(theta, r.c, r.s);
    r.tx = x; r.ty = y;
    return r;
}

// so2.hpp:401-404 log() = atan2(imag, real) ; src/pose2d.cpp:118-121 rotation()
inline double se2_rotation(const SE2& a) { return std::atan2(a.s, a.c); }

// se2.hpp:262-265 operator*= : fastMultiply (se2.hpp:154-157, so2.hpp:168-176) then
// normalize (se2.hpp:189-191).  operator* (se2.hpp:233-237) copies then *=.
inline SE2 se2_mul(const SE2& a, const SE2& b)
{
    SE2 r = a;
    // translation() += so2()*(other.translation())      so2.hpp:262-266
    r.tx += a.c * b.tx - a.s * b.ty;
    r.ty += a.s * b.tx + a.c * b.ty;
    // complex multiplication                              so2.hpp:168-176
    double lhs_real = a.c, lhs_imag = a.s;
    r.c = lhs_real * b.c - lhs_imag * b.s;
    r.s = lhs_real * b.s + lhs_imag * b.c;
    so2_normalize(r.c, r.s);
    return r;
}

// se2.hpp:163-167 inverse(): invR = SO2(real,-imag) [2-arg ctor normalizes];
// translation = invR * (t * -1)
inline SE2 se2_inverse(const SE2& a)
{
    SE2 r;
    r.c = a.c; r.s = -a.s;
    so2_normalize(r.c, r.s);
    double mx = a.tx * -1.0, my = a.ty * -1.0;
    r.tx = r.c * mx - r.s * my;
    r.ty = r.s * mx + r.c * my;
    return r;
}

// se2.hpp:389-411 exp([vx,vy,theta])
inline SE2 se2_exp(double vx, double vy, double theta)
{
    SE2 r;
    so2_exp(theta, r.c, r.s);
    double sin_theta_by_theta, one_minus_cos_theta_by_theta;
    if (std::abs(theta) < 1e-10) {
        double theta_sq = theta * theta;
        sin_theta_by_theta = 1. - (1. / 6.) * theta_sq;
        one_minus_cos_theta_by_theta = 0.5 * theta - (1. / 24.) * theta * theta_sq;
    } else {
        sin_theta_by_theta = r.s / theta;
        one_minus_cos_theta_by_theta = (1. - r.c) / theta;
    }
    r.tx = sin_theta_by_theta * vx - one_minus_cos_theta_by_theta * vy;
    r.ty = one_minus_cos_theta_by_theta * vx + sin_theta_by_theta * vy;
    return r;
}

// src/pose2d.cpp:76-96 : a + b = a*b ; a - b = a^-1 * b
inline SE2 pose_plus(const SE2& a, const SE2& b) { return se2_mul(a, b); }
inline SE2 pose_minus(const SE2& a, const SE2& b) { return se2_mul(se2_inverse(a), b); }

// -------------------------------------------------------------------------------------
// Eigen::Affine3d restated (linear 3x3 + translation).
// -------------------------------------------------------------------------------------
struct Affine3 {
    double R[3][3];
    double t[3];
};

// Eigen QuaternionBase::toRotationMatrix (published algorithm, Eigen 3.3 Quaternion.h)
inline void quat_to_matrix(const double q[4] /*w,x,y,z*/, double R[3][3])
{
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0][0] = 1.0 - (tyy + tzz); R[0][1] = txy - twz;         R[0][2] = txz + twy;
    R[1][0] = txy + twz;         R[1][1] = 1.0 - (txx + tzz); R[1][2] = tyz - twx;
    R[2][0] = txz - twy;         R[2][1] = tyz + twx;         R[2][2] = 1.0 - (txx + tyy);
}

// Eigen AngleAxis::toRotationMatrix for axis = UnitZ (published algorithm, AngleAxis.h)
inline void angle_axis_z(double angle, double R[3][3])
{
    const double ax[3] = {0.0, 0.0, 1.0};
    const double sn = std::sin(angle), c = std::cos(angle);
    const double sin_axis[3] = {sn * ax[0], sn * ax[1], sn * ax[2]};
    const double cos1_axis[3] = {(1.0 - c) * ax[0], (1.0 - c) * ax[1], (1.0 - c) * ax[2]};
    double tmp;
    tmp = cos1_axis[0] * ax[1]; R[0][1] = tmp - sin_axis[2]; R[1][0] = tmp + sin_axis[2];
    tmp = cos1_axis[0] * ax[2]; R[0][2] = tmp + sin_axis[1]; R[2][0] = tmp - sin_axis[1];
    tmp = cos1_axis[1] * ax[2]; R[1][2] = tmp - sin_axis[0]; R[2][1] = tmp + sin_axis[0];
    R[0][0] = cos1_axis[0] * ax[0] + c;
    R[1][1] = cos1_axis[1] * ax[1] + c;
    R[2][2] = cos1_axis[2] * ax[2] + c;
}

// Translation3d(t) * rotation  -> linear = R, translation = t
inline Affine3 affine_from(const double t[3], const double R[3][3])
{
    Affine3 a;
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) a.R[i][j] = R[i][j]; a.t[i] = t[i]; }
    return a;
}

// Affine * Affine: linear = A.R*B.R ; translation = A.R*B.t + A.t  (sums in index order)
inline Affine3 affine_mul(const Affine3& A, const Affine3& B)
{
    Affine3 r;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j)
            r.R[i][j] = (A.R[i][0] * B.R[0][j] + A.R[i][1] * B.R[1][j]) + A.R[i][2] * B.R[2][j];
        r.t[i] = ((A.R[i][0] * B.t[0] + A.R[i][1] * B.t[1]) + A.R[i][2] * B.t[2]) + A.t[i];
    }
    return r;
}

// Affine * point = linear*p + translation
inline V3d affine_apply(const Affine3& A, const V3d& p)
{
    V3d r;
    r.x = ((A.R[0][0] * p.x + A.R[0][1] * p.y) + A.R[0][2] * p.z) + A.t[0];
    r.y = ((A.R[1][0] * p.x + A.R[1][1] * p.y) + A.R[1][2] * p.z) + A.t[1];
    r.z = ((A.R[2][0] * p.x + A.R[2][1] * p.y) + A.R[2][2] * p.z) + A.t[2];
    return r;
}

// include/lama/types.h:111-120 PointCloudXYZ
struct Scan {
    std::vector<V3d> points;
    double sensor_origin[3] = {0, 0, 0};
    double sensor_orientation[4] = {1, 0, 0, 0}; // w,x,y,z
};

// moving_tf = Translation3d(sensor_origin_) * sensor_orientation_
// (src/match_surface_2d.cpp:49, src/pf_slam2d.cpp:397,444)
inline Affine3 moving_tf(const Scan& s)
{
    double R[3][3];
    quat_to_matrix(s.sensor_orientation, R);
    return affine_from(s.sensor_origin, R);
}

// fixed_tf = Translation3d(x,y,0) * AngleAxisd(rotation, UnitZ)
// (src/match_surface_2d.cpp:51-54, src/pf_slam2d.cpp:399-402,445)
inline Affine3 fixed_tf(const SE2& pose)
{
    double R[3][3];
    angle_axis_z(se2_rotation(pose), R);
    const double t[3] = {pose.tx, pose.ty, 0.0};
    return affine_from(t, R);
}

// -------------------------------------------------------------------------------------
// Container  (include/lama/sdm/container.h:47-123,167-183; src/sdm/container.cpp:39-95)
// 2-D only: SIZE = 1 << (2*log2dim).
// -------------------------------------------------------------------------------------
struct Container {
    uint32_t element_size = 0;
    uint32_t SIZE = 0;
    uint32_t WORD_COUNT = 0;
    std::vector<uint8_t> data;   // calloc'd: zero-initialised (container.cpp:78)
    std::vector<uint64_t> mask;  // calloc'd (container.cpp:82)

    Container(uint32_t log2dim, uint32_t elem)
        : element_size(elem), SIZE(1u << (2 * log2dim)), WORD_COUNT(std::max(uint32_t(SIZE >> 6), uint32_t(1))),
          data(size_t(SIZE) * elem, 0), mask(WORD_COUNT, 0) {}

    bool is_on(uint32_t i) const { return 0 != (mask[i >> 6] & (uint64_t(1) << (i & 63))); }
    void set_on(uint32_t i) { mask[i >> 6] |= (uint64_t(1) << (i & 63)); }

    // container.h:102-106  non-const get sets the mask bit
    uint8_t* get(uint32_t idx)
    {
        if (!is_on(idx)) set_on(idx);
        return data.data() + size_t(idx) * element_size;
    }
    // container.h:119-123  const get returns nullptr when the bit is off
    const uint8_t* get(uint32_t idx) const
    {
        if (!is_on(idx)) return nullptr;
        return data.data() + size_t(idx) * element_size;
    }
};

// per-(particle,scan) instrumentation used for the roofline model
struct TouchSets {
    bool enabled = false;
    std::unordered_set<uint64_t> s_match, s_occ, s_bf;
    void clear() { s_match.clear(); s_occ.clear(); s_bf.clear(); }
};

// -------------------------------------------------------------------------------------
// Map  (include/lama/sdm/map.h:60-189 ; src/sdm/map.cpp:42-107,198-227,371-455)
// LRU/compression/IO are out of scope (default use_compression=false, pf_slam2d.h:178).
// -------------------------------------------------------------------------------------
class Map {
public:
    static const uint64_t UNIVERSAL_CONSTANT = 2642244; // map.h:68

    double resolution;
    double scale;
    size_t cell_memory_size;
    uint32_t patch_length;
    uint32_t patch_volume;
    int log2dim;
    double off; // translation of tf_ on every axis: (UNIVERSAL_CONSTANT>>1)*patch_length  map.cpp:55-58

    // map.h:109 ; value = shared (copy-on-write) container, include/lama/cow_ptr.h
    std::unordered_map<uint64_t, std::shared_ptr<Container>> patches;

    // optional: which patch ids non-const get() touched (brushfire / raycast footprint)
    std::unordered_set<uint64_t>* touch_rw = nullptr;
    // optional: which (existing) patch ids const get() touched (match footprint)
    mutable std::unordered_set<uint64_t>* touch_ro = nullptr;

    Map(double res, size_t cell_size, uint32_t patch_size)
        : resolution(res), scale(1.0 / res), cell_memory_size(cell_size),
          patch_length(1u << ((int)std::log2((double)patch_size))),
          patch_volume(patch_length * patch_length)
    {
        log2dim = (int)std::log2((double)patch_length);
        off = double(UNIVERSAL_CONSTANT >> 1) * double(patch_length);
    }

    // map.cpp:72-107 copy: patches share containers (COW)
    Map(const Map& o)
        : resolution(o.resolution), scale(o.scale), cell_memory_size(o.cell_memory_size),
          patch_length(o.patch_length), patch_volume(o.patch_volume), log2dim(o.log2dim), off(o.off),
          patches(o.patches)
    {}
    virtual ~Map() {}

    // tf_ * v with tf_ = Translation(off) * Scaling(scale)  (map.cpp:58)
    // (the off-diagonal zeros of the Scaling only ever add +-0.0)
    inline double tfc(double v) const { return scale * v + off; }

    // map.h:125-126
    inline V3u w2m(const V3d& p) const
    {
        V3u r;
        r.x = (uint32_t)(tfc(p.x) + 0.5);
        r.y = (uint32_t)(tfc(p.y) + 0.5);
        r.z = (uint32_t)(tfc(p.z) + 0.5);
        return r;
    }
    // map.h:137-138
    inline V3d w2m_nocast(const V3d& p) const { return V3d{tfc(p.x), tfc(p.y), tfc(p.z)}; }

    // map.h:153-161 (2-D branch)
    inline uint64_t m2p(const V3u& c) const
    {
        return uint64_t(c.x >> log2dim) * UNIVERSAL_CONSTANT + uint64_t(c.y >> log2dim);
    }
    // map.h:182-189 (2-D: MASK3D = 0)
    inline uint32_t m2c(const V3u& c) const
    {
        const uint32_t m = ((1u << log2dim) - 1);
        return (c.x & m) | ((c.y & m) << log2dim);
    }
    // map.h:166-177 (2-D)
    inline V3u p2m(uint64_t idx) const
    {
        return V3u{uint32_t((idx / UNIVERSAL_CONSTANT) << log2dim), uint32_t((idx % UNIVERSAL_CONSTANT) << log2dim), 0};
    }

    // map.h:147-148: tf_inv_ * m with tf_inv_ = tf_.inverse() (map.cpp:59).  Eigen's affine inverse of
    // Translation(off) * Scaling(s): linear = 3x3 cofactor inverse of diag(s,s,s) -> (s*s) * (1 / ((s*s)*s)),
    // translation = -(linear * off)  (published algorithm, Eigen 3.3 Inverse_SSE/InverseImpl.h compute_inverse_size3)
    inline double tf_inv_linear() const { return (scale * scale) * (1.0 / ((scale * scale) * scale)); }
    inline V3d m2w(const V3u& c) const
    {
        const double l = tf_inv_linear(), t = -(l * off);
        return V3d{l * (double)c.x + t, l * (double)c.y + t, l * (double)c.z + t};
    }
    // map.cpp:139-157 (integer bounds of the allocated patches) and map.h:221-225 (world bounds)
    void bounds(V3u& mn, V3u& mx) const
    {
        mn = V3u{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        mx = V3u{0, 0, 0};
        for (auto& kv : patches) {
            const V3u a = p2m(kv.first);
            mn.x = std::min(mn.x, a.x); mn.y = std::min(mn.y, a.y); mn.z = std::min(mn.z, a.z);
            mx.x = std::max(mx.x, a.x); mx.y = std::max(mx.y, a.y); mx.z = std::max(mx.z, a.z);
        }
        mx.x += patch_length; mx.y += patch_length; mx.z += patch_length;
    }
    void bounds(V3d& mn, V3d& mx) const
    {
        V3u a, b; bounds(a, b);
        mn = m2w(a); mx = m2w(b);
    }

    // map.cpp:489-575 Map::write / Map::read (uncompressed containers; IOHeader map.h:95-103 written as it lies in
    // memory: 32 bytes on LP64) + Container::write/read (container.cpp:143-176: cells then the mask words)
    struct IOHeader { uint32_t magic; uint16_t version; uint32_t cell_size; uint32_t patch_length; size_t num_patches; float resolution; bool is_3d; };
    virtual void writeParameters(std::ofstream&) const {}
    virtual void readParameters(std::ifstream&) {}
    bool write(const std::string& filename) const
    {
        std::ofstream f(filename.c_str(), std::ios::out | std::ios::binary | std::ios::trunc);
        if (!f.is_open()) return false;
        IOHeader header;
        std::memset(&header, 0, sizeof(header));
        header.magic = 0x6d64732e; header.version = 0x0103; header.cell_size = (uint32_t)cell_memory_size;
        header.patch_length = patch_length; header.num_patches = patches.size(); header.resolution = (float)resolution; header.is_3d = false;
        f.write((char*)&header, sizeof(IOHeader));
        if (!f) return false;
        this->writeParameters(f);
        for (auto it = patches.begin(); it != patches.end(); ++it) {
            f.write((char*)&(it->first), sizeof(uint64_t));
            f.write((const char*)it->second->data.data(), (std::streamsize)it->second->data.size());
            f.write((const char*)it->second->mask.data(), (std::streamsize)(sizeof(uint64_t) * it->second->mask.size()));
        }
        f.close();
        return true;
    }
    bool read(const std::string& filename)
    {
        std::ifstream f(filename.c_str(), std::ios::in | std::ios::binary);
        if (!f.is_open()) return false;
        IOHeader header;
        f.read((char*)&header, sizeof(IOHeader));
        if (!f) return false;
        if (header.magic != 0x6d64732e || header.version != 0x0103) return false;
        if ((header.cell_size != cell_memory_size) || header.is_3d) return false;
        resolution = header.resolution;
        scale = 1.0 / resolution;
        patch_length = header.patch_length;
        patch_volume = patch_length * patch_length;
        log2dim = (int)std::log2((double)patch_length);
        off = double(UNIVERSAL_CONSTANT >> 1) * double(patch_length);
        this->readParameters(f);
        for (size_t i = 0; i < header.num_patches; ++i) {
            uint64_t idx;
            f.read((char*)&idx, sizeof(idx));
            if (!f) return false;
            auto c = std::make_shared<Container>((uint32_t)log2dim, (uint32_t)cell_memory_size);
            f.read((char*)c->data.data(), (std::streamsize)c->data.size());
            f.read((char*)c->mask.data(), (std::streamsize)(sizeof(uint64_t) * c->mask.size()));
            patches[idx] = c;
        }
        prev_patch_ = nullptr; prev_idx_ = ~uint64_t(0);
        return true;
    }
    // map.cpp:465-488 deletePatchAt (no LRU here), map.cpp:361-367 visit_all_patches
    bool deletePatchAt(const V3u& c)
    {
        auto it = patches.find(m2p(c));
        if (it == patches.end()) return false;
        patches.erase(it);
        prev_patch_ = nullptr; prev_idx_ = ~uint64_t(0);
        return true;
    }
    template <typename F>
    void visit_all_patches(F&& walker) const { for (auto& kv : patches) walker(p2m(kv.first)); }
    // map.cpp:352-359 visit_all_cells: every cell whose mask bit is on
    template <typename F>
    void visit_all_cells(F&& walker) const
    {
        for (auto& kv : patches) {
            const V3u anchor = p2m(kv.first);
            for (uint32_t c = 0; c < patch_volume; ++c)
                if ((kv.second->mask[c >> 6] >> (c & 63)) & 1ull)
                    walker(V3u{anchor.x + (c & (patch_length - 1)), anchor.y + (c >> log2dim), anchor.z});
        }
    }

    // map.cpp:371-412 (non-compressed branch) + COWPtr::operator-> detach (cow_ptr.h:86-114)
    uint8_t* get(const V3u& c)
    {
        uint64_t idx = m2p(c);
        if (prev_idx_ != idx || prev_patch_ == nullptr) {
            auto it = patches.find(idx);
            if (it == patches.end())
                it = patches.insert(std::make_pair(idx, std::make_shared<Container>(log2dim, (uint32_t)cell_memory_size))).first;
            prev_idx_ = idx;
            prev_patch_ = &(it->second);
            if (touch_rw) touch_rw->insert(idx);
        }
        if (prev_patch_->use_count() > 1)           // detach(): deep copy a shared patch
            *prev_patch_ = std::make_shared<Container>(**prev_patch_);
        return (*prev_patch_)->get(m2c(c));
    }

    // map.cpp:414-455 (non-compressed branch)
    const uint8_t* get(const V3u& c) const
    {
        uint64_t idx = m2p(c);
        if (prev_idx_ != idx) {
            auto it = patches.find(idx);
            if (it == patches.end()) {
                prev_idx_ = idx;
                prev_patch_ = nullptr;
                return nullptr;
            }
            prev_idx_ = idx;
            prev_patch_ = const_cast<std::shared_ptr<Container>*>(&(it->second));
            if (touch_ro) touch_ro->insert(idx);
        } else if (prev_patch_ == nullptr) {
            return nullptr;
        }
        return static_cast<const Container*>(prev_patch_->get())->get(m2c(c));
    }

    // map.cpp:198-227  integer Bresenham, both end points excluded
    template <class F>
    void computeRay(const V3u& from, const V3u& to, F&& callback)
    {
        if (from == to) return;
        int64_t error[3] = {0, 0, 0};
        int64_t coord[3] = {(int64_t)from.x, (int64_t)from.y, (int64_t)from.z};
        int64_t delta[3] = {(int64_t)to.x - coord[0], (int64_t)to.y - coord[1], (int64_t)to.z - coord[2]};
        int64_t step[3];
        for (int j = 0; j < 3; ++j) step[j] = (delta[j] < 0) ? -1 : 1;
        for (int j = 0; j < 3; ++j) delta[j] = std::llabs(delta[j]);
        int n = (int)std::max(delta[0], std::max(delta[1], delta[2]));
        for (int i = 0; i < n - 1; ++i) {
            for (int j = 0; j < 3; ++j) error[j] += delta[j];
            for (int j = 0; j < 3; ++j) {
                if ((error[j] << 1) < n) continue;
                coord[j] += step[j];
                error[j] -= n;
            }
            callback(V3u{(uint32_t)coord[0], (uint32_t)coord[1], (uint32_t)coord[2]});
        }
    }

    // map.cpp:115-125
    size_t memory() const
    {
        double total = 0.0;
        for (auto& kv : patches) {
            total += sizeof(uint64_t) + 16 /*sizeof(COWPtr<Container>)*/ + sizeof(void*);
            total += (double)(kv.second->data.size()) / (double)kv.second.use_count();
        }
        return (size_t)total;
    }

    void reset_cache() const { prev_idx_ = uint64_t(-1); prev_patch_ = nullptr; }

private:
    mutable uint64_t prev_idx_ = uint64_t(-1);                    // map.h:374
    mutable std::shared_ptr<Container>* prev_patch_ = nullptr;    // map.h:375 (guarded by prev_idx_)
};

// -------------------------------------------------------------------------------------
// FrequencyOccupancyMap  (include/lama/sdm/frequency_occupancy_map.h:43-46,
//                         src/sdm/frequency_occupancy_map.cpp:38-91)
// -------------------------------------------------------------------------------------
struct frequency { uint16_t occupied; uint16_t visited; };
static_assert(sizeof(frequency) == 4, "frequency must be 4 bytes (SURVEY A.9-2)");

class FrequencyOccupancyMap : public Map {
public:
    FrequencyOccupancyMap(double res, uint32_t patch_size = 32) : Map(res, sizeof(frequency), patch_size) {}
    FrequencyOccupancyMap(const FrequencyOccupancyMap& o) : Map(o) {}

    static double prob(const frequency& f)             // :38-45
    {
        if (f.visited == 0) return 0.25;
        return ((double)f.occupied) / ((double)f.visited);
    }
    bool setFree(const V3u& c)                          // :65-74
    {
        frequency* cell = (frequency*)get(c);
        bool free = prob(*cell) < 0.25;
        cell->visited++;
        if (free) return false;
        return prob(*cell) < 0.25;
    }
    bool setOccupied(const V3u& c)                      // :81-91
    {
        frequency* cell = (frequency*)get(c);
        bool occupied = prob(*cell) > 0.25;
        cell->occupied++;
        cell->visited++;
        if (occupied) return false;
        return prob(*cell) > 0.25;
    }
    bool isFree(const V3u& c) const                     // :119-125
    {
        const frequency* cell = (const frequency*)get(c);
        if (cell == 0) return false;
        return prob(*cell) < 0.25;
    }
    bool isOccupied(const V3u& c) const                 // :132-138
    {
        const frequency* cell = (const frequency*)get(c);
        if (cell == 0) return false;
        return prob(*cell) > 0.25;
    }
    bool isUnknown(const V3u& c) const                  // :145-151
    {
        const frequency* cell = (const frequency*)get(c);
        if (cell == 0) return true;
        return cell->visited == 0;
    }
    double getProbability(const V3u& c) const           // :166-172
    {
        const frequency* cell = (const frequency*)get(c);
        if (cell == 0) return 0.25;
        return prob(*cell);
    }
};

// -------------------------------------------------------------------------------------
// DynamicDistanceMap  (include/lama/sdm/dynamic_distance_map.h:48-105,
//                      src/sdm/dynamic_distance_map.cpp:36-330)
// -------------------------------------------------------------------------------------
#pragma pack(push, 1)
// -------------------------------------------------------------------------------------
// SimpleOccupancyMap  (src/sdm/simple_occupancy_map.cpp:36-131): int8 tri-state cell, -1 free / 0 unknown / 1 occupied
// -------------------------------------------------------------------------------------
class SimpleOccupancyMap : public Map {
public:
    SimpleOccupancyMap(double res, uint32_t patch_size = 32) : Map(res, sizeof(int8_t), patch_size) {}
    bool setFree(const V3u& c) { int8_t* cell = (int8_t*)get(c); if (*cell == -1) return false; *cell = -1; return true; }     // :53-61
    bool setOccupied(const V3u& c) { int8_t* cell = (int8_t*)get(c); if (*cell == 1) return false; *cell = 1; return true; }   // :68-76
    bool setUnknown(const V3u& c) { int8_t* cell = (int8_t*)get(c); if (*cell == 0) return false; *cell = 0; return true; }    // :83-91
    bool isFree(const V3u& c) const { const int8_t* cell = (const int8_t*)get(c); return cell != 0 && *cell == -1; }           // :97-104
    bool isFree(const V3d& p) const { return isFree(w2m(p)); }                                                                 // :92-95
    bool isOccupied(const V3u& c) const { const int8_t* cell = (const int8_t*)get(c); return cell != 0 && *cell == 1; }        // :111-118
};

// -------------------------------------------------------------------------------------
// ProbabilisticOccupancyMap  (include/lama/sdm/probabilistic_occupancy_map.h:40-100, src/sdm/probabilistic_occupancy_map.cpp:36-195)
// log-odds cell {float prob}; the parameters are doubles holding float-rounded values (logods() returns float).
// -------------------------------------------------------------------------------------
class ProbabilisticOccupancyMap : public Map {
public:
    static float logods(const float& prob) { return (float)std::log(prob / (1.0 - prob)); }      // :43-46
    static float prob_of(const float& l) { return (float)(1.0 - 1.0 / (1.0 + std::exp(l))); }      // :38-41 (std::exp(float))
    ProbabilisticOccupancyMap(double res, uint32_t patch_size = 32) : Map(res, sizeof(float), patch_size)
    {
        miss_ = logods(0.4); hit_ = logods(0.7);                        // :53-59
        clamp_min_ = logods(0.12); clamp_max_ = logods(0.97);
        occ_thresh_ = 0.0 * logods(0.5);
    }
    bool setFree(const V3u& c)                                          // :82-91
    {
        float* cell = (float*)get(c);
        bool free = *cell < occ_thresh_;
        *cell = (float)std::max(*cell + miss_, clamp_min_);
        if (free) return false;
        return (*cell < occ_thresh_);
    }
    bool setOccupied(const V3u& c)                                      // :98-107
    {
        float* cell = (float*)get(c);
        bool occupied = *cell > occ_thresh_;
        *cell = (float)std::min(*cell + hit_, clamp_max_);
        if (occupied) return false;
        return (*cell > occ_thresh_);
    }
    bool isFree(const V3u& c) const { const float* cell = (const float*)get(c); return cell != 0 && *cell < occ_thresh_; }       // :131-137
    bool isOccupied(const V3u& c) const { const float* cell = (const float*)get(c); return cell != 0 && *cell > occ_thresh_; }   // :144-150
    double miss_, hit_, clamp_min_, clamp_max_, occ_thresh_;
};

// lama::random (src/random.cpp:34-73): ONE process-wide std::mt19937; uniform() builds a fresh
// std::uniform_real_distribution<double>(0,1) per call.  The reference seeds it from std::random_device unless
// random::setSeed is called; tests always seed.
namespace random {
inline std::mt19937& gen() { static std::mt19937 g(5489u); return g; }
inline void setSeed(uint32_t seed) { gen().seed(seed); }
inline double uniform() { std::uniform_real_distribution<double> d(0.0, 1.0); return d(gen()); }
}

struct distance_t {            // dynamic_distance_map.h:48-53 (Vector3s = 3 x int16)
    int16_t obstacle[3];
    uint16_t sqdist;
    bool valid_obstacle;
    bool is_queued;
};
#pragma pack(pop)
static_assert(sizeof(distance_t) == 10, "distance_t must be 10 bytes (SURVEY A.9-2)");

struct BrushfireStats {
    uint64_t raise_pops = 0, lower_pops = 0, lower_fired = 0, pushes = 0, max_queue = 0, tie_overwrites = 0;
    uint64_t max_queue_last = 0;   // of the last update() only
};

// test switch: maps created while this is set use update_canonical() (see there); never set for parity claims
inline bool& canonical_default() { static bool v = false; return v; }

// Instrumentation only (tools/research): when non-null, every priority-queue operation of the distance map is appended as
// {op, prio, x, y}: op 0 = update() begins, 1 = push lower_, 2 = push raise_, 3 = pop raise_, 4 = pop lower_ (lower() did not
// run), 5 = pop lower_ (lower() ran), 6 = update() ends.
struct BfTraceRec { uint32_t op, prio, x, y; };
inline std::vector<BfTraceRec>*& bf_trace() { static thread_local std::vector<BfTraceRec>* t = nullptr; return t; }
inline void bf_trace_add(uint32_t op, uint32_t prio, const V3u& l) { if (bf_trace()) bf_trace()->push_back({op, prio, l.x, l.y}); }
// Research only: when set, DynamicDistanceMap::update() runs this instead of the reference's loop (tools/research/lse_proto.hpp
// checks a level-synchronous restatement against the faithful one).  Never set by tests, bench or the C API.
class DynamicDistanceMap;
inline uint32_t (*&bf_update_hook())(DynamicDistanceMap&) { static uint32_t (*h)(DynamicDistanceMap&) = nullptr; return h; }

class DynamicDistanceMap : public Map {
public:
    DynamicDistanceMap(double res, uint32_t patch_size = 32)       // :36-47
        : Map(res, sizeof(distance_t), patch_size), max_sqdist_(100) { canonical = canonical_default(); }
    DynamicDistanceMap(const DynamicDistanceMap& o)                // :49-61 (queues NOT copied)
        : Map(o), canonical(o.canonical), max_sqdist_(o.max_sqdist_) {}

    uint32_t max_sqdist() const { return max_sqdist_; }

    void setMaxDistance(double distance)                           // :149-153
    {
        max_sqdist_ = (uint32_t)std::ceil(distance * scale);
        max_sqdist_ *= max_sqdist_;
    }
    double maxDistance() const { return std::sqrt((double)max_sqdist_) * resolution; } // :155-158
    void writeParameters(std::ofstream& stream) const override { stream.write((char*)&max_sqdist_, sizeof(max_sqdist_)); }   // :200-203
    void readParameters(std::ifstream& stream) override { stream.read((char*)&max_sqdist_, sizeof(max_sqdist_)); }          // :205-208

    double distance(const V3u& c) const                            // :140-147
    {
        const distance_t* cell = (const distance_t*)get(c);
        if (cell == 0 || !cell->valid_obstacle)
            return std::sqrt((double)max_sqdist_) * resolution;
        return std::sqrt((double)cell->sqdist) * resolution;
    }

    // :66-138, 2-D branch :77-93
    double distance(const V3d& coordinates, V3d* gradient) const
    {
        V3d m = w2m_nocast(coordinates);
        V3u d{(uint32_t)m.x, (uint32_t)m.y, (uint32_t)m.z};
        double mu0 = m.x - (double)d.x, mu1 = m.y - (double)d.y;
        double muinv0 = 1.0 - mu0, muinv1 = 1.0 - mu1;

        double v0 = distance(d);
        double v1 = distance(V3u{d.x + 1, d.y, d.z});
        double v2 = distance(V3u{d.x, d.y + 1, d.z});
        double v3 = distance(V3u{d.x + 1, d.y + 1, d.z});

        double dist = v0 * muinv0 * muinv1 + v1 * muinv1 * mu0 + v2 * muinv0 * mu1 + v3 * mu0 * mu1;
        if (gradient) {
            gradient->x = -((v0 - v1) * muinv1 + (v2 - v3) * mu1) * scale;
            gradient->y = -((v0 - v2) * muinv0 + (v1 - v3) * mu0) * scale;
            gradient->z = 0;
        }
        return dist;
    }

    void addObstacle(const V3u& location)                          // :212-226
    {
        distance_t* cell = (distance_t*)get(location);
        if (cell->valid_obstacle && cell->sqdist == 0) return;
        cell->sqdist = 0;
        cell->obstacle[0] = cell->obstacle[1] = cell->obstacle[2] = 0;
        cell->valid_obstacle = true;
        cell->is_queued = true;
        lower_.push({0, location});
        bf_trace_add(1, 0, location);
        ++stats.pushes;
    }
    void removeObstacle(const V3u& location)                       // :228-242
    {
        distance_t* cell = (distance_t*)get(location);
        if (!(cell->valid_obstacle && cell->sqdist == 0)) return;
        cell->sqdist = 0;
        cell->obstacle[0] = cell->obstacle[1] = cell->obstacle[2] = 0;
        cell->valid_obstacle = false;
        cell->is_queued = true;
        raise_.push({0, location});
        bf_trace_add(2, 0, location);
        ++stats.pushes;
    }

    uint32_t update()                                              // :160-197
    {
        if (bf_update_hook()) return bf_update_hook()(*this);
        if (canonical) return update_canonical();
        uint32_t processed = 0;
        stats.max_queue_last = 0;
        bf_trace_add(0, 0, V3u{0, 0, 0});
        while (!raise_.empty()) {
            stats.max_queue = std::max<uint64_t>(stats.max_queue, raise_.size() + lower_.size());
            stats.max_queue_last = std::max<uint64_t>(stats.max_queue_last, std::max(raise_.size(), lower_.size()));
            bf_trace_add(3, (uint32_t)raise_.top().first, raise_.top().second);
            V3u location = raise_.top().second; raise_.pop();
            distance_t* current = (distance_t*)get(location);
            ++processed; ++stats.raise_pops;
            raise(location, current);
        }
        while (!lower_.empty()) {
            stats.max_queue = std::max<uint64_t>(stats.max_queue, lower_.size());
            stats.max_queue_last = std::max<uint64_t>(stats.max_queue_last, lower_.size());
            const uint32_t tprio_ = (uint32_t)lower_.top().first;
            V3u location = lower_.top().second; lower_.pop();
            distance_t* current = (distance_t*)get(location);
            ++processed; ++stats.lower_pops;
            const uint64_t fired0_ = stats.lower_fired;
            if (bf_trace()) bf_trace()->push_back({4u, tprio_, location.x, location.y});
            const size_t trace_at_ = bf_trace() ? bf_trace()->size() - 1 : 0;
            if (current->valid_obstacle) {
                V3u obs = offs(location, current->obstacle);
                const distance_t* obstacle = (distance_t*)get(obs);
                current = (distance_t*)get(location);   // (re-fetch: no semantic effect)
                if (obstacle->sqdist == 0)                         // :191 (valid_obstacle NOT tested)
                    lower(location, current);
            }
            if (bf_trace() && stats.lower_fired != fired0_) (*bf_trace())[trace_at_].op = 5u;
        }
        bf_trace_add(6, 0, V3u{0, 0, 0});
        return processed;
    }

    BrushfireStats stats;

    // ---------------------------------------------------------------------------------------------------------
    // NOT the reference's algorithm: level-synchronous variant of update() with a canonical tie rule, kept here to
    // MEASURE how far an order-independent (hence parallelisable) brushfire deviates from the faithful one above.
    // Raise wave: breadth-first rounds, decisions taken on the state at the start of a round.  Lower wave: all
    // queued cells of one priority level fire "simultaneously"; offers to the same neighbour are folded in the order
    // (smaller candidate first, then direction index of the offering move), using the reference's overwrite rule.
    // ---------------------------------------------------------------------------------------------------------
    uint32_t update_canonical()
    {
        uint32_t processed = 0;
        std::vector<V3u> frontier;
        while (!raise_.empty()) { frontier.push_back(raise_.top().second); raise_.pop(); }
        std::vector<std::vector<V3u>> bucket(max_sqdist_ + 1);
        while (!lower_.empty()) { bucket[std::min<uint32_t>((uint32_t)lower_.top().first, max_sqdist_)].push_back(lower_.top().second); lower_.pop(); }
        auto key = [](const V3u& c) { return ((uint64_t)c.y << 32) | c.x; };
        while (!frontier.empty()) {
            std::sort(frontier.begin(), frontier.end(), [&](const V3u& a, const V3u& b) { return key(a) < key(b); });
            struct Dec { V3u n; bool ovalid; };
            std::vector<Dec> decs;
            for (const V3u& loc : frontier) {
                ++processed;
                for (int i = 0; i < 4; ++i) {
                    int64_t d[3]; delta(i, d);
                    V3u nl{(uint32_t)((int64_t)loc.x + d[0]), (uint32_t)((int64_t)loc.y + d[1]), loc.z};
                    distance_t* n = (distance_t*)get(nl);
                    if (n->is_queued || !n->valid_obstacle) continue;
                    V3u obs = offs(nl, n->obstacle);
                    const distance_t* o = (distance_t*)get(obs);
                    decs.push_back({nl, o->valid_obstacle});
                }
            }
            std::vector<V3u> next;
            for (const Dec& dc : decs) {
                distance_t* n = (distance_t*)get(dc.n);
                if (n->is_queued || !n->valid_obstacle) continue;       // already handled through another neighbour
                if (!dc.ovalid) {
                    next.push_back(dc.n);
                    n->sqdist = 0; n->obstacle[0] = n->obstacle[1] = n->obstacle[2] = 0; n->valid_obstacle = false; n->is_queued = true;
                } else {
                    bucket[n->sqdist].push_back(dc.n);
                    n->is_queued = true;
                }
            }
            for (const V3u& loc : frontier) ((distance_t*)get(loc))->is_queued = false;
            frontier.swap(next);
        }
        for (uint32_t lev = 0; lev < max_sqdist_; ++lev) {
            std::vector<V3u>& ent = bucket[lev];
            if (ent.empty()) continue;
            std::sort(ent.begin(), ent.end(), [&](const V3u& a, const V3u& b) { return key(a) < key(b); });
            ent.erase(std::unique(ent.begin(), ent.end(), [](const V3u& a, const V3u& b) { return a == b; }), ent.end());
            struct Offer { V3u n; uint32_t cand; int dir; int16_t ox, oy; };
            std::vector<Offer> offers;
            std::vector<V3u> fired;
            for (const V3u& loc : ent) {
                ++processed;
                distance_t* cur = (distance_t*)get(loc);
                if (!cur->valid_obstacle) continue;
                V3u obs = offs(loc, cur->obstacle);
                const distance_t* oc = (distance_t*)get(obs);
                cur = (distance_t*)get(loc);
                if (oc->sqdist != 0 || !cur->is_queued) continue;
                fired.push_back(loc);
                for (int i = 0; i < 4; ++i) {
                    int64_t d[3]; delta(i, d);
                    if (d[0] * cur->obstacle[0] > 0 || d[1] * cur->obstacle[1] > 0) continue;
                    V3u nl{(uint32_t)((int64_t)loc.x + d[0]), (uint32_t)((int64_t)loc.y + d[1]), loc.z};
                    (void)get(nl);
                    cur = (distance_t*)get(loc);
                    const int64_t ox = (int64_t)loc.x + cur->obstacle[0], oy = (int64_t)loc.y + cur->obstacle[1];
                    const int64_t dx = (int64_t)nl.x - ox, dy = (int64_t)nl.y - oy;
                    offers.push_back({nl, (uint32_t)(dx * dx + dy * dy), i, (int16_t)(ox - (int64_t)nl.x), (int16_t)(oy - (int64_t)nl.y)});
                }
            }
            std::sort(offers.begin(), offers.end(), [&](const Offer& a, const Offer& b) {
                if (key(a.n) != key(b.n)) return key(a.n) < key(b.n);
                if (a.cand != b.cand) return a.cand < b.cand;
                return a.dir < b.dir;
            });
            for (const Offer& of : offers) {
                distance_t* n = (distance_t*)get(of.n);
                const uint32_t cmp = n->valid_obstacle ? n->sqdist : max_sqdist_;
                bool over = of.cand < cmp;
                if (!over && of.cand == n->sqdist) {
                    V3u nobs = offs(of.n, n->obstacle);
                    const distance_t* o = (distance_t*)get(nobs);
                    n = (distance_t*)get(of.n);
                    if (!n->valid_obstacle || !(o->valid_obstacle && o->sqdist == 0)) over = true;
                }
                if (over) {
                    bucket[std::min<uint32_t>(of.cand, max_sqdist_)].push_back(of.n);
                    n->sqdist = (uint16_t)of.cand; n->valid_obstacle = true; n->obstacle[0] = of.ox; n->obstacle[1] = of.oy; n->obstacle[2] = 0; n->is_queued = true;
                }
            }
            for (const V3u& loc : fired) ((distance_t*)get(loc))->is_queued = false;
        }
        return processed;
    }
    bool canonical = false;     // when set, update() dispatches to update_canonical() (experiments only)
    friend struct BfResearchAccess;   // tools/research only

private:
    typedef std::pair<int, V3u> queue_pair_t;                      // .h:90
    struct compare_prio {                                          // .h:92-95
        bool operator()(const queue_pair_t& l, const queue_pair_t& r) const { return l.first > r.first; }
    };
    typedef std::priority_queue<queue_pair_t, std::vector<queue_pair_t>, compare_prio> queue_t; // .h:97-98

    static V3u offs(const V3u& loc, const int16_t o[3])
    {
        return V3u{(uint32_t)((int64_t)loc.x + o[0]), (uint32_t)((int64_t)loc.y + o[1]), (uint32_t)((int64_t)loc.z + o[2])};
    }

    // deltas: dynamic_distance_map.cpp:40-43 (first 4 used in 2-D)
    static void delta(int i, int64_t d[3])
    {
        static const int64_t D[4][3] = {{1, 0, 0}, {0, 1, 0}, {-1, 0, 0}, {0, -1, 0}};
        d[0] = D[i][0]; d[1] = D[i][1]; d[2] = D[i][2];
    }

    void raise(const V3u& location, distance_t* current)           // :244-279
    {
        for (int i = 0; i < 4; ++i) {
            int64_t d[3]; delta(i, d);
            V3u newloc{(uint32_t)((int64_t)location.x + d[0]), (uint32_t)((int64_t)location.y + d[1]), (uint32_t)((int64_t)location.z + d[2])};
            distance_t* neighbor = (distance_t*)get(newloc);
            if (neighbor->is_queued || !neighbor->valid_obstacle) continue;

            V3u obs = offs(newloc, neighbor->obstacle);
            const distance_t* obstacle = (distance_t*)get(obs);
            neighbor = (distance_t*)get(newloc);
            if (!obstacle->valid_obstacle) {
                raise_.push({neighbor->sqdist, newloc});
                bf_trace_add(2, neighbor->sqdist, newloc);
                ++stats.pushes;
                neighbor->sqdist = 0;
                neighbor->obstacle[0] = neighbor->obstacle[1] = neighbor->obstacle[2] = 0;
                neighbor->valid_obstacle = false;
                neighbor->is_queued = true;
            } else if (!neighbor->is_queued) {
                lower_.push({neighbor->sqdist, newloc});
                bf_trace_add(1, neighbor->sqdist, newloc);
                ++stats.pushes;
                neighbor->is_queued = true;
            }
        }
        current = (distance_t*)get(location);
        current->is_queued = false;
    }

    void lower(const V3u& location, distance_t* current)           // :281-330
    {
        if (!current->is_queued) return;
        ++stats.lower_fired;
        const int16_t cobs[3] = {current->obstacle[0], current->obstacle[1], current->obstacle[2]};
        for (int i = 0; i < 4; ++i) {
            int64_t d[3]; delta(i, d);
            // only update away from the obstacle                    :296
            if (d[0] * cobs[0] > 0 || d[1] * cobs[1] > 0 || d[2] * cobs[2] > 0) continue;

            int64_t newloc[3] = {(int64_t)location.x + d[0], (int64_t)location.y + d[1], (int64_t)location.z + d[2]};
            V3u nl{(uint32_t)newloc[0], (uint32_t)newloc[1], (uint32_t)newloc[2]};
            distance_t* neighbor = (distance_t*)get(nl);

            int64_t obs[3] = {(int64_t)location.x + cobs[0], (int64_t)location.y + cobs[1], (int64_t)location.z + cobs[2]};
            int64_t dist[3] = {newloc[0] - obs[0], newloc[1] - obs[1], newloc[2] - obs[2]};
            uint32_t new_sqdist = (uint32_t)(dist[0] * dist[0] + dist[1] * dist[1] + dist[2] * dist[2]);
            uint32_t cmp_sqdist = neighbor->valid_obstacle ? neighbor->sqdist : max_sqdist_;

            bool overwrite = (new_sqdist < cmp_sqdist);
            if (!overwrite && new_sqdist == neighbor->sqdist) {     // :311-317
                V3u nobs = offs(nl, neighbor->obstacle);
                const distance_t* obstacle = (distance_t*)get(nobs);
                neighbor = (distance_t*)get(nl);
                if (!neighbor->valid_obstacle || !(obstacle->valid_obstacle && obstacle->sqdist == 0)) {
                    overwrite = true;
                    ++stats.tie_overwrites;
                }
            }
            if (overwrite) {                                        // :319-326
                lower_.push({(int)new_sqdist, nl});
                bf_trace_add(1, new_sqdist, nl);
                ++stats.pushes;
                neighbor->sqdist = (uint16_t)new_sqdist;
                neighbor->valid_obstacle = true;
                neighbor->obstacle[0] = (int16_t)(obs[0] - newloc[0]);
                neighbor->obstacle[1] = (int16_t)(obs[1] - newloc[1]);
                neighbor->obstacle[2] = (int16_t)(obs[2] - newloc[2]);
                neighbor->is_queued = true;
            }
        }
        current = (distance_t*)get(location);
        current->is_queued = false;
    }

    queue_t lower_;
    queue_t raise_;
    uint32_t max_sqdist_;
};

// -------------------------------------------------------------------------------------
// nlls: CauchyWeight, GaussNewton, Solver  (src/nlls/*.cpp)
// -------------------------------------------------------------------------------------
// sdm::export_to_png's build_image (src/sdm/export.cpp:46-95), 2-D: the pixel array (rows = y, top row = min y)
struct ExportImage { uint32_t width = 0, height = 0; std::vector<uint8_t> data; };
template <typename OccMap>
inline void build_image_occ(const OccMap& occ, ExportImage& image)
{
    V3u mn, mx;
    occ.bounds(mn, mx);
    image.width = mx.x - mn.x; image.height = mx.y - mn.y;
    image.data.assign((size_t)image.width * image.height, 90);
    occ.visit_all_cells([&](const V3u& coords) {
        const uint32_t u = coords.x - mn.x, v = coords.y - mn.y;
        uint8_t& px = image.data[u + (size_t)v * image.width];
        if (occ.isFree(coords)) px = 255;
        else if (occ.isOccupied(coords)) px = 0;
        else px = 127;
    });
}
inline void build_image_dm(const DynamicDistanceMap& dm, ExportImage& image)
{
    V3u mn, mx;
    dm.bounds(mn, mx);
    image.width = mx.x - mn.x; image.height = mx.y - mn.y;
    image.data.assign((size_t)image.width * image.height, 127);
    dm.visit_all_cells([&](const V3u& coords) {
        const uint32_t u = coords.x - mn.x, v = coords.y - mn.y;
        image.data[u + (size_t)v * image.width] = (uint8_t)(dm.distance(coords) * 255 / dm.maxDistance());
    });
}

struct CauchyWeight {                                              // robust_cost.cpp:66-73
    double c_;
    explicit CauchyWeight(double param) : c_(1.0 / (param * param)) {}
    double value(double x) const { return (1.0 / (1.0 + x * x * c_)); }
};

// Eigen 3.3 LDLT<Matrix3d,Lower> (ldlt_inplace<Lower>::unblocked + LDLT::_solve_impl),
// used by gauss_newton.cpp:66 `A.selfadjointView<Lower>().ldlt().solve(-g)`.
inline void ldlt3_solve(const double Ain[3][3] /*lower used*/, const double b[3], double x[3])
{
    double m[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = (j <= i) ? Ain[i][j] : Ain[j][i];
    int tr[3];
    double temp[3];
    const int size = 3;
    bool zero_matrix = false;
    for (int k = 0; k < size; ++k) {
        // largest |diagonal| in the remaining corner (first maximum)
        int big = k; double best = std::fabs(m[k][k]);
        for (int i = k + 1; i < size; ++i) if (std::fabs(m[i][i]) > best) { best = std::fabs(m[i][i]); big = i; }
        tr[k] = big;
        if (k != big) {
            int s = size - big - 1;
            for (int j = 0; j < k; ++j) std::swap(m[k][j], m[big][j]);
            for (int i = 0; i < s; ++i) std::swap(m[size - s + i][k], m[size - s + i][big]);
            std::swap(m[k][k], m[big][big]);
            for (int i = k + 1; i < big; ++i) { double tmp = m[i][k]; m[i][k] = m[big][i]; m[big][i] = tmp; }
        }
        int rs = size - k - 1;
        if (k > 0) {
            for (int j = 0; j < k; ++j) temp[j] = m[j][j] * m[k][j];
            double acc = 0.0;
            for (int j = 0; j < k; ++j) acc = (j == 0) ? m[k][0] * temp[0] : acc + m[k][j] * temp[j];
            m[k][k] -= acc;
            for (int i = 0; i < rs; ++i) {
                double a2 = 0.0;
                for (int j = 0; j < k; ++j) a2 = (j == 0) ? m[k + 1 + i][0] * temp[0] : a2 + m[k + 1 + i][j] * temp[j];
                m[k + 1 + i][k] -= a2;
            }
        }
        double realAkk = m[k][k];
        bool pivot_is_valid = (std::fabs(realAkk) > 0.0);
        if (k == 0 && !pivot_is_valid) { for (int j = 0; j < size; ++j) tr[j] = j; zero_matrix = true; break; }
        if (rs > 0 && pivot_is_valid) for (int i = 0; i < rs; ++i) m[k + 1 + i][k] /= realAkk;
    }
    (void)zero_matrix;
    // solve: dst = P b
    double d[3] = {b[0], b[1], b[2]};
    for (int k = 0; k < size; ++k) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
    // L^-1 (unit lower, forward substitution)
    for (int i = 1; i < size; ++i) {
        double acc = 0.0;
        for (int j = 0; j < i; ++j) acc = (j == 0) ? m[i][0] * d[0] : acc + m[i][j] * d[j];
        d[i] -= acc;
    }
    // pseudo-inverse of D; tolerance = 1/highest()
    const double tol = 1.0 / 1.7976931348623157e308;
    for (int i = 0; i < size; ++i) { if (std::fabs(m[i][i]) > tol) d[i] /= m[i][i]; else d[i] = 0.0; }
    // L^-T (unit upper, backward substitution)
    for (int i = size - 2; i >= 0; --i) {
        double acc = 0.0; bool first = true;
        for (int j = i + 1; j < size; ++j) { acc = first ? m[j][i] * d[j] : acc + m[j][i] * d[j]; first = false; }
        d[i] -= acc;
    }
    // P^T
    for (int k = size - 1; k >= 0; --k) if (tr[k] != k) std::swap(d[k], d[tr[k]]);
    x[0] = d[0]; x[1] = d[1]; x[2] = d[2];
}

struct SolveStats { uint32_t iterations = 0; uint32_t evals = 0; };

// MatchSurface2D (src/match_surface_2d.cpp:35-122) as a plain struct
struct MatchSurface2D {
    const DynamicDistanceMap* surface_;
    const Scan* scan_;
    SE2 state_;

    MatchSurface2D(const DynamicDistanceMap* s, const Scan* scan, const SE2& est) : surface_(s), scan_(scan), state_(est) {}

    // :42-90.  J is row-major N x 3 here (values identical to the reference's col-major matrix)
    void eval(std::vector<double>& residuals, std::vector<double>* J) const
    {
        Affine3 mtf = moving_tf(*scan_);
        Affine3 ftf = fixed_tf(state_);          // AngleAxisd(state_.so2().log(), UnitZ)
        Affine3 tf = affine_mul(ftf, mtf);
        const size_t n = scan_->points.size();
        residuals.resize(n);
        if (J) J->resize(n * 3);
        V3d grad;
        for (size_t i = 0; i < n; ++i) {
            V3d hit = affine_apply(tf, scan_->points[i]);
            hit.z = 0.0;
            residuals[i] = surface_->distance(hit, &grad);
            if (J) {
                (*J)[3 * i + 0] = grad.x;
                (*J)[3 * i + 1] = grad.y;
                (*J)[3 * i + 2] = grad.y * hit.x - grad.x * hit.y;
            }
        }
    }
    // :92-116 error(): root mean square of the NON-interpolated cell distances at w2m(tf * p_i) (z of the hit is kept)
    void cell_distances(double* out) const
    {
        Affine3 tf = affine_mul(fixed_tf(state_), moving_tf(*scan_));
        for (size_t i = 0; i < scan_->points.size(); ++i) out[i] = surface_->distance(surface_->w2m(affine_apply(tf, scan_->points[i])));
    }
    double error() const
    {
        std::vector<double> d(scan_->points.size());
        cell_distances(d.data());
        double s = 0;
        for (double v : d) s += v * v;
        return std::sqrt(s / (double)d.size());
    }
    // :118-122
    void update(const double h[3]) { state_ = se2_mul(se2_exp(h[0], h[1], h[2]), state_); }
};

// GaussNewton (src/nlls/gauss_newton.cpp:38-91) + Solver::solve (src/nlls/solver.cpp:53-107)
// with robust cost = CauchyWeight.  Returns number of iterations (applied + reverted steps).
inline SolveStats solve_gn(MatchSurface2D& problem, uint32_t max_iterations, const CauchyWeight& robust)
{
    const double eps1 = 1e-4, eps2 = 1e-4;             // gauss_newton.cpp:38-42
    SolveStats st;
    std::vector<double> r, ur, J;
    double h[3] = {0, 0, 0};
    bool stop_ = false;                                 // reset() :49-52
    double chi2_ = 0.0;
    bool valid = true;
    uint32_t iter = 0;
    while (!stop_ && iter < max_iterations) {           // solver.cpp:67
        if (valid) {
            problem.eval(r, &J); ++st.evals;            // :71
            const size_t rows = r.size();
            for (size_t i = 0; i < rows; ++i) {         // :74-79
                double w = std::sqrt(robust.value(r[i]));
                r[i] *= w;
                J[3 * i + 0] *= w; J[3 * i + 1] *= w; J[3 * i + 2] *= w;
            }
        }
        // strategy->step(r, J)                          gauss_newton.cpp:53-73
        {
            const size_t rows = r.size();
            double g[3] = {0, 0, 0};
            for (size_t i = 0; i < rows; ++i) { g[0] += J[3 * i] * r[i]; g[1] += J[3 * i + 1] * r[i]; g[2] += J[3 * i + 2] * r[i]; }
            chi2_ = 0.0;
            for (size_t i = 0; i < rows; ++i) chi2_ += r[i] * r[i];
            double max_abs_g = std::max(std::fabs(g[0]), std::max(std::fabs(g[1]), std::fabs(g[2])));
            if (max_abs_g < eps1) {
                stop_ = true;
                h[0] = h[1] = h[2] = 0.0;
            } else {
                double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                for (size_t i = 0; i < rows; ++i)
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) A[a][b] += J[3 * i + a] * J[3 * i + b];
                double mg[3] = {-g[0], -g[1], -g[2]};
                ldlt3_solve(A, mg, h);
                double max_abs_h = std::max(std::fabs(h[0]), std::max(std::fabs(h[1]), std::fabs(h[2])));
                if (max_abs_h < eps2) stop_ = true;
            }
        }
        if (stop_) break;                               // solver.cpp:84-86 (h NOT applied)

        problem.update(h);                              // :89
        problem.eval(ur, nullptr); ++st.evals;          // :90
        double ur2 = 0.0;
        for (size_t i = 0; i < ur.size(); ++i) {        // :92-96
            double w = std::sqrt(robust.value(ur[i]));
            ur[i] *= w;
        }
        for (size_t i = 0; i < ur.size(); ++i) ur2 += ur[i] * ur[i];
        // strategy->valid(ur)                           gauss_newton.cpp:75-86
        {
            double dF = chi2_ - ur2;
            if (dF > 0) valid = true;
            else { stop_ = true; valid = false; }
        }
        if (!valid) {                                   // solver.cpp:99-102
            double mh[3] = {-h[0], -h[1], -h[2]};
            problem.update(mh);
        }
        ++iter;
    }
    st.iterations = iter;
    return st;
}

// Eigen LLT on selfadjointView<Upper> + solve for a 3x3 (published algorithm, Eigen 3.3 Cholesky/LLT.h llt_inplace::unblocked
// and the triangular solves): L L^T = A, L y = b, L^T x = y.
inline void llt3_solve(const double A[3][3] /*upper used*/, const double b[3], double x[3])
{
    const double l00 = std::sqrt(A[0][0]);
    const double l10 = A[0][1] / l00, l20 = A[0][2] / l00;
    const double l11 = std::sqrt(A[1][1] - l10 * l10);
    const double l21 = (A[1][2] - l20 * l10) / l11;
    const double l22 = std::sqrt(A[2][2] - (l20 * l20 + l21 * l21));
    const double y0 = b[0] / l00;
    const double y1 = (b[1] - l10 * y0) / l11;
    const double y2 = (b[2] - (l20 * y0 + l21 * y1)) / l22;
    x[2] = y2 / l22;
    x[1] = (y1 - l21 * x[2]) / l11;
    x[0] = (y0 - (l10 * x[1] + l20 * x[2])) / l00;
}

// Solver::solve (src/nlls/solver.cpp:53-107) with the LevenbergMarquard strategy (src/nlls/levenberg_marquardt.cpp:38-107,
// eps1 = eps2 = tau = 1e-4) and CauchyWeight -- what Slam2D / Loc2D run with Options::strategy = "lm".
inline SolveStats solve_lm(MatchSurface2D& problem, uint32_t max_iterations, const CauchyWeight& robust)
{
    const double eps1 = 1e-4, eps2 = 1e-4, tau = 1e-4;
    SolveStats st;
    std::vector<double> r, ur, J;
    double h[3] = {0, 0, 0}, g[3] = {0, 0, 0};
    double mu_ = -1, v_ = 2.0, chi2_ = 0.0;             // reset() :49-54
    bool stop_ = false, valid = true;
    uint32_t iter = 0;
    while (!stop_ && iter < max_iterations) {
        if (valid) {
            problem.eval(r, &J); ++st.evals;
            for (size_t i = 0; i < r.size(); ++i) {
                double w = std::sqrt(robust.value(r[i]));
                r[i] *= w;
                J[3 * i + 0] *= w; J[3 * i + 1] *= w; J[3 * i + 2] *= w;
            }
        }
        {                                               // step() :56-84
            const size_t rows = r.size();
            chi2_ = 0.0;
            for (size_t i = 0; i < rows; ++i) chi2_ += r[i] * r[i];
            g[0] = g[1] = g[2] = 0;
            for (size_t i = 0; i < rows; ++i) { g[0] += J[3 * i] * r[i]; g[1] += J[3 * i + 1] * r[i]; g[2] += J[3 * i + 2] * r[i]; }
            const double max_abs_g = std::max(std::fabs(g[0]), std::max(std::fabs(g[1]), std::fabs(g[2])));
            if (max_abs_g < eps1) {
                stop_ = true;
                h[0] = h[1] = h[2] = 0.0;
            } else {
                double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                for (size_t i = 0; i < rows; ++i)
                    for (int a = 0; a < 3; ++a)
                        for (int b = 0; b < 3; ++b) A[a][b] += J[3 * i + a] * J[3 * i + b];
                if (mu_ < 0) mu_ = tau * std::max(A[0][0], std::max(A[1][1], A[2][2]));
                A[0][0] += mu_; A[1][1] += mu_; A[2][2] += mu_;
                const double mg[3] = {-g[0], -g[1], -g[2]};
                llt3_solve(A, mg, h);
                const double max_abs_h = std::max(std::fabs(h[0]), std::max(std::fabs(h[1]), std::fabs(h[2])));
                if (max_abs_h < eps2) stop_ = true;
            }
        }
        if (stop_) break;
        problem.update(h);
        problem.eval(ur, nullptr); ++st.evals;
        double ur2 = 0.0;
        for (size_t i = 0; i < ur.size(); ++i) { double w = std::sqrt(robust.value(ur[i])); ur[i] *= w; }
        for (size_t i = 0; i < ur.size(); ++i) ur2 += ur[i] * ur[i];
        {                                               // valid() :86-101
            const double dF = chi2_ - ur2;
            const double dL = 0.5 * ((h[0] * (mu_ * h[0] - g[0]) + h[1] * (mu_ * h[1] - g[1])) + h[2] * (mu_ * h[2] - g[2]));
            if (dL > 0.0 && dF > 0.0) {
                mu_ = mu_ * std::max(1.0 / 3.0, 1 - std::pow(2 * (dF / dL) - 1, 3));
                v_ = 2.0;
                valid = true;
            } else {
                mu_ = mu_ * v_; v_ = 2 * v_;
                valid = false;
            }
        }
        if (!valid) { const double mh[3] = {-h[0], -h[1], -h[2]}; problem.update(mh); }
        ++iter;
    }
    st.iterations = iter;
    return st;
}

// -------------------------------------------------------------------------------------
// Map update shared by PFSlam2D::updateParticleMaps (src/pf_slam2d.cpp:439-509) and Slam2D::updateMaps
// (src/slam2d.cpp:247-321, identical statement for statement except the transient-map pruning): ray-cast the scan
// from `pose` into the occupancy map, forward obstacle events to the distance map, then dm.update().
// -------------------------------------------------------------------------------------
inline uint32_t update_maps_body(DynamicDistanceMap& dm, FrequencyOccupancyMap& occ, const Scan& surface, const SE2& pose,
                                 double truncated_ray_, double truncated_range_, uint64_t* ray_cells_out = nullptr,
                                 uint64_t* events_out = nullptr)
{
    Affine3 mtf = moving_tf(surface);
    Affine3 ftf = fixed_tf(pose);
    Affine3 tf = affine_mul(ftf, mtf);
    V3d wso{tf.t[0], tf.t[1], tf.t[2]};
    uint64_t ray_cells = 0, events = 0;
    const size_t num_points = surface.points.size();
    for (size_t i = 0; i < num_points; ++i) {
        V3d start = wso;
        V3d hit = affine_apply(tf, surface.points[i]);
        V3d AB{0, 0, 0};
        double ray_length = 1.0;
        bool mark_hit = true;
        if (truncated_range_ > 0.0) {                               // :467-479
            AB = V3d{hit.x - start.x, hit.y - start.y, hit.z - start.z};
            ray_length = std::sqrt(AB.x * AB.x + AB.y * AB.y + AB.z * AB.z);
            if (truncated_range_ < ray_length) {
                hit = V3d{start.x + AB.x / ray_length * truncated_range_, start.y + AB.y / ray_length * truncated_range_, start.z + AB.z / ray_length * truncated_range_};
                mark_hit = false;
            }
        }
        if (mark_hit && (truncated_ray_ > 0.0)) {                   // :481-491
            if (truncated_range_ == 0.0) {
                AB = V3d{hit.x - start.x, hit.y - start.y, hit.z - start.z};
                ray_length = std::sqrt(AB.x * AB.x + AB.y * AB.y + AB.z * AB.z);
            }
            if (truncated_ray_ < ray_length)
                start = V3d{hit.x - AB.x / ray_length * truncated_ray_, hit.y - AB.y / ray_length * truncated_ray_, hit.z - AB.z / ray_length * truncated_ray_};
        }
        V3u mhit = occ.w2m(hit);                                    // :493
        if (mark_hit) {
            bool changed = occ.setOccupied(mhit);
            if (changed) { dm.addObstacle(mhit); ++events; }
        }
        occ.computeRay(occ.w2m(start), mhit, [&](const V3u& coord) {
            ++ray_cells;
            bool changed = occ.setFree(coord);
            if (changed) { dm.removeObstacle(coord); ++events; }
        });
    }
    uint32_t processed = dm.update();                               // :508
    if (ray_cells_out) *ray_cells_out = ray_cells;
    if (events_out) *events_out = events;
    return processed;
}

// -------------------------------------------------------------------------------------
// ThreadPool -- same dispatch MODEL as src/thread_pool.cpp:52-114 (one task per particle per
// region, wait() barrier); plain mutex queue instead of moodycamel (vendor lib, not restated).
// -------------------------------------------------------------------------------------
class ThreadPool {
public:
    void init(size_t size)
    {
        if (size == 0) size = std::thread::hardware_concurrency();
        for (size_t i = 0; i < size; ++i)
            workers.emplace_back([this] {
                for (;;) {
                    std::function<void()> task;
                    {
                        std::unique_lock<std::mutex> lock(m);
                        cv.wait(lock, [this] { return stop || !tasks.empty(); });
                        if (stop && tasks.empty()) return;
                        task = std::move(tasks.front());
                        tasks.pop_front();
                    }
                    task();
                    {
                        std::unique_lock<std::mutex> lock(m);
                        if (--pending == 0) done.notify_all();
                    }
                }
            });
    }
    ~ThreadPool()
    {
        { std::unique_lock<std::mutex> lock(m); stop = true; }
        cv.notify_all();
        for (auto& t : workers) t.join();
    }
    void enqueue(std::function<void()>&& f)
    {
        { std::unique_lock<std::mutex> lock(m); ++pending; tasks.emplace_back(std::move(f)); }
        cv.notify_one();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lock(m);
        done.wait(lock, [this] { return pending == 0; });
    }
    size_t size() const { return workers.size(); }
private:
    std::vector<std::thread> workers;
    std::deque<std::function<void()>> tasks;
    std::mutex m;
    std::condition_variable cv, done;
    size_t pending = 0;
    bool stop = false;
};

// -------------------------------------------------------------------------------------
// PFSlam2D  (include/lama/pf_slam2d.h, src/pf_slam2d.cpp)
// -------------------------------------------------------------------------------------
struct PFOptions {                                     // pf_slam2d.h:132-185
    uint32_t particles = 30;
    double srr = 0.1, str = 0.2, stt = 0.1, srt = 0.2;
    double meas_sigma = 0.05, meas_sigma_gain = 3;
    double trans_thresh = 0.5, rot_thresh = 0.5;
    double l2_max = 0.5;
    double truncated_ray = 0.0, truncated_range = 0.0;
    double resolution = 0.05;
    uint32_t patch_size = 32;
    uint32_t max_iter = 100;
    int32_t threads = -1;
    uint32_t seed = 0;
};

struct ParticleCounters {   // per particle, last update() (instrumentation; not in the reference)
    uint32_t iterations = 0, evals = 0;
    uint64_t ray_cells = 0, occ_events = 0, bf_processed = 0;
    uint32_t n_match = 0, n_occ = 0, n_bf = 0, n_match_or_bf = 0;
};

struct Particle {                                      // pf_slam2d.h:67-86
    double weight = 0, normalized_weight = 0, weight_sum = 0;
    SE2 pose;
    std::vector<SE2> poses;
    std::shared_ptr<DynamicDistanceMap> dm;
    std::shared_ptr<FrequencyOccupancyMap> occ;
    ParticleCounters ctr;
    std::unordered_set<uint64_t> match_touch;   // instrumentation only
};

struct UpdateTimes { double total = 0, solving = 0, normalizing = 0, resampling = 0, mapping = 0; bool resampled = false; };

class PFSlam2D {
public:
    explicit PFSlam2D(const PFOptions& o) : options_(o), gen_(o.seed)   // pf_slam2d.cpp:105-137
    {
        truncated_ray_ = o.truncated_ray;
        truncated_range_ = o.truncated_range;
        if (options_.threads <= 1) thread_pool_ = nullptr;              // :123-128
        else { thread_pool_.reset(new ThreadPool); thread_pool_->init(options_.threads); }
        // seed == 0 -> random_device in the reference (:131-134); tests always pass a seed
        gen_.seed(options_.seed);
    }

    void setPrior(const SE2& prior) { pose_ = prior; }                  // :146-149

    bool count_touches = false;  // enable roofline instrumentation (slower)

    // src/pf_slam2d.cpp:178-312
    bool update(const Scan& surface, const SE2& odometry, double timestamp)
    {
        using clk = std::chrono::steady_clock;
        auto t0 = clk::now();
        last_times = UpdateTimes();
        current_surface_ = &surface;

        if (!has_first_scan) {                                          // :185-228
            odom_ = odometry;
            timestamps_.push_back(timestamp);
            const uint32_t P = options_.particles;
            particles_[0].assign(P, Particle());
            current_particle_set_ = 0;
            Particle& p0 = particles_[0][0];
            p0.poses.push_back(pose_);
            p0.pose = pose_;
            p0.weight = 0.0; p0.weight_sum = 0.0;
            p0.dm = std::make_shared<DynamicDistanceMap>(options_.resolution, options_.patch_size);
            p0.dm->setMaxDistance(options_.l2_max);
            p0.occ = std::make_shared<FrequencyOccupancyMap>(options_.resolution, options_.patch_size);
            updateParticleMaps(&p0);
            for (uint32_t i = 1; i < P; ++i) {
                Particle& p = particles_[0][i];
                p.poses.push_back(pose_);
                p.pose = pose_;
                p.weight = 0.0; p.weight_sum = 0.0;
                p.dm = std::make_shared<DynamicDistanceMap>(*p0.dm);
                p.occ = std::make_shared<FrequencyOccupancyMap>(*p0.occ);
            }
            has_first_scan = true;
            last_times.total = last_times.mapping = sec(t0);
            return true;
        }

        // 1. predict from odometry                                      :231-236
        SE2 odelta = pose_minus(odom_, odometry);
        odom_ = odometry;
        const uint32_t P = options_.particles;
        auto& cur = particles_[current_particle_set_];
        for (uint32_t i = 0; i < P; ++i) drawFromMotion(odelta, cur[i].pose);

        acc_trans_ += std::sqrt(odelta.tx * odelta.tx + odelta.ty * odelta.ty);   // :239
        acc_rot_ += std::fabs(se2_rotation(odelta));                               // :240
        if (acc_trans_ <= options_.trans_thresh && acc_rot_ <= options_.rot_thresh) return false;
        acc_trans_ = 0; acc_rot_ = 0;

        // 2. scan matching                                               :252-269
        auto t1 = clk::now();
        if (thread_pool_) {
            for (uint32_t i = 0; i < P; ++i) thread_pool_->enqueue([this, i]() { scanMatch(&particles_[current_particle_set_][i]); });
            thread_pool_->wait();
        } else {
            for (uint32_t i = 0; i < P; ++i) scanMatch(&cur[i]);
        }
        last_times.solving = sec(t1);

        // 3. normalize                                                   :271-277
        auto t2 = clk::now();
        normalize();
        last_times.normalizing = sec(t2);

        // 4. resample                                                    :279-287
        if (neff_ < (options_.particles * 0.5)) {
            auto t3 = clk::now();
            resample();
            last_times.resampling = sec(t3);
            last_times.resampled = true;
            ++num_resamples;
        }

        // 5. update maps                                                 :289-302
        auto t4 = clk::now();
        auto& cur2 = particles_[current_particle_set_];
        if (thread_pool_) {
            for (uint32_t i = 0; i < P; ++i) thread_pool_->enqueue([this, i]() { updateParticleMaps(&particles_[current_particle_set_][i]); });
            thread_pool_->wait();
        } else {
            for (uint32_t i = 0; i < P; ++i) updateParticleMaps(&cur2[i]);
        }
        last_times.mapping = sec(t4);
        last_times.total = sec(t0);
        return true;
    }

    size_t getBestParticleIdx() const                                   // :314-330
    {
        const auto& cur = particles_[current_particle_set_];
        size_t best_idx = 0;
        double best_ws = cur[0].weight_sum;
        for (uint32_t i = 1; i < options_.particles; ++i)
            if (best_ws < cur[i].weight_sum) { best_ws = cur[i].weight_sum; best_idx = i; }
        return best_idx;
    }
    SE2 getPose() const { return particles_[current_particle_set_][getBestParticleIdx()].pose; }
    double getNeff() const { return neff_; }
    std::vector<Particle>& particles() { return particles_[current_particle_set_]; }
    const PFOptions& options() const { return options_; }
    bool hasFirstScan() const { return has_first_scan; }

    // ---- stage-wise entry points for teacher-forced parity tests (same code as update()) ----
    void stage_set_scan(const Scan& s) { current_surface_ = &s; }
    void stage_scan_match_all() { for (auto& p : particles()) scanMatch(&p); }
    void stage_update_maps_all() { for (auto& p : particles()) updateParticleMaps(&p); }
    void stage_normalize() { normalize(); }
    void stage_resample_with(const std::vector<int32_t>& idx) { resample_apply(idx); }
    std::vector<int32_t> stage_resample_indices(double u) { return resample_indices(u); }

    // src/pf_slam2d.cpp:365-391
    void drawFromMotion(const SE2& delta, SE2& pose)
    {
        double sigma, x, y, yaw;
        double sxy = 0.3 * options_.stt;
        const double dx = delta.tx, dy = delta.ty, drot = se2_rotation(delta);
        sigma = options_.stt * std::fabs(dx) + options_.str * std::fabs(drot) + sxy * std::fabs(dy);
        x = dx + normal(sigma);
        sigma = options_.stt * std::fabs(dy) + options_.str * std::fabs(drot) + sxy * std::fabs(dx);
        y = dy + normal(sigma);
        sigma = options_.srr * std::fabs(drot) + options_.srt * std::sqrt(dx * dx + dy * dy);
        yaw = drot + normal(sigma);
        yaw = std::fmod(yaw, 2 * M_PI);
        if (yaw > M_PI) yaw -= 2 * M_PI;
        pose = se2_mul(pose, se2_from_xyr(x, y, yaw));                  // pose += Pose2D(x,y,yaw)
    }

    // src/pf_slam2d.cpp:393-414
    double calculateLikelihood(const Particle& particle) const
    {
        const Scan& surface = *current_surface_;
        Affine3 tf = affine_mul(fixed_tf(particle.pose), moving_tf(surface));
        double likelihood = 0;
        for (size_t i = 0; i < surface.points.size(); ++i) {
            V3d hit = affine_apply(tf, surface.points[i]);
            double dist = particle.dm->distance(hit, nullptr);
            likelihood += -(dist * dist) / options_.meas_sigma;
        }
        return likelihood;
    }

    // src/pf_slam2d.cpp:416-437
    void scanMatch(Particle* particle)
    {
        std::unordered_set<uint64_t> touched;
        if (count_touches) { particle->dm->reset_cache(); particle->dm->touch_ro = &touched; }
        MatchSurface2D ms(particle->dm.get(), current_surface_, particle->pose);
        CauchyWeight cauchy(0.15);
        SolveStats st = solve_gn(ms, options_.max_iter, cauchy);
        particle->pose = ms.state_;
        particle->poses.push_back(particle->pose);
        double l = calculateLikelihood(*particle);
        particle->weight_sum += l;
        particle->weight += l;
        particle->ctr.iterations = st.iterations;
        particle->ctr.evals = st.evals + 1;
        if (count_touches) {
            particle->dm->touch_ro = nullptr;
            particle->ctr.n_match = (uint32_t)touched.size();
            particle->match_touch = std::move(touched);
        }
    }

    // src/pf_slam2d.cpp:439-509
    void updateParticleMaps(Particle* particle)
    {
        const Scan& surface = *current_surface_;
        std::unordered_set<uint64_t> t_occ, t_bf;
        if (count_touches) {
            particle->occ->reset_cache(); particle->dm->reset_cache();
            particle->occ->touch_rw = &t_occ; particle->dm->touch_rw = &t_bf;
        }
        uint64_t ray_cells = 0, events = 0;
        uint32_t processed = update_maps_body(*particle->dm, *particle->occ, surface, particle->pose, truncated_ray_, truncated_range_,
                                              &ray_cells, &events);
        particle->ctr.ray_cells = ray_cells;
        particle->ctr.occ_events = events;
        particle->ctr.bf_processed = processed;
        if (count_touches) {
            particle->occ->touch_rw = nullptr; particle->dm->touch_rw = nullptr;
            particle->ctr.n_occ = (uint32_t)t_occ.size();
            particle->ctr.n_bf = (uint32_t)t_bf.size();
            std::unordered_set<uint64_t> u = t_bf;
            u.insert(particle->match_touch.begin(), particle->match_touch.end());
            particle->ctr.n_match_or_bf = (uint32_t)u.size();
        }
    }

    // src/pf_slam2d.cpp:511-535
    void normalize()
    {
        auto& cur = particles_[current_particle_set_];
        double gain = 1.0 / (options_.meas_sigma_gain * options_.particles);
        double max_l = cur[0].weight;
        const uint32_t P = options_.particles;
        for (uint32_t i = 1; i < P; ++i) if (max_l < cur[i].weight) max_l = cur[i].weight;
        double sum = 0;
        for (uint32_t i = 0; i < P; ++i) {
            cur[i].normalized_weight = std::exp(gain * (cur[i].weight - max_l));
            sum += cur[i].normalized_weight;
        }
        neff_ = 0;
        for (uint32_t i = 0; i < P; ++i) {
            cur[i].normalized_weight /= sum;
            neff_ += cur[i].normalized_weight * cur[i].normalized_weight;
        }
        neff_ = 1.0 / neff_;
    }

    // src/pf_slam2d.cpp:537-556 (index generation)
    std::vector<int32_t> resample_indices(double u01)
    {
        auto& cur = particles_[current_particle_set_];
        const uint32_t P = options_.particles;
        std::vector<int32_t> sample_idx(P);                             // zero-initialised (:540)
        double interval = 1.0 / (double)P;
        double target = interval * u01;
        double cw = 0.0;
        uint32_t n = 0;
        for (size_t i = 0; i < P; ++i) {
            cw += cur[i].normalized_weight;
            while (cw > target) {
                if (n >= P) break;   // the reference would write out of bounds here; cannot occur for sum(nw)~1
                sample_idx[n++] = (int32_t)i;
                target += interval;
            }
        }
        return sample_idx;
    }

    // src/pf_slam2d.cpp:558-574 (particle-set construction)
    void resample_apply(const std::vector<int32_t>& sample_idx)
    {
        const uint32_t P = options_.particles;
        uint8_t ps = 1 - current_particle_set_;
        particles_[ps].assign(P, Particle());
        auto& cur = particles_[current_particle_set_];
        for (size_t i = 0; i < P; ++i) {
            uint32_t idx = (uint32_t)sample_idx[i];
            particles_[ps][i] = cur[idx];
            particles_[ps][i].weight = 0.0;
            particles_[ps][i].weight_sum = cur[idx].weight_sum;
            particles_[ps][i].dm = std::make_shared<DynamicDistanceMap>(*cur[idx].dm);
            particles_[ps][i].occ = std::make_shared<FrequencyOccupancyMap>(*cur[idx].occ);
        }
        cur.clear();
        current_particle_set_ = ps;
        last_sample_idx = sample_idx;
    }

    void resample()
    {
        double u = std::uniform_real_distribution<double>(0.0, 1.0)(gen_);   // random.cpp:51-55
        resample_apply(resample_indices(u));
    }

    double normal(double stddev)                                        // random.cpp:69-73
    {
        std::normal_distribution<double> distribution(0.0, stddev);
        return distribution(gen_);
    }

    UpdateTimes last_times;
    uint32_t num_resamples = 0;
    std::vector<int32_t> last_sample_idx;

private:
    static double sec(std::chrono::steady_clock::time_point t0)
    { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

    PFOptions options_;
    std::mt19937 gen_;                                                  // random.cpp:38-39 (process-global there)
    std::vector<Particle> particles_[2];
    uint8_t current_particle_set_ = 0;
    SE2 odom_, pose_;
    double acc_trans_ = 0, acc_rot_ = 0;
    bool has_first_scan = false;
    double truncated_ray_ = 0, truncated_range_ = 0;
    double neff_ = 0;
    std::deque<double> timestamps_;
    const Scan* current_surface_ = nullptr;
    std::unique_ptr<ThreadPool> thread_pool_;
};

// -------------------------------------------------------------------------------------
// Slam2D  (include/lama/slam2d.h:91-188, src/slam2d.cpp:92-198, 247-321).  Online SLAM = one pose, one map pair.
// Strategy "gn" only (GaussNewton + CauchyWeight(0.15), src/slam2d.cpp:103-106); transient_map pruning
// (:322-373) is out of scope.
// -------------------------------------------------------------------------------------
struct AABB {                                                           // include/lama/aabb.h:41-74
    double center[3], hwidth[3];
    AABB(const double mn[3], const double mx[3])
    {
        for (int k = 0; k < 3; ++k) { const double local = mx[k] - mn[k]; hwidth[k] = local * 0.5; center[k] = mn[k] + hwidth[k]; }
    }
    bool testIntersection(const AABB& o) const
    {
        bool r = true;
        for (int k = 0; k < 3; ++k) r = r && (std::abs(center[k] - o.center[k]) <= (hwidth[k] + o.hwidth[k]));
        return r;
    }
};

struct SlamOptions {                                    // slam2d.h:91-125
    double trans_thresh = 0.5, rot_thresh = 0.5;
    double l2_max = 0.5;
    double truncated_ray = 0.0, truncated_range = 0.0;
    double resolution = 0.05;
    uint32_t patch_size = 32;
    uint32_t max_iter = 100;
    bool transient_map = false;
    bool lm = false;                                     // Options::strategy == "lm" (src/slam2d.cpp:226-233)
};

class Slam2D {
public:
    explicit Slam2D(const SlamOptions& o) : opt_(o), dm_(o.resolution, o.patch_size), occ_(o.resolution, o.patch_size)
    {
        dm_.setMaxDistance(o.l2_max);                                   // src/slam2d.cpp:94-95
    }
    void setPose(const SE2& p) { pose_ = p; }
    void set_lm(bool on) { opt_.lm = on; }
    SE2 getPose() const { return pose_; }
    uint32_t getNumberOfProcessedCells() const { return processed_; }
    DynamicDistanceMap& dm() { return dm_; }
    FrequencyOccupancyMap& occ() { return occ_; }
    SolveStats last_solve;

    bool enoughMotion(const SE2& odometry) const                        // :129-141
    {
        if (!has_first_scan) return true;
        SE2 odelta = pose_minus(odom_, odometry);
        if (std::sqrt(odelta.tx * odelta.tx + odelta.ty * odelta.ty) <= opt_.trans_thresh && std::fabs(se2_rotation(odelta)) <= opt_.rot_thresh)
            return false;
        return true;
    }

    bool update(const Scan& surface, const SE2& odometry, double /*timestamp*/)   // :143-198
    {
        if (!has_first_scan) {
            odom_ = odometry;
            updateMaps(surface);
            has_first_scan = true;
            return true;
        }
        SE2 odelta = pose_minus(odom_, odometry);
        SE2 ppose = pose_plus(pose_, odelta);
        if (std::sqrt(odelta.tx * odelta.tx + odelta.ty * odelta.ty) <= opt_.trans_thresh && std::fabs(se2_rotation(odelta)) <= opt_.rot_thresh)
            return false;
        pose_ = ppose;
        odom_ = odometry;
        MatchSurface2D ms(&dm_, &surface, pose_);
        CauchyWeight cauchy(0.15);
        last_solve = opt_.lm ? solve_lm(ms, opt_.max_iter, cauchy) : solve_gn(ms, opt_.max_iter, cauchy);
        pose_ = ms.state_;
        updateMaps(surface);
        return true;
    }

private:
    void updateMaps(const Scan& surface)                                // :247-379
    {
        processed_ = update_maps_body(dm_, occ_, surface, pose_, opt_.truncated_ray, opt_.truncated_range);
        deleted_last = 0;
        if (!opt_.transient_map) return;
        // 4. transient map (:322-379): box of the (range-truncated) hits, symmetric about the pose at TWICE the largest
        // offset, expanded by twice the distance map's range; patches that do not meet it are deleted from both maps
        const Affine3 tf = affine_mul(fixed_tf(pose_), moving_tf(surface));
        double mn[3], mx[3];
        for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<double>::max(); mx[k] = -std::numeric_limits<double>::max(); }
        for (size_t i = 0; i < surface.points.size(); ++i) {
            V3d hit = affine_apply(tf, surface.points[i]);
            if (opt_.truncated_range > 0.0) {
                const V3d AB{hit.x - tf.t[0], hit.y - tf.t[1], hit.z - tf.t[2]};
                const double ray_length = std::sqrt((AB.x * AB.x + AB.y * AB.y) + AB.z * AB.z);
                if (opt_.truncated_range < ray_length)
                    hit = V3d{tf.t[0] + AB.x / ray_length * opt_.truncated_range, tf.t[1] + AB.y / ray_length * opt_.truncated_range,
                              tf.t[2] + AB.z / ray_length * opt_.truncated_range};
            }
            mn[0] = std::min(mn[0], hit.x); mn[1] = std::min(mn[1], hit.y); mn[2] = std::min(mn[2], hit.z);
            mx[0] = std::max(mx[0], hit.x); mx[1] = std::max(mx[1], hit.y); mx[2] = std::max(mx[2], hit.z);
        }
        mn[2] = mx[2] = 0;
        const double xdist = std::max(pose_.tx - mn[0], mx[0] - pose_.tx) * 2.0;
        const double ydist = std::max(pose_.ty - mn[1], mx[1] - pose_.ty) * 2.0;
        mn[0] = pose_.tx - xdist; mn[1] = pose_.ty - ydist;
        mx[0] = pose_.tx + xdist; mx[1] = pose_.ty + ydist;
        AABB a(mn, mx);
        for (int k = 0; k < 3; ++k) a.hwidth[k] += 2.0 * dm_.maxDistance();
        std::vector<V3u> to_remove;
        dm_.visit_all_patches([&](const V3u& origin) {
            const uint32_t length = occ_.patch_length;
            V3d ws = occ_.m2w(origin);
            V3d we = occ_.m2w(V3u{origin.x + length, origin.y + length, origin.z + 0});
            ws.z = we.z = 0.0;
            const double b0[3] = {ws.x, ws.y, ws.z}, b1[3] = {we.x, we.y, we.z};
            if (a.testIntersection(AABB(b0, b1))) return;
            to_remove.push_back(origin);
        });
        for (auto& coord : to_remove) { occ_.deletePatchAt(coord); if (dm_.deletePatchAt(coord)) ++deleted_last; }
    }
public:
    uint32_t deleted_last = 0;
private:
    SlamOptions opt_;
    DynamicDistanceMap dm_;
    FrequencyOccupancyMap occ_;
    SE2 odom_, pose_;
    bool has_first_scan = false;
    uint32_t processed_ = 0;
};

// -------------------------------------------------------------------------------------
// LidarOdometry2D  (include/lama/lidar_odometry_2d.h:45-80, src/lidar_odometry_2d.cpp:42-200): scan-to-map odometry on a
// log-odds occupancy map + distance map (max distance 1 m) that keeps only the patches near the latest scan.
// -------------------------------------------------------------------------------------
// LidarOdometry2D::updateMaps, ray-cast part (src/lidar_odometry_2d.cpp:85-126): hit -> setOccupied/addObstacle, the
// last metre of the ray -> setFree/removeObstacle, then distance_map->update().  mn/mx: bounding box of the hits.
inline uint32_t lidar_update_maps_body(DynamicDistanceMap& dm_, ProbabilisticOccupancyMap& occ_, const Scan& surface, const SE2& odom,
                                       double mn[3], double mx[3])
{
    const Affine3 mtf = moving_tf(surface);
    const Affine3 ftf = fixed_tf(odom);
    const Affine3 tf = affine_mul(ftf, mtf);
    const V3d wso{tf.t[0], tf.t[1], tf.t[2]};
    for (int k = 0; k < 3; ++k) { mn[k] = std::numeric_limits<double>::max(); mx[k] = -std::numeric_limits<double>::max(); }
    for (size_t i = 0; i < surface.points.size(); ++i) {
        V3d start = wso;
        const V3d hit = affine_apply(tf, surface.points[i]);
        const V3d AB{hit.x - start.x, hit.y - start.y, hit.z - start.z};
        const double ray_length = std::sqrt((AB.x * AB.x + AB.y * AB.y) + AB.z * AB.z);
        if (ray_length >= 1.0) start = V3d{hit.x - AB.x / ray_length, hit.y - AB.y / ray_length, hit.z - AB.z / ray_length};
        const V3u mhit = occ_.w2m(hit);
        mn[0] = std::min(mn[0], hit.x); mn[1] = std::min(mn[1], hit.y); mn[2] = std::min(mn[2], hit.z);
        mx[0] = std::max(mx[0], hit.x); mx[1] = std::max(mx[1], hit.y); mx[2] = std::max(mx[2], hit.z);
        if (occ_.setOccupied(mhit)) dm_.addObstacle(mhit);
        occ_.computeRay(occ_.w2m(start), mhit, [&](const V3u& coord) {
            if (occ_.setFree(coord)) dm_.removeObstacle(coord);
        });
    }
    return dm_.update();
}

class LidarOdometry2D {
public:
    explicit LidarOdometry2D(double resolution = 0.05, uint32_t max_iter = 100)      // :42-52
        : dm_(resolution), occ_(resolution), max_iter_(max_iter)
    {
        dm_.setMaxDistance(1.0);
    }
    DynamicDistanceMap& dm() { return dm_; }
    ProbabilisticOccupancyMap& occ() { return occ_; }
    SE2 odom, map_update_odom;
    SolveStats last_solve;
    uint32_t deleted_last = 0, map_updates = 0;

    bool update(const Scan& surface, double /*timestamp*/)               // :60-83
    {
        if (!has_first_scan) {
            updateMaps(surface);
            has_first_scan = true;
            return true;
        }
        MatchSurface2D ms(&dm_, &surface, odom);
        CauchyWeight cauchy(0.15);
        last_solve = solve_gn(ms, max_iter_, cauchy);
        odom = ms.state_;
        SE2 odelta = pose_minus(map_update_odom, odom);
        if (std::sqrt(odelta.tx * odelta.tx + odelta.ty * odelta.ty) > 0.1 || std::abs(se2_rotation(odelta)) > 0.5) {
            updateMaps(surface);
            map_update_odom = odom;
        }
        return true;
    }

    void updateMaps(const Scan& surface)                                 // :85-200
    {
        ++map_updates;
        double mn[3], mx[3];
        lidar_update_maps_body(dm_, occ_, surface, odom, mn, mx);
        // transient map: drop the patches whose box does not meet the (symmetrised, expanded) box of the scan  :128-199
        mn[2] = mx[2] = 0;
        const double xdist = std::max(odom.tx - mn[0], mx[0] - odom.tx);
        const double ydist = std::max(odom.ty - mn[1], mx[1] - odom.ty);
        mn[0] = odom.tx - xdist; mn[1] = odom.ty - ydist;
        mx[0] = odom.tx + xdist; mx[1] = odom.ty + ydist;
        AABB a(mn, mx);
        for (int k = 0; k < 3; ++k) a.hwidth[k] += 2.0 * dm_.maxDistance();
        std::vector<V3u> to_remove;
        dm_.visit_all_patches([&](const V3u& origin) {
            const uint32_t length = occ_.patch_length;
            V3d ws = occ_.m2w(origin);
            V3d we = occ_.m2w(V3u{origin.x + length, origin.y + length, origin.z + 0});
            ws.z = we.z = 0.0;
            const double b0[3] = {ws.x, ws.y, ws.z}, b1[3] = {we.x, we.y, we.z};
            AABB b(b0, b1);
            if (a.testIntersection(b)) return;
            to_remove.push_back(origin);
        });
        deleted_last = 0;
        for (auto& coord : to_remove) { occ_.deletePatchAt(coord); if (dm_.deletePatchAt(coord)) ++deleted_last; }
    }

private:
    DynamicDistanceMap dm_;
    ProbabilisticOccupancyMap occ_;
    uint32_t max_iter_;
    bool has_first_scan = false;
};

// -------------------------------------------------------------------------------------
// Loc2D  (include/lama/loc2d.h:47-165, src/loc2d.cpp:46-192): localisation on a fixed distance map.
// Restated: Init, setPose, enoughMotion, update() incl. Solve(..., &cov) and RMSE, triggerGlobalLocalization /
// globalLocalization (:194-197, :249-286) and addSamplingCovariance (:199-247).  Strategy "lm" is out of scope.
// -------------------------------------------------------------------------------------
// Solver::calculateCovariance (src/nlls/solver.cpp:133-150): rank of J by column-pivoted Householder QR
// (Eigen ColPivHouseholderQR::rank(): pivots above epsilon * diagonalSize() = 3 eps relative to the largest pivot --
// Eigen 3.3 is not in this image, the threshold is restated from its published source); full rank -> (J^T J)^-1,
// else V diag(|sv| > 1e-3 ? 1/sv^2 : 3.0) V^T from the thin SVD of J (svd_cov3 below).
inline int colpiv_qr_rank(std::vector<double> J /*n x 3 row-major, copied*/, size_t n)
{
    double r[3];
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < 3; ++k) {
        int best = k; double bn = -1;
        for (int c = k; c < 3; ++c) { double s2 = 0; for (size_t i = k; i < n; ++i) s2 += J[3 * i + perm[c]] * J[3 * i + perm[c]]; if (s2 > bn) { bn = s2; best = c; } }
        std::swap(perm[k], perm[best]);
        const int c = perm[k];
        double norm = std::sqrt(bn);
        if (norm == 0) { r[k] = 0; continue; }
        const double x0 = J[3 * k + c];
        const double alpha = x0 > 0 ? -norm : norm;
        std::vector<double> v(n - k);
        for (size_t i = k; i < n; ++i) v[i - k] = J[3 * i + c];
        v[0] -= alpha;
        double vn2 = 0; for (double t : v) vn2 += t * t;
        if (vn2 > 0)
            for (int cc = k; cc < 3; ++cc) {
                const int col = perm[cc];
                double dot = 0; for (size_t i = k; i < n; ++i) dot += v[i - k] * J[3 * i + col];
                const double f = 2.0 * dot / vn2;
                for (size_t i = k; i < n; ++i) J[3 * i + col] -= f * v[i - k];
            }
        r[k] = std::fabs(alpha);
    }
    const double maxp = std::max(r[0], std::max(r[1], r[2]));
    const double thr = 2.220446049250313e-16 * 3.0 * maxp;
    int rank = 0;
    for (int k = 0; k < 3; ++k) if (r[k] > thr) ++rank;
    return rank;
}

// The rank-deficient branch (src/nlls/solver.cpp:143-149): singular values and right singular vectors of the n x 3
// Jacobian by one-sided (Hestenes) Jacobi rotations -- the same decomposition Eigen::JacobiSVD returns, up to the signs /
// order of the columns of V, which V D V^T does not depend on.
inline void svd_cov3(std::vector<double> U /*n x 3 row-major, copied*/, size_t n, double cov[9])
{
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (size_t i = 0; i < n; ++i) { alpha += U[3 * i + p] * U[3 * i + p]; beta += U[3 * i + q] * U[3 * i + q]; gamma += U[3 * i + p] * U[3 * i + q]; }
                if (gamma == 0.0 || std::fabs(gamma) <= 1e-300 + 2.220446049250313e-16 * std::sqrt(alpha * beta)) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
                for (size_t i = 0; i < n; ++i) {
                    const double up = U[3 * i + p], uq = U[3 * i + q];
                    U[3 * i + p] = c * up - sn * uq; U[3 * i + q] = sn * up + c * uq;
                }
                for (int i = 0; i < 3; ++i) {
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - sn * vq; V[i][q] = sn * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
    double d[3];
    for (int j = 0; j < 3; ++j) {
        double s2 = 0; for (size_t i = 0; i < n; ++i) s2 += U[3 * i + j] * U[3 * i + j];
        const double sv = std::sqrt(s2);
        d[j] = sv > 1.e-3 ? 1.0 / (sv * sv) : 3.0;                                      // :147-148
    }
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) cov[3 * a + b] = V[a][0] * d[0] * V[b][0] + V[a][1] * d[1] * V[b][1] + V[a][2] * d[2] * V[b][2];
}

inline void inverse3(const double A[3][3], double out[9])
{
    const double a = A[0][0], b = A[0][1], c = A[0][2], d = A[1][0], e = A[1][1], f = A[1][2], g = A[2][0], h = A[2][1], i = A[2][2];
    const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    out[0] = (e * i - f * h) / det; out[1] = (c * h - b * i) / det; out[2] = (b * f - c * e) / det;
    out[3] = (f * g - d * i) / det; out[4] = (a * i - c * g) / det; out[5] = (c * d - a * f) / det;
    out[6] = (d * h - e * g) / det; out[7] = (b * g - a * h) / det; out[8] = (a * e - b * d) / det;
}

struct LocOptions {                                     // src/loc2d.cpp:46-58
    double trans_thresh = 0.5, rot_thresh = 0.5, l2_max = 1.0, resolution = 0.05;
    uint32_t patch_size = 32, max_iter = 100;
    uint32_t gloc_particles = 3000, gloc_iters = 10;
    double gloc_thresh = 0.15, cov_blend = 0.0;
    bool lm = false;                                     // Options::strategy == "lm" (src/loc2d.cpp:288-294)
};

class Loc2D {
public:
    explicit Loc2D(const LocOptions& o) : opt_(o), dm_(o.resolution, o.patch_size), occ_(o.resolution, o.patch_size)     // Init :61-108
    {
        dm_.setMaxDistance(o.l2_max);
        for (int k = 0; k < 9; ++k) cov_[k] = (k % 4 == 0) ? 1.0 : 0.0;
        cov_blend_ = std::max(std::min(o.cov_blend, 1.0), 0.0);                        // :90
        const double sstep = dm_.resolution;                                            // :95-107
        steps_.push_back({0.0, 0.0});
        for (int i = 1; i <= 20; ++i) {
            steps_.push_back({i * sstep, 0.0});   steps_.push_back({0.0, i * sstep});
            steps_.push_back({-i * sstep, 0.0});  steps_.push_back({0.0, -i * sstep});
            steps_.push_back({i * sstep, i * sstep});   steps_.push_back({-i * sstep, i * sstep});
            steps_.push_back({i * sstep, -i * sstep});  steps_.push_back({-i * sstep, -i * sstep});
        }
    }
    DynamicDistanceMap& dm() { return dm_; }
    SimpleOccupancyMap& occ() { return occ_; }
    void set_lm(bool on) { opt_.lm = on; }
    void triggerGlobalLocalization() { do_gloc_ = true; }                              // :194-197
    bool globalLocalizationIsActive() const { return do_gloc_; }
    // instrumentation for the parity tests: the candidates of the last globalLocalization call
    std::vector<SE2> gloc_poses;
    std::vector<double> gloc_errors;
    std::vector<double> sampling_l;
    void setPose(const SE2& p) { pose_ = p; has_first_scan = false; }                  // loc2d.h:136-137
    SE2 getPose() const { return pose_; }
    const double* getCovar() const { return cov_; }
    double getRMSE() const { return rmse_; }
    uint32_t lastIterations() const { return iters_; }
    bool rankDeficient() const { return rank_deficient_; }

    bool enoughMotion(const SE2& odometry) const                                       // :113-124
    {
        if (!has_first_scan) return true;
        SE2 odelta = pose_minus(odom_, odometry);
        if (std::sqrt(odelta.tx * odelta.tx + odelta.ty * odelta.ty) <= opt_.trans_thresh && std::fabs(se2_rotation(odelta)) <= opt_.rot_thresh) return false;
        return true;
    }

    bool update(const Scan& surface, const SE2& odometry, double, bool force_update = false)   // :126-192
    {
        if (!has_first_scan) {
            odom_ = odometry;
            has_first_scan = true;
            if (!force_update) return true;
            rmse_ = rmse_at(surface, pose_);
        }
        SE2 odelta = pose_minus(odom_, odometry);
        SE2 ppose = pose_plus(pose_, odelta);
        if (!force_update && !enoughMotion(odometry)) return false;
        pose_ = ppose;
        odom_ = odometry;
        if (do_gloc_) {                                                                 // :157-168
            if (gloc_cur_iter_ < opt_.gloc_iters) {
                ++gloc_cur_iter_;
                globalLocalization(surface);
            } else {
                do_gloc_ = false;
                gloc_cur_iter_ = 0;
            }
        }
        MatchSurface2D ms(&dm_, &surface, pose_);
        CauchyWeight cauchy(0.15);
        SolveStats st = opt_.lm ? solve_lm(ms, opt_.max_iter, cauchy) : solve_gn(ms, opt_.max_iter, cauchy);
        iters_ = st.iterations;
        // covariance branch of Solver::solve (src/nlls/solver.cpp:109-116)
        std::vector<double> r, J;
        ms.eval(r, &J);
        for (size_t i = 0; i < r.size(); ++i) {
            const double w = std::sqrt(cauchy.value(r[i]));
            J[3 * i] *= w; J[3 * i + 1] *= w; J[3 * i + 2] *= w;
        }
        double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
        for (size_t i = 0; i < r.size(); ++i)
            for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) A[a][b] += J[3 * i + a] * J[3 * i + b];
        rank_deficient_ = colpiv_qr_rank(J, r.size()) != 3;
        if (!rank_deficient_) inverse3(A, cov_); else svd_cov3(J, r.size(), cov_);
        pose_ = ms.state_;
        if (cov_blend_ > 0.0) addSamplingCovariance(surface);                           // :175-176
        rmse_ = rmse_at(surface, pose_);
        if (do_gloc_ && rmse_ < opt_.gloc_thresh) { do_gloc_ = false; gloc_cur_iter_ = 0; }   // :182-189
        return true;
    }

    // :249-286.  The candidates come from the process-wide random::uniform() stream: x, y until the cell is free in
    // the occupancy map, then the heading; the first candidate with the smallest squared residual norm wins.
    void globalLocalization(const Scan& surface)
    {
        V3d mn, mx;
        occ_.bounds(mn, mx);
        const double diff0 = mx.x - mn.x, diff1 = mx.y - mn.y;
        double best_error = std::numeric_limits<double>::max();
        gloc_poses.clear(); gloc_errors.clear();
        for (uint32_t i = 0; i < opt_.gloc_particles; ++i) {
            double x, y, a;
            for (;;) {
                x = mn.x + random::uniform() * diff0;
                y = mn.y + random::uniform() * diff1;
                if (!occ_.isFree(V3d{x, y, 0.0})) continue;
                a = random::uniform() * 2 * M_PI - M_PI;
                break;
            }
            const SE2 p = se2_from_xyr(x, y, a);
            MatchSurface2D ms(&dm_, &surface, p);
            std::vector<double> r;
            ms.eval(r, nullptr);
            double error = 0;
            for (double v : r) error += v * v;
            gloc_poses.push_back(p); gloc_errors.push_back(error);
            if (error < best_error) { best_error = error; pose_ = p; }
        }
    }

    // :199-247  (Olson 2009 sampling covariance around the solution, blended into the xy block of cov_)
    void addSamplingCovariance(const Scan& surface)
    {
        double K[2][2] = {{0, 0}, {0, 0}}, u[2] = {0, 0}, ssum = 0;
        const Affine3 mtf = moving_tf(surface);
        const size_t num_points = surface.points.size();
        const size_t step = std::max(num_points / 100, size_t(1));
        double Raa[3][3];
        angle_axis_z(se2_rotation(pose_), Raa);
        sampling_l.clear();
        for (size_t k = 0; k < steps_.size(); ++k) {
            const double x = pose_.tx + steps_[k][0], y = pose_.ty + steps_[k][1];
            const double trans[3] = {x, y, 0.0};
            const Affine3 tf = affine_mul(affine_from(trans, Raa), mtf);
            double l = 0.0;
            for (size_t i = 0; i < num_points; i += step) {
                const V3d hit = affine_apply(tf, surface.points[i]);
                const double dist = dm_.distance(dm_.w2m(hit));
                const double e = std::exp(-(dist * dist) / 0.01);
                l += e * e * e;
            }
            sampling_l.push_back(l);
            K[0][0] = K[0][0] + x * x * l; K[0][1] = K[0][1] + x * y * l;               // trans.head<2>() * trans.head<2>()^T * l
            K[1][0] = K[1][0] + y * x * l; K[1][1] = K[1][1] + y * y * l;
            u[0] = u[0] + x * l; u[1] = u[1] + y * l;
            ssum = ssum + l;
        }
        const double a1 = 1.0 / ssum, a2 = 1.0 / (ssum * ssum);
        const double sc[2][2] = {{a1 * K[0][0] - a2 * u[0] * u[0], a1 * K[0][1] - a2 * u[0] * u[1]},
                                 {a1 * K[1][0] - a2 * u[1] * u[0], a1 * K[1][1] - a2 * u[1] * u[1]}};
        const double alpha = cov_blend_;
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 2; ++c) cov_[3 * r + c] = alpha * sc[r][c] + (1.0 - alpha) * cov_[3 * r + c];
    }

private:
    double rmse_at(const Scan& surface, const SE2& pose) const                          // :139-142, :178-180
    {
        MatchSurface2D ms(&dm_, &surface, pose);
        std::vector<double> r;
        ms.eval(r, nullptr);
        double s2 = 0;
        for (double x : r) s2 += x * x;
        return std::sqrt(s2 / ((double)(surface.points.size() - 1)));
    }
    LocOptions opt_;
    DynamicDistanceMap dm_;
    SimpleOccupancyMap occ_;
    std::vector<std::array<double, 2>> steps_;
    double cov_blend_ = 0.0;
    bool do_gloc_ = false;
    uint32_t gloc_cur_iter_ = 0;
    SE2 odom_, pose_;
    double cov_[9];
    double rmse_ = 0.0;
    uint32_t iters_ = 0;
    bool has_first_scan = false, rank_deficient_ = false;
};

// -------------------------------------------------------------------------------------
// SE2 pose-graph linearisation: what minisam's linearzationLowerHessian
// (vendor/minisam/minisam/nonlinear/linearization.cpp:150-272,290-341) accumulates for the factor graphs of
// SimplePGO::optimize (src/simple_pgo.cpp:48-105) and GraphSlam2D::optimizePoseGraph (src/graph_slam2d.cpp:394-430):
// PriorFactor<SE2d> (slam/PriorFactor.h:50-62) and BetweenFactor<SE2d> (slam/BetweenFactor.h:50-68) with
// DiagonalLoss (core/LossFunction.cpp:95-113), Sophus traits (geometry/Sophus.h:45-74), SE2 log/Adj
// (include/lama/sophus/se2.hpp:125-133,519-542).  Output as dense 3x3 blocks (the sparse scatter is the caller's):
//   err[f]   : whitened error of factor f
//   Hdiag[v] : sum over the factors touching v, in factor order, of J_v^T J_v   (row-major 3x3)
//   Hoff[f]  : J_i^T J_j of a between factor (rows: columns of J_i; zero for a prior)
//   b[v]     : Atb segment = - sum J_v^T err
// -------------------------------------------------------------------------------------
struct PgoFactor { int32_t i, j; SE2 meas; double sqrt_info[3]; };     // j < 0: prior on i

inline void se2_log(const SE2& g, double out[3])                        // se2.hpp:519-542
{
    const double theta = std::atan2(g.s, g.c);
    out[2] = theta;
    const double halftheta = 0.5 * theta;
    double h;
    const double real_minus_one = g.c - 1.;
    if (std::abs(real_minus_one) < 1e-10) h = 1. - (1. / 12) * theta * theta;
    else h = -(halftheta * g.s) / (real_minus_one);
    out[0] = h * g.tx + halftheta * g.ty;                                // V_inv * translation
    out[1] = -halftheta * g.tx + h * g.ty;
}
inline void se2_adj(const SE2& g, double A[3][3])                       // se2.hpp:125-133
{
    A[0][0] = g.c; A[0][1] = -g.s; A[0][2] = g.ty;
    A[1][0] = g.s; A[1][1] = g.c;  A[1][2] = -g.tx;
    A[2][0] = 0;   A[2][1] = 0;    A[2][2] = 1;
}
inline void pgo_factor_terms(const PgoFactor& f, const std::vector<SE2>& x, double e[3], double Ji[3][3], double Jj[3][3])
{
    if (f.j < 0) {                                                       // PriorFactor: Local(prior, x), identity Jacobian
        se2_log(se2_mul(se2_inverse(f.meas), x[f.i]), e);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { Ji[r][c] = r == c ? 1.0 : 0.0; Jj[r][c] = 0.0; }
    } else {                                                             // BetweenFactor
        const SE2& v1 = x[f.i]; const SE2& v2 = x[f.j];
        const SE2 diff = se2_mul(se2_inverse(v1), v2);
        se2_log(se2_mul(se2_inverse(f.meas), diff), e);
        double Hinv[3][3], Hcmp1[3][3];
        se2_adj(v1, Hinv);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Hinv[r][c] = -Hinv[r][c];       // InverseJacobian: -Adj
        se2_adj(se2_inverse(v2), Hcmp1);                                                        // ComposeJacobians: H1 = s2.inverse().Adj()
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            Ji[r][c] = (Hcmp1[r][0] * Hinv[0][c] + Hcmp1[r][1] * Hinv[1][c]) + Hcmp1[r][2] * Hinv[2][c];
            Jj[r][c] = r == c ? 1.0 : 0.0;
        }
    }
    for (int r = 0; r < 3; ++r) {                                        // DiagonalLoss::weightInPlace
        e[r] = e[r] * f.sqrt_info[r];
        for (int c = 0; c < 3; ++c) { Ji[r][c] *= f.sqrt_info[r]; Jj[r][c] *= f.sqrt_info[r]; }
    }
}
inline void pgo_linearize(const std::vector<SE2>& x, const std::vector<PgoFactor>& factors,
                          std::vector<double>& err, std::vector<double>& Hdiag, std::vector<double>& Hoff, std::vector<double>& b, double& chi2)
{
    const size_t N = x.size(), F = factors.size();
    err.assign(3 * F, 0.0); Hdiag.assign(9 * N, 0.0); Hoff.assign(9 * F, 0.0); b.assign(3 * N, 0.0);
    chi2 = 0.0;
    for (size_t k = 0; k < F; ++k) {
        const PgoFactor& f = factors[k];
        double e[3], Ji[3][3], Jj[3][3];
        pgo_factor_terms(f, x, e, Ji, Jj);
        for (int r = 0; r < 3; ++r) { err[3 * k + r] = e[r]; chi2 += e[r] * e[r]; }
        auto JtJ = [](const double A[3][3], const double B[3][3], int a, int c) { return (A[0][a] * B[0][c] + A[1][a] * B[1][c]) + A[2][a] * B[2][c]; };
        auto Jte = [&](const double A[3][3], int a) { return (A[0][a] * e[0] + A[1][a] * e[1]) + A[2][a] * e[2]; };
        for (int a = 0; a < 3; ++a) {
            b[3 * f.i + a] -= Jte(Ji, a);
            for (int c = 0; c < 3; ++c) Hdiag[9 * f.i + 3 * a + c] += JtJ(Ji, Ji, a, c);
        }
        if (f.j >= 0)
            for (int a = 0; a < 3; ++a) {
                b[3 * f.j + a] -= Jte(Jj, a);
                for (int c = 0; c < 3; ++c) { Hdiag[9 * f.j + 3 * a + c] += JtJ(Jj, Jj, a, c); Hoff[9 * k + 3 * a + c] = JtJ(Ji, Jj, a, c); }
            }
    }
}

} // namespace orc
