// sdm_io.cpp -- `.sdm` files and export images for maps downloaded from the device (include/lama/sdm_io.h).
#include "lama/sdm_io.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace lama {
namespace sdm {

namespace {

// include/lama/sdm/map.h:72,75,95-103.  The reference writes the struct as it lies in memory on x86-64 / LP64:
// magic @0, version @4, (2 pad), cell_size @8, patch_length @12, num_patches (size_t) @16, resolution (float) @24,
// is_3d (bool) @28, (3 pad) -> 32 bytes.
constexpr uint32_t MAGIC = 0x6d64732e;
constexpr uint16_t IO_VERSION = 0x0103;
constexpr uint64_t UNIVERSAL_CONSTANT = 2642244;
struct IOHeader {
    uint32_t magic;
    uint16_t version;
    uint32_t cell_size;
    uint32_t patch_length;
    size_t num_patches;
    float resolution;
    bool is_3d;
};
static_assert(sizeof(IOHeader) == 32, "IOHeader must match the reference's in-memory layout");

} // namespace

bool write(const HostMap& m, const std::string& filename)               // src/sdm/map.cpp:489-527
{
    const size_t pb = (size_t)m.patchVolume() * m.cellSize();
    if (m.cells.size() != m.ids.size() * pb || m.masks.size() != m.ids.size() * 16 || m.patch_length != 32) return false;
    std::ofstream f(filename.c_str(), std::ios::out | std::ios::binary | std::ios::trunc);
    if (!f.is_open()) return false;
    IOHeader h;
    std::memset(&h, 0, sizeof(h));
    h.magic = MAGIC; h.version = IO_VERSION; h.cell_size = m.cellSize(); h.patch_length = m.patch_length;
    h.num_patches = m.ids.size(); h.resolution = (float)m.resolution; h.is_3d = false;
    f.write((const char*)&h, sizeof(h));
    if (!f) return false;
    if (m.kind == kDistanceMap) f.write((const char*)&m.max_sqdist, sizeof(m.max_sqdist));      // writeParameters
    for (size_t k = 0; k < m.ids.size(); ++k) {
        f.write((const char*)&m.ids[k], sizeof(uint64_t));
        f.write((const char*)&m.cells[k * pb], (std::streamsize)pb);                             // Container::write
        f.write((const char*)&m.masks[k * 16], 16 * sizeof(uint64_t));
    }
    f.close();
    return !f.fail();
}

bool read(HostMap& m, const std::string& filename)                      // src/sdm/map.cpp:529-575
{
    std::ifstream f(filename.c_str(), std::ios::in | std::ios::binary);
    if (!f.is_open()) return false;
    IOHeader h;
    f.read((char*)&h, sizeof(h));
    if (!f) return false;
    if (h.magic != MAGIC || h.version != IO_VERSION) return false;
    if (h.is_3d || h.patch_length != 32) return false;                   // the device path is 2-D, 32 x 32 patches
    if (h.cell_size == 10) m.kind = kDistanceMap;
    else if (h.cell_size == 4) m.kind = kFrequencyOccupancyMap;
    else if (h.cell_size == 1) m.kind = kSimpleOccupancyMap;
    else return false;
    m.resolution = h.resolution;                                         // (float -> double, as the reference)
    m.patch_length = h.patch_length;
    if (m.kind == kDistanceMap) { f.read((char*)&m.max_sqdist, sizeof(m.max_sqdist)); if (!f) return false; }
    const size_t pb = (size_t)m.patchVolume() * m.cellSize();
    m.ids.assign(h.num_patches, 0);
    m.cells.assign(h.num_patches * pb, 0);
    m.masks.assign(h.num_patches * 16, 0);
    for (size_t k = 0; k < h.num_patches; ++k) {
        f.read((char*)&m.ids[k], sizeof(uint64_t));
        if (!f) return false;
        f.read((char*)&m.cells[k * pb], (std::streamsize)pb);
        f.read((char*)&m.masks[k * 16], 16 * sizeof(uint64_t));
        if (!f) return false;
    }
    return true;
}

void build_image(const HostMap& m, Image& image)                         // src/sdm/export.cpp:46-95
{
    // Map::bounds (src/sdm/map.cpp:139-157): patch anchors p2m(idx), max + patch_length
    uint32_t lo[2] = {0xFFFFFFFFu, 0xFFFFFFFFu}, hi[2] = {0, 0};
    for (uint64_t idx : m.ids) {
        const uint32_t ax = (uint32_t)((idx / UNIVERSAL_CONSTANT) << 5), ay = (uint32_t)((idx % UNIVERSAL_CONSTANT) << 5);
        lo[0] = std::min(lo[0], ax); lo[1] = std::min(lo[1], ay);
        hi[0] = std::max(hi[0], ax); hi[1] = std::max(hi[1], ay);
    }
    if (m.ids.empty()) { image.width = image.height = 0; image.data.clear(); return; }
    hi[0] += m.patch_length; hi[1] += m.patch_length;
    image.width = hi[0] - lo[0]; image.height = hi[1] - lo[1];
    image.data.assign((size_t)image.width * image.height, m.kind == kDistanceMap ? 127 : 90);
    const uint32_t cs = m.cellSize();
    const size_t pb = (size_t)m.patchVolume() * cs;
    const double maxdist = std::sqrt((double)m.max_sqdist) * m.resolution;       // DynamicDistanceMap::maxDistance :155-158
    for (size_t k = 0; k < m.ids.size(); ++k) {
        const uint32_t ax = (uint32_t)((m.ids[k] / UNIVERSAL_CONSTANT) << 5), ay = (uint32_t)((m.ids[k] % UNIVERSAL_CONSTANT) << 5);
        for (uint32_t c = 0; c < 1024; ++c) {
            if (!((m.masks[k * 16 + (c >> 6)] >> (c & 63)) & 1ull)) continue;     // Container::begin_on: cells whose mask bit is on
            const uint32_t u = ax + (c & 31) - lo[0], v = ay + (c >> 5) - lo[1];
            const uint8_t* cell = &m.cells[k * pb + (size_t)c * cs];
            uint8_t px;
            if (m.kind == kDistanceMap) {
                uint16_t sq; std::memcpy(&sq, cell + 6, 2);                       // distance_t {int16 obstacle[3]; uint16 sqdist; bool valid; bool queued}
                const bool valid = cell[8] != 0;
                const double d = valid ? std::sqrt((double)sq) * m.resolution : maxdist;   // :140-147
                px = (uint8_t)(d * 255 / maxdist);
            } else if (m.kind == kFrequencyOccupancyMap) {
                uint16_t occ, vis; std::memcpy(&occ, cell, 2); std::memcpy(&vis, cell + 2, 2);
                // FrequencyOccupancyMap::prob / isFree / isOccupied (src/sdm/frequency_occupancy_map.cpp:38-45,119-138)
                const double p = vis == 0 ? 0.25 : ((double)occ) / ((double)vis);
                px = p < 0.25 ? 255 : (p > 0.25 ? 0 : 127);
            } else if (m.kind == kProbabilisticOccupancyMap) {
                float lp; std::memcpy(&lp, cell, 4);
                // ProbabilisticOccupancyMap::isFree / isOccupied (src/sdm/probabilistic_occupancy_map.cpp:131-150), occ_thresh = 0
                px = (double)lp < 0.0 ? 255 : ((double)lp > 0.0 ? 0 : 127);
            } else {
                const int8_t s = (int8_t)cell[0];
                px = s == -1 ? 255 : (s == 1 ? 0 : 127);
            }
            image(u, v) = px;
        }
    }
}

// ---- minimal PNG writer (8-bit grayscale, zlib stream of stored blocks) ----
namespace {
uint32_t crc32(const uint8_t* p, size_t n, uint32_t crc = 0)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
void be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back((x >> 16) & 0xFF); v.push_back((x >> 8) & 0xFF); v.push_back(x & 0xFF); }
void chunk(std::vector<uint8_t>& out, const char* type, const std::vector<uint8_t>& body)
{
    be32(out, (uint32_t)body.size());
    std::vector<uint8_t> td(type, type + 4);
    td.insert(td.end(), body.begin(), body.end());
    out.insert(out.end(), td.begin(), td.end());
    be32(out, crc32(td.data(), td.size()));
}
} // namespace

bool write_png(const Image& im, const std::string& filename)
{
    if (im.width == 0 || im.height == 0 || im.data.size() != (size_t)im.width * im.height) return false;
    std::vector<uint8_t> raw;                                   // filter byte 0 + row
    raw.reserve((size_t)(im.width + 1) * im.height);
    for (uint32_t v = 0; v < im.height; ++v) { raw.push_back(0); raw.insert(raw.end(), &im.data[(size_t)v * im.width], &im.data[(size_t)v * im.width] + im.width); }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t c : raw) { a = (a + c) % 65521u; b = (b + a) % 65521u; }
    for (size_t off = 0; off < raw.size(); off += 65535) {
        const size_t len = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + len == raw.size() ? 1 : 0);
        z.push_back(len & 0xFF); z.push_back(len >> 8); z.push_back(~len & 0xFF); z.push_back((~len >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + (long)off, raw.begin() + (long)(off + len));
    }
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, im.width); be32(ihdr, im.height);
    ihdr.push_back(8); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    std::ofstream f(filename.c_str(), std::ios::out | std::ios::binary | std::ios::trunc);
    if (!f.is_open()) return false;
    f.write((const char*)out.data(), (std::streamsize)out.size());
    f.close();
    return !f.fail();
}

bool export_to_png(const HostMap& m, const std::string& filename)      // src/sdm/export.cpp:97-109
{
    Image im;
    build_image(m, im);
    return write_png(im, filename);
}

} // namespace sdm
} // namespace lama
