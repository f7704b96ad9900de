// lama/sdm_io.h -- host-side map formats for maps downloaded from the device:
//   * the reference's `.sdm` binary (Map::write / Map::read, src/sdm/map.cpp:489-575; header include/lama/sdm/map.h:95-103;
//     per patch: 64-bit patch index, the cells, the 16 mask words -- Container::write, src/sdm/container.cpp:143-165),
//     so that a map built on the GPU can be loaded by the reference (and vice versa);
//   * the grayscale images of sdm::export_to_png (src/sdm/export.cpp:46-110) for occupancy and distance maps.
// Pure host code, no device involvement: the input is what PFSlam2D/Slam2D::download*Map (lama_hip_pf_download_map) return.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace lama {
namespace sdm {

enum MapKind : int32_t {
    kDistanceMap = 0,            // DynamicDistanceMap: distance_t, 10 B / cell (+ max_sqdist as map parameter)
    kFrequencyOccupancyMap = 1,  // FrequencyOccupancyMap: frequency {uint16 occupied, uint16 visited}, 4 B / cell
    kSimpleOccupancyMap = 2,     // SimpleOccupancyMap: int8 tri-state, 1 B / cell
    kProbabilisticOccupancyMap = 3   // ProbabilisticOccupancyMap: prob_tag {float log-odds}, 4 B / cell (LidarOdometry2D)
};

struct HostMap {
    MapKind kind = kDistanceMap;
    double resolution = 0.05;
    uint32_t patch_length = 32;
    uint32_t max_sqdist = 100;          // DynamicDistanceMap::max_sqdist_ (written by writeParameters, :200-208)
    std::vector<uint64_t> ids;          // Map::m2p patch indices
    std::vector<uint8_t> cells;         // patch_volume * cell_size bytes per patch
    std::vector<uint64_t> masks;        // 16 words per patch (32 x 32 cells)

    uint32_t cellSize() const { return kind == kDistanceMap ? 10u : (kind == kSimpleOccupancyMap ? 1u : 4u); }
    uint32_t patchVolume() const { return patch_length * patch_length; }
    size_t numPatches() const { return ids.size(); }
};

// Map::write / Map::read.  read() accepts any of the three cell sizes and sets `kind` from it (4-byte cells read as
// kFrequencyOccupancyMap: the file does not say which of the two 4-byte policies wrote it -- set `kind` afterwards if needed).
bool write(const HostMap& map, const std::string& filename);
bool read(HostMap& map, const std::string& filename);

// lama::Image restricted to one channel (include/lama/image.h): rows top to bottom, pixel (u, v) at data[u + v * width]
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> data;
    uint8_t& operator()(uint32_t u, uint32_t v) { return data[u + (size_t)v * width]; }
    uint8_t operator()(uint32_t u, uint32_t v) const { return data[u + (size_t)v * width]; }
};

// build_image of src/sdm/export.cpp:46-95: occupancy -> free 255 / occupied 0 / other visited cells 127 on background 90;
// distance -> distance * 255 / maxDistance on background 127.  Only cells whose mask bit is on are visited.
void build_image(const HostMap& map, Image& image);
bool write_png(const Image& image, const std::string& filename);     // 8-bit grayscale, stored (uncompressed) deflate
bool export_to_png(const HostMap& map, const std::string& filename);

} // namespace sdm
} // namespace lama
