// lama/sdm/export.h -- the reference's include path (include/lama/sdm/export.h:44-50): sdm::export_to_png for an occupancy or a
// distance map.  The encoder works on the map's host snapshot (lama/sdm_io.h, src/sdm/export.cpp:46-110 restated).
#pragma once
#include <string>
#include "../sdm_io.h"
#include "../sdm_maps.h"

namespace lama {
namespace sdm {

inline bool export_to_png(const OccupancyMap& occ, const std::string& filename, double /*zed*/ = 0.0) { return export_to_png(occ.snapshot(), filename); }
inline bool export_to_png(const DynamicDistanceMap& dm, const std::string& filename, double /*zed*/ = 0.0) { return export_to_png(dm.snapshot(), filename); }

} // namespace sdm
} // namespace lama
