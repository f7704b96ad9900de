// dm_builder.hpp -- the FIRST build of lama::Loc2D's distance map, on the host.
//
// Loc2D::Init's caller adds every occupied cell of a static map to the distance map and calls update() once
// (/root/reference/src/loc2d.cpp:61-108 with iris_lama_ros' InitLoc2DFromOccupancyGridMsg: addObstacle per occupied cell, then
// distance_map->update()).  On an EMPTY map that is one serial chain of brushfire pops -- 1.9 M of them for a 100 m x 60 m floor plan --
// whose queue starts with every obstacle cell in it; BASELINE.json's configs[0] calls this configuration "plumbing, no GPU", and one
// chain has nothing to parallelise over: a host core replays it in ~0.1 us per pop, the device's wave pair in
// 0.67 us and, for a queue beyond its LDS stages, the one-lane kernel in 7 us.  So the host facade builds the FIRST map here and
// uploads it (lama_hip_pf_upload_map, the device side of Map::read); every later update() of the same map -- obstacles added to or
// removed from a map that already exists -- runs on the device as before (lama_hip_map_add_obstacles).
//
// This is product code written from scratch, not the test oracle: a dense tile store over the bounding box of what the build can
// touch instead of the reference's patch hash map, the queue std::priority_queue itself (the reference's type, so the pop order among
// equal priorities -- an artefact of libstdc++'s heap -- is the reference's by construction).  Semantics followed statement by
// statement: addObstacle /root/reference/src/sdm/dynamic_distance_map.cpp:212-226, update :160-197 (lower loop), lower :281-330, the
// allocation / mask side effects of the non-const Map::get /root/reference/src/sdm/map.cpp:371-412 and Container::get
// /root/reference/include/lama/sdm/container.h:102-106.
#pragma once
#include <cstdint>
#include <vector>

#include "lama/sdm_io.h"

namespace lama {
namespace detail {

// cells_xy: n (x, y) pairs of map coordinates in addObstacle order.  Fills `out` (kind, resolution, patch_length, max_sqdist must be
// set by the caller; ids / cells / masks are replaced) with the reference's records of every patch the build allocates, and
// returns the number of cells update() processed (its return value, :196).  Returns false -- and leaves `out` untouched -- when
// the dense store would exceed `max_store_cells` (a map whose obstacles span more than that many cells is built on the device).
bool build_distance_map(const uint32_t* cells_xy, size_t n, uint32_t max_sqdist, sdm::HostMap& out, uint32_t& processed,
                        uint64_t max_store_cells = 1ull << 28);

}  // namespace detail
}  // namespace lama
