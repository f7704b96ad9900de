// lama/sdm/probabilistic_occupancy_map.h -- the reference's include path (include/lama/sdm/probabilistic_occupancy_map.h).  On this path the map classes are host views
// of maps that live in HBM (plus the two host-writable ones lama::Loc2D exposes); they are all defined in lama/sdm_maps.h.
#pragma once
#include "../sdm_maps.h"
