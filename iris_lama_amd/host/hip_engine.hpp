// hip_engine.hpp -- run-time binding of the device C-ABI (include/lama_hip.h) for the host classes.
// The loader only ever binds liblama_hip.so and throws if it cannot.  (A host library compiled with -DLAMA_TESTING -- the
// test-suite's own build, never the shipped one -- can be pointed at another implementation of the C-ABI, so that the
// multi-rank host logic is exercised on machines without a GPU.)
#pragma once

#include <memory>
#include <string>

#include "lama_hip.h"

namespace lama {

struct HipEngine {
    void* dl = nullptr;
    std::string origin;
    decltype(&lama_hip_default_cfg) default_cfg = nullptr;
    decltype(&lama_hip_device_count) device_count = nullptr;
    decltype(&lama_hip_ctx_create) ctx_create = nullptr;
    decltype(&lama_hip_ctx_destroy) ctx_destroy = nullptr;
    decltype(&lama_hip_last_error) last_error = nullptr;
    decltype(&lama_hip_pf_init) pf_init = nullptr;
    decltype(&lama_hip_pf_set_poses) pf_set_poses = nullptr;
    decltype(&lama_hip_pf_get_poses) pf_get_poses = nullptr;
    decltype(&lama_hip_pf_scan_match) pf_scan_match = nullptr;
    decltype(&lama_hip_pf_resample) pf_resample = nullptr;
    decltype(&lama_hip_pf_update_maps) pf_update_maps = nullptr;
    decltype(&lama_hip_pf_update_maps_begin) pf_update_maps_begin = nullptr;
    decltype(&lama_hip_sync) sync = nullptr;
    decltype(&lama_hip_pf_map_patches) pf_map_patches = nullptr;
    decltype(&lama_hip_pf_download_map) pf_download_map = nullptr;
    decltype(&lama_hip_match_batch) match_batch = nullptr;
    decltype(&lama_hip_pf_export_particle) pf_export_particle = nullptr;
    decltype(&lama_hip_pf_import_particle) pf_import_particle = nullptr;
    decltype(&lama_hip_get_counters) get_counters = nullptr;
    decltype(&lama_hip_pf_upload_map) pf_upload_map = nullptr;
    decltype(&lama_hip_reset_counters) reset_counters = nullptr;
    decltype(&lama_hip_map_add_obstacles) map_add_obstacles = nullptr;
    decltype(&lama_hip_match_solve) match_solve = nullptr;
    decltype(&lama_hip_eval_batch) eval_batch = nullptr;
    decltype(&lama_hip_pf_patch_ids) pf_patch_ids = nullptr;
    decltype(&lama_hip_pf_delete_patches) pf_delete_patches = nullptr;
    decltype(&lama_hip_map_sample_likelihood) map_sample_likelihood = nullptr;
    decltype(&lama_hip_match_eval) match_eval = nullptr;
    decltype(&lama_hip_match_cell_distances) match_cell_distances = nullptr;
    decltype(&lama_hip_match_solve_with) match_solve_with = nullptr;
    decltype(&lama_hip_pf_export_particles) pf_export_particles = nullptr;
    decltype(&lama_hip_pf_import_particles) pf_import_particles = nullptr;
    decltype(&lama_hip_blob_alloc) blob_alloc = nullptr;
    decltype(&lama_hip_blob_free) blob_free = nullptr;
    decltype(&lama_hip_blob_copy) blob_copy = nullptr;
    ~HipEngine();
};

// Binds liblama_hip.so (next to liblama_host.so unless an explicit path is given).  Throws
// std::runtime_error if the library or any symbol is missing.
std::shared_ptr<HipEngine> loadHipEngine(const std::string& explicit_path = std::string());

// The engine a newly constructed host object binds: liblama_hip.so through loadHipEngine().
std::shared_ptr<HipEngine> defaultEngine();
// The same for a distance map of the given reach: ceil(l2_max / resolution) <= 127 cells binds liblama_hip.so; beyond that, up to the
// 255 cells the reference's uint16_t sqdist can hold, liblama_hip_wide.so (the same sources compiled with -DLAMA_WIDE_DM: a 4-byte
// distance plane, 9-bit obstacle offsets in the queue entries; csrc/lama_dev.h).  More than 255 cells is refused by ctx_create.
std::shared_ptr<HipEngine> defaultEngine(double l2_max, double resolution);
#ifdef LAMA_TESTING
// Test builds of the host library only (-DLAMA_TESTING, tests/cpu_engine/Makefile): an engine that the objects constructed
// afterwards bind instead (nullptr = default loader).  The shipped liblama_host.so is compiled without it.
void setEngineOverride(std::shared_ptr<HipEngine> e);
#endif

} // namespace lama
