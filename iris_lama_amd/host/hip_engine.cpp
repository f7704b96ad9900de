#include "hip_engine.hpp"

#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <stdexcept>

namespace lama {

HipEngine::~HipEngine()
{
    // the library stays loaded for the life of the process (HIP runtimes do not like dlclose)
}

static std::string siblingPath(const char* name)
{
    Dl_info info;
    if (dladdr((void*)&loadHipEngine, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t k = p.find_last_of('/');
        if (k != std::string::npos) return p.substr(0, k + 1) + name;
    }
    return name;
}

static bool needs_wide(double l2_max, double resolution)
{
    // DynamicDistanceMap::setMaxDistance (src/sdm/dynamic_distance_map.cpp:149-153), as lama_hip_ctx_create computes it
    return resolution > 0.0 && std::ceil(l2_max * (1.0 / resolution)) > 127.0;
}

#ifdef LAMA_TESTING      // the test-suite's host library only (tests/cpu_engine/Makefile): the shipped liblama_host.so has no such hook
static std::shared_ptr<HipEngine> g_override;
void setEngineOverride(std::shared_ptr<HipEngine> e) { g_override = std::move(e); }
std::shared_ptr<HipEngine> defaultEngine() { return g_override ? g_override : loadHipEngine(); }
std::shared_ptr<HipEngine> defaultEngine(double l2_max, double resolution)
{
    if (g_override) return g_override;
    return needs_wide(l2_max, resolution) ? loadHipEngine(siblingPath("liblama_hip_wide.so")) : loadHipEngine();
}
#else
std::shared_ptr<HipEngine> defaultEngine() { return loadHipEngine(); }
std::shared_ptr<HipEngine> defaultEngine(double l2_max, double resolution)
{
    return needs_wide(l2_max, resolution) ? loadHipEngine(siblingPath("liblama_hip_wide.so")) : loadHipEngine();
}
#endif

std::shared_ptr<HipEngine> loadHipEngine(const std::string& explicit_path)
{
    const std::string path = explicit_path.empty() ? siblingPath("liblama_hip.so") : explicit_path;
    void* dl = dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
    if (!dl) throw std::runtime_error(std::string("lama: cannot load the device library ") + path + ": " + dlerror() +
                                      " (build it with hipcc --offload-arch=gfx950; there is no CPU fallback)");
    auto e = std::make_shared<HipEngine>();
    e->dl = dl;
    e->origin = path;
#define BIND(field, sym)                                                                                  \
    e->field = reinterpret_cast<decltype(e->field)>(dlsym(dl, #sym));                                     \
    if (!e->field) throw std::runtime_error(std::string("lama: symbol " #sym " missing in ") + path);
    BIND(default_cfg, lama_hip_default_cfg)
    BIND(device_count, lama_hip_device_count)
    BIND(ctx_create, lama_hip_ctx_create)
    BIND(ctx_destroy, lama_hip_ctx_destroy)
    BIND(last_error, lama_hip_last_error)
    BIND(pf_init, lama_hip_pf_init)
    BIND(pf_set_poses, lama_hip_pf_set_poses)
    BIND(pf_get_poses, lama_hip_pf_get_poses)
    BIND(pf_scan_match, lama_hip_pf_scan_match)
    BIND(pf_resample, lama_hip_pf_resample)
    BIND(pf_update_maps, lama_hip_pf_update_maps)
    BIND(pf_update_maps_begin, lama_hip_pf_update_maps_begin)
    BIND(sync, lama_hip_sync)
    BIND(pf_map_patches, lama_hip_pf_map_patches)
    BIND(pf_download_map, lama_hip_pf_download_map)
    BIND(pf_upload_map, lama_hip_pf_upload_map)
    BIND(match_batch, lama_hip_match_batch)
    BIND(pf_export_particle, lama_hip_pf_export_particle)
    BIND(pf_import_particle, lama_hip_pf_import_particle)
    BIND(get_counters, lama_hip_get_counters)
    BIND(reset_counters, lama_hip_reset_counters)
    BIND(map_add_obstacles, lama_hip_map_add_obstacles)
    BIND(match_solve, lama_hip_match_solve)
    BIND(eval_batch, lama_hip_eval_batch)
    BIND(pf_patch_ids, lama_hip_pf_patch_ids)
    BIND(pf_delete_patches, lama_hip_pf_delete_patches)
    BIND(map_sample_likelihood, lama_hip_map_sample_likelihood)
    BIND(match_eval, lama_hip_match_eval)
    BIND(match_cell_distances, lama_hip_match_cell_distances)
    BIND(match_solve_with, lama_hip_match_solve_with)
    BIND(pf_export_particles, lama_hip_pf_export_particles)
    BIND(pf_import_particles, lama_hip_pf_import_particles)
    BIND(blob_alloc, lama_hip_blob_alloc)
    BIND(blob_free, lama_hip_blob_free)
    BIND(blob_copy, lama_hip_blob_copy)
#undef BIND
    return e;
}

} // namespace lama
