"""ctypes bindings of include/lama_hip.h and include/lama_host.h."""
import ctypes as C
import math
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "lib", "liblama_hip.so")
HIP_LIB_WIDE = os.path.join(_HERE, "lib", "liblama_hip_wide.so")
HOST_LIB = os.path.join(_HERE, "lib", "liblama_host.so")
_PRODUCT_HOST_LIB = HOST_LIB

MAP_DISTANCE, MAP_OCCUPANCY = 0, 1

# the reference's record formats (what lama_hip_pf_download_map returns)
DIST_T = np.dtype([("obstacle", "<i2", (3,)), ("sqdist", "<u2"), ("valid", "u1"), ("queued", "u1")])
FREQ_T = np.dtype([("occupied", "<u2"), ("visited", "<u2")])


class LamaError(RuntimeError):
    pass


class HipCfg(C.Structure):
    _fields_ = [("particles", C.c_uint32), ("resolution", C.c_double), ("patch_size", C.c_uint32),
                ("l2_max", C.c_double), ("meas_sigma", C.c_double), ("max_iter", C.c_uint32),
                ("truncated_ray", C.c_double), ("truncated_range", C.c_double), ("device", C.c_int32),
                ("window_patches", C.c_uint32), ("dm_patch_capacity", C.c_uint32),
                ("occ_patch_capacity", C.c_uint32), ("queue_capacity", C.c_uint32), ("profile", C.c_uint32),
                ("active_capacity", C.c_uint32), ("sequential_raycast", C.c_uint32), ("brushfire_mode", C.c_uint32),
                ("brushfire_waves", C.c_uint32), ("occupancy_policy", C.c_uint32), ("ray_rule", C.c_uint32),
                ("solver_strategy", C.c_uint32)]


class HipCounters(C.Structure):
    _fields_ = [("ms_scan_match", C.c_double), ("launches_scan_match", C.c_uint64),
                ("ms_update_maps", C.c_double), ("launches_update_maps", C.c_uint64),
                ("ms_raycast", C.c_double), ("launches_raycast", C.c_uint64),
                ("ms_brushfire", C.c_double), ("launches_brushfire", C.c_uint64),
                ("ms_resample", C.c_double), ("launches_resample", C.c_uint64),
                ("gn_iterations", C.c_uint64), ("gn_evals", C.c_uint64), ("ray_cells", C.c_uint64),
                ("bf_cells", C.c_uint64), ("dm_patches", C.c_uint64), ("occ_patches", C.c_uint64),
                ("ms_eval_batch", C.c_double), ("launches_eval_batch", C.c_uint64), ("arena_growths", C.c_uint64), ("window_shifts", C.c_uint64), ("wrap_guard_scans", C.c_uint64),
                ("brushfire_mode", C.c_uint32), ("brushfire_waves", C.c_uint32),
                ("sequential_raycast_scans", C.c_uint64), ("parallel_raycast_scans", C.c_uint64),
                ("brushfire_handovers", C.c_uint32), ("replay_handovers", C.c_uint32),
                ("window_patches", C.c_uint32), ("window_growths", C.c_uint32),
                ("bf_longest_chain_sum", C.c_uint64), ("bf_longest_chain_last", C.c_uint64), ("brushfire_early", C.c_uint64), ("brushfire_routed", C.c_uint64),
                ("pool_growths", C.c_uint64), ("resample_clones", C.c_uint64), ("resample_bytes", C.c_uint64),
                ("hbm_bytes_allocated", C.c_uint64), ("hbm_bytes_used", C.c_uint64), ("hbm_bytes_total", C.c_uint64),
                ("peer_access", C.c_uint32), ("struct_bytes", C.c_uint32), ("peer_copy_ms", C.c_double), ("peer_copy_bytes", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HIP_SYMBOLS = [
    "lama_hip_default_cfg", "lama_hip_device_count", "lama_hip_ctx_create", "lama_hip_ctx_destroy",
    "lama_hip_last_error", "lama_hip_pf_init", "lama_hip_pf_set_poses", "lama_hip_pf_get_poses",
    "lama_hip_pf_scan_match", "lama_hip_pf_resample", "lama_hip_pf_update_maps", "lama_hip_pf_map_patches",
    "lama_hip_pf_download_map", "lama_hip_pf_upload_map", "lama_hip_match_batch", "lama_hip_pf_export_particle",
    "lama_hip_pf_import_particle", "lama_hip_get_counters", "lama_hip_get_counters_sized", "lama_hip_counters_bytes", "lama_hip_reset_counters",
    "lama_hip_map_add_obstacles", "lama_hip_match_solve", "lama_hip_eval_batch", "lama_hip_map_sample_likelihood",
    "lama_hip_pgo_create", "lama_hip_pgo_destroy", "lama_hip_pgo_last_error", "lama_hip_pgo_linearize",
    "lama_hip_pf_patch_ids", "lama_hip_pf_delete_patches", "lama_hip_pf_update_maps_begin", "lama_hip_sync", "lama_hip_ctx_device",
    "lama_hip_pf_map_checksums", "lama_hip_match_eval", "lama_hip_match_cell_distances", "lama_hip_match_solve_with",
    "lama_hip_blob_alloc", "lama_hip_blob_free", "lama_hip_blob_copy", "lama_hip_pf_export_particles", "lama_hip_pf_import_particles",
]

_hip = None
_hip_wide = None


def _torch_runtime_first():
    """A process that also uses PyTorch-ROCm must load torch's HIP runtime FIRST: torch bundles its own libamdhip64, and when
    /opt/rocm's copy (which liblama_hip.so links) is already loaded, torch.cuda finds "no HIP GPUs" afterwards (measured on the
    MI355X box; the other order works: liblama_hip.so then binds to the copy torch brought).  Importing torch is enough; done
    before the device library is loaded -- directly or through liblama_host.so -- whenever torch is installed."""
    if "torch" not in sys.modules and not os.environ.get("LAMA_NO_TORCH_PRELOAD"):
        try:
            import importlib.util
            if importlib.util.find_spec("torch") is not None:
                import torch  # noqa: F401
        except Exception:      # torch is optional: a broken install must not make liblama_hip.so unloadable (set LAMA_NO_TORCH_PRELOAD=1
            pass               # to skip the import altogether, e.g. for pure-ctypes users who never touch torch)


def hip_lib(wide=False):
    """Load liblama_hip.so -- or, with wide=True, liblama_hip_wide.so, the build of the same sources for distance maps with an
    l2_max of 128 .. 255 cells (csrc/lama_dev.h).  Raises if it has not been built: there is no fallback path."""
    global _hip, _hip_wide
    if wide:
        if _hip_wide is None:
            if not os.path.exists(HIP_LIB_WIDE):
                raise LamaError(f"{HIP_LIB_WIDE} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. There is no CPU fallback.")
            _torch_runtime_first()
            L = C.CDLL(HIP_LIB_WIDE)
            _bind_hip(L)
            _hip_wide = L
        return _hip_wide
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise LamaError(f"{HIP_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _torch_runtime_first()
        L = C.CDLL(HIP_LIB)
        _bind_hip(L)
        _hip = L
    return _hip


def needs_wide(l2_max, resolution):
    """True when a distance map of this reach needs liblama_hip_wide.so: ceil(l2_max / resolution) > 127 cells (as
    lama_hip_ctx_create and the host classes compute it, DynamicDistanceMap::setMaxDistance)."""
    return resolution > 0.0 and math.ceil(l2_max * (1.0 / resolution)) > 127


def is_device_library(path):
    """Is this engine origin one of the two device libraries (and not the test double of tests/cpu_engine)?"""
    return path.endswith("liblama_hip.so") or path.endswith("liblama_hip_wide.so")


def _lib_of_origin(path):
    if path.endswith("liblama_hip.so"):
        return hip_lib()
    if path.endswith("liblama_hip_wide.so"):
        return hip_lib(wide=True)
    L = C.CDLL(path)                   # test double bound through set_engine_library()
    _bind_hip(L)
    return L


def hip_lib_or_none():
    return _hip


def _bind_hip(L):
    if True:
        vp, i32, u32, u64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64
        L.lama_hip_default_cfg.argtypes = [vp]
        L.lama_hip_default_cfg.restype = None
        L.lama_hip_device_count.argtypes = [vp]
        L.lama_hip_ctx_create.argtypes = [vp, vp]
        L.lama_hip_ctx_destroy.argtypes = [vp]
        L.lama_hip_ctx_destroy.restype = None
        L.lama_hip_last_error.argtypes = [vp]
        L.lama_hip_last_error.restype = C.c_char_p
        L.lama_hip_pf_init.argtypes = [vp, vp, u32, vp, vp, vp]
        L.lama_hip_pf_set_poses.argtypes = [vp, vp]
        L.lama_hip_pf_get_poses.argtypes = [vp, vp]
        L.lama_hip_pf_scan_match.argtypes = [vp, vp, u32, vp, vp, vp, vp, vp]
        L.lama_hip_pf_resample.argtypes = [vp, vp]
        L.lama_hip_pf_update_maps.argtypes = [vp, vp, u32, vp, vp]
        L.lama_hip_pf_update_maps_begin.argtypes = [vp, vp, u32, vp, vp]
        L.lama_hip_sync.argtypes = [vp]
        L.lama_hip_ctx_device.argtypes = [vp]
        L.lama_hip_ctx_device.restype = C.c_int32
        L.lama_hip_pf_map_patches.argtypes = [vp, u32, i32, vp]
        L.lama_hip_pf_download_map.argtypes = [vp, u32, i32, u32, vp, vp, vp, vp]
        L.lama_hip_pf_upload_map.argtypes = [vp, u32, i32, u32, vp, vp, vp]
        L.lama_hip_match_batch.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, vp]
        L.lama_hip_pf_export_particle.argtypes = [vp, u32, vp, u64, vp]
        L.lama_hip_pf_import_particle.argtypes = [vp, u32, vp, u64]
        L.lama_hip_pf_export_particles.argtypes = [vp, u32, vp, vp, vp, vp]
        L.lama_hip_pf_import_particles.argtypes = [vp, u32, vp, vp, vp]
        L.lama_hip_get_counters.argtypes = [vp, vp]
        L.lama_hip_get_counters_sized.argtypes = [vp, vp, u32]
        L.lama_hip_counters_bytes.argtypes = []
        L.lama_hip_counters_bytes.restype = C.c_uint32
        L.lama_hip_reset_counters.argtypes = [vp]
        L.lama_hip_map_add_obstacles.argtypes = [vp, u32, vp, u32]
        L.lama_hip_match_solve.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp, vp, i32]
        L.lama_hip_eval_batch.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, vp, vp]
        L.lama_hip_pf_patch_ids.argtypes = [vp, u32, i32, u32, vp, vp]
        L.lama_hip_pf_delete_patches.argtypes = [vp, u32, vp, u32, vp]
        L.lama_hip_map_sample_likelihood.argtypes = [vp, u32, vp, u32, vp, vp, C.c_double, vp, u32, u32, vp]
        L.lama_hip_match_eval.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp, vp]
        L.lama_hip_match_cell_distances.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp]
        L.lama_hip_match_solve_with.argtypes = [vp, u32, vp, u32, vp, vp, vp, vp, vp, i32, u32]
        has_cks = hasattr(L, "lama_hip_pf_map_checksums")   # device-only diagnostic (absent from the engine test double)
        if has_cks:
            L.lama_hip_pf_map_checksums.argtypes = [vp, i32, vp]
        has_pgo = hasattr(L, "lama_hip_pgo_create")       # the engine test double (tests/cpu_engine) has no pose-graph part
        if has_pgo:
            L.lama_hip_pgo_create.argtypes = [i32, u32, vp, vp, vp, vp, u32, vp]
            L.lama_hip_pgo_destroy.argtypes = [vp]
            L.lama_hip_pgo_destroy.restype = None
            L.lama_hip_pgo_last_error.argtypes = [vp]
            L.lama_hip_pgo_last_error.restype = C.c_char_p
            L.lama_hip_pgo_linearize.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        for s in HIP_SYMBOLS:
            if s.startswith("lama_hip_pgo_") and not has_pgo:
                continue
            if s == "lama_hip_pf_map_checksums" and not has_cks:
                continue
            if s not in ("lama_hip_default_cfg", "lama_hip_ctx_destroy", "lama_hip_last_error", "lama_hip_pgo_destroy",
                         "lama_hip_pgo_last_error"):
                getattr(L, s).restype = i32


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_IDQ = np.array([1.0, 0.0, 0.0, 0.0])
_Z3 = np.zeros(3)


def default_cfg(**kw):
    cfg = HipCfg()
    hip_lib().lama_hip_default_cfg(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def device_count():
    n = C.c_int32(0)
    hip_lib().lama_hip_device_count(C.byref(n))
    return n.value


class HipContext:
    """One device context = one shard of the particle pool (include/lama_hip.h)."""

    def __init__(self, cfg):
        self.L = hip_lib(wide=needs_wide(cfg.l2_max, cfg.resolution))
        self.cfg = cfg
        self.P = cfg.particles
        h = C.c_void_p()
        rc = self.L.lama_hip_ctx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise LamaError(f"lama_hip_ctx_create failed with status {rc} (no usable HIP device or invalid cfg)")
        self.h = h

    def close(self):
        if getattr(self, "h", None) and not getattr(self, "_is_borrowed", False):
            self.L.lama_hip_ctx_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc != 0:
            raise LamaError(f"status {rc}: {self.L.lama_hip_last_error(self.h).decode()}")

    @staticmethod
    def _scan(pts, origin, quat):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        origin = np.ascontiguousarray(_Z3 if origin is None else origin, dtype=np.float64)
        quat = np.ascontiguousarray(_IDQ if quat is None else quat, dtype=np.float64)
        return pts, origin, quat

    def init(self, pts, pose0, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        pose0 = np.ascontiguousarray(pose0, dtype=np.float64)
        self._chk(self.L.lama_hip_pf_init(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(pose0)))

    def set_poses(self, poses):
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        assert poses.shape == (self.P, 4)
        self._chk(self.L.lama_hip_pf_set_poses(self.h, _p(poses)))

    def get_poses(self):
        out = np.zeros((self.P, 4))
        self._chk(self.L.lama_hip_pf_get_poses(self.h, _p(out)))
        return out

    def scan_match(self, pts, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        poses = np.zeros((self.P, 4))
        ll = np.zeros(self.P)
        it = np.zeros(self.P, dtype=np.int32)
        self._chk(self.L.lama_hip_pf_scan_match(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(poses), _p(ll), _p(it)))
        return poses, ll, it

    def resample(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        assert idx.shape == (self.P,)
        self._chk(self.L.lama_hip_pf_resample(self.h, _p(idx)))

    def update_maps(self, pts, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        self._chk(self.L.lama_hip_pf_update_maps(self.h, _p(pts), len(pts), _p(origin), _p(quat)))

    def download_map(self, particle, kind):
        """{reference patch id: (cells[1024] in the reference record format, mask[16])}"""
        n = C.c_uint32(0)
        self._chk(self.L.lama_hip_pf_map_patches(self.h, particle, kind, C.byref(n)))
        n = n.value
        ids = np.zeros(n, dtype=np.uint64)
        dt = DIST_T if kind == MAP_DISTANCE else FREQ_T
        cells = np.zeros((n, 1024), dtype=dt)
        masks = np.zeros((n, 16), dtype=np.uint64)
        got = C.c_uint32(0)
        self._chk(self.L.lama_hip_pf_download_map(self.h, particle, kind, n, _p(ids), _p(cells), _p(masks), C.byref(got)))
        assert got.value == n
        return {int(ids[k]): (cells[k], masks[k]) for k in range(n)}

    def upload_map(self, particle, kind, patches):
        """The inverse of download_map: replace the particle's map of `kind` by {patch id: (cells[1024], mask[16])}."""
        ids = np.array(sorted(patches), dtype=np.uint64)
        dt = DIST_T if kind == MAP_DISTANCE else FREQ_T
        cells = np.zeros((len(ids), 1024), dtype=dt)
        masks = np.zeros((len(ids), 16), dtype=np.uint64)
        for k, i in enumerate(ids):
            cells[k], masks[k] = patches[int(i)]
        self._chk(self.L.lama_hip_pf_upload_map(self.h, particle, kind, len(ids), _p(ids), _p(cells), _p(masks)))

    def match_batch(self, particle, pts, poses, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        out = np.zeros(len(poses))
        self._chk(self.L.lama_hip_match_batch(self.h, particle, _p(pts), len(pts), _p(origin), _p(quat), _p(poses), len(poses), _p(out)))
        return out

    def update_maps_begin(self, pts, origin=None, quat=None):
        """Queues the map update and returns; status and counters are collected by sync() or the next call."""
        pts, origin, quat = self._scan(pts, origin, quat)
        self._chk(self.L.lama_hip_pf_update_maps_begin(self.h, _p(pts), len(pts), _p(origin), _p(quat)))

    def sync(self):
        self._chk(self.L.lama_hip_sync(self.h))

    def device(self):
        """HIP device ordinal of the context."""
        return int(self.L.lama_hip_ctx_device(self.h))

    def map_checksums(self, kind):
        """One 64-bit checksum per particle of its distance / occupancy map (patch set, cells, masks), computed on the device."""
        out = np.zeros(self.P, dtype=np.uint64)
        self._chk(self.L.lama_hip_pf_map_checksums(self.h, kind, _p(out)))
        return out

    def patch_ids(self, particle, kind):
        n = C.c_uint32(0)
        self._chk(self.L.lama_hip_pf_patch_ids(self.h, particle, kind, 0, None, C.byref(n)))
        ids = np.zeros(n.value, dtype=np.uint64)
        self._chk(self.L.lama_hip_pf_patch_ids(self.h, particle, kind, n.value, _p(ids), C.byref(n)))
        return ids

    def delete_patches(self, particle, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint64)
        d = C.c_uint32(0)
        self._chk(self.L.lama_hip_pf_delete_patches(self.h, particle, _p(ids), len(ids), C.byref(d)))
        return d.value

    def eval_batch(self, particle, pts, poses, origin=None, quat=None):
        """-> (squared residual norm, log-likelihood) per pose (Loc2D::globalLocalization's candidate evaluation)"""
        pts, origin, quat = self._scan(pts, origin, quat)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        sq, ll = np.zeros(len(poses)), np.zeros(len(poses))
        self._chk(self.L.lama_hip_eval_batch(self.h, particle, _p(pts), len(pts), _p(origin), _p(quat), _p(poses), len(poses), _p(sq), _p(ll)))
        return sq, ll

    def sample_likelihood(self, particle, pts, yaw, xy, point_step, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        out = np.zeros(len(xy))
        self._chk(self.L.lama_hip_map_sample_likelihood(self.h, particle, _p(pts), len(pts), _p(origin), _p(quat), float(yaw), _p(xy),
                                                        len(xy), int(point_step), _p(out)))
        return out

    def add_obstacles(self, particle, cells_xy):
        cells = np.ascontiguousarray(cells_xy, dtype=np.uint32).reshape(-1, 2)
        self._chk(self.L.lama_hip_map_add_obstacles(self.h, particle, _p(cells), len(cells)))

    def match_solve(self, particle, pts, pose4, origin=None, quat=None, solve=True):
        """-> (pose, JtJ lower [00,10,11,20,21,22], sum r^2 (unweighted), iterations)"""
        pts, origin, quat = self._scan(pts, origin, quat)
        pose = np.array(pose4, dtype=np.float64)
        out = np.zeros(7)
        it = C.c_int32(0)
        self._chk(self.L.lama_hip_match_solve(self.h, particle, _p(pts), len(pts), _p(origin), _p(quat), _p(pose), _p(out),
                                              C.byref(it), 1 if solve else 0))
        return pose, out[:6].copy(), float(out[6]), it.value

    def export_bytes(self, particle):
        n = C.c_uint64(0)
        self._chk(self.L.lama_hip_pf_export_particle(self.h, particle, None, 0, C.byref(n)))
        return n.value

    def export_particle(self, particle, device_ptr, cap):
        n = C.c_uint64(0)
        self._chk(self.L.lama_hip_pf_export_particle(self.h, particle, C.c_void_p(device_ptr), cap, C.byref(n)))
        return n.value

    def import_particle(self, particle, device_ptr, nbytes):
        self._chk(self.L.lama_hip_pf_import_particle(self.h, particle, C.c_void_p(device_ptr), nbytes))

    def export_sizes(self, particles):
        """Blob sizes of a batch of particles (host mirror of the patch counts: no device work)."""
        pa = np.ascontiguousarray(particles, dtype=np.uint32)
        out = np.zeros(len(pa), dtype=np.uint64)
        self._chk(self.L.lama_hip_pf_export_particles(self.h, len(pa), _p(pa), None, None, _p(out)))
        return out

    def export_particles(self, particles, device_ptrs, caps):
        """All outgoing particles of a resample in one launch (device_ptrs: one buffer address per particle)."""
        pa = np.ascontiguousarray(particles, dtype=np.uint32)
        ptrs = np.ascontiguousarray(device_ptrs, dtype=np.uint64)
        cp = np.ascontiguousarray(caps, dtype=np.uint64)
        out = np.zeros(len(pa), dtype=np.uint64)
        self._chk(self.L.lama_hip_pf_export_particles(self.h, len(pa), _p(pa), _p(ptrs), _p(cp), _p(out)))
        return out

    def import_particles(self, particles, device_ptrs, nbytes):
        pa = np.ascontiguousarray(particles, dtype=np.uint32)
        ptrs = np.ascontiguousarray(device_ptrs, dtype=np.uint64)
        nb = np.ascontiguousarray(nbytes, dtype=np.uint64)
        self._chk(self.L.lama_hip_pf_import_particles(self.h, len(pa), _p(pa), _p(ptrs), _p(nb)))

    def counters(self):
        c = HipCounters()
        self._chk(self.L.lama_hip_get_counters(self.h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._chk(self.L.lama_hip_reset_counters(self.h))


# ------------------------------------------------------------------------------------------------ host library
_host = None


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB):
            raise LamaError(f"{HOST_LIB} is missing: run `make -C iris_lama_amd host`")
        _torch_runtime_first()                                # (the host library loads liblama_hip.so when a device object is created)
        L = C.CDLL(HOST_LIB)
        L.lama_corridor_generate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lama_corridor_generate.restype = C.c_int
        _host = L
    return _host


def corridor_log(steps=40, beams=1080):
    """Seeded synthetic corridor log: pts (steps+1, beams, 3), odom xyr, truth xyr."""
    pts = np.zeros((steps + 1, beams, 3))
    odom = np.zeros((steps + 1, 3))
    truth = np.zeros((steps + 1, 3))
    rc = host_lib().lama_corridor_generate(steps, beams, _p(pts), _p(odom), _p(truth))
    if rc != 0:
        raise LamaError("lama_corridor_generate failed")
    return pts, odom, truth


# ------------------------------------------------------------------------------------------------ lama::PFSlam2D
class PFOptions(C.Structure):
    _fields_ = [("particles", C.c_uint32), ("srr", C.c_double), ("str", C.c_double), ("stt", C.c_double),
                ("srt", C.c_double), ("meas_sigma", C.c_double), ("meas_sigma_gain", C.c_double),
                ("trans_thresh", C.c_double), ("rot_thresh", C.c_double), ("l2_max", C.c_double),
                ("truncated_ray", C.c_double), ("truncated_range", C.c_double), ("resolution", C.c_double),
                ("patch_size", C.c_uint32), ("max_iter", C.c_uint32), ("seed", C.c_uint32),
                ("create_summary", C.c_int32), ("gpu_device", C.c_int32), ("shard_rank", C.c_uint32),
                ("shard_world", C.c_uint32), ("profile", C.c_int32), ("brushfire_mode", C.c_uint32),
                ("window_patches", C.c_uint32), ("dm_patch_capacity", C.c_uint32), ("occ_patch_capacity", C.c_uint32),
                ("queue_capacity", C.c_uint32), ("gpus", C.c_int32)]


HOST_SYMBOLS = [
    "lama_corridor_generate", "lama_pf_default_options", "lama_pf_engine_origin",
    "lama_pf_create", "lama_pf_destroy", "lama_pf_last_error", "lama_pf_set_prior", "lama_pf_update",
    "lama_pf_update_begin", "lama_pf_local_range", "lama_pf_local_loglik", "lama_pf_plan_resample",
    "lama_pf_apply_resample", "lama_pf_update_maps", "lama_pf_device_context", "lama_pf_get_poses",
    "lama_pf_set_pose", "lama_pf_get_weights", "lama_pf_set_weights", "lama_pf_neff", "lama_pf_best",
    "lama_pf_best_pose_xyr", "lama_pf_num_resamples", "lama_pf_exchange_times", "lama_pf_shard_context", "lama_pf_memory_usage", "lama_pf_summary",
    "lama_pf_last_times", "lama_pf_draw_from_motion", "lama_pf_normalize", "lama_pf_resample_indices",
    "lama_pose_minus", "lama_pose_from_xyr",
    "lama_slam_default_options", "lama_slam_create", "lama_slam_destroy", "lama_slam_last_error", "lama_slam_set_pose",
    "lama_slam_get_pose", "lama_slam_update", "lama_slam_enough_motion", "lama_slam_processed_cells",
    "lama_slam_iterations", "lama_slam_device_context", "lama_slam_engine_origin", "lama_slam_deleted_patches", "lama_loc_create3",
    "lama_slam_view_bounds", "lama_slam_view_cells", "lama_slam_view_occupancy", "lama_slam_view_distance_cells", "lama_slam_view_distance_points",
    "lama_slam_match_eval", "lama_slam_match_solve",
    "lama_loc_create", "lama_loc_destroy", "lama_loc_last_error", "lama_loc_engine_origin", "lama_loc_set_obstacles_world",
    "lama_loc_write_distance_map", "lama_loc_read_distance_map", "lama_loc_device_context",
    "lama_loc_set_pose", "lama_loc_get_pose", "lama_loc_update", "lama_loc_covar", "lama_loc_rmse", "lama_loc_iterations",
    "lama_loc_create2", "lama_loc_occ_set_cells", "lama_loc_occ_bounds", "lama_loc_trigger_global_localization",
    "lama_loc_global_localization_active", "lama_loc_gloc_candidates", "lama_loc_sampling_likelihoods",
    "lama_random_set_seed", "lama_random_uniform",
    "lama_sdm_write", "lama_sdm_read", "lama_sdm_image", "lama_sdm_export_png", "lama_dm_build", "lama_dm_build_fetch",
    "lama_lo_create", "lama_lo_destroy", "lama_lo_last_error", "lama_lo_engine_origin", "lama_lo_update", "lama_lo_get_odom",
    "lama_lo_iterations", "lama_lo_deleted_patches", "lama_lo_device_context",
]


def _bind_host(L):
    vp, i32, u32, d = C.c_void_p, C.c_int32, C.c_uint32, C.c_double
    sig = {
        "lama_pf_default_options": (None, [vp]),
        "lama_pf_engine_origin": (C.c_char_p, [vp]), "lama_pf_create": (vp, [vp, vp, i32]),
        "lama_pf_destroy": (None, [vp]), "lama_pf_last_error": (C.c_char_p, [vp]),
        "lama_pf_set_prior": (None, [vp, d, d, d]), "lama_pf_update": (i32, [vp, vp, u32, vp, vp, vp, d]),
        "lama_pf_update_begin": (i32, [vp, vp, u32, vp, vp, vp, d]), "lama_pf_local_range": (i32, [vp, vp, vp]),
        "lama_pf_local_loglik": (i32, [vp, vp]), "lama_pf_plan_resample": (i32, [vp, vp, vp]),
        "lama_pf_apply_resample": (i32, [vp, vp]), "lama_pf_update_maps": (i32, [vp]),
        "lama_pf_device_context": (vp, [vp]), "lama_pf_get_poses": (i32, [vp, vp]),
        "lama_pf_set_pose": (i32, [vp, u32, vp]), "lama_pf_get_weights": (i32, [vp, vp, vp, vp]),
        "lama_pf_set_weights": (i32, [vp, vp, vp]), "lama_pf_neff": (d, [vp]), "lama_pf_best": (i32, [vp]),
        "lama_pf_best_pose_xyr": (i32, [vp, vp]), "lama_pf_num_resamples": (u32, [vp]),
        "lama_pf_exchange_times": (i32, [vp, vp]), "lama_pf_shard_context": (vp, [vp, u32]),
        "lama_pf_memory_usage": (C.c_uint64, [vp]), "lama_pf_summary": (i32, [vp, vp, i32]),
        "lama_pf_last_times": (i32, [vp, vp]), "lama_pf_draw_from_motion": (i32, [vp, vp, vp]),
        "lama_pf_normalize": (d, [vp]), "lama_pf_resample_indices": (i32, [vp, d, vp]),
        "lama_pose_minus": (None, [vp, vp, vp]), "lama_pose_from_xyr": (None, [d, d, d, vp]),
        "lama_slam_default_options": (None, [vp]), "lama_slam_create": (vp, [vp, vp, i32]), "lama_slam_destroy": (None, [vp]),
        "lama_slam_last_error": (C.c_char_p, [vp]), "lama_slam_set_pose": (None, [vp, d, d, d]),
        "lama_slam_get_pose": (i32, [vp, vp]), "lama_slam_update": (i32, [vp, vp, u32, vp, vp, vp, d]),
        "lama_slam_enough_motion": (i32, [vp, vp]), "lama_slam_processed_cells": (u32, [vp]),
        "lama_slam_iterations": (u32, [vp]), "lama_slam_device_context": (vp, [vp]), "lama_slam_deleted_patches": (u32, [vp]),
        "lama_slam_engine_origin": (C.c_char_p, [vp]),
        "lama_slam_view_bounds": (i32, [vp, i32, vp, vp, vp, vp]), "lama_slam_view_cells": (C.c_int64, [vp, i32, vp, C.c_uint64]),
        "lama_slam_view_occupancy": (i32, [vp, C.c_uint64, vp, vp, vp, vp, vp]),
        "lama_slam_view_distance_cells": (i32, [vp, C.c_uint64, vp, vp]), "lama_slam_view_distance_points": (i32, [vp, C.c_uint64, vp, vp]),
        "lama_slam_match_eval": (i32, [vp, vp, u32, vp, vp, vp, vp, vp, vp]),
        "lama_slam_match_solve": (i32, [vp, vp, u32, vp, vp, vp, C.c_char_p, C.c_char_p, d, u32, vp, vp]),
        "lama_loc_create": (vp, [d, d, d, d, u32, i32, vp, i32]), "lama_loc_destroy": (None, [vp]),
        "lama_loc_last_error": (C.c_char_p, [vp]), "lama_loc_engine_origin": (C.c_char_p, [vp]),
        "lama_loc_set_obstacles_world": (i32, [vp, vp, u32]), "lama_loc_write_distance_map": (i32, [vp, C.c_char_p]), "lama_loc_read_distance_map": (i32, [vp, C.c_char_p]), "lama_loc_device_context": (vp, [vp]), "lama_loc_set_pose": (None, [vp, d, d, d]),
        "lama_loc_get_pose": (i32, [vp, vp]), "lama_loc_update": (i32, [vp, vp, u32, vp, vp, vp, d, i32]),
        "lama_loc_covar": (i32, [vp, vp]), "lama_loc_rmse": (d, [vp]), "lama_loc_iterations": (u32, [vp]),
        "lama_loc_create2": (vp, [d, d, d, d, u32, u32, u32, d, d, i32, vp, i32]),
        "lama_loc_create3": (vp, [d, d, d, d, u32, u32, u32, d, d, C.c_char_p, i32, vp, i32]),
        "lama_loc_occ_set_cells": (i32, [vp, vp, u32, i32]), "lama_loc_occ_bounds": (i32, [vp, vp]),
        "lama_loc_trigger_global_localization": (None, [vp]), "lama_loc_global_localization_active": (i32, [vp]),
        "lama_loc_gloc_candidates": (u32, [vp, vp, vp, u32]), "lama_loc_sampling_likelihoods": (u32, [vp, vp, u32]),
        "lama_random_set_seed": (None, [u32]), "lama_random_uniform": (d, []),
        "lama_lo_create": (vp, [d, u32, i32, vp, i32]), "lama_lo_destroy": (None, [vp]), "lama_lo_last_error": (C.c_char_p, [vp]),
        "lama_lo_engine_origin": (C.c_char_p, [vp]), "lama_lo_update": (i32, [vp, vp, u32, vp, vp, d]),
        "lama_lo_get_odom": (i32, [vp, vp]), "lama_lo_iterations": (u32, [vp]), "lama_lo_deleted_patches": (u32, [vp]),
        "lama_lo_device_context": (vp, [vp]),
        "lama_sdm_write": (i32, [C.c_char_p, i32, d, u32, u32, vp, vp, vp]),
        "lama_sdm_read": (i32, [C.c_char_p, vp, vp, vp, u32, vp, vp, vp, vp]),
        "lama_sdm_image": (i32, [i32, d, u32, u32, vp, vp, vp, vp, vp, vp, C.c_uint64]),
        "lama_sdm_export_png": (i32, [i32, d, u32, u32, vp, vp, vp, C.c_char_p]),
        "lama_dm_build": (C.c_int64, [vp, C.c_uint64, u32, vp]), "lama_dm_build_fetch": (i32, [vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args


_host_bound = False


def _hostlib():
    global _host_bound
    L = host_lib()
    if not _host_bound:
        _bind_host(L)
        _host_bound = True
    return L


def use_host_library(path=None):
    """Load the host C-ABI (include/lama_host.h) from `path` instead of iris_lama_amd/lib/liblama_host.so (None = back to the
    product library).  The product never calls this; the test-suite loads its own -DLAMA_TESTING build of the same sources
    (tests/_testhost.py), the only build that can bind anything but liblama_hip.so.  Objects created before the switch keep
    working only as long as no call is made through them afterwards: switch between, not during, uses."""
    global _host, _host_bound, HOST_LIB
    HOST_LIB = path or _PRODUCT_HOST_LIB
    _host = None
    _host_bound = False


def pf_options(**kw):
    o = PFOptions()
    _hostlib().lama_pf_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def pose_from_xyr(x, y, yaw):
    out = np.zeros(4)
    _hostlib().lama_pose_from_xyr(float(x), float(y), float(yaw), _p(out))
    return out


class PFSlam2D:
    """ctypes view of the host-side lama::PFSlam2D (include/lama/pf_slam2d.h)."""

    def __init__(self, opts):
        self.L = _hostlib()
        self.opts = opts
        self.P = opts.particles
        err = C.create_string_buffer(512)
        h = self.L.lama_pf_create(C.byref(opts), err, 512)
        if not h:
            raise LamaError(err.value.decode())
        self.h = C.c_void_p(h)
        lo, hi = C.c_uint32(0), C.c_uint32(0)
        self.L.lama_pf_local_range(self.h, C.byref(lo), C.byref(hi))
        self.lo, self.hi = lo.value, hi.value

    def close(self):
        if getattr(self, "h", None):
            self.L.lama_pf_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise LamaError(self.L.lama_pf_last_error(self.h).decode())
        return rc

    def engine_origin(self):
        return self.L.lama_pf_engine_origin(self.h).decode()

    def set_prior(self, x, y, yaw):
        self.L.lama_pf_set_prior(self.h, float(x), float(y), float(yaw))

    @staticmethod
    def _scan(pts, origin, quat):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        origin = np.ascontiguousarray(_Z3 if origin is None else origin, dtype=np.float64)
        quat = np.ascontiguousarray(_IDQ if quat is None else quat, dtype=np.float64)
        return pts, origin, quat

    def update(self, pts, odom_xyr, ts=0.0, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        od = np.ascontiguousarray(odom_xyr, dtype=np.float64)
        return bool(self._chk(self.L.lama_pf_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(od), float(ts))))

    def update_begin(self, pts, odom_xyr, ts=0.0, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        od = np.ascontiguousarray(odom_xyr, dtype=np.float64)
        return self._chk(self.L.lama_pf_update_begin(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(od), float(ts)))

    def local_loglik(self):
        out = np.zeros(self.hi - self.lo)
        self.L.lama_pf_local_loglik(self.h, _p(out))
        return out

    def plan_resample(self, all_loglik):
        all_loglik = np.ascontiguousarray(all_loglik, dtype=np.float64)
        assert all_loglik.shape == (self.P,)
        idx = np.zeros(self.P, dtype=np.int32)
        r = self._chk(self.L.lama_pf_plan_resample(self.h, _p(all_loglik), _p(idx)))
        return idx if r else None

    def apply_resample(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        self._chk(self.L.lama_pf_apply_resample(self.h, _p(idx)))

    def update_maps(self):
        self._chk(self.L.lama_pf_update_maps(self.h))

    def device_context(self):
        return self.L.lama_pf_device_context(self.h)

    def poses(self):
        out = np.zeros((self.P, 4))
        self.L.lama_pf_get_poses(self.h, _p(out))
        return out

    def set_pose(self, i, pose4):
        self._chk(self.L.lama_pf_set_pose(self.h, i, _p(np.ascontiguousarray(pose4, dtype=np.float64))))

    def weights(self):
        w, nw, ws = np.zeros(self.P), np.zeros(self.P), np.zeros(self.P)
        self.L.lama_pf_get_weights(self.h, _p(w), _p(nw), _p(ws))
        return w, nw, ws

    def set_weights(self, w=None, ws=None):
        self.L.lama_pf_set_weights(self.h, _p(np.ascontiguousarray(w)) if w is not None else None,
                                   _p(np.ascontiguousarray(ws)) if ws is not None else None)

    def neff(self):
        return self.L.lama_pf_neff(self.h)

    def best(self):
        return self.L.lama_pf_best(self.h)

    def best_pose_xyr(self):
        out = np.zeros(3)
        self.L.lama_pf_best_pose_xyr(self.h, _p(out))
        return out

    def num_resamples(self):
        return self.L.lama_pf_num_resamples(self.h)

    def exchange_times(self):
        """Options::gpus > 1: what the last update spent exchanging (seconds) and shipped between shards."""
        t = np.zeros(8)
        n = self.L.lama_pf_exchange_times(self.h, _p(t))
        return dict(shards=n, gather_s=t[0], ship_s=t[1], import_s=t[2], shipped_particles=int(t[3]), shipped_bytes=int(t[4]),
                    local_copies_s=t[5], phase_begin_s=t[6], phase_maps_s=t[7])

    def shard_context(self, r):
        """HipContext view of shard r's device context (Options::gpus > 1), None when out of range."""
        h = self.L.lama_pf_shard_context(self.h, r)
        if not h:
            return None
        ctx = self.hip_context()
        ctx.h = C.c_void_p(h)
        G = self.opts.gpus
        ctx.P = ((r + 1) * self.P + G - 1) // G - (r * self.P + G - 1) // G
        return ctx

    def memory_usage(self):
        return self.L.lama_pf_memory_usage(self.h)

    def summary(self):
        n = self.L.lama_pf_summary(self.h, None, 0)
        if n <= 0:
            return ""
        buf = C.create_string_buffer(n)
        self.L.lama_pf_summary(self.h, buf, n)
        return buf.value.decode()

    def last_times(self):
        t = np.zeros(5)
        self.L.lama_pf_last_times(self.h, _p(t))
        return dict(total=t[0], solving=t[1], normalizing=t[2], resampling=t[3], mapping=t[4])

    def draw_from_motion(self, delta4, pose4):
        pose = np.array(pose4, dtype=np.float64)
        self._chk(self.L.lama_pf_draw_from_motion(self.h, _p(np.ascontiguousarray(delta4, dtype=np.float64)), _p(pose)))
        return pose

    def normalize(self):
        return self.L.lama_pf_normalize(self.h)

    def resample_indices(self, u):
        out = np.zeros(self.P, dtype=np.int32)
        self.L.lama_pf_resample_indices(self.h, float(u), _p(out))
        return out

    def hip_context(self):
        """Borrowed HipContext view of the local shard's device context (export/import, counters, maps)."""
        ctx = HipContext.__new__(HipContext)
        ctx.L = _lib_of_origin(self.engine_origin())
        ctx.cfg = None
        ctx.P = self.hi - self.lo
        ctx.h = C.c_void_p(self.device_context())
        ctx._is_borrowed = True
        return ctx


class SlamOptions(C.Structure):
    _fields_ = [("trans_thresh", C.c_double), ("rot_thresh", C.c_double), ("l2_max", C.c_double),
                ("truncated_ray", C.c_double), ("truncated_range", C.c_double), ("resolution", C.c_double),
                ("patch_size", C.c_uint32), ("max_iter", C.c_uint32), ("gpu_device", C.c_int32), ("transient_map", C.c_int32),
                ("lm", C.c_int32)]


class Slam2D:
    """ctypes view of the host-side lama::Slam2D (include/lama/slam2d.h): online SLAM, one pose + one map pair."""

    def __init__(self, **kw):
        self.L = _hostlib()
        o = SlamOptions()
        self.L.lama_slam_default_options(C.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        err = C.create_string_buffer(512)
        h = self.L.lama_slam_create(C.byref(o), err, 512)
        if not h:
            raise LamaError(err.value.decode())
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.lama_slam_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def deleted_patches(self):
        return self.L.lama_slam_deleted_patches(self.h)

    def engine_origin(self):
        return self.L.lama_slam_engine_origin(self.h).decode()

    def set_pose(self, x, y, yaw):
        self.L.lama_slam_set_pose(self.h, float(x), float(y), float(yaw))

    def pose(self):
        out = np.zeros(4)
        self.L.lama_slam_get_pose(self.h, _p(out))
        return out

    def update(self, pts, odom_xyr, ts=0.0, origin=None, quat=None):
        pts, origin, quat = PFSlam2D._scan(pts, origin, quat)
        od = np.ascontiguousarray(odom_xyr, dtype=np.float64)
        rc = self.L.lama_slam_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(od), float(ts))
        if rc < 0:
            raise LamaError(self.L.lama_slam_last_error(self.h).decode())
        return bool(rc)

    def enough_motion(self, odom_xyr):
        return bool(self.L.lama_slam_enough_motion(self.h, _p(np.ascontiguousarray(odom_xyr, dtype=np.float64))))

    def processed_cells(self):
        return self.L.lama_slam_processed_cells(self.h)

    def iterations(self):
        return self.L.lama_slam_iterations(self.h)

    # ---- Slam2D::getOccupancyMap() / getDistanceMap(): host snapshots with the reference's const map API (lama/sdm_maps.h)
    def view_bounds(self, which):
        mn, mx = np.zeros(3, dtype=np.uint32), np.zeros(3, dtype=np.uint32)
        wmn, wmx = np.zeros(3), np.zeros(3)
        if self.L.lama_slam_view_bounds(self.h, int(which), _p(mn), _p(mx), _p(wmn), _p(wmx)) < 0:
            return None
        return mn, mx, wmn, wmx

    def view_cells(self, which):
        n = self.L.lama_slam_view_cells(self.h, int(which), None, 0)
        if n < 0:
            return None
        out = np.zeros((n, 2), dtype=np.uint32)
        self.L.lama_slam_view_cells(self.h, int(which), _p(out), n)
        return out

    def view_occupancy(self, cells_xy):
        c = np.ascontiguousarray(cells_xy, dtype=np.uint32)
        n = len(c)
        fr, oc, un = (np.zeros(n, dtype=np.uint8) for _ in range(3))
        pr = np.zeros(n)
        if self.L.lama_slam_view_occupancy(self.h, n, _p(c), _p(fr), _p(oc), _p(un), _p(pr)) < 0:
            return None
        return fr.astype(bool), oc.astype(bool), un.astype(bool), pr

    def view_distance_cells(self, cells_xy):
        c = np.ascontiguousarray(cells_xy, dtype=np.uint32)
        out = np.zeros(len(c))
        return out if self.L.lama_slam_view_distance_cells(self.h, len(c), _p(c), _p(out)) >= 0 else None

    def view_distance_points(self, pts_xy):
        q = np.ascontiguousarray(pts_xy, dtype=np.float64)
        out = np.zeros((len(q), 3))
        return out if self.L.lama_slam_view_distance_points(self.h, len(q), _p(q), _p(out)) >= 0 else None

    # ---- lama::MatchSurface2D / lama::Solve on Slam2D::getDistanceMap() (include/lama/match_surface_2d.h, nlls/solver.h)
    def match_eval(self, pts, pose4, origin=None, quat=None, jac=True):
        pts, origin, quat = PFSlam2D._scan(pts, origin, quat)
        n = len(pts)
        r, J, rmse = np.zeros(n), (np.zeros((3, n)) if jac else None), C.c_double(0)
        rc = self.L.lama_slam_match_eval(self.h, _p(pts), n, _p(origin), _p(quat), _p(np.ascontiguousarray(pose4, dtype=np.float64)),
                                         _p(r), _p(J), C.byref(rmse))
        if rc < 0:
            raise LamaError(self.L.lama_slam_last_error(self.h).decode())
        return r, (J.T.copy() if jac else None), rmse.value

    def match_solve(self, pts, pose4, strategy="gn", weight="cauchy", weight_param=0.15, max_iterations=100, origin=None, quat=None):
        pts, origin, quat = PFSlam2D._scan(pts, origin, quat)
        pose = np.array(pose4, dtype=np.float64)
        cov, it = np.zeros(9), C.c_uint32(0)
        rc = self.L.lama_slam_match_solve(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(pose), strategy.encode(), weight.encode(),
                                          float(weight_param), int(max_iterations), _p(cov), C.byref(it))
        if rc < 0:
            raise LamaError(self.L.lama_slam_last_error(self.h).decode())
        return pose, cov.reshape(3, 3), it.value

    def hip_context(self):
        ctx = HipContext.__new__(HipContext)
        ctx.L = _lib_of_origin(self.engine_origin())
        ctx.cfg = None
        ctx.P = 1
        ctx.h = C.c_void_p(self.L.lama_slam_device_context(self.h))
        ctx._is_borrowed = True
        return ctx


class Loc2D:
    """ctypes view of the host-side lama::Loc2D (include/lama/loc2d.h): localisation on a fixed distance map."""

    def __init__(self, trans_thresh=0.5, rot_thresh=0.5, l2_max=1.0, resolution=0.05, max_iter=100, gpu_device=0,
                 gloc_particles=3000, gloc_iters=10, gloc_thresh=0.15, cov_blend=0.0, strategy="gn"):
        self.L = _hostlib()
        err = C.create_string_buffer(512)
        h = self.L.lama_loc_create3(trans_thresh, rot_thresh, l2_max, resolution, max_iter, gloc_particles, gloc_iters,
                                    gloc_thresh, cov_blend, strategy.encode(), gpu_device, err, 512)
        if not h:
            raise LamaError(err.value.decode())
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.lama_loc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _chk(self, rc):
        if rc < 0:
            raise LamaError(self.L.lama_loc_last_error(self.h).decode())
        return rc

    def engine_origin(self):
        return self.L.lama_loc_engine_origin(self.h).decode()

    def set_obstacles_world(self, xy):
        xy = np.ascontiguousarray(xy, dtype=np.float64).reshape(-1, 2)
        self._chk(self.L.lama_loc_set_obstacles_world(self.h, _p(xy), len(xy)))

    def write_distance_map(self, filename):
        self._chk(self.L.lama_loc_write_distance_map(self.h, str(filename).encode()))

    def hip_context(self):
        """Borrowed HipContext view of the device context (map downloads in tests)."""
        ctx = HipContext.__new__(HipContext)
        ctx.L = _lib_of_origin(self.engine_origin())
        ctx.cfg = None
        ctx.P = 1
        ctx.h = C.c_void_p(self.L.lama_loc_device_context(self.h))
        ctx._is_borrowed = True
        return ctx

    def read_distance_map(self, filename):
        """distance_map->read(file): the file's patches replace the device map (no addObstacle / update() replay)."""
        self._chk(self.L.lama_loc_read_distance_map(self.h, str(filename).encode()))

    def set_pose(self, x, y, yaw):
        self.L.lama_loc_set_pose(self.h, float(x), float(y), float(yaw))

    def pose(self):
        out = np.zeros(4)
        self.L.lama_loc_get_pose(self.h, _p(out))
        return out

    def update(self, pts, odom_xyr, ts=0.0, force=False, origin=None, quat=None):
        pts, origin, quat = PFSlam2D._scan(pts, origin, quat)
        od = np.ascontiguousarray(odom_xyr, dtype=np.float64)
        return bool(self._chk(self.L.lama_loc_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(od), float(ts), 1 if force else 0)))

    def covar(self):
        out = np.zeros(9)
        self.L.lama_loc_covar(self.h, _p(out))
        return out.reshape(3, 3)

    def rmse(self):
        return self.L.lama_loc_rmse(self.h)

    def iterations(self):
        return self.L.lama_loc_iterations(self.h)

    # ---- global localisation / sampling covariance (src/loc2d.cpp:194-286)
    def occ_set_cells(self, cells_xy, state):
        """occupancy_map->setFree (-1) / setUnknown (0) / setOccupied (1) on map cells."""
        cells = np.ascontiguousarray(cells_xy, dtype=np.uint32).reshape(-1, 2)
        self._chk(self.L.lama_loc_occ_set_cells(self.h, _p(cells), len(cells), int(state)))

    def occ_bounds(self):
        out = np.zeros(6)
        self.L.lama_loc_occ_bounds(self.h, _p(out))
        return out[:3].copy(), out[3:].copy()

    def trigger_global_localization(self):
        self.L.lama_loc_trigger_global_localization(self.h)

    def global_localization_active(self):
        return bool(self.L.lama_loc_global_localization_active(self.h))

    def gloc_candidates(self):
        n = self.L.lama_loc_gloc_candidates(self.h, None, None, 0)
        poses, err = np.zeros((n, 4)), np.zeros(n)
        if n:
            self.L.lama_loc_gloc_candidates(self.h, _p(poses), _p(err), n)
        return poses, err

    def sampling_likelihoods(self):
        n = self.L.lama_loc_sampling_likelihoods(self.h, None, 0)
        out = np.zeros(n)
        if n:
            self.L.lama_loc_sampling_likelihoods(self.h, _p(out), n)
        return out


def random_set_seed(seed):
    """lama::random::setSeed (process-wide generator used by Loc2D::globalLocalization)."""
    _hostlib().lama_random_set_seed(int(seed))


# ---------------------------------------------------------------------------------------------------------------
# lama::sdm map formats (include/lama/sdm_io.h) on downloaded maps: {patch id: (cells, mask)} as HipContext.download_map returns
# ---------------------------------------------------------------------------------------------------------------
SDM_CELL_BYTES = {MAP_DISTANCE: 10, MAP_OCCUPANCY: 4, 2: 1, 3: 4}      # 3 = ProbabilisticOccupancyMap (float log-odds)


def _sdm_arrays(patches, kind):
    ids = np.array(sorted(patches), dtype=np.uint64)
    cb = SDM_CELL_BYTES[kind] * 1024
    cells = np.zeros((len(ids), cb), dtype=np.uint8)
    masks = np.zeros((len(ids), 16), dtype=np.uint64)
    for k, i in enumerate(ids):
        c, m = patches[int(i)]
        cells[k] = np.ascontiguousarray(c).view(np.uint8).reshape(-1)
        masks[k] = m
    return ids, cells, masks


def sdm_write(filename, patches, kind, resolution=0.05, max_sqdist=100):
    ids, cells, masks = _sdm_arrays(patches, kind)
    if _hostlib().lama_sdm_write(filename.encode(), kind, resolution, max_sqdist, len(ids), _p(ids), _p(cells), _p(masks)) != 0:
        raise LamaError(f"cannot write {filename}")


def sdm_read(filename):
    """-> (kind, resolution, max_sqdist, {patch id: (cells bytes, mask)})"""
    L = _hostlib()
    kind, n, msq, res = C.c_int32(0), C.c_uint32(0), C.c_uint32(0), C.c_double(0)
    if L.lama_sdm_read(filename.encode(), C.byref(kind), C.byref(res), C.byref(msq), 0, None, None, None, C.byref(n)) != 0:
        raise LamaError(f"cannot read {filename}")
    cb = SDM_CELL_BYTES[kind.value] * 1024
    ids = np.zeros(n.value, dtype=np.uint64)
    cells = np.zeros((n.value, cb), dtype=np.uint8)
    masks = np.zeros((n.value, 16), dtype=np.uint64)
    L.lama_sdm_read(filename.encode(), None, None, None, n.value, _p(ids), _p(cells), _p(masks), None)
    return kind.value, res.value, msq.value, {int(ids[k]): (cells[k], masks[k]) for k in range(n.value)}


def dm_build(cells_xy, max_sqdist):
    """The first build of Loc2D's distance map as the host facade does it (iris_lama_amd/host/dm_builder.hpp): addObstacle for the
    cells in order on an empty map + one update().  -> (cells processed, {patch id: (distance_t[1024] structured, mask[16])}),
    or None when the host does not build it (the facade then uses the device chain)."""
    L = _hostlib()
    cells_xy = np.ascontiguousarray(cells_xy, dtype=np.uint32)
    done = C.c_uint32(0)
    n = L.lama_dm_build(_p(cells_xy), len(cells_xy), int(max_sqdist), C.byref(done))
    if n < 0:
        return None
    ids = np.zeros(n, dtype=np.uint64)
    cells = np.zeros((n, 10240), dtype=np.uint8)
    masks = np.zeros((n, 16), dtype=np.uint64)
    if n:
        L.lama_dm_build_fetch(_p(ids), _p(cells), _p(masks))
    dist_t = np.dtype([("obstacle", "<i2", (3,)), ("sqdist", "<u2"), ("valid", "u1"), ("queued", "u1")])
    return done.value, {int(ids[k]): (cells[k].view(dist_t), masks[k]) for k in range(n)}


def sdm_image(patches, kind, resolution=0.05, max_sqdist=100):
    ids, cells, masks = _sdm_arrays(patches, kind)
    w, h = C.c_uint32(0), C.c_uint32(0)
    L = _hostlib()
    L.lama_sdm_image(kind, resolution, max_sqdist, len(ids), _p(ids), _p(cells), _p(masks), C.byref(w), C.byref(h), None, 0)
    out = np.zeros((h.value, w.value), dtype=np.uint8)
    L.lama_sdm_image(kind, resolution, max_sqdist, len(ids), _p(ids), _p(cells), _p(masks), C.byref(w), C.byref(h), _p(out), out.size)
    return out


def sdm_export_png(filename, patches, kind, resolution=0.05, max_sqdist=100):
    ids, cells, masks = _sdm_arrays(patches, kind)
    if _hostlib().lama_sdm_export_png(kind, resolution, max_sqdist, len(ids), _p(ids), _p(cells), _p(masks), filename.encode()) != 0:
        raise LamaError(f"cannot write {filename}")


# ---------------------------------------------------------------------------------------------------------------
# SE2 pose-graph linearisation on the device (include/lama_hip.h, lama_hip_pgo_*)
# ---------------------------------------------------------------------------------------------------------------
class PoseGraph:
    """Factors: fi, fj (fj = -1: prior on fi), meas [F,4] = {c, s, tx, ty}, sqrt_info [F,3] (DiagonalLoss: 1 / sigma)."""

    def __init__(self, num_poses, fi, fj, meas, sqrt_info, device=0):
        self.L = hip_lib()
        self.N = int(num_poses)
        self.fi = np.ascontiguousarray(fi, dtype=np.int32)
        self.fj = np.ascontiguousarray(fj, dtype=np.int32)
        self.meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 4)
        self.sqrt_info = np.ascontiguousarray(sqrt_info, dtype=np.float64).reshape(-1, 3)
        self.F = len(self.fi)
        h = C.c_void_p()
        rc = self.L.lama_hip_pgo_create(device, self.N, _p(self.fi), _p(self.fj), _p(self.meas), _p(self.sqrt_info), self.F, C.byref(h))
        if rc != 0 or not h:
            raise LamaError(f"lama_hip_pgo_create failed (status {rc}): no usable MI355X / HIP device or invalid graph; there is no CPU fallback")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.lama_hip_pgo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def linearize(self, poses, want_err=True, want_off=True):
        """-> dict(err [F,3], Hdiag [N,3,3], Hoff [F,3,3], b [N,3], chi2, kernel_ms)"""
        poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(self.N, 4)
        err = np.zeros((self.F, 3)) if want_err else None
        hoff = np.zeros((self.F, 3, 3)) if want_off else None
        hd, b = np.zeros((self.N, 3, 3)), np.zeros((self.N, 3))
        chi2, ms = C.c_double(0), C.c_double(0)
        rc = self.L.lama_hip_pgo_linearize(self.h, _p(poses), _p(err), _p(hd), _p(hoff), _p(b), C.byref(chi2), C.byref(ms))
        if rc != 0:
            raise LamaError(self.L.lama_hip_pgo_last_error(self.h).decode())
        return {"err": err, "Hdiag": hd, "Hoff": hoff, "b": b, "chi2": chi2.value, "kernel_ms": ms.value}


class LidarOdometry2D:
    """ctypes view of the host-side lama::LidarOdometry2D (include/lama/lidar_odometry_2d.h)."""

    def __init__(self, resolution=0.05, max_iter=100, gpu_device=0):
        self.L = _hostlib()
        err = C.create_string_buffer(512)
        h = self.L.lama_lo_create(resolution, max_iter, gpu_device, err, 512)
        if not h:
            raise LamaError(err.value.decode())
        self.h = C.c_void_p(h)

    def close(self):
        if getattr(self, "h", None):
            self.L.lama_lo_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def engine_origin(self):
        return self.L.lama_lo_engine_origin(self.h).decode()

    def update(self, pts, ts=0.0, origin=None, quat=None):
        pts, origin, quat = PFSlam2D._scan(pts, origin, quat)
        rc = self.L.lama_lo_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), float(ts))
        if rc < 0:
            raise LamaError(self.L.lama_lo_last_error(self.h).decode())
        return bool(rc)

    def odom(self):
        out = np.zeros(4)
        self.L.lama_lo_get_odom(self.h, _p(out))
        return out

    def iterations(self):
        return self.L.lama_lo_iterations(self.h)

    def deleted_patches(self):
        return self.L.lama_lo_deleted_patches(self.h)

    def hip_context(self):
        """Borrowed HipContext view of the device context (map downloads in tests)."""
        ctx = HipContext.__new__(HipContext)
        ctx.L = _lib_of_origin(self.engine_origin())
        ctx.cfg = None
        ctx.P = 1
        ctx.h = C.c_void_p(self.L.lama_lo_device_context(self.h))
        ctx._is_borrowed = True
        return ctx
