"""ctypes bindings of include/lama_hip.h and include/lama_host.h."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "lib", "liblama_hip.so")
HOST_LIB = os.path.join(_HERE, "lib", "liblama_host.so")

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
                ("occ_patch_capacity", C.c_uint32), ("queue_capacity", C.c_uint32), ("profile", C.c_uint32)]


class HipCounters(C.Structure):
    _fields_ = [("ms_scan_match", C.c_double), ("launches_scan_match", C.c_uint64),
                ("ms_update_maps", C.c_double), ("launches_update_maps", C.c_uint64),
                ("ms_resample", C.c_double), ("launches_resample", C.c_uint64),
                ("gn_iterations", C.c_uint64), ("gn_evals", C.c_uint64), ("ray_cells", C.c_uint64),
                ("bf_cells", C.c_uint64), ("dm_patches", C.c_uint64), ("occ_patches", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


HIP_SYMBOLS = [
    "lama_hip_default_cfg", "lama_hip_device_count", "lama_hip_ctx_create", "lama_hip_ctx_destroy",
    "lama_hip_last_error", "lama_hip_pf_init", "lama_hip_pf_set_poses", "lama_hip_pf_get_poses",
    "lama_hip_pf_scan_match", "lama_hip_pf_resample", "lama_hip_pf_update_maps", "lama_hip_pf_map_patches",
    "lama_hip_pf_download_map", "lama_hip_match_batch", "lama_hip_pf_export_particle",
    "lama_hip_pf_import_particle", "lama_hip_get_counters", "lama_hip_reset_counters",
]

_hip = None


def hip_lib():
    """Load liblama_hip.so (raises if it has not been built: there is no fallback path)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise LamaError(f"{HIP_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        L = C.CDLL(HIP_LIB)
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
        L.lama_hip_pf_map_patches.argtypes = [vp, u32, i32, vp]
        L.lama_hip_pf_download_map.argtypes = [vp, u32, i32, u32, vp, vp, vp, vp]
        L.lama_hip_match_batch.argtypes = [vp, u32, vp, u32, vp, vp, vp, u32, vp]
        L.lama_hip_pf_export_particle.argtypes = [vp, u32, vp, u64, vp]
        L.lama_hip_pf_import_particle.argtypes = [vp, u32, vp, u64]
        L.lama_hip_get_counters.argtypes = [vp, vp]
        L.lama_hip_reset_counters.argtypes = [vp]
        for s in HIP_SYMBOLS:
            if s not in ("lama_hip_default_cfg", "lama_hip_ctx_destroy", "lama_hip_last_error"):
                getattr(L, s).restype = i32
        _hip = L
    return _hip


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
        self.L = hip_lib()
        self.cfg = cfg
        self.P = cfg.particles
        h = C.c_void_p()
        rc = self.L.lama_hip_ctx_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise LamaError(f"lama_hip_ctx_create failed with status {rc} (no usable HIP device or invalid cfg)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
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

    def match_batch(self, particle, pts, poses, origin=None, quat=None):
        pts, origin, quat = self._scan(pts, origin, quat)
        poses = np.ascontiguousarray(poses, dtype=np.float64)
        out = np.zeros(len(poses))
        self._chk(self.L.lama_hip_match_batch(self.h, particle, _p(pts), len(pts), _p(origin), _p(quat), _p(poses), len(poses), _p(out)))
        return out

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
