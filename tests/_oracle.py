"""ctypes view of the CPU oracle (oracle/_build/liblama_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = os.path.join(ROOT, "oracle", "_build", "liblama_oracle.so")

DIST_T = np.dtype([("obstacle", "<i2", (3,)), ("sqdist", "<u2"), ("valid", "u1"), ("queued", "u1")])
FREQ_T = np.dtype([("occupied", "<u2"), ("visited", "<u2")])
assert DIST_T.itemsize == 10 and FREQ_T.itemsize == 4


class PFOptions(C.Structure):
    _fields_ = [("particles", C.c_uint32), ("srr", C.c_double), ("str", C.c_double), ("stt", C.c_double),
                ("srt", C.c_double), ("meas_sigma", C.c_double), ("meas_sigma_gain", C.c_double),
                ("trans_thresh", C.c_double), ("rot_thresh", C.c_double), ("l2_max", C.c_double),
                ("truncated_ray", C.c_double), ("truncated_range", C.c_double), ("resolution", C.c_double),
                ("patch_size", C.c_uint32), ("max_iter", C.c_uint32), ("threads", C.c_int32), ("seed", C.c_uint32)]


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(_LIB)
        vp, d, u32, i32, u64 = C.c_void_p, C.c_double, C.c_uint32, C.c_int, C.c_uint64
        sig = {
            "orc_se2_from_xyr": (None, [d, d, d, vp]), "orc_se2_rotation": (d, [vp]), "orc_cauchy": (d, [d, d]),
            "orc_m2p": (u64, [d, u32, vp]), "orc_m2c": (u32, [d, u32, vp]), "orc_w2m": (None, [d, u32, vp, vp]),
            "orc_compute_ray": (i32, [vp, vp, vp, i32]),
            "orc_dm_new": (vp, [d, u32, d]), "orc_dm_clone": (vp, [vp]), "orc_dm_free": (None, [vp]),
            "orc_dm_max_sqdist": (u32, [vp]),
            "orc_dm_add_obstacle": (None, [vp, u32, u32, u32]), "orc_dm_remove_obstacle": (None, [vp, u32, u32, u32]),
            "orc_dm_update": (u32, [vp]), "orc_dm_distance_cell": (d, [vp, u32, u32, u32]),
            "orc_dm_distance": (d, [vp, vp, vp]), "orc_dm_patch_ids": (i32, [vp, vp, i32]),
            "orc_dm_patch_read": (i32, [vp, u64, vp, vp]), "orc_dm_stats": (None, [vp, vp]),
            "orc_occ_new": (vp, [d, u32]), "orc_occ_clone": (vp, [vp]), "orc_occ_free": (None, [vp]),
            "orc_occ_set_free": (i32, [vp, u32, u32, u32]), "orc_occ_set_occupied": (i32, [vp, u32, u32, u32]),
            "orc_occ_probability": (d, [vp, u32, u32, u32]), "orc_occ_patch_ids": (i32, [vp, vp, i32]),
            "orc_occ_patch_read": (i32, [vp, u64, vp, vp]),
            "orc_eval": (None, [vp, vp, i32, vp, vp, vp, vp, vp]),
            "orc_solve": (i32, [vp, vp, i32, vp, vp, vp, u32, vp]),
            "orc_loglik": (d, [vp, vp, i32, vp, vp, vp, d]),
            "orc_pf_new": (vp, [vp]), "orc_pf_free": (None, [vp]), "orc_pf_set_prior": (None, [vp, vp]),
            "orc_pf_count_touches": (None, [vp, i32]),
            "orc_pf_update": (i32, [vp, vp, i32, vp, vp, vp, d]), "orc_pf_times": (None, [vp, vp, vp]),
            "orc_pf_num_resamples": (u32, [vp]), "orc_pf_neff": (d, [vp]), "orc_pf_best": (i32, [vp]),
            "orc_pf_get_poses": (None, [vp, vp]), "orc_pf_set_poses": (None, [vp, vp]),
            "orc_pf_get_weights": (None, [vp, vp, vp, vp]), "orc_pf_set_weights": (None, [vp, vp, vp]),
            "orc_pf_particle_dm": (vp, [vp, i32]), "orc_pf_particle_occ": (vp, [vp, i32]),
            "orc_pf_counters": (None, [vp, i32, vp]), "orc_pf_last_sample_idx": (i32, [vp, vp, i32]),
            "orc_pf_stage_set_scan": (None, [vp, vp, i32, vp, vp]), "orc_pf_stage_scan_match": (None, [vp]),
            "orc_pf_stage_update_maps": (None, [vp]), "orc_pf_stage_normalize": (d, [vp]),
            "orc_pf_stage_resample_indices": (None, [vp, d, vp]), "orc_pf_stage_resample_with": (None, [vp, vp, i32]),
            "orc_pf_draw_from_motion": (None, [vp, vp, vp]),
            "orc_scan_tf": (None, [vp, vp, vp, vp]), "orc_ldlt3_solve": (None, [vp, vp, vp]),
            "orc_se2_exp": (None, [vp, vp]), "orc_se2_mul": (None, [vp, vp, vp]), "orc_se2_inverse": (None, [vp, vp]),
            "orc_pose_minus": (None, [vp, vp, vp]),
            "orc_slam_new": (vp, [d, d, d, d, d, d, u32, u32]), "orc_slam_new2": (vp, [d, d, d, d, d, d, u32, u32, i32]),
            "orc_slam_deleted_last": (u32, [vp]), "orc_slam_set_lm": (None, [vp, i32]), "orc_loc_set_lm": (None, [vp, i32]), "orc_slam_free": (None, [vp]),
            "orc_slam_set_pose": (None, [vp, vp]), "orc_slam_get_pose": (None, [vp, vp]),
            "orc_slam_update": (i32, [vp, vp, i32, vp, vp, vp, d]), "orc_slam_enough_motion": (i32, [vp, vp]),
            "orc_slam_processed_cells": (u32, [vp]), "orc_slam_iterations": (u32, [vp]),
            "orc_slam_dm": (vp, [vp]), "orc_slam_occ": (vp, [vp]),
            "orc_loc_new": (vp, [d, d, d, d, u32, u32]), "orc_loc_free": (None, [vp]), "orc_loc_dm": (vp, [vp]),
            "orc_loc_set_pose": (None, [vp, vp]), "orc_loc_get_pose": (None, [vp, vp]),
            "orc_loc_update": (i32, [vp, vp, i32, vp, vp, vp, d, i32]), "orc_loc_covar": (None, [vp, vp]),
            "orc_loc_rmse": (d, [vp]), "orc_loc_iterations": (u32, [vp]), "orc_loc_rank_deficient": (i32, [vp]),
            "orc_lo_new": (vp, [d, u32]), "orc_lo_free": (None, [vp]), "orc_lo_update": (i32, [vp, vp, i32, vp, vp, d]),
            "orc_lo_get_odom": (None, [vp, vp]), "orc_lo_set_odom": (None, [vp, vp]), "orc_lo_dm": (vp, [vp]), "orc_lo_occ": (vp, [vp]),
            "orc_lo_deleted_last": (u32, [vp]), "orc_lo_map_updates": (u32, [vp]), "orc_lo_iterations": (u32, [vp]),
            "orc_pocc_patch_ids": (i32, [vp, vp, i32]), "orc_pocc_patch_read": (i32, [vp, u64, vp, vp]), "orc_pocc_free": (None, [vp]),
            "orc_pocc_params": (None, [vp]),
            "orc_pgo_linearize": (None, [vp, u32, vp, vp, vp, vp, u32, vp, vp, vp, vp, vp]),
            "orc_map_write": (i32, [vp, C.c_char_p]), "orc_map_read": (i32, [vp, C.c_char_p]),
            "orc_map_image": (i32, [vp, i32, vp, vp, vp, C.c_uint64]),
            "orc_loc_new2": (vp, [d, d, d, d, u32, u32, u32, u32, d, d]), "orc_random_set_seed": (None, [u32]),
            "orc_random_uniform": (d, []), "orc_loc_occ_set": (None, [vp, vp, u32, i32]), "orc_loc_occ_bounds": (None, [vp, vp]),
            "orc_loc_trigger_gloc": (None, [vp]), "orc_loc_gloc_active": (i32, [vp]),
            "orc_loc_gloc_candidates": (u32, [vp, vp, vp, u32]), "orc_loc_sampling_l": (u32, [vp, vp, u32]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


IDENT_Q = np.array([1.0, 0, 0, 0])
ZERO3 = np.zeros(3)


def set_canonical_default(on):
    """Maps created afterwards use the level-synchronous brushfire variant (NOT the reference's tie order)."""
    lib().orc_set_canonical_default(1 if on else 0)


def se2(x, y, r):
    out = np.zeros(4)
    lib().orc_se2_from_xyr(x, y, r, _p(out))
    return out


def se2_exp(v):
    out = np.zeros(4)
    lib().orc_se2_exp(_p(np.ascontiguousarray(v, dtype=np.float64)), _p(out))
    return out


def se2_mul(a, b):
    out = np.zeros(4)
    lib().orc_se2_mul(_p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), _p(out))
    return out


def se2_inverse(a):
    out = np.zeros(4)
    lib().orc_se2_inverse(_p(np.ascontiguousarray(a)), _p(out))
    return out


def default_options(**kw):
    o = PFOptions()
    lib().orc_pf_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _bind_map_queries():
    L = lib()
    if getattr(L, "_map_queries_bound", False):
        return L
    vp = C.c_void_p
    for pre in ("orc_occ", "orc_dm"):
        getattr(L, pre + "_bounds").restype = None
        getattr(L, pre + "_bounds").argtypes = [vp, vp, vp, vp, vp]
        getattr(L, pre + "_cells").restype = C.c_int64
        getattr(L, pre + "_cells").argtypes = [vp, vp, C.c_uint64]
    L.orc_occ_state.restype = C.c_int
    L.orc_occ_state.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32]
    L._map_queries_bound = True
    return L


class _MapBase:
    def bounds(self):
        """Map::bounds: (min cell, max cell, min world, max world)."""
        L = _bind_map_queries()
        mn, mx = np.zeros(3, dtype=np.uint32), np.zeros(3, dtype=np.uint32)
        wmn, wmx = np.zeros(3), np.zeros(3)
        getattr(L, self._pre + "_bounds")(self.h, _p(mn), _p(mx), _p(wmn), _p(wmx))
        return mn, mx, wmn, wmx

    def cells(self):
        """Map::visit_all_cells: (n, 2) cell coordinates whose mask bit is on."""
        L = _bind_map_queries()
        n = getattr(L, self._pre + "_cells")(self.h, None, 0)
        out = np.zeros((n, 2), dtype=np.uint32)
        getattr(L, self._pre + "_cells")(self.h, _p(out), n)
        return out

    _pre = None
    _dtype = None

    def __init__(self, handle, owned=True):
        self.h = C.c_void_p(handle)
        self.owned = owned

    def __del__(self):
        if getattr(self, "owned", False) and self.h:
            getattr(lib(), self._pre + "_free")(self.h)
            self.h = None

    def patch_ids(self):
        n = getattr(lib(), self._pre + "_patch_ids")(self.h, None, 0)
        ids = np.zeros(n, dtype=np.uint64)
        getattr(lib(), self._pre + "_patch_ids")(self.h, _p(ids), n)
        return ids

    def patch(self, pid):
        cells = np.zeros(1024, dtype=self._dtype)
        mask = np.zeros(16, dtype=np.uint64)
        r = getattr(lib(), self._pre + "_patch_read")(self.h, int(pid), _p(cells), _p(mask))
        assert r == cells.nbytes, r
        return cells, mask

    def dump(self):
        """{patch_id: (cells[1024], mask[16])}"""
        return {int(p): self.patch(p) for p in self.patch_ids()}

    # Map::write / Map::read (src/sdm/map.cpp:489-575) and sdm::export_to_png's pixel array (src/sdm/export.cpp:46-95)
    def write(self, filename):
        assert lib().orc_map_write(self.h, filename.encode()) == 1

    def read(self, filename):
        return lib().orc_map_read(self.h, filename.encode()) == 1

    def image(self):
        kind = 0 if self._pre == "orc_dm" else 1
        w, h = C.c_uint32(0), C.c_uint32(0)
        lib().orc_map_image(self.h, kind, C.byref(w), C.byref(h), None, 0)
        out = np.zeros((h.value, w.value), dtype=np.uint8)
        lib().orc_map_image(self.h, kind, C.byref(w), C.byref(h), _p(out), out.size)
        return out


class DM(_MapBase):
    _pre = "orc_dm"
    _dtype = DIST_T

    @classmethod
    def new(cls, res=0.05, patch=32, l2_max=0.5):
        return cls(lib().orc_dm_new(res, patch, l2_max))

    def clone(self):
        return DM(lib().orc_dm_clone(self.h))

    def add(self, x, y, z=0):
        lib().orc_dm_add_obstacle(self.h, x, y, z)

    def remove(self, x, y, z=0):
        lib().orc_dm_remove_obstacle(self.h, x, y, z)

    def update(self):
        return lib().orc_dm_update(self.h)

    def max_sqdist(self):
        return lib().orc_dm_max_sqdist(self.h)

    def distance_cell(self, x, y, z=0):
        return lib().orc_dm_distance_cell(self.h, x, y, z)

    def distance(self, p, grad=False):
        p = np.ascontiguousarray(p, dtype=np.float64)
        g = np.zeros(3)
        d = lib().orc_dm_distance(self.h, _p(p), _p(g) if grad else None)
        return (d, g) if grad else d

    def stats(self):
        s = np.zeros(7, dtype=np.uint64)
        lib().orc_dm_stats(self.h, _p(s))
        return dict(zip(["raise_pops", "lower_pops", "lower_fired", "pushes", "max_queue", "tie_overwrites", "max_queue_last"], s.tolist()))


class Occ(_MapBase):
    _pre = "orc_occ"
    _dtype = FREQ_T

    @classmethod
    def new(cls, res=0.05, patch=32):
        return cls(lib().orc_occ_new(res, patch))

    def set_free(self, x, y, z=0):
        return bool(lib().orc_occ_set_free(self.h, x, y, z))

    def set_occupied(self, x, y, z=0):
        return bool(lib().orc_occ_set_occupied(self.h, x, y, z))

    def probability(self, x, y, z=0):
        return lib().orc_occ_probability(self.h, x, y, z)

    def state(self, x, y, z=0):
        """(isFree, isOccupied, isUnknown)"""
        b = _bind_map_queries().orc_occ_state(self.h, int(x), int(y), int(z))
        return bool(b & 1), bool(b & 2), bool(b & 4)


PROB_T = np.dtype([("prob", "<f4")])


class POcc(_MapBase):
    """ProbabilisticOccupancyMap (log-odds float cells)."""
    _pre = "orc_pocc"
    _dtype = PROB_T


def pocc_params():
    """miss, hit, clamp_min, clamp_max, occ_thresh of ProbabilisticOccupancyMap (float-rounded doubles)"""
    out = np.zeros(5)
    lib().orc_pocc_params(_p(out))
    return out


class LidarOdometry:
    """Oracle LidarOdometry2D (src/lidar_odometry_2d.cpp)."""

    def __init__(self, resolution=0.05, max_iter=100):
        self.h = C.c_void_p(lib().orc_lo_new(resolution, max_iter))

    def __del__(self):
        if self.h:
            lib().orc_lo_free(self.h)
            self.h = None

    def update(self, pts, ts=0.0, origin=ZERO3, quat=IDENT_Q):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        return bool(lib().orc_lo_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), ts))

    def odom(self):
        out = np.zeros(4)
        lib().orc_lo_get_odom(self.h, _p(out))
        return out

    def set_odom(self, pose4):
        lib().orc_lo_set_odom(self.h, _p(np.ascontiguousarray(pose4, dtype=np.float64)))

    def dm(self):
        return DM(lib().orc_lo_dm(self.h), owned=False)

    def occ(self):
        return POcc(lib().orc_lo_occ(self.h), owned=False)

    def deleted_last(self):
        return lib().orc_lo_deleted_last(self.h)

    def map_updates(self):
        return lib().orc_lo_map_updates(self.h)

    def iterations(self):
        return lib().orc_lo_iterations(self.h)


def compute_ray(frm, to):
    frm = np.asarray(frm, dtype=np.uint32)
    to = np.asarray(to, dtype=np.uint32)
    n = lib().orc_compute_ray(_p(frm), _p(to), None, 0)
    out = np.zeros((max(n, 1), 3), dtype=np.uint32)
    lib().orc_compute_ray(_p(frm), _p(to), _p(out), n)
    return out[:n]


def w2m(p, res=0.05, patch=32):
    out = np.zeros(3, dtype=np.uint32)
    lib().orc_w2m(res, patch, _p(np.ascontiguousarray(p, dtype=np.float64)), _p(out))
    return out


def eval_(dm, pts, pose, origin=ZERO3, quat=IDENT_Q, jac=True):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    n = len(pts)
    r = np.zeros(n)
    J = np.zeros((n, 3)) if jac else None
    lib().orc_eval(dm.h, _p(pts), n, _p(origin), _p(quat), _p(np.ascontiguousarray(pose)), _p(r), _p(J))
    return (r, J) if jac else r


def solve(dm, pts, pose, max_iter=100, origin=ZERO3, quat=IDENT_Q):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    pose = np.array(pose, dtype=np.float64)
    ev = C.c_uint32(0)
    it = lib().orc_solve(dm.h, _p(pts), len(pts), _p(origin), _p(quat), _p(pose), max_iter, C.byref(ev))
    return pose, it, ev.value


def solve_full(dm, pts, pose, max_iter=100, lm=False, origin=ZERO3, quat=IDENT_Q):
    """Solve(GN | LM, Cauchy(0.15), &cov): (pose, iterations, 3x3 covariance)."""
    L = lib()
    L.orc_solve_full.restype = C.c_int
    L.orc_solve_full.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    pose = np.array(pose, dtype=np.float64)
    cov = np.zeros(9)
    it = L.orc_solve_full(dm.h, _p(pts), len(pts), _p(origin), _p(quat), _p(pose), max_iter, 1 if lm else 0, _p(cov))
    return pose, it, cov.reshape(3, 3)


def match_error(dm, pts, pose, origin=ZERO3, quat=IDENT_Q):
    """MatchSurface2D::error(): rms of the non-interpolated cell distances."""
    L = lib()
    L.orc_match_error.restype = C.c_double
    L.orc_match_error.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    return L.orc_match_error(dm.h, _p(pts), len(pts), _p(origin), _p(quat), _p(np.ascontiguousarray(pose, dtype=np.float64)))


def loglik(dm, pts, pose, sigma=0.05, origin=ZERO3, quat=IDENT_Q):
    pts = np.ascontiguousarray(pts, dtype=np.float64)
    return lib().orc_loglik(dm.h, _p(pts), len(pts), _p(origin), _p(quat), _p(np.ascontiguousarray(pose)), sigma)


class PF:
    def __init__(self, opts):
        self.opts = opts
        self.P = opts.particles
        self.h = C.c_void_p(lib().orc_pf_new(C.byref(opts)))
        self._keep = None

    def __del__(self):
        if self.h and lib is not None:          # (at interpreter shutdown the module's globals may be gone already)
            lib().orc_pf_free(self.h)
            self.h = None

    def set_prior(self, pose4):
        lib().orc_pf_set_prior(self.h, _p(np.ascontiguousarray(pose4)))

    def count_touches(self, on=True):
        lib().orc_pf_count_touches(self.h, 1 if on else 0)

    def update(self, pts, odom4, ts=0.0, origin=ZERO3, quat=IDENT_Q):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        return bool(lib().orc_pf_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(np.ascontiguousarray(odom4)), ts))

    def times(self):
        t = np.zeros(5)
        rs = C.c_int(0)
        lib().orc_pf_times(self.h, _p(t), C.byref(rs))
        return dict(total=t[0], solving=t[1], normalizing=t[2], resampling=t[3], mapping=t[4], resampled=bool(rs.value))

    def poses(self):
        out = np.zeros((self.P, 4))
        lib().orc_pf_get_poses(self.h, _p(out))
        return out

    def set_poses(self, poses):
        lib().orc_pf_set_poses(self.h, _p(np.ascontiguousarray(poses, dtype=np.float64)))

    def weights(self):
        w, nw, ws = np.zeros(self.P), np.zeros(self.P), np.zeros(self.P)
        lib().orc_pf_get_weights(self.h, _p(w), _p(nw), _p(ws))
        return w, nw, ws

    def set_weights(self, w=None, ws=None):
        lib().orc_pf_set_weights(self.h, _p(np.ascontiguousarray(w)) if w is not None else None,
                                 _p(np.ascontiguousarray(ws)) if ws is not None else None)

    def neff(self):
        return lib().orc_pf_neff(self.h)

    def best(self):
        return lib().orc_pf_best(self.h)

    def num_resamples(self):
        return lib().orc_pf_num_resamples(self.h)

    def dm(self, i):
        return DM(lib().orc_pf_particle_dm(self.h, i), owned=False)

    def occ(self, i):
        return Occ(lib().orc_pf_particle_occ(self.h, i), owned=False)

    def counters(self, i):
        c = np.zeros(9, dtype=np.uint64)
        lib().orc_pf_counters(self.h, i, _p(c))
        return dict(zip(["iterations", "evals", "ray_cells", "occ_events", "bf_processed", "n_match", "n_occ", "n_bf", "n_match_or_bf"], c.tolist()))

    def map_checksums(self, kind):
        """Per-particle map checksums, the function lama_hip_pf_map_checksums computes on the device (kind: 0 distance, 1 occupancy,
        2 distance as liblama_hip_wide.so packs it: l2_max beyond 127 cells)."""
        L = lib()
        L.orc_pf_map_checksums.restype = None
        L.orc_pf_map_checksums.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        out = np.zeros(self.P, dtype=np.uint64)
        L.orc_pf_map_checksums(self.h, int(kind), _p(out))
        return out

    def last_sample_idx(self):
        out = np.zeros(self.P, dtype=np.int32)
        n = lib().orc_pf_last_sample_idx(self.h, _p(out), self.P)
        return out[:n]

    # stage-wise
    def stage_set_scan(self, pts, origin=ZERO3, quat=IDENT_Q):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        lib().orc_pf_stage_set_scan(self.h, _p(pts), len(pts), _p(origin), _p(quat))

    def stage_scan_match(self):
        lib().orc_pf_stage_scan_match(self.h)

    def stage_update_maps(self):
        lib().orc_pf_stage_update_maps(self.h)

    def stage_normalize(self):
        return lib().orc_pf_stage_normalize(self.h)

    def stage_resample_indices(self, u):
        out = np.zeros(self.P, dtype=np.int32)
        lib().orc_pf_stage_resample_indices(self.h, u, _p(out))
        return out

    def stage_resample_with(self, idx):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        lib().orc_pf_stage_resample_with(self.h, _p(idx), len(idx))

    def draw_from_motion(self, delta4, pose4):
        pose = np.array(pose4, dtype=np.float64)
        lib().orc_pf_draw_from_motion(self.h, _p(np.ascontiguousarray(delta4)), _p(pose))
        return pose


class Slam:
    """Oracle Slam2D (src/slam2d.cpp)."""

    def __init__(self, trans_thresh=0.5, rot_thresh=0.5, l2_max=0.5, truncated_ray=0.0, truncated_range=0.0,
                 resolution=0.05, patch_size=32, max_iter=100, transient_map=False):
        self.h = C.c_void_p(lib().orc_slam_new2(trans_thresh, rot_thresh, l2_max, truncated_ray, truncated_range,
                                                resolution, patch_size, max_iter, 1 if transient_map else 0))

    def deleted_last(self):
        return lib().orc_slam_deleted_last(self.h)

    def set_lm(self, on=True):
        """Options::strategy = "lm" (Levenberg-Marquardt) instead of Gauss-Newton"""
        lib().orc_slam_set_lm(self.h, 1 if on else 0)

    def __del__(self):
        if self.h:
            lib().orc_slam_free(self.h)
            self.h = None

    def set_pose(self, pose4):
        lib().orc_slam_set_pose(self.h, _p(np.ascontiguousarray(pose4, dtype=np.float64)))

    def pose(self):
        out = np.zeros(4)
        lib().orc_slam_get_pose(self.h, _p(out))
        return out

    def update(self, pts, odom4, ts=0.0, origin=ZERO3, quat=IDENT_Q):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        return bool(lib().orc_slam_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(np.ascontiguousarray(odom4)), ts))

    def enough_motion(self, odom4):
        return bool(lib().orc_slam_enough_motion(self.h, _p(np.ascontiguousarray(odom4))))

    def processed_cells(self):
        return lib().orc_slam_processed_cells(self.h)

    def iterations(self):
        return lib().orc_slam_iterations(self.h)

    def dm(self):
        return DM(lib().orc_slam_dm(self.h), owned=False)

    def occ(self):
        return Occ(lib().orc_slam_occ(self.h), owned=False)


class Loc:
    """Oracle Loc2D (src/loc2d.cpp) incl. global localisation and sampling covariance."""

    def __init__(self, trans_thresh=0.5, rot_thresh=0.5, l2_max=1.0, resolution=0.05, patch_size=32, max_iter=100,
                 gloc_particles=3000, gloc_iters=10, gloc_thresh=0.15, cov_blend=0.0):
        self.h = C.c_void_p(lib().orc_loc_new2(trans_thresh, rot_thresh, l2_max, resolution, patch_size, max_iter,
                                               gloc_particles, gloc_iters, gloc_thresh, cov_blend))

    def __del__(self):
        if self.h:
            lib().orc_loc_free(self.h)
            self.h = None

    def dm(self):
        return DM(lib().orc_loc_dm(self.h), owned=False)

    def set_pose(self, pose4):
        lib().orc_loc_set_pose(self.h, _p(np.ascontiguousarray(pose4, dtype=np.float64)))

    def pose(self):
        out = np.zeros(4)
        lib().orc_loc_get_pose(self.h, _p(out))
        return out

    def update(self, pts, odom4, ts=0.0, force=False, origin=ZERO3, quat=IDENT_Q):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        return bool(lib().orc_loc_update(self.h, _p(pts), len(pts), _p(origin), _p(quat), _p(np.ascontiguousarray(odom4)), ts, 1 if force else 0))

    def covar(self):
        out = np.zeros(9)
        lib().orc_loc_covar(self.h, _p(out))
        return out.reshape(3, 3)

    def rmse(self):
        return lib().orc_loc_rmse(self.h)

    def iterations(self):
        return lib().orc_loc_iterations(self.h)

    def rank_deficient(self):
        return bool(lib().orc_loc_rank_deficient(self.h))

    def occ_set_cells(self, cells_xy, state):
        cells = np.ascontiguousarray(cells_xy, dtype=np.uint32).reshape(-1, 2)
        lib().orc_loc_occ_set(self.h, _p(cells), len(cells), int(state))

    def occ_bounds(self):
        out = np.zeros(6)
        lib().orc_loc_occ_bounds(self.h, _p(out))
        return out[:3].copy(), out[3:].copy()

    def set_lm(self, on=True):
        lib().orc_loc_set_lm(self.h, 1 if on else 0)

    def trigger_global_localization(self):
        lib().orc_loc_trigger_gloc(self.h)

    def global_localization_active(self):
        return bool(lib().orc_loc_gloc_active(self.h))

    def gloc_candidates(self):
        n = lib().orc_loc_gloc_candidates(self.h, None, None, 0)
        poses, err = np.zeros((n, 4)), np.zeros(n)
        if n:
            lib().orc_loc_gloc_candidates(self.h, _p(poses), _p(err), n)
        return poses, err

    def sampling_likelihoods(self):
        n = lib().orc_loc_sampling_l(self.h, None, 0)
        out = np.zeros(n)
        if n:
            lib().orc_loc_sampling_l(self.h, _p(out), n)
        return out


def random_set_seed(seed):
    lib().orc_random_set_seed(int(seed))


def pgo_linearize(poses, fi, fj, meas, sqrt_info):
    """Oracle restatement of minisam's linearzationLowerHessian for SE2 prior/between factors (dense blocks)."""
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4)
    fi = np.ascontiguousarray(fi, dtype=np.int32); fj = np.ascontiguousarray(fj, dtype=np.int32)
    meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 4)
    sq = np.ascontiguousarray(sqrt_info, dtype=np.float64).reshape(-1, 3)
    N, F = len(poses), len(fi)
    err, hd, ho, b = np.zeros((F, 3)), np.zeros((N, 3, 3)), np.zeros((F, 3, 3)), np.zeros((N, 3))
    chi2 = C.c_double(0)
    lib().orc_pgo_linearize(_p(poses), N, _p(fi), _p(fj), _p(meas), _p(sq), F, _p(err), _p(hd), _p(ho), _p(b), C.byref(chi2))
    return {"err": err, "Hdiag": hd, "Hoff": ho, "b": b, "chi2": chi2.value}
