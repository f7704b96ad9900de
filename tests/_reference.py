"""ctypes view of the REFERENCE ITSELF compiled from /root/reference (oracle/_ref/liblama_ref.so, recipe oracle/Makefile.ref,
C view oracle/ref_capi.cpp).  TEST INFRASTRUCTURE ONLY: it pins the CPU oracle (tests/test_oracle_vs_reference.py) and
generates the golden vectors under tests/golden/ (tools/make_reference_golden.py)."""
import ctypes as C
import os

import numpy as np

from _oracle import DIST_T, FREQ_T, PFOptions, IDENT_Q, ZERO3, _p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(ROOT, "oracle", "_ref", "liblama_ref.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(LIB_PATH)
        vp, d, u32, i32, u64 = C.c_void_p, C.c_double, C.c_uint32, C.c_int, C.c_uint64
        sig = {
            "ref_pose_from_xyr": (None, [d, d, d, vp]), "ref_pose_plus_xyr": (None, [vp, vp, vp]), "ref_pose_minus_xyr": (None, [vp, vp, vp]),
            "ref_se2_exp": (None, [vp, vp]), "ref_pose_rotation": (d, [d, d, d]), "ref_cauchy": (d, [d, d]),
            "ref_random_set_seed": (None, [u32]), "ref_random_uniform": (d, []), "ref_random_normal": (d, [d]),
            "ref_dm_new": (vp, [d, u32, d]), "ref_dm_clone": (vp, [vp]), "ref_dm_free": (None, [vp]),
            "ref_dm_add_obstacle": (None, [vp, u32, u32, u32]), "ref_dm_remove_obstacle": (None, [vp, u32, u32, u32]),
            "ref_dm_update": (u32, [vp]), "ref_dm_max_distance": (d, [vp]), "ref_dm_distance_cell": (d, [vp, u32, u32, u32]),
            "ref_dm_distance": (d, [vp, vp, vp]),
            "ref_map_patch_ids": (i32, [vp, vp, i32]), "ref_map_patch_read": (i32, [vp, u64, vp, vp]),
            "ref_occ_patch_ids": (i32, [vp, vp, i32]), "ref_occ_patch_read": (i32, [vp, u64, vp, vp]),
            "ref_dm_w2m": (None, [vp, vp, vp]), "ref_dm_m2w": (None, [vp, vp, vp]), "ref_dm_m2p": (u64, [vp, vp]), "ref_dm_m2c": (u32, [vp, vp]),
            "ref_compute_ray": (i32, [vp, vp, vp, vp, i32]),
            "ref_occ_new": (vp, [d, u32]), "ref_occ_free": (None, [vp]),
            "ref_occ_set_free": (i32, [vp, u32, u32, u32]), "ref_occ_set_occupied": (i32, [vp, u32, u32, u32]),
            "ref_occ_probability": (d, [vp, u32, u32, u32]),
            "ref_eval": (None, [vp, vp, i32, vp, vp, vp, vp, vp]),
            "ref_pgo_linearize": (i32, [vp, u32, vp, vp, vp, vp, u32, vp, vp, vp]),
            "ref_solve": (None, [vp, vp, i32, vp, vp, vp, u32, i32, vp, vp]),
            "ref_pf_new": (vp, [vp]), "ref_pf_free": (None, [vp]), "ref_pf_set_prior": (None, [vp, vp]),
            "ref_pf_update": (i32, [vp, vp, i32, vp, vp, vp, d]), "ref_pf_neff": (d, [vp]), "ref_pf_best": (i32, [vp]),
            "ref_pf_get_pose": (None, [vp, vp]), "ref_pf_get_poses": (None, [vp, vp]), "ref_pf_get_weights": (None, [vp, vp, vp, vp]),
            "ref_pf_particle_dm": (vp, [vp, i32]), "ref_pf_particle_occ": (vp, [vp, i32]),
            "ref_slam_new": (vp, [d, d, d, d, d, d, u32, u32, i32, i32]), "ref_slam_free": (None, [vp]),
            "ref_slam_set_pose": (None, [vp, vp]), "ref_slam_get_pose": (None, [vp, vp]),
            "ref_slam_update": (i32, [vp, vp, i32, vp, vp, vp, d]), "ref_slam_dm": (vp, [vp]), "ref_slam_occ": (vp, [vp]),
            "ref_loc_new": (vp, [d, d, d, d, u32, u32, i32]), "ref_loc_free": (None, [vp]), "ref_loc_dm": (vp, [vp]),
            "ref_loc_occ_set": (None, [vp, vp, u32, i32]), "ref_loc_set_pose": (None, [vp, vp]), "ref_loc_get_pose": (None, [vp, vp]),
            "ref_loc_update": (i32, [vp, vp, i32, vp, vp, vp, d, i32]), "ref_loc_covar": (None, [vp, vp]), "ref_loc_rmse": (d, [vp]),
            "ref_dm_write": (i32, [vp, C.c_char_p]), "ref_dm_read": (i32, [vp, C.c_char_p]),
            "ref_occ_write": (i32, [vp, C.c_char_p]), "ref_occ_read": (i32, [vp, C.c_char_p]),
            "ref_dm_export_png": (i32, [vp, C.c_char_p]), "ref_occ_export_png": (i32, [vp, C.c_char_p]),
            "ref_image_read": (i32, [C.c_char_p, vp, vp, vp, u64]),
            "ref_loc_new2": (vp, [d, d, d, d, u32, u32, u32, u32, d, d]), "ref_loc_occ_set_state": (None, [vp, vp, u32, i32]),
            "ref_loc_trigger_gloc": (None, [vp]), "ref_loc_gloc_active": (i32, [vp]),
            "ref_lo_new": (vp, [d, u32]), "ref_lo_free": (None, [vp]), "ref_lo_update": (i32, [vp, vp, i32, vp, vp, d]),
            "ref_lo_get_odom": (None, [vp, vp]), "ref_lo_dm": (vp, [vp]), "ref_lo_occ": (vp, [vp]),
            "ref_pocc_patch_ids": (i32, [vp, vp, i32]), "ref_pocc_patch_read": (i32, [vp, u64, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


class Map:
    """A reference sdm map (borrowed or owned handle): patch ids and raw cell records in the reference's own struct layout."""
    cell_dtype = DIST_T
    _ids, _read = "ref_map_patch_ids", "ref_map_patch_read"

    def __init__(self, handle, owned=False, free=None):
        self.h, self.owned, self._free = handle, owned, free

    def __del__(self):
        if self.owned and self.h and self._free:
            getattr(lib(), self._free)(self.h)
            self.h = None

    def patch_ids(self):
        n = getattr(lib(), self._ids)(self.h, None, 0)
        ids = np.zeros(max(n, 1), dtype=np.uint64)
        getattr(lib(), self._ids)(self.h, _p(ids), n)
        return np.sort(ids[:n])

    def patch(self, pid):
        cells = np.zeros(1024, dtype=self.cell_dtype)
        mask = np.zeros(16, dtype=np.uint64)
        n = getattr(lib(), self._read)(self.h, int(pid), _p(cells), _p(mask))
        assert n == 1024
        return cells, mask

    def dump(self):
        return {int(i): self.patch(i) for i in self.patch_ids()}


class DM(Map):
    @classmethod
    def new(cls, res=0.05, patch=32, l2_max=0.5):
        return cls(lib().ref_dm_new(res, patch, l2_max), True, "ref_dm_free")

    def add(self, x, y, z=0):
        lib().ref_dm_add_obstacle(self.h, x, y, z)

    def remove(self, x, y, z=0):
        lib().ref_dm_remove_obstacle(self.h, x, y, z)

    def update(self):
        return lib().ref_dm_update(self.h)

    def distance_cell(self, x, y, z=0):
        return lib().ref_dm_distance_cell(self.h, x, y, z)

    def distance(self, p, grad=False):
        g = np.zeros(3)
        v = lib().ref_dm_distance(self.h, _p(_a(p)), _p(g) if grad else None)
        return (v, g) if grad else v

    def w2m(self, p):
        out = np.zeros(3, dtype=np.uint32)
        lib().ref_dm_w2m(self.h, _p(_a(p)), _p(out))
        return out

    def m2w(self, c):
        out = np.zeros(3)
        lib().ref_dm_m2w(self.h, _p(np.ascontiguousarray(c, dtype=np.uint32)), _p(out))
        return out

    def compute_ray(self, frm, to):
        f, t = np.ascontiguousarray(frm, dtype=np.uint32), np.ascontiguousarray(to, dtype=np.uint32)
        n = lib().ref_compute_ray(self.h, _p(f), _p(t), None, 0)
        out = np.zeros((max(n, 1), 3), dtype=np.uint32)
        lib().ref_compute_ray(self.h, _p(f), _p(t), _p(out), n)
        return out[:n]


class Occ(Map):
    cell_dtype = FREQ_T
    _ids, _read = "ref_occ_patch_ids", "ref_occ_patch_read"

    @classmethod
    def new(cls, res=0.05, patch=32):
        return cls(lib().ref_occ_new(res, patch), True, "ref_occ_free")

    def set_free(self, x, y, z=0):
        return lib().ref_occ_set_free(self.h, x, y, z)

    def set_occupied(self, x, y, z=0):
        return lib().ref_occ_set_occupied(self.h, x, y, z)

    def probability(self, x, y, z=0):
        return lib().ref_occ_probability(self.h, x, y, z)


class POcc(Map):
    from _oracle import PROB_T as cell_dtype
    _ids, _read = "ref_pocc_patch_ids", "ref_pocc_patch_read"


def image_read(filename):
    """Decode an image file with the reference's reader (grey pixels, rows top to bottom)."""
    w, h = C.c_uint32(0), C.c_uint32(0)
    assert lib().ref_image_read(filename.encode(), C.byref(w), C.byref(h), None, 0)
    out = np.zeros((h.value, w.value), dtype=np.uint8)
    assert lib().ref_image_read(filename.encode(), C.byref(w), C.byref(h), _p(out), out.size)
    return out


def pose_from_xyr(x, y, r):
    out = np.zeros(4)
    lib().ref_pose_from_xyr(x, y, r, _p(out))
    return out


def eval_(dm, pts, xyr, origin=ZERO3, quat=IDENT_Q, jac=True):
    pts = _a(pts)
    n = len(pts)
    r = np.zeros(n)
    J = np.zeros((3, n)) if jac else None
    lib().ref_eval(dm.h, _p(pts), n, _p(_a(origin)), _p(_a(quat)), _p(_a(xyr)), _p(r), _p(J) if jac else None)
    return (r, J.T.copy()) if jac else r


def solve(dm, pts, xyr, max_iter=100, lm=False, cov=False, origin=ZERO3, quat=IDENT_Q):
    pts = _a(pts)
    pose = np.zeros(4)
    c = np.zeros(9) if cov else None
    lib().ref_solve(dm.h, _p(pts), len(pts), _p(_a(origin)), _p(_a(quat)), _p(_a(xyr)), max_iter, 1 if lm else 0, _p(pose), _p(c) if cov else None)
    return (pose, c.reshape(3, 3)) if cov else pose


class PF:
    def __init__(self, opts: PFOptions):
        self.h = lib().ref_pf_new(C.byref(opts))
        self.P = opts.particles

    def __del__(self):
        if self.h:
            lib().ref_pf_free(self.h)
            self.h = None

    def set_prior(self, xyr):
        lib().ref_pf_set_prior(self.h, _p(_a(xyr)))

    def update(self, pts, odom_xyr, ts=0.0, origin=ZERO3, quat=IDENT_Q):
        pts = _a(pts)
        return bool(lib().ref_pf_update(self.h, _p(pts), len(pts), _p(_a(origin)), _p(_a(quat)), _p(_a(odom_xyr)), ts))

    def poses(self):
        out = np.zeros((self.P, 4))
        lib().ref_pf_get_poses(self.h, _p(out))
        return out

    def pose(self):
        out = np.zeros(4)
        lib().ref_pf_get_pose(self.h, _p(out))
        return out

    def weights(self):
        w, nw, ws = np.zeros(self.P), np.zeros(self.P), np.zeros(self.P)
        lib().ref_pf_get_weights(self.h, _p(w), _p(nw), _p(ws))
        return w, nw, ws

    def neff(self):
        return lib().ref_pf_neff(self.h)

    def best(self):
        return lib().ref_pf_best(self.h)

    def dm(self, i):
        return DM(lib().ref_pf_particle_dm(self.h, i))

    def occ(self, i):
        return Occ(lib().ref_pf_particle_occ(self.h, i))


def pgo_linearize(poses, fi, fj, meas, sqrt_info):
    """minisam's own linearzationLowerHessian (vendor/minisam, compiled into liblama_ref.so) on a graph built like
    SimplePGO::optimize's: dense symmetric H (3N x 3N, pose order), b = Atb (N x 3), whitened errors (F x 3)."""
    poses = np.ascontiguousarray(poses, dtype=np.float64).reshape(-1, 4)
    fi = np.ascontiguousarray(fi, dtype=np.int32); fj = np.ascontiguousarray(fj, dtype=np.int32)
    meas = np.ascontiguousarray(meas, dtype=np.float64).reshape(-1, 4)
    sq = np.ascontiguousarray(sqrt_info, dtype=np.float64).reshape(-1, 3)
    N, F = len(poses), len(fi)
    H, b, err = np.zeros((3 * N, 3 * N)), np.zeros((N, 3)), np.zeros((F, 3))
    rc = lib().ref_pgo_linearize(_p(poses), N, _p(fi), _p(fj), _p(meas), _p(sq), F, _p(H), _p(b), _p(err))
    assert rc == 0, "the reference's linearisation threw"
    return {"H": H, "b": b, "err": err}
