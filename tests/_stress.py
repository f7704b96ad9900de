"""Randomised map-exactness case shared by the GPU parity tests and the lane-simulator tests (test infrastructure)."""
import numpy as np

import _oracle as O
from _cmp import DM_FIELDS, OCC_FIELDS, assert_maps_equal


def random_room_scan(rng, pose_xyr, n_beams, kind):
    """Ranges of a lidar at pose (x, y, yaw) inside a random star-shaped room (radius as a random trigonometric series)."""
    ang = np.deg2rad(np.linspace(-135.0, 135.0, n_beams))
    x0, y0, yaw = pose_xyr
    th = yaw + ang
    lo, hi = np.zeros(n_beams), np.full(n_beams, kind["R"] * 2.5)
    for _ in range(40):                                   # bisection along every beam until it leaves the room
        mid = 0.5 * (lo + hi)
        px, py = x0 + mid * np.cos(th), y0 + mid * np.sin(th)
        phi = np.arctan2(py, px)
        rr = kind["R"] * (1.0 + sum(c * np.cos(m * phi + p) for m, c, p in kind["coef"]))
        inside = np.hypot(px, py) < rr
        lo = np.where(inside, mid, lo)
        hi = np.where(inside, hi, mid)
    r = lo + rng.normal(0, 0.01, n_beams)
    return np.stack([r * np.cos(ang), r * np.sin(ang), np.zeros(n_beams)], axis=1)


def random_rooms_case(F, seed, small=False, l2_max=None):
    """Random star-shaped room, random beam count, ray truncation options and particle count; 4 scans from perturbed poses (walls are
    re-drawn a cell off: raise waves as well as lower waves) with a resample in between: occupancy and distance maps bit-identical to
    the oracle.  small: rooms of 2.5 - 4.5 m, at most 360 beams and 3 particles (the lane simulator).  l2_max: the reach of the distance
    map in metres (default: the option's default, 0.5 m = 10 cells; beyond 6.35 m = 127 cells the wide device library runs)."""
    rng = np.random.default_rng(1000 + seed)
    kind = {"R": rng.uniform(2.5, 4.5 if small else 14.0), "coef": [(m, rng.uniform(0.02, 0.12), rng.uniform(0, 2 * np.pi)) for m in (2, 3, 5, 7)]}
    n_beams = int(rng.choice([90, 180, 360] if small else [90, 360, 720, 1080]))
    P = int(rng.choice([1, 2, 3] if small else [1, 3, 6]))
    trunc_ray = float(rng.choice([0.0, 0.0, 2.0]))
    trunc_range = float(rng.choice([0.0, 0.0, kind["R"] * 0.8]))
    seq_ray = int(rng.choice([1, 2]))
    bf_waves = int(rng.choice([1, 2]))
    bf_mode = 0
    base = np.array([rng.uniform(-0.3, 0.3) * kind["R"], rng.uniform(-0.3, 0.3) * kind["R"], rng.uniform(-np.pi, np.pi)])
    more = {} if l2_max is None else {"l2_max": float(l2_max)}
    opts = O.default_options(particles=P, seed=seed + 1, truncated_ray=trunc_ray, truncated_range=trunc_range, **more)
    pf = O.PF(opts)
    scan0 = random_room_scan(rng, base, n_beams, kind)
    pose0 = O.se2(*base)
    pf.set_prior(pose0)
    assert pf.update(scan0, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, truncated_ray=trunc_ray, truncated_range=trunc_range, sequential_raycast=seq_ray,
                                     brushfire_waves=bf_waves, brushfire_mode=bf_mode, dm_patch_capacity=1024, occ_patch_capacity=1024, **more))
    ctx.init(scan0, pose0)
    for k in range(4):
        truth = base + np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3)])
        scan = random_room_scan(rng, truth, n_beams, kind)
        poses = np.stack([O.se2(*(truth + rng.normal(0, [0.02, 0.02, 0.01]))) for _ in range(P)])
        pf.set_poses(poses)
        pf.stage_set_scan(scan)
        pf.stage_update_maps()
        ctx.set_poses(poses)
        ctx.update_maps(scan)
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"seed {seed} scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"seed {seed} scan {k} dm p{i}")
        if P > 1 and k == 1:
            idx = rng.integers(0, P, P).astype(np.int32)
            pf.stage_resample_with(idx)
            ctx.resample(idx)
        base = truth
    c = ctx.counters()
    ctx.close()
    return c
