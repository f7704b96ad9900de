"""-m gpu: stage-wise ("teacher-forced") parity of the HIP path against the CPU oracle, through the C-ABI.

  * maps (uint16 occupancy counters, DM sqdist / obstacle offset / flags, Container masks, patch sets):
    BIT-EXACT for identical poses;
  * scan-match poses: |d| <= POSE_TOL (fp64; differences come from libm-vs-ocml trig and the reduction
    order of the 1080-row normal equations), identical Gauss-Newton iteration counts;
  * log-likelihood: relative 1e-9.
"""
import numpy as np
import pytest

import _oracle as O
from _cmp import DM_FIELDS, OCC_FIELDS, assert_maps_equal

pytestmark = pytest.mark.gpu

POSE_TOL = 1e-8      # metres / radians, per solve, when no GN decision flips
LL_RTOL = 1e-9


@pytest.fixture(scope="module")
def F():
    import iris_lama_amd.ffi as f
    if f.device_count() < 1:
        pytest.fail("no HIP device visible: the gpu tests need the MI355X box (there is no CPU fallback)")
    return f


def _perturbed(rng, base, P, sxy=0.03, sth=0.01):
    out = np.zeros((P, 4))
    for i in range(P):
        out[i] = O.se2_mul(base, O.se2(rng.normal(0, sxy), rng.normal(0, sxy), rng.normal(0, sth)))
    return out


def _angle(p):
    return np.arctan2(p[..., 1], p[..., 0])


@pytest.mark.parametrize("P,steps", [(8, 12), (40, 4)])
def test_stagewise_parity_corridor(F, P, steps):
    pts, odom, truth = F.corridor_log(steps, 1080)
    rng = np.random.default_rng(5)
    opts = O.default_options(particles=P, seed=7)
    pf = O.PF(opts)
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0)

    ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
    ctx.init(pts[0], pose0)
    for i in (0, P - 1):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"init occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"init dm p{i}")

    flips = 0
    for k in range(1, steps + 1):
        base = O.se2(*truth[k])
        start = _perturbed(rng, base, P)
        # ---- stage (i): scan match on identical maps + identical start poses
        pf.set_poses(start)
        pf.set_weights(w=np.zeros(P), ws=np.zeros(P))
        pf.stage_set_scan(pts[k])
        pf.stage_scan_match()
        o_poses = pf.poses()
        o_ll = pf.weights()[0]
        o_it = np.array([pf.counters(i)["iterations"] for i in range(P)])
        ctx.set_poses(start)
        g_poses, g_ll, g_it = ctx.scan_match(pts[k])
        same = g_it == o_it
        flips += int((~same).sum())
        dxy = np.abs(g_poses[:, 2:] - o_poses[:, 2:]).max(axis=1)
        dth = np.abs(_angle(g_poses) - _angle(o_poses))
        assert (dxy[same] <= POSE_TOL).all() and (dth[same] <= POSE_TOL).all(), (k, dxy, dth)
        assert np.allclose(g_ll[same], o_ll[same], rtol=LL_RTOL, atol=0), (k, g_ll, o_ll)
        # a flipped GN decision moves the pose by at most the eps2 scale
        assert (dxy[~same] <= 5e-4).all() and (dth[~same] <= 5e-4).all()

        # ---- stage (iii): resample with a fixed index vector every other step
        if k % 2 == 0:
            idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
            pf.stage_resample_with(idx)
            ctx.resample(idx)
            o_poses = pf.poses()
            assert np.array_equal(ctx.get_poses(), g_poses[idx])      # the pose travels with the particle

        # ---- stage (ii): map update with teacher-forced (oracle) poses: bit-exact
        ctx.set_poses(o_poses)
        pf.stage_update_maps()
        ctx.update_maps(pts[k])
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    assert flips <= P * steps // 10, f"too many Gauss-Newton decision flips: {flips}"
    c = ctx.counters()
    assert c["launches_scan_match"] == steps and c["launches_update_maps"] == steps + 1
    print("counters", c, "GN flips", flips)
    ctx.close()


def test_match_batch_matches_oracle_loglik(F):
    pts, odom, truth = F.corridor_log(1, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=1, seed=1))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=1))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(1)
    poses = _perturbed(rng, O.se2(*truth[1]), 64, 0.2, 0.05)
    got = ctx.match_batch(0, pts[1], poses)
    want = np.array([O.loglik(pf.dm(0), pts[1], q) for q in poses])
    assert np.allclose(got, want, rtol=LL_RTOL, atol=0)
    ctx.close()
