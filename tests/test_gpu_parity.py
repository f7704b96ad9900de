"""-m gpu: stage-wise ("teacher-forced") parity of the HIP path against the CPU oracle, through the C-ABI.

  * maps (uint16 occupancy counters, DM sqdist / obstacle offset / flags, Container masks, patch sets):
    BIT-EXACT for identical poses;
  * scan-match poses: |d| <= POSE_TOL (fp64; differences come from libm-vs-ocml trig and the reduction
    order of the 1080-row normal equations), identical Gauss-Newton iteration counts;
  * log-likelihood: relative 1e-9.
"""
import ctypes as C
import os

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


def _perturbed(rng, base, P, sxy=0.03, sth=0.01, drifted=0):
    """P poses around `base`; the last `drifted` of them with four times the noise (the particles whose brushfire chains are long)"""
    out = np.zeros((P, 4))
    for i in range(P):
        k = 4.0 if i >= P - drifted else 1.0
        out[i] = O.se2_mul(base, O.se2(rng.normal(0, k * sxy), rng.normal(0, k * sxy), rng.normal(0, k * sth)))
    return out


def _angle(p):
    return np.arctan2(p[..., 1], p[..., 0])


@pytest.mark.parametrize("P,steps,seq_ray,bf_waves,bf_mode", [(8, 12, 2, 0, 0), (8, 12, 1, 1, 0), (40, 4, 0, 2, 0), (40, 4, 0, 1, 0),
                                                              (300, 3, 0, 0, 0)])
def test_stagewise_parity_corridor(F, P, steps, seq_ray, bf_waves, bf_mode):
    """Both ray-cast forms and both wave layouts of the exact brushfire; P = 300 is one of BASELINE's particle counts."""
    _stagewise(F, P, steps, seq_ray, bf_waves, bf_mode)


@pytest.mark.parametrize("route", ["1,0,110,8", "1,0,0,256", "1,0,0,4", "1000000,0,150,64"])
def test_long_chains_are_routed_to_the_big_queue_stage(F, monkeypatch, route):
    """k_bf_route: particles with many more obstacle events than the pool's mean are the long brushfire chains; they run in the
    big-queue stage on a second stream beside the first stage, which skips them.  Forced on 40 particles: (a) threshold 110 % of the
    mean, 8 places; (b) everybody over the threshold; (c) the same with 4 places: they go to the longest chains; (d) routing off --
    bit-exact maps in all four."""
    monkeypatch.setenv("LAMA_HIP_BF_ROUTE", route)
    c = _stagewise(F, 40, 6, 0, 2, 0)
    if route.startswith("1,"):
        assert c["brushfire_routed"] > 0, c
    else:
        assert c["brushfire_routed"] == 0, c


def test_routing_and_early_lane_change_nothing_at_3000_particles(F, monkeypatch):
    """BASELINE config 2's pool free-running for 25 scans -- long enough for a few particles to drift and grow brushfire chains
    three to seven times the pool's mean -- with the default routing / early lane and with both switched off: poses, weights and the
    device-side checksums of every particle's maps are identical, and the counters show that the lanes were used."""
    P, steps = 3000, 25
    pts, odom, _ = F.corridor_log(steps, 1080)
    out = []
    for route in (None, "64,48,150,0,0"):
        if route is None:
            monkeypatch.delenv("LAMA_HIP_BF_ROUTE", raising=False)
        else:
            monkeypatch.setenv("LAMA_HIP_BF_ROUTE", route)
        pf = F.PFSlam2D(F.pf_options(particles=P, seed=42))
        pf.set_prior(*odom[0])
        for k in range(steps + 1):
            pf.update(pts[k], odom[k], float(k))
        c = pf.hip_context()
        out.append(dict(poses=pf.poses().copy(), w=pf.weights()[0].copy(), dm=c.map_checksums(F.MAP_DISTANCE), occ=c.map_checksums(F.MAP_OCCUPANCY),
                        ctr=c.counters()))
        pf.close()
    a, b = out
    assert a["ctr"]["brushfire_routed"] > 0 and a["ctr"]["brushfire_early"] > 0, a["ctr"]
    assert b["ctr"]["brushfire_routed"] == 0 and b["ctr"]["brushfire_early"] == 0, b["ctr"]
    for k in ("poses", "w", "dm", "occ"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("cap", [0, 24])
def test_drifted_poses_long_chains_default_routing(F, cap, monkeypatch):
    """72 particles mapped WITHOUT scan matching, six of them from poses off by up to ~25 cm / 9 degrees: those re-draw walls and
    remove others in every update (long raise waves, queues that outgrow the first brushfire stage), and the default routing (>= 64
    particles) sends them -- the particles with the most obstacle events -- to the big-queue stage: maps bit-exact, and the counters
    show that the path ran."""
    P, steps = 72, 6
    pts, odom, truth = F.corridor_log(steps, 1080)
    rng = np.random.default_rng(11)
    pf = O.PF(O.default_options(particles=P, seed=7))
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0)
    monkeypatch.setenv("LAMA_HIP_BF_ROUTE", "64,48,150,64,1")      # the defaults, but the early lane from 64 particles on (default: 1024)
    # cap = 24: arenas that the run outgrows several times (updates that are aborted in their allocation phase and repeated after
    # growth) while particles sit in the routed / early lanes
    ctx = F.HipContext(F.default_cfg(particles=P, profile=1, dm_patch_capacity=cap, occ_patch_capacity=cap))
    ctx.init(pts[0], pose0)
    for k in range(1, steps + 1):
        poses = _perturbed(rng, O.se2(*truth[k]), P, 0.02, 0.006, drifted=6)
        pf.set_poses(poses)
        pf.stage_set_scan(pts[k])
        pf.stage_update_maps()
        ctx.set_poses(poses)
        ctx.update_maps(pts[k])
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    c = ctx.counters()
    ctx.close()
    print("counters", c)
    assert c["brushfire_routed"] > 0 and c["bf_longest_chain_sum"] > 2 * c["bf_cells"] / P, c
    assert c["brushfire_early"] > 0, c          # routed in one update, early lane in the next (no resample in between)
    if cap:
        assert c["arena_growths"] > 0, c          # (with the default floor of 128 patches the drifted particles may grow once, too)


def test_early_lane_and_routing_at_1536_particles_against_the_oracle(F, monkeypatch):
    """a DIRECT oracle comparison of the routed / early-lane path at its default thresholds (pools of 1,024 particles
    and more; no environment override).  1,536 particles, teacher forced, ten scans with scan matching (its log-likelihoods feed
    the early lane) and a resample in the middle; eight particles start every scan far off (the long chains).  After every scan
    the device's checksum of EVERY particle's two maps equals the oracle's, a sample is compared cell by cell, and the counters
    show that the big-queue stage and the early lane ran."""
    monkeypatch.delenv("LAMA_HIP_BF_ROUTE", raising=False)
    P, steps, drifted = 1536, 10, 8
    pts, odom, truth = F.corridor_log(steps, 1080)
    rng = np.random.default_rng(17)
    pf = O.PF(O.default_options(particles=P, seed=7, threads=min(64, os.cpu_count() or 1)))
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
    ctx.init(pts[0], pose0)
    for k in range(1, steps + 1):
        start = _perturbed(rng, O.se2(*truth[k]), P, 0.03, 0.01, drifted)
        pf.set_poses(start)
        pf.set_weights(w=np.zeros(P), ws=np.zeros(P))
        pf.stage_set_scan(pts[k])
        pf.stage_scan_match()
        ctx.set_poses(start)
        g_poses, g_ll, g_it = ctx.scan_match(pts[k])
        o_poses = pf.poses()
        same = g_it == np.array([pf.counters(i)["iterations"] for i in range(P)])
        assert same.mean() > 0.999 and np.abs(g_poses[same] - o_poses[same]).max() <= POSE_TOL
        if k == 5:
            idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
            pf.stage_resample_with(idx)
            ctx.resample(idx)
            o_poses = pf.poses()
        # the drifted particles map from where they started (far off: the scan disagrees with their map, they re-draw it)
        o_poses[P - drifted:] = start[P - drifted:] if k != 5 else o_poses[P - drifted:]
        pf.set_poses(o_poses)
        ctx.set_poses(o_poses)
        pf.stage_update_maps()
        ctx.update_maps(pts[k])
        assert np.array_equal(ctx.map_checksums(F.MAP_DISTANCE), pf.map_checksums(0)), k
        assert np.array_equal(ctx.map_checksums(F.MAP_OCCUPANCY), pf.map_checksums(1)), k
    for i in (0, 1, 700, P - drifted - 1, P - drifted, P - 2, P - 1):
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"dm p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"occ p{i}")
    c = ctx.counters()
    ctx.close()
    assert c["brushfire_routed"] > 0 and c["brushfire_early"] > 0, c
    assert c["bf_longest_chain_sum"] > 2 * c["bf_cells"] / P, c


def _stagewise(F, P, steps, seq_ray, bf_waves, bf_mode, sxy=0.03, sth=0.01, drifted=0, cfg_extra=None, **map_opts):
    pts, odom, truth = F.corridor_log(steps, 1080)
    rng = np.random.default_rng(5)
    opts = O.default_options(particles=P, seed=7, **map_opts)
    pf = O.PF(opts)
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0)

    ctx = F.HipContext(F.default_cfg(particles=P, profile=1, sequential_raycast=seq_ray, brushfire_waves=bf_waves, brushfire_mode=bf_mode, **map_opts, **(cfg_extra or {})))
    ctx.init(pts[0], pose0)
    for i in (0, P - 1):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"init occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"init dm p{i}")

    flips = 0
    for k in range(1, steps + 1):
        base = O.se2(*truth[k])
        start = _perturbed(rng, base, P, sxy, sth, drifted)
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
    assert flips == 0, f"Gauss-Newton decision flips against the oracle: {flips}"     # identical iteration counts on these seeds
    c = ctx.counters()
    assert c["launches_scan_match"] == steps and c["launches_update_maps"] == steps + 1
    print("counters", c, "GN flips", flips)
    ctx.close()
    return c


@pytest.mark.parametrize("l2_max,bf_waves", [(8.0, 2), (12.75, 2), (6.6, 1)])
def test_stagewise_parity_l2_max_beyond_127_cells(F, l2_max, bf_waves):
    """VERDICT r05 item 7 -- an option limit lifted: l2_max of 160, 255 and 132 cells (the reference's uint16_t sqdist ends at 255^2).
    Such a context runs liblama_hip_wide.so: the same sources with a 4-byte distance plane and 9-bit obstacle offsets in the queue
    entries (csrc/lama_dev.h), picked by the reach of the map when the context is created.  The three stages -- scan match, resampling
    (clones of 4-byte planes), map update with an exact brushfire that now floods the whole corridor and 8 - 13 m around it -- against
    the oracle, bit for bit, and the device-side checksum in the wide packing."""
    assert F.needs_wide(l2_max, 0.05) and not F.needs_wide(6.35, 0.05)
    P, steps = 3, 3
    c = _stagewise(F, P, steps, 2, bf_waves, 0, l2_max=l2_max, cfg_extra=dict(queue_capacity=1 << 20))
    assert c["launches_update_maps"] == steps + 1
    # the default library refuses the same configuration when it is called directly
    h = C.c_void_p()
    cfg = F.default_cfg(particles=2, l2_max=l2_max)
    assert F.hip_lib().lama_hip_ctx_create(C.byref(cfg), C.byref(h)) == -1 and not h


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("LAMA_STRESS_SEEDS_WIDE", "6")))))
def test_randomized_rooms_maps_bit_exact_wide_library(F, seed):
    """The randomised rooms (2.5 - 14 m, 90 - 1080 beams, truncation options, both ray-casts, one- and two-wave brushfire, a resample
    in between) with a distance map that reaches 6.4 - 12.75 m: every scan floods the whole room; raise waves too (walls are re-drawn
    a cell off)."""
    l2 = [6.4, 7.0, 9.0, 10.0, 12.0, 12.75][seed % 6]
    from _stress import random_rooms_case
    random_rooms_case(F, 50 + seed, l2_max=l2)


def test_host_classes_pick_the_wide_library(F):
    """lama::Slam2D and lama::PFSlam2D with l2_max = 7 m (140 cells) bind liblama_hip_wide.so by themselves and run free next to the
    oracle; particle blobs (4-byte distance plane) travel between two wide contexts; Loc2D builds its 7 m map on the host and
    uploads it to a wide context."""
    steps = 6
    pts, odom, truth = F.corridor_log(steps, 1080)
    h = F.Slam2D(l2_max=7.0)
    assert h.engine_origin().endswith("liblama_hip_wide.so")
    o = O.Slam(l2_max=7.0)
    h.set_pose(*odom[0])
    o.set_pose(O.se2(*odom[0]))
    for k in range(steps + 1):
        assert h.update(pts[k], odom[k], float(k)) == o.update(pts[k], O.se2(*odom[k]), float(k))
        assert np.abs(h.pose() - o.pose()).max() < 1e-7, k
    ctx = h.hip_context()
    assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), o.occ().dump(), OCC_FIELDS, "slam occ (wide)")
    assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), o.dm().dump(), DM_FIELDS, "slam dm (wide)")
    h.close()

    P = 6
    hp = F.PFSlam2D(F.pf_options(particles=P, seed=11, l2_max=7.0))
    assert hp.engine_origin().endswith("liblama_hip_wide.so")
    op = O.PF(O.default_options(particles=P, seed=11, l2_max=7.0))
    hp.set_prior(*odom[0])
    op.set_prior(O.se2(*odom[0]))
    for k in range(4):
        assert hp.update(pts[k], odom[k], float(k)) == op.update(pts[k], O.se2(*odom[k]), float(k))
    ctx = hp.hip_context()
    assert np.array_equal(ctx.map_checksums(F.MAP_OCCUPANCY), op.map_checksums(1))
    assert np.array_equal(ctx.map_checksums(F.MAP_DISTANCE), op.map_checksums(2))          # (2: the wide library's packing)
    # a particle's blob into a second wide context and back out: identical maps
    other = F.HipContext(F.default_cfg(particles=2, l2_max=7.0))
    other.init(pts[0], O.se2(*odom[0]))
    import torch
    n = ctx.export_bytes(3)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    ctx.export_particle(3, buf.data_ptr(), n)
    other.import_particle(1, buf.data_ptr(), n)
    assert other.map_checksums(F.MAP_DISTANCE)[1] == ctx.map_checksums(F.MAP_DISTANCE)[3]
    assert_maps_equal(other.download_map(1, F.MAP_DISTANCE), ctx.download_map(3, F.MAP_DISTANCE), DM_FIELDS, "imported particle (wide)")
    other.close()
    hp.close()

    from _worlds import corridor_obstacles
    obst = corridor_obstacles()
    ol = O.Loc(l2_max=7.0)
    dm = ol.dm()
    for x, y in obst:
        c = O.w2m([x, y, 0.0])
        dm.add(int(c[0]), int(c[1]))
    dm.update()
    hl = F.Loc2D(l2_max=7.0)
    hl.set_obstacles_world(obst)
    assert hl.engine_origin().endswith("liblama_hip_wide.so")
    assert_maps_equal(hl.hip_context().download_map(0, F.MAP_DISTANCE), dm.dump(), DM_FIELDS, "Loc2D 7 m distance map after Init")
    start = truth[0] + np.array([0.05, -0.04, 0.01])
    ol.set_pose(O.se2(*start))
    hl.set_pose(*start)
    for k in range(4):
        assert ol.update(pts[k], O.se2(*odom[k]), float(k), force=(k == 0)) == hl.update(pts[k], odom[k], float(k), force=(k == 0))
        assert np.abs(ol.pose() - hl.pose()).max() < 1e-7, k
    hl.close()


def test_clones_keep_distance_patches_far_beyond_the_hits(F):
    """with l2_max above 64 cells (3.2 m at 0.05 m) the brushfire allocates distance-map patches more than two patches
    beyond the scan's reach; a resample copies only the directory rows of the mapped box, which therefore has to include them.
    A round room of 6 m radius (hits in every direction, also along y: the rows are what the copy cuts), l2_max = 4.2 m (84 cells,
    guard radius 3 patches), a resample between two updates, every map of every clone against the oracle."""
    n, radius, P = 1080, 6.0, 4
    ang = np.deg2rad(-135.0 + 0.25 * np.arange(n))
    rng = np.random.default_rng(11)

    def scan():
        r = radius + rng.normal(0, 0.01, n)
        return np.stack([r * np.cos(ang), r * np.sin(ang), np.zeros(n)], axis=1)

    pose0 = O.se2(0.3, -0.2, 0.1)
    pf = O.PF(O.default_options(particles=P, seed=1, l2_max=4.2))
    pf.set_prior(pose0)
    pts = scan()
    pf.update(pts, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, l2_max=4.2))
    ctx.init(pts, pose0)
    for k, idx in ((1, None), (2, np.array([0, 0, 2, 2], dtype=np.int32)), (3, np.array([1, 1, 1, 3], dtype=np.int32))):
        pts = scan()
        poses = np.stack([O.se2_mul(pose0, O.se2(0.04 * i + 0.01 * k, -0.03 * i, 0.01 * i)) for i in range(P)])
        pf.set_poses(poses)
        ctx.set_poses(poses)
        if idx is not None:
            pf.stage_resample_with(idx)
            ctx.resample(idx)
            assert np.array_equal(ctx.get_poses(), pf.poses())
        pf.stage_set_scan(pts)
        pf.stage_update_maps()
        ctx.update_maps(pts)
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    c = ctx.counters()
    assert c["resample_clones"] >= 4
    ctx.close()


def test_parallel_raycast_with_truncation_and_mounted_sensor(F, monkeypatch):
    """The parallel ray-cast counts the free-cell visits per occupancy patch (k_ray_patches); against the oracle with truncation
    and a mounted sensor so that start cells differ per beam.  No environment variable may change what runs: the second pass sets
    the switches earlier builds honoured and must report (and compute) the same thing."""
    P, steps = 6, 8
    pts, odom, truth = F.corridor_log(steps, 1080)
    origin, quat = np.array([0.15, -0.1, 0.2]), np.array([np.cos(0.1), 0.0, 0.0, np.sin(0.1)])
    rng = np.random.default_rng(11)
    for mode in ("0", "1"):
        if mode == "1":
            for name in ("LAMA_HIP_RAY_MODE", "LAMA_HIP_BRUSHFIRE_MODE", "LAMA_HIP_SEQUENTIAL_RAYCAST", "LAMA_HIP_BF_CACHE"):
                monkeypatch.setenv(name, "1")
        pf = O.PF(O.default_options(particles=P, seed=7, truncated_ray=4.0, truncated_range=9.0))
        pose0 = O.se2(*odom[0])
        pf.set_prior(pose0)
        assert pf.update(pts[0], pose0, 0.0, origin=origin, quat=quat)
        ctx = F.HipContext(F.default_cfg(particles=P, truncated_ray=4.0, truncated_range=9.0))
        ctx.init(pts[0], pose0, origin=origin, quat=quat)
        for k in range(1, steps + 1):
            poses = _perturbed(rng, O.se2(*truth[k]), P, 0.02, 0.005)
            pf.set_poses(poses)
            pf.stage_set_scan(pts[k], origin=origin, quat=quat)
            pf.stage_update_maps()
            ctx.set_poses(poses)
            ctx.update_maps(pts[k], origin=origin, quat=quat)
            for i in range(P):
                assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"mode {mode} scan {k} occ p{i}")
                assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"mode {mode} scan {k} dm p{i}")
        c = ctx.counters()
        assert c["brushfire_mode"] == 0 and c["sequential_raycast_scans"] == 0 and c["parallel_raycast_scans"] == steps + 1, c
        ctx.close()


@pytest.mark.parametrize("cap,seq_ray", [(96, 0), (8, 0), (8, 1)])
def test_patch_arenas_grow_on_demand(F, cap, seq_ray):
    """The reference's maps allocate patches without bound (src/sdm/map.cpp:400-411); the device arenas start small here and
    must be doubled on the way -- maps stay bit-identical to the oracle's, the growth counter moves, resample() still works.
    cap = 8: a SINGLE scan (the first one needs ~55 occupancy / ~70 distance-map patches) asks for far more than is free: the
    update's allocation phase reports it before any cell is modified, the arenas are doubled and the update runs again
    (both ray-cast forms)."""
    P, steps = 4, 14
    pts, odom, truth = F.corridor_log(steps, 1080)
    pf = O.PF(O.default_options(particles=P, seed=3))
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, dm_patch_capacity=cap, occ_patch_capacity=cap, sequential_raycast=seq_ray))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(11)
    for k in range(1, steps + 1):
        poses = _perturbed(rng, O.se2(*truth[k]), P)
        if k % 3 == 0:
            idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
            pf.stage_resample_with(idx)
            ctx.resample(idx)
        pf.set_poses(poses)
        pf.stage_set_scan(pts[k])
        pf.stage_update_maps()
        ctx.set_poses(poses)
        ctx.update_maps(pts[k])
    for i in range(P):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"dm p{i}")
    c = ctx.counters()
    assert c["arena_growths"] >= 1, c
    assert c["dm_patches"] > cap * P / 2
    ctx.close()


def test_map_window_follows_the_robot(F):
    """The reference's maps have no extent.  The device window has one (here 40 patches = 64 m) but it is re-centred when a scan
    could leave it.  The corridor scans are integrated from poses that also drift sideways by 1 m per scan (not a consistent
    world, but the oracle integrates the same thing: pose + sensor reach leaves the window centred on the first pose after a few
    scans, the window moves, the maps stay bit-identical to the oracle's."""
    P, steps = 3, 44
    pts, odom, truth = F.corridor_log(steps, 1080)
    pf = O.PF(O.default_options(particles=P, seed=3))
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, window_patches=40))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(12)
    for k in range(1, steps + 1):
        poses = _perturbed(rng, O.se2(truth[k][0], truth[k][1] + 1.0 * k, truth[k][2]), P)
        if k % 5 == 0:
            idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
            pf.stage_resample_with(idx)
            ctx.resample(idx)
        pf.set_poses(poses)
        pf.stage_set_scan(pts[k])
        pf.stage_update_maps()
        ctx.set_poses(poses)
        ctx.update_maps(pts[k])
        if k in (20, 44):
            for i in range(P):
                assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
                assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    c = ctx.counters()
    assert c["window_shifts"] >= 1, c
    # scan matching against the shifted window still sees the map
    g_poses, g_ll, g_it = ctx.scan_match(pts[steps])
    pf.stage_set_scan(pts[steps]); pf.set_weights(w=np.zeros(P), ws=np.zeros(P)); pf.stage_scan_match()
    assert np.abs(g_poses - pf.poses()).max() <= 1e-6        # (a smeared map: the solver wanders, both sides alike)
    ctx.close()


def test_map_extent_is_not_limited_by_the_window_450m(F):
    """the reference allocates patches anywhere (src/sdm/map.cpp:400-411).  One particle is driven 450 m down a
    generated corridor (460 m x 4 m, pillars every 4 m, a 1080-beam 270-degree scanner with 30 m range: beams without a return are
    dropped, so the scans are ragged): the device window starts at 128 patches (204.8 m), must GROW past the old limit of 248
    patches, the arenas grow with the map, and the maps stay bit-identical to the oracle's."""
    from _worlds import long_corridor_segments, segment_world_scan
    P, dist, step = 1, 450.0, 0.75
    n_scans = int(dist / step)
    segs = long_corridor_segments()
    rng = np.random.default_rng(77)

    def scan_at(k):
        x, y, yaw = 2.0 + step * k, 2.0 + 0.3 * np.sin(0.05 * k), 0.04 * np.sin(0.11 * k)
        return O.se2(x, y, yaw), segment_world_scan(segs, x, y, yaw, noise=rng.normal(0.0, 0.01, 1080))
    pose0, scan0 = scan_at(0)
    pf = O.PF(O.default_options(particles=P, seed=5))
    pf.set_prior(pose0)
    assert pf.update(scan0, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P))
    ctx.init(scan0, pose0)
    for k in range(1, n_scans + 1):
        pose, scan = scan_at(k)
        pf.set_poses(pose[None, :]); pf.stage_set_scan(scan); pf.stage_update_maps()
        ctx.set_poses(pose[None, :]); ctx.update_maps(scan)
        if k in (150, 400):          # on the way
            assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), pf.occ(0).dump(), OCC_FIELDS, f"scan {k} occ")
    c = ctx.counters()
    assert c["window_growths"] >= 1 and c["window_patches"] > 280, c          # 450 m + sensor reach = more than 280 patches of 1.6 m
    assert c["arena_growths"] >= 1
    assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), pf.occ(0).dump(), OCC_FIELDS, "occ after 450 m")
    assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), pf.dm(0).dump(), DM_FIELDS, "dm after 450 m")
    # scan matching on the grown window sees the map
    pose, scan = scan_at(n_scans)
    g_poses, g_ll, g_it = ctx.scan_match(scan)
    pf.stage_set_scan(scan); pf.set_weights(w=np.zeros(P), ws=np.zeros(P)); pf.stage_scan_match()
    assert np.abs(g_poses - pf.poses()).max() <= 1e-6
    ctx.close()


def test_particle_blob_between_contexts_with_different_windows(F):
    """Every shard's window follows its own particles, so a shipped particle may come from a window that sits elsewhere: the blob
    carries its origin and the importer translates the directories -- the imported particle's maps are the exported ones."""
    import torch
    pts, odom, truth = F.corridor_log(3, 1080)
    a = F.HipContext(F.default_cfg(particles=2, window_patches=64))
    b = F.HipContext(F.default_cfg(particles=2, window_patches=64))
    a.init(pts[0], O.se2(*odom[0]))
    b.init(pts[3], O.se2(truth[3][0] + 9.0, truth[3][1] + 5.0, 0.4))          # another place: another window origin
    a.set_poses(np.tile(O.se2(*truth[1]), (2, 1)))
    a.update_maps(pts[1])
    n = a.export_bytes(1)
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    a.export_particle(1, buf.data_ptr(), n)
    b.import_particle(0, buf.data_ptr(), n)
    for kind, fields in ((F.MAP_DISTANCE, DM_FIELDS), (F.MAP_OCCUPANCY, OCC_FIELDS)):
        assert_maps_equal(b.download_map(0, kind), a.download_map(1, kind), fields, f"kind {kind}")
    assert np.array_equal(b.get_poses()[0], a.get_poses()[1])
    a.close(); b.close()


def test_import_reads_what_the_last_copy_wrote_into_a_reused_buffer(F):
    """The bug of round 6: blobs arrive through copies that somebody else's stream carries out, into a buffer whose
    address the caller's allocator reuses -- the import must see the bytes of the LAST copy, never cached lines of the blob that lay
    there before.  Two different blobs of one particle (after scan 1 and after scan 3: the second one holds more map) alternate in
    ONE device buffer, each written by torch's own copy (not the library's stream), 60 imports: the distance and occupancy maps of
    the receiving slot must be the exporter's of that moment, every time.  (In ONE process the library of before the fix passes this
    too -- the stale lines needed several processes sharing the device, tools/flake_partition.sh -- so this test states the contract;
    the many-processes partition test is what guards it.)"""
    import torch
    pts, odom, truth = F.corridor_log(3, 1080)
    a = F.HipContext(F.default_cfg(particles=2))
    b = F.HipContext(F.default_cfg(particles=2))
    a.init(pts[0], O.se2(*odom[0]))
    b.init(pts[0], O.se2(*odom[0]))
    blobs, want = [], []
    for k in (1, 2, 3):
        a.set_poses(np.tile(O.se2(*truth[k]), (2, 1)))
        a.update_maps(pts[k])
        if k in (1, 3):
            n = a.export_bytes(0)
            buf = torch.empty(n, dtype=torch.uint8, device="cuda")
            a.export_particle(0, buf.data_ptr(), n)
            blobs.append(buf.cpu().pin_memory())
            want.append((a.map_checksums(F.MAP_DISTANCE)[0], a.map_checksums(F.MAP_OCCUPANCY)[0]))
    assert want[0] != want[1]
    dev = torch.empty(max(len(h) for h in blobs), dtype=torch.uint8, device="cuda")      # ONE buffer, reused for every import
    side = torch.cuda.Stream()
    for rep in range(30):
        for j in (0, 1):
            h = blobs[j]
            if rep % 2:                                   # a copy stream of its own (what a communication library does) ...
                with torch.cuda.stream(side):
                    dev[:len(h)].copy_(h, non_blocking=True)
            else:                                         # ... or the null stream
                dev[:len(h)].copy_(h)
            torch.cuda.synchronize()
            b.import_particle(1, dev.data_ptr(), len(h))
            got = (b.map_checksums(F.MAP_DISTANCE)[1], b.map_checksums(F.MAP_OCCUPANCY)[1])
            assert got == want[j], (rep, j, got, want[j])
    a.close(); b.close()


def test_batched_import_grows_the_receiving_arenas(F):
    """A shard whose arenas are still small receives particles from a shard whose arenas have grown: the batched import enlarges
    the receiver first (all slots keep their maps), several slots take the same blob, and the receiver carries on updating."""
    import torch
    pts, odom, truth = F.corridor_log(4, 1080)
    pose0 = O.se2(*odom[0])
    P = 3
    pf = O.PF(O.default_options(particles=P, seed=2))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    a = F.HipContext(F.default_cfg(particles=P))                                             # default arenas
    b = F.HipContext(F.default_cfg(particles=P, dm_patch_capacity=8, occ_patch_capacity=8))  # grows on demand
    a.init(pts[0], pose0)
    b.init(pts[0], pose0)
    rng = np.random.default_rng(3)
    for k in (1, 2):
        start = np.stack([O.se2_mul(O.se2(*truth[k]), O.se2(*rng.normal(0, [0.05, 0.05, 0.02]))) for _ in range(P)])
        pf.set_poses(start); pf.stage_set_scan(pts[k]); pf.stage_update_maps()
        a.set_poses(start); a.update_maps(pts[k])
    src = np.array([2, 0], dtype=np.uint32)
    sizes = a.export_sizes(src)
    bufs = [torch.empty(int(n), dtype=torch.uint8, device="cuda") for n in sizes]
    assert np.array_equal(a.export_particles(src, [x.data_ptr() for x in bufs], sizes), sizes)
    b.import_particles([0, 1, 2], [bufs[0].data_ptr(), bufs[0].data_ptr(), bufs[1].data_ptr()], [sizes[0], sizes[0], sizes[1]])
    pf.stage_resample_with(np.array([2, 2, 0], dtype=np.int32))
    assert np.array_equal(b.get_poses(), pf.poses())
    for rnd in range(2):
        for i in range(P):
            assert_maps_equal(b.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"round {rnd} occ p{i}")
            assert_maps_equal(b.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"round {rnd} dm p{i}")
        if rnd == 0:
            nxt = np.stack([O.se2_mul(O.se2(*truth[3]), O.se2(*rng.normal(0, [0.05, 0.05, 0.02]))) for _ in range(P)])
            pf.set_poses(nxt); pf.stage_set_scan(pts[3]); pf.stage_update_maps()
            b.set_poses(nxt); b.update_maps(pts[3])
    a.close(); b.close()


def test_visited_counter_wrap_inside_a_scan(F):
    """uint16 `visited` wraps silently in the reference (src/sdm/frequency_occupancy_map.cpp:65-74).  The parallel ray-cast adds
    a scan's visits in any order, which is only the same thing while no counter wraps INSIDE a scan: 2000 beams down one corridor
    of cells, scan after scan from the same pose, reach 65536 visits in the 33rd scan -- the device must notice (it keeps a bound
    of the largest counter), cast those scans beam by beam, and stay bit-identical to the oracle through the wrap."""
    P, n = 2, 2000
    ang = np.linspace(-0.004, 0.004, n)
    scan = np.stack([6.0 * np.cos(ang), 6.0 * np.sin(ang), np.zeros(n)], axis=1)
    pose0 = O.se2(1.0, 1.0, 0.3)
    pf = O.PF(O.default_options(particles=P, seed=5))
    pf.set_prior(pose0)
    assert pf.update(scan, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P))
    ctx.init(scan, pose0)
    poses = np.stack([pose0, O.se2(1.0, 1.0, 0.3)])
    for k in range(1, 36):
        pf.set_poses(poses)
        pf.stage_set_scan(scan)
        pf.stage_update_maps()
        ctx.set_poses(poses)
        ctx.update_maps(scan)
        if k in (30, 31, 32, 33, 35):
            for i in range(P):
                assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
                assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    occ = ctx.download_map(0, F.MAP_OCCUPANCY)
    c = ctx.counters()
    assert c["wrap_guard_scans"] >= 1, c          # the guard fired ...
    assert c["wrap_guard_scans"] <= 6, c          # ... and only around the wrap, not from the first scan on
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


def test_free_running_host_class_tracks_oracle_and_truth(F):
    """Full lama::PFSlam2D::update on the GPU, free running on the corridor log, next to the oracle with the same
    seed.  GPU and CPU trig differ in the last ulp, so after the first scan the trajectories are not bit-equal;
    they must stay within millimetres of each other and of the ground truth."""
    P, steps = 30, 40
    pts, odom, truth = F.corridor_log(steps, 1080)
    h = F.PFSlam2D(F.pf_options(particles=P, seed=42))
    assert h.engine_origin().endswith("liblama_hip.so")
    o = O.PF(O.default_options(particles=P, seed=42))
    h.set_prior(*odom[0])
    o.set_prior(O.se2(*odom[0]))
    worst = 0.0
    for k in range(steps + 1):
        assert h.update(pts[k], odom[k], float(k)) == o.update(pts[k], O.se2(*odom[k]), float(k))
        gp = h.best_pose_xyr()
        op = o.poses()[o.best()]
        assert np.hypot(gp[0] - truth[k][0], gp[1] - truth[k][1]) < 0.03, (k, gp, truth[k])
        worst = max(worst, np.hypot(gp[0] - op[2], gp[1] - op[3]))
    assert worst < 1e-6, worst
    assert abs(h.neff() - o.neff()) < 1e-3
    print("free-running: max |gpu best - oracle best| = %.2e m" % worst)
    print(h.summary())
    h.close()


def test_free_running_3000_particles_tracks_oracle(F):
    """BASELINE config 2 size, free running (no teacher forcing): lama::PFSlam2D with 3000 particles on the GPU next to the
    oracle with the same seed for 6 scans, default gain and the forced-resample gain: the host RNG streams are identical, so
    every particle's pose must stay within 1e-6 m of the oracle's, with the same resampling decisions and best particle."""
    import os
    P, steps = 3000, 5
    pts, odom, truth = F.corridor_log(steps, 1080)
    for gain in (3.0, 1e-4):                       # 1 / (gain * P) is what enters the weights (src/pf_slam2d.cpp:514)
        h = F.PFSlam2D(F.pf_options(particles=P, seed=9, meas_sigma_gain=gain))
        o = O.PF(O.default_options(particles=P, seed=9, meas_sigma_gain=gain, threads=min(64, os.cpu_count() or 1)))
        h.set_prior(*odom[0])
        o.set_prior(O.se2(*odom[0]))
        for k in range(steps + 1):
            assert h.update(pts[k], odom[k], float(k)) == o.update(pts[k], O.se2(*odom[k]), float(k))
            assert np.abs(h.poses() - o.poses()).max() < 1e-6, (gain, k)
            assert h.num_resamples() == o.num_resamples(), (gain, k)
            assert h.best() == o.best(), (gain, k)
            assert abs(h.neff() - o.neff()) <= 1e-6 * max(1.0, o.neff()), (gain, k)
        if gain < 1.0:
            assert h.num_resamples() > 0
        gp = h.best_pose_xyr()
        assert np.hypot(gp[0] - truth[steps][0], gp[1] - truth[steps][1]) < 0.05
        h.close()


def test_long_free_running_3000_particles_every_map_every_scan(F):
    """BASELINE's largest pool compared DIRECTLY, not transitively -- 3000 particles free running (no teacher
    forcing) over 27 scans next to the oracle, with a gain that makes the filter resample on its own (scans 17 and 24) and with the
    routed stage and the early lane at their default thresholds; after EVERY scan the device-side checksums of both maps of ALL 3000
    particles equal the oracle's (patch set, every cell, every mask bit), poses within 1e-6, same resampling decisions, same best
    particle.  (Free-running poses differ from the oracle's by ~1e-11 m; a hit within that distance of a cell border would
    legitimately change a map -- expected 0.03 times in this run.)"""
    import os
    P, steps, gain = 3000, 26, 0.0012
    pts, odom, truth = F.corridor_log(steps, 1080)
    h = F.PFSlam2D(F.pf_options(particles=P, seed=9, meas_sigma_gain=gain))
    o = O.PF(O.default_options(particles=P, seed=9, meas_sigma_gain=gain, threads=min(64, os.cpu_count() or 1)))
    h.set_prior(*odom[0])
    o.set_prior(O.se2(*odom[0]))
    worst = 0.0
    for k in range(steps + 1):
        assert h.update(pts[k], odom[k], float(k)) == o.update(pts[k], O.se2(*odom[k]), float(k))
        worst = max(worst, float(np.abs(h.poses() - o.poses()).max()))
        assert worst < 1e-6, k
        assert h.num_resamples() == o.num_resamples(), k
        assert h.best() == o.best(), k
        ctx = h.hip_context()
        assert np.array_equal(ctx.map_checksums(F.MAP_DISTANCE), o.map_checksums(0)), f"distance maps after scan {k}"
        assert np.array_equal(ctx.map_checksums(F.MAP_OCCUPANCY), o.map_checksums(1)), f"occupancy maps after scan {k}"
    c = h.hip_context().counters()
    assert o.num_resamples() >= 2                                  # the filter resampled by itself
    assert c["brushfire_routed"] > 0 and c["brushfire_early"] > 0, c      # long chains went through the routed stage / the early lane
    assert c["resample_clones"] > 0
    print("largest pose difference to the oracle over the run", worst, "resamples", o.num_resamples(),
          "routed / early", c["brushfire_routed"], c["brushfire_early"])
    h.close()


def test_sharded_two_ranks_one_gpu_gloo(F):
    """G = 2 logical shards on ONE device (two processes, gloo collectives, blobs staged through the GPU):
    exercises export/import of particles in HBM and the sharded driver against the real HIP library."""
    import os, pickle, tempfile
    import torch.multiprocessing as mp
    from test_distributed_cpu import _free_port
    from _dist_worker import run
    world, P, steps, beams, gain = 2, 10, 8, 1080, 0.01
    out = tempfile.mkdtemp()
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=run, args=(r, world, port, "gloo", None, P, steps, beams, gain, out, 0)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    res = [pickle.load(open(os.path.join(out, f"rank{r}.pkl"), "rb")) for r in range(world)]
    assert all(r["origin"].endswith("liblama_hip.so") for r in res)
    assert res[0]["resamples"] == res[1]["resamples"] > 0
    assert sum(r["shipped"] for r in res) > 0
    # both ranks agree on the replicated state bit for bit
    for k in range(steps + 1):
        a, b = res[0]["hist"][k], res[1]["hist"][k]
        assert np.array_equal(a["w"], b["w"]) and np.array_equal(a["ws"], b["ws"]) and a["best"] == b["best"]
    # partition invariance against a single-shard run of the same library (same seed, same log)
    pts, odom, _ = F.corridor_log(steps, beams)
    h = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain))
    h.set_prior(*odom[0])
    for k in range(steps + 1):
        h.update(pts[k], odom[k], float(k))
        allp = np.concatenate([res[0]["hist"][k]["poses"], res[1]["hist"][k]["poses"]])
        assert np.array_equal(allp, h.poses()), k       # bit-identical for G = 1 and G = 2
    c = h.hip_context()
    for r in res:
        for i, (dm, occ) in r["maps"].items():
            assert_maps_equal(dm, c.download_map(i, F.MAP_DISTANCE), DM_FIELDS, f"dm p{i}")
            assert_maps_equal(occ, c.download_map(i, F.MAP_OCCUPANCY), OCC_FIELDS, f"occ p{i}")
    h.close()


def _visible_gpus():
    import torch
    return torch.cuda.device_count()


def _partition_invariance(F, world, backend, devices, P=3000, steps=5, gain=0.0001, l2_max=None):
    """`world` processes, one shard each (on the HIP devices `devices[r]`), against a single-shard run of the same library: poses,
    weights, resampling decisions and a device-side checksum of every particle's maps must agree bit for bit."""
    import os, pickle, tempfile
    import torch.multiprocessing as mp
    from test_distributed_cpu import _free_port
    from _dist_worker import run
    beams = 1080
    out = tempfile.mkdtemp()
    port = _free_port()
    ctx = mp.get_context("spawn")
    if l2_max is not None:
        os.environ["LAMA_TEST_L2_MAX"] = str(l2_max)          # (inherited by the spawned ranks)
    try:
        procs = [ctx.Process(target=run, args=(r, world, port, backend, None, P, steps, beams, gain, out, devices[r], True)) for r in range(world)]
        for p in procs:
            p.start()
    finally:
        os.environ.pop("LAMA_TEST_L2_MAX", None)
    for p in procs:
        p.join(900)
    if any(p.exitcode != 0 for p in procs) and not all(os.path.exists(os.path.join(out, f"rank{r}.up")) for r in range(world)):
        for p in procs:
            if p.is_alive():
                p.terminate()
        pytest.skip(f"the {backend} process group of {world} ranks could not be brought up on this box (exit codes {[p.exitcode for p in procs]})")
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = [pickle.load(open(os.path.join(out, f"rank{r}.pkl"), "rb")) for r in range(world)]
    assert [(r["lo"], r["hi"]) for r in res] == [(k * P // world, (k + 1) * P // world) for k in range(world)]
    assert [r["device"] for r in res] == list(devices) and all(r["backend"] == backend for r in res)
    assert len({r["resamples"] for r in res}) == 1 and res[0]["resamples"] > 0
    assert sum(r["shipped"] for r in res) > 0, "no clone crossed a shard border: the test would not exercise the shipping"
    pts, odom, _ = F.corridor_log(steps, beams)
    h = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, **({} if l2_max is None else {"l2_max": l2_max})))
    assert l2_max is None or h.engine_origin().endswith("liblama_hip_wide.so")
    h.set_prior(*odom[0])
    for k in range(steps + 1):
        h.update(pts[k], odom[k], float(k))
        allp = np.concatenate([r["hist"][k]["poses"] for r in res])
        assert np.array_equal(allp, h.poses()), k
        w, nw, ws = h.weights()
        for r in res:
            assert np.array_equal(r["hist"][k]["w"], w) and np.array_equal(r["hist"][k]["ws"], ws) and r["hist"][k]["best"] == h.best()
    c = h.hip_context()
    for kind, name in ((0, "distance"), (1, "occupancy")):
        got, want = np.concatenate([r["sums"][kind] for r in res]), c.map_checksums(F.MAP_DISTANCE if kind == 0 else F.MAP_OCCUPANCY)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, f"{name} maps of particles {bad[:20].tolist()} ({bad.size} of {P}) differ between the {world} shards and the single context"
    assert h.num_resamples() == res[0]["resamples"]
    h.close()


@pytest.mark.parametrize("world", sorted({4, 6, 8} | {int(w) for w in os.environ.get("LAMA_TEST_EXTRA_WORLDS", "").split(",") if w}))
def test_config3_split_partition_invariance(F, world):
    """BASELINE configs[2]: 3000 particles in G = 4 / 6 / 8 contiguous shards (750 / 500 / 375 per shard -- the 8-GPU split), here as G
    processes on ONE device over gloo, with a small meas_sigma_gain so that the filter resamples and clones cross shard
    borders.  Every shard must agree with a single-shard run of the same library bit for bit: poses, weights, resampling
    decisions, and a device-side checksum of every particle's maps.  (Round 6: this test is what caught the import kernel reading
    blobs that a foreign copy stream had written with plain loads -- 10 - 27 % of its runs diverged at G >= 5 with one binary of the
    round; tools/flake_partition.sh repeats it, LAMA_TEST_EXTRA_WORLDS adds process counts.)"""
    _partition_invariance(F, world, "gloo", [0] * world)


def test_partition_invariance_with_the_wide_library(F):
    """The same with a distance map that reaches 7 m (140 cells): every rank binds liblama_hip_wide.so, the blobs that cross shard
    borders carry 4-byte distance planes; 600 particles in 4 processes on one device against one context."""
    _partition_invariance(F, 4, "gloo", [0] * 4, P=600, steps=4, l2_max=7.0)


def test_config3_on_distinct_devices_over_rccl(F):
    """The same on REAL devices when the box has them (auto-skipped on a one-GPU box): one process per GPU, min(8, visible)
    ranks over `nccl` (= RCCL: the all-gather of the log-likelihoods and the batched isend / irecv of the particle blobs run
    device to device over xGMI), against the single-shard run."""
    n = min(8, _visible_gpus())
    if n < 2:
        pytest.skip("one visible device: RCCL with several ranks needs one GPU per rank (the sharding itself runs in test_config3_split_partition_invariance)")
    _partition_invariance(F, n, "nccl", list(range(n)))


def _multi_gpu_object_vs_one_shard(F, gpus, P, gain, distinct_devices=False):
    steps = 6 if P >= 3000 else 12
    pts, odom, _ = F.corridor_log(steps, 1080)
    a = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain))
    b = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, gpus=gpus))
    assert b.engine_origin().endswith("liblama_hip.so")
    devs = [b.shard_context(r).device() for r in range(gpus)]
    if distinct_devices:
        assert sorted(devs) == list(range(gpus)), devs            # shard r on HIP device r: peer copies really cross devices
    a.set_prior(*odom[0]); b.set_prior(*odom[0])
    shipped = 0
    for k in range(steps + 1):
        assert a.update(pts[k], odom[k], float(k)) == b.update(pts[k], odom[k], float(k))
        x = b.exchange_times()
        assert x["shards"] == gpus
        shipped += x["shipped_particles"]
        assert np.array_equal(a.poses(), b.poses()), k
        for u, v in zip(a.weights(), b.weights()):
            assert np.array_equal(u, v), k
        assert a.best() == b.best() and (k == 0 or a.neff() == b.neff())
    assert a.num_resamples() > 0 and a.num_resamples() == b.num_resamples()
    assert shipped > 0, "no clone crossed a shard border: the test would not exercise the shipping"
    ca = a.hip_context()
    for kind in (F.MAP_DISTANCE, F.MAP_OCCUPANCY):
        assert np.array_equal(np.concatenate([b.shard_context(r).map_checksums(kind) for r in range(gpus)]), ca.map_checksums(kind))
    assert np.array_equal(a.best_pose_xyr(), b.best_pose_xyr())
    a.close(); b.close()


@pytest.mark.parametrize("gpus,P,gain", [(2, 30, 0.01), (3, 301, 0.001), (8, 3000, 0.0001)])
def test_multi_gpu_object_is_bit_identical_to_one_shard(F, gpus, P, gain):
    """lama::PFSlam2D with Options::gpus > 1 (one process, a host thread + device context per shard; on a one-GPU box all of them
    on that device): update() is the whole sharded step in C++ -- no Python, no process group.  With a gain that makes the filter
    resample and clone across shard borders it must agree with the gpus = 1 object bit for bit: poses, weights, Neff, best particle,
    resampling decisions and a device-side checksum of every particle's maps.  (8, 3000) is BASELINE configs[2]'s split."""
    _multi_gpu_object_vs_one_shard(F, gpus, P, gain)


def test_runs_of_the_multi_gpu_object_are_bit_identical_to_each_other(F):
    """Run-to-run determinism under concurrency: eight contexts on one device, driven by eight host threads, 3000 particles, the
    first scans of the log (regions move after the first scan: the allocation guard's bound asks for head room) and a resample --
    thirty times; every run must give every particle the map checksums, poses and weights of the first run after every update.
    (Round 5: with the particle table read through the scalar data cache, 8 % of such runs gave one to three particles a distance
    map of their own -- a stale cache line sends a kernel to the region the particle has left; the table is read with agent-scope
    loads since.  One context alone never showed it.)"""
    P, gpus, gain, steps, runs = 3000, 8, 1e-4, 3, 30
    pts, odom, _ = F.corridor_log(steps, 1080)
    base = None
    for run in range(runs):
        a = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, gpus=gpus))
        a.set_prior(*odom[0])
        rec = []
        for k in range(steps + 1):
            a.update(pts[k], odom[k], float(k))
            dm = np.concatenate([a.shard_context(r).map_checksums(F.MAP_DISTANCE) for r in range(gpus)])
            oc = np.concatenate([a.shard_context(r).map_checksums(F.MAP_OCCUPANCY) for r in range(gpus)])
            rec.append((a.poses().copy(), a.weights()[1].copy(), dm, oc))
        n_res = a.num_resamples()
        a.close()
        if base is None:
            base = rec
            assert n_res > 0
            continue
        for k in range(steps + 1):
            for name, x, y in zip(("poses", "weights", "distance maps", "occupancy maps"), rec[k], base[k]):
                assert np.array_equal(x, y), f"run {run}, update {k}: {name} of particles {np.nonzero(np.atleast_2d(x.T != y.T).any(axis=0))[0][:8]} differ from the first run's"


def test_multi_gpu_object_on_distinct_devices(F):
    """The same object with one shard per REAL device (auto-skipped on a one-GPU box): Options::gpus = min(8, visible devices),
    3000 particles -- the clones that cross a shard border travel with hipMemcpyPeerAsync between different GPUs."""
    n = min(8, _visible_gpus())
    if n < 2:
        pytest.skip("one visible device: the shards of the object would share it (covered by test_multi_gpu_object_is_bit_identical_to_one_shard)")
    _multi_gpu_object_vs_one_shard(F, n, 3000, 0.0001, distinct_devices=True)


@pytest.mark.parametrize("seq_ray", [2, 1])
@pytest.mark.parametrize("radius", [4.0, 8.0, 12.0, 20.0])
def test_large_queue_paths_round_room(F, radius, seq_ray):
    """Round rooms of growing radius: the brushfire queue outgrows its LDS window, first mid-run (spill to
    k_brushfire_slow) and then from the start; the maps must stay bit-exact on every path."""
    n = 1080
    ang = np.deg2rad(-135.0 + 0.25 * np.arange(n))
    rng = np.random.default_rng(int(radius))
    r = radius + rng.normal(0, 0.01, n)
    pts = np.stack([r * np.cos(ang), r * np.sin(ang), np.zeros(n)], axis=1)
    pose0 = O.se2(0.3, -0.2, 0.1)
    P = 3
    pf = O.PF(O.default_options(particles=P, seed=1))
    pf.set_prior(pose0)
    pf.update(pts, pose0)
    # seq_ray 2 -> helper-wave brushfire (auto), seq_ray 1 -> single-wave brushfire: spill paths of both
    ctx = F.HipContext(F.default_cfg(particles=P, dm_patch_capacity=1024, occ_patch_capacity=1024, sequential_raycast=seq_ray,
                                     brushfire_waves=0 if seq_ray == 2 else 1))
    ctx.init(pts, pose0)
    # second scan from slightly different poses: removals (raise wave) + additions
    r2 = radius + rng.normal(0, 0.01, n)
    pts2 = np.stack([r2 * np.cos(ang), r2 * np.sin(ang), np.zeros(n)], axis=1)
    poses = np.stack([O.se2_mul(pose0, O.se2(0.05 * i, -0.03 * i, 0.01 * i)) for i in range(P)])
    pf.set_poses(poses)
    pf.stage_set_scan(pts2)
    pf.stage_update_maps()
    ctx.set_poses(poses)
    ctx.update_maps(pts2)
    for i in range(P):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"dm p{i}")
    print("radius", radius, "oracle max queue", pf.dm(0).stats()["max_queue"])
    ctx.close()


def test_slam2d_online_gpu_vs_oracle(F):
    """cfg 4: lama::Slam2D on the GPU (one-particle context, same kernels) free running next to the oracle's Slam2D."""
    steps = 20
    pts, odom, truth = F.corridor_log(steps, 1080)
    h = F.Slam2D()
    assert h.engine_origin().endswith("liblama_hip.so")
    o = O.Slam()
    h.set_pose(*odom[0])
    o.set_pose(O.se2(*odom[0]))
    for k in range(steps + 1):
        assert h.update(pts[k], odom[k], float(k)) == o.update(pts[k], O.se2(*odom[k]), float(k))
        d = np.abs(h.pose() - o.pose()).max()
        assert d < 1e-7, (k, d)
        if k > 0:
            assert h.iterations() == o.iterations()
        g = h.pose()
        assert np.hypot(g[2] - truth[k][0], g[3] - truth[k][1]) < 0.03
    # the maps agree bit for bit as long as the poses did not flip a cell; at 1e-7 m they do not on this log
    ctx = h.hip_context()
    assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), o.occ().dump(), OCC_FIELDS, "slam occ")
    assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), o.dm().dump(), DM_FIELDS, "slam dm")
    h.close()


def test_loc2d_gpu_vs_oracle(F):
    """cfg 1: lama::Loc2D on the device (static distance map built by addObstacle + brushfire on the GPU, solve with
    covariance and RMSE) against the oracle."""
    from _worlds import corridor_obstacles
    obst = corridor_obstacles()
    steps = 12
    pts, odom, truth = F.corridor_log(steps, 1080)
    o = O.Loc()
    dm = o.dm()
    for x, y in obst:
        c = O.w2m([x, y, 0.0])
        dm.add(int(c[0]), int(c[1]))
    dm.update()
    h = F.Loc2D()
    h.set_obstacles_world(obst)
    assert h.engine_origin().endswith("liblama_hip.so")
    # round 6: the FIRST build of the map is replayed by the host facade and uploaded; what the DEVICE then holds is the oracle's map
    assert_maps_equal(h.hip_context().download_map(0, F.MAP_DISTANCE), dm.dump(), DM_FIELDS, "Loc2D distance map after Init")
    start = truth[0] + np.array([0.05, -0.04, 0.01])
    o.set_pose(O.se2(*start))
    h.set_pose(*start)
    for k in range(steps + 1):
        assert o.update(pts[k], O.se2(*odom[k]), float(k), force=(k == 0)) == h.update(pts[k], odom[k], float(k), force=(k == 0))
        assert np.abs(o.pose() - h.pose()).max() < 1e-7, k
        assert o.iterations() == h.iterations()
        assert abs(o.rmse() - h.rmse()) < 1e-9
        assert np.allclose(o.covar(), h.covar(), rtol=1e-6, atol=1e-12)
    # a LATER update of the map that now exists -- a box of new obstacles in the corridor -- runs on the device (lama_hip_map_add_obstacles
    # on top of the uploaded map): still the oracle's map, and the localisation goes on identically
    box = np.array([(9.0 + 0.05 * i, 1.6 + 0.05 * j) for i in range(8) for j in range(8) if i in (0, 7) or j in (0, 7)])
    for x, y in box:
        c = O.w2m([x, y, 0.0])
        dm.add(int(c[0]), int(c[1]))
    dm.update()
    h.set_obstacles_world(box)
    assert_maps_equal(h.hip_context().download_map(0, F.MAP_DISTANCE), dm.dump(), DM_FIELDS, "Loc2D distance map after a later update")
    assert o.update(pts[steps], O.se2(*odom[steps]), 99.0, force=True) == h.update(pts[steps], odom[steps], 99.0, force=True)
    assert np.abs(o.pose() - h.pose()).max() < 1e-7
    h.close()


def test_first_map_build_device_chain_equals_host_build(F):
    """The same first build -- every occupied cell of a floor plan added to an empty map, one update() -- through the C-ABI as ONE exact
    brushfire on the device (lama_hip_map_add_obstacles) and through lama::Loc2D (host replay + upload): the same distance map, which
    is the oracle's, and the same number of processed cells."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import floor_plan_cells
    cells = floor_plan_cells(24.0, 16.0)
    dm = O.DM.new(l2_max=1.0)
    for x, y in cells:
        dm.add(int(x), int(y))
    n = dm.update()
    ctx = F.HipContext(F.default_cfg(particles=1, l2_max=1.0, queue_capacity=1 << 20))
    ctx.add_obstacles(0, cells)
    assert ctx.counters()["bf_cells"] == n
    chain = ctx.download_map(0, F.MAP_DISTANCE)
    ctx.close()
    assert_maps_equal(chain, dm.dump(), DM_FIELDS, "device chain")
    loc = F.Loc2D(l2_max=1.0)
    loc.set_obstacles_world((cells.astype(np.float64) - (2642244 >> 1) * 32) * 0.05)
    assert_maps_equal(loc.hip_context().download_map(0, F.MAP_DISTANCE), dm.dump(), DM_FIELDS, "host build, uploaded")
    loc.close()


def test_loc2d_loads_a_prebuilt_distance_map(F, tmp_path):
    """BASELINE config 1 is Loc2D on a PRE-BUILT distance map: distance_map->write(file) / distance_map->read(file) in the reference's
    .sdm format (Map::write / Map::read, src/sdm/map.cpp:489-575).  read() puts the file's patches on the device
    (lama_hip_pf_upload_map) without replaying addObstacle + update(): the loaded map is the built one bit for bit (downloaded and
    compared with the ORACLE's map), a file written by the oracle loads the same way, and localisation on the loaded map gives the
    poses, iteration counts and covariances of localisation on the map that was built on the device."""
    from _worlds import corridor_obstacles
    obst = corridor_obstacles()
    steps = 6
    pts, odom, truth = F.corridor_log(steps, 1080)
    odm = O.DM.new(l2_max=1.0)
    for x, y in obst:
        c = O.w2m([x, y, 0.0])
        odm.add(int(c[0]), int(c[1]))
    odm.update()
    built = F.Loc2D()
    built.set_obstacles_world(obst)
    f_dev, f_orc = str(tmp_path / "device.sdm"), str(tmp_path / "oracle.sdm")
    built.write_distance_map(f_dev)
    odm.write(f_orc)
    start = truth[0] + np.array([0.05, -0.04, 0.01])
    runs = []
    for src in (None, f_dev, f_orc):
        h = built if src is None else F.Loc2D()
        if src is not None:
            h.read_distance_map(src)
            ctx = h.hip_context()
            assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), odm.dump(), DM_FIELDS, f"map loaded from {os.path.basename(src)}")
            assert ctx.counters()["bf_cells"] == 0                       # nothing was replayed
        h.set_pose(*start)
        out = []
        for k in range(steps + 1):
            h.update(pts[k], odom[k], float(k), force=(k == 0))
            out.append((h.pose().copy(), h.iterations(), h.rmse(), h.covar().copy()))
        runs.append(out)
        if src is not None:
            h.close()
    built.close()
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and np.array_equal(a[3], b[3])
    # the C-ABI call itself, on a particle filter's context: a particle's maps replaced by another particle's (download -> upload)
    P = 3
    ctx = F.HipContext(F.default_cfg(particles=P))
    ctx.init(pts[0], O.se2(*odom[0]))
    rng = np.random.default_rng(2)
    poses = np.stack([O.se2_mul(O.se2(*truth[1]), O.se2(*rng.normal(0, [0.2, 0.1, 0.03]))) for _ in range(P)])
    ctx.set_poses(poses); ctx.update_maps(pts[1])
    d0, o0 = ctx.download_map(0, F.MAP_DISTANCE), ctx.download_map(0, F.MAP_OCCUPANCY)
    ctx.upload_map(2, F.MAP_DISTANCE, d0); ctx.upload_map(2, F.MAP_OCCUPANCY, o0)
    assert_maps_equal(ctx.download_map(2, F.MAP_DISTANCE), d0, DM_FIELDS, "uploaded dm")
    assert_maps_equal(ctx.download_map(2, F.MAP_OCCUPANCY), o0, OCC_FIELDS, "uploaded occ")
    ck = ctx.map_checksums(F.MAP_DISTANCE)
    assert ck[2] == ck[0] and ck[1] != ck[0]
    ctx.set_poses(np.stack([poses[0], poses[1], poses[0]])); ctx.update_maps(pts[2])          # ... and the uploaded maps are live
    assert_maps_equal(ctx.download_map(2, F.MAP_DISTANCE), ctx.download_map(0, F.MAP_DISTANCE), DM_FIELDS, "after an update")
    assert_maps_equal(ctx.download_map(2, F.MAP_OCCUPANCY), ctx.download_map(0, F.MAP_OCCUPANCY), OCC_FIELDS, "after an update")
    ctx.close()


def test_3000_particles_map_a_hall_of_10000_m2_with_in_place_resampling(F):
    """memory and resampling that scale with CHANGE.  3000 particles map a generated hall of 108 m x 108 m (16
    bays, a 1080-beam scanner with 27 m range) along a tour through all bays -- more than 10,000 m^2 per particle, ~46 MB of maps
    each -- and are resampled every third scan with index vectors that kill about 40 % of the pool.  One particle set: survivors
    keep their home and regions, clones go to the homes of the dead, regions grow one particle at a time in pooled planes that
    grow chunk by chunk.  Checked: a sample of final particles is bit-exact against the ORACLE replaying each one's lineage (the
    poses of its ancestors, scan by scan -- a particle's map depends on nothing else); clones that have not been updated since
    have the checksum of their source; no capacity error; allocated HBM <= 2 x used; only the clones were copied."""
    import json
    from _worlds import hall_segments, hall_tour, segment_world_scan
    P, resample_every = 3000, 3
    SIDE = 108.0                                                             # 11,664 m^2 of floor; what a particle has SEEN of it is checked below
    segs, tour = hall_segments(SIDE), hall_tour(SIDE)
    rng = np.random.default_rng(21)
    scans = [segment_world_scan(segs, x, y, yaw, max_range=27.0, noise=rng.normal(0.0, 0.01, 1080)) for x, y, yaw in tour]
    off = rng.normal(0.0, [0.12, 0.12, 0.01], size=(P, 3))                  # every slot's own offset from the true pose
    off[0] = 0.0

    def poses_at(k, slots):
        c, s_ = np.cos(tour[k][2]), np.sin(tour[k][2])
        x = tour[k][0] + c * off[slots, 0] - s_ * off[slots, 1]
        y = tour[k][1] + s_ * off[slots, 0] + c * off[slots, 1]
        th = tour[k][2] + off[slots, 2]
        return np.stack([np.cos(th), np.sin(th), x, y], axis=1)
    ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
    ctx.init(scans[0], O.se2(*tour[0]))
    all_slots = np.arange(P)
    history = []                                                             # per scan: the resample index vector applied before it (or None)
    for k in range(1, len(tour)):
        idx = None
        if k % resample_every == 0:
            w = rng.random(P) ** 3                                           # skewed weights: many particles die, some are drawn often
            cs = np.cumsum(w / w.sum())
            idx = np.searchsorted(cs, (rng.random() + np.arange(P)) / P).clip(0, P - 1).astype(np.int32)      # systematic resampling
            before = ctx.map_checksums(F.MAP_OCCUPANCY)
            ctx.resample(idx)
            assert np.array_equal(ctx.map_checksums(F.MAP_OCCUPANCY), before[idx])        # every new particle IS its source, all 3000
        history.append(idx)
        ctx.set_poses(poses_at(k, all_slots))
        ctx.update_maps(scans[k])
    c = ctx.counters()
    assert c["pool_growths"] >= 2 and c["arena_growths"] >= 5, c
    used, alloc = c["hbm_bytes_used"], c["hbm_bytes_allocated"]
    assert used > 100e9 and alloc <= 2.0 * used, (used, alloc)
    n_res = sum(1 for h in history if h is not None)
    n_clone = sum(P - len(np.unique(h)) for h in history if h is not None)
    assert c["resample_clones"] == (P - 1) + n_clone and n_clone > 0.25 * P * n_res, (c["resample_clones"], n_clone, n_res)
    # lineages of a sample of final particles, replayed by the oracle (one oracle particle per sampled lineage, no resampling there)
    sample = np.array(sorted(set([0, 1, 2, 777, 1500, 2222, 2998, 2999] + list(rng.integers(0, P, 8)))))
    lineage = np.zeros((len(tour), len(sample)), dtype=np.int64)            # slot of the ancestor at scan k
    cur = sample.copy()
    for k in range(len(tour) - 1, 0, -1):
        lineage[k] = cur
        if history[k - 1] is not None:
            cur = history[k - 1][cur]
    lineage[0] = 0
    S = len(sample)
    pf = O.PF(O.default_options(particles=S, seed=1, threads=min(S, os.cpu_count() or 1)))
    pf.set_prior(O.se2(*tour[0]))
    assert pf.update(scans[0], O.se2(*tour[0]))
    for k in range(1, len(tour)):
        pf.set_poses(poses_at(k, lineage[k]))
        pf.stage_set_scan(scans[k])
        pf.stage_update_maps()
    area = 0.0
    for j, p in enumerate(sample):
        occ = ctx.download_map(int(p), F.MAP_OCCUPANCY)
        assert_maps_equal(occ, pf.occ(j).dump(), OCC_FIELDS, f"occ of particle {p}")
        assert_maps_equal(ctx.download_map(int(p), F.MAP_DISTANCE), pf.dm(j).dump(), DM_FIELDS, f"dm of particle {p}")
        if j == 0:
            area = sum(int((cells["visited"] != 0).sum()) for cells, _ in occ.values()) * 0.05 * 0.05
    report = {"particles": P, "scans": len(tour), "resamples": n_res, "clones_copied": int(c["resample_clones"]), "mapped_m2_per_particle": area,
              "hbm_bytes_used": int(used), "hbm_bytes_allocated": int(alloc), "allocated_over_used": alloc / used,
              "pool_chunks_added": int(c["pool_growths"]), "region_growth_batches": int(c["arena_growths"]),
              "resample_ms_mean": c["ms_resample"] / max(c["launches_resample"], 1), "clone_bytes_per_resample": c["resample_bytes"] / max(c["launches_resample"], 1),
              "update_maps_ms_mean": c["ms_update_maps"] / max(c["launches_update_maps"], 1), "dm_patches": int(c["dm_patches"]), "occ_patches": int(c["occ_patches"])}
    print("big world:", json.dumps(report))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        json.dump(report, open(os.path.join(out, "r05_big_world.json"), "w"), indent=1)
    ctx.close()
    assert area >= 10000.0, area                                             # cells a particle has visited, in m^2


def test_slam2d_map_accessors_gpu(F):
    """Slam2D::getOccupancyMap() / getDistanceMap(): snapshots of the device maps answer the reference's const map queries
    (bounds, visit_all_cells, isFree / isOccupied / isUnknown / getProbability, distance, gradient) exactly like the oracle's
    maps built from the same scans at the same (teacher-forced) poses."""
    from _cmp import check_slam_views
    steps = 8
    pts, odom, truth = F.corridor_log(steps, 1080)
    o, h = O.Slam(), F.Slam2D()
    assert h.engine_origin().endswith("liblama_hip.so")
    o.set_pose(O.se2(*odom[0])); h.set_pose(*odom[0])
    for k in range(steps + 1):
        assert o.update(pts[k], O.se2(*odom[k]), float(k)) == h.update(pts[k], odom[k], float(k))
        assert np.abs(o.pose() - h.pose()).max() < 1e-8
        p = o.pose()
        h.set_pose(p[2], p[3], float(np.arctan2(p[1], p[0])))      # teacher forcing keeps the integer maps comparable
        o.set_pose(h.pose())
    check_slam_views(o, h, np.random.default_rng(4))
    h.close()


def test_match_surface_2d_and_solver_api_gpu(F):
    """lama::MatchSurface2D::eval / error as device kernels and lama::Solve as ONE fused device launch, through the reference's
    class API, against the oracle: residuals 1e-9, Jacobian 1e-6, poses 1e-8, equal iteration counts, covariance rel 1e-6;
    configurations without a device kernel raise."""
    from _cmp import check_match_surface_and_solver
    steps = 6
    pts, odom, truth = F.corridor_log(steps, 1080)
    o, h = O.Slam(), F.Slam2D()
    assert h.engine_origin().endswith("liblama_hip.so")
    o.set_pose(O.se2(*odom[0])); h.set_pose(*odom[0])
    for k in range(steps + 1):
        assert o.update(pts[k], O.se2(*odom[k]), float(k)) == h.update(pts[k], odom[k], float(k))
        p = o.pose()
        h.set_pose(p[2], p[3], float(np.arctan2(p[1], p[0])))
        o.set_pose(h.pose())
    check_match_surface_and_solver(O, o, h, pts[steps], np.random.default_rng(8), pose_tol=1e-8, exact=False)
    h.close()


def test_loc2d_rank_deficient_covariance_gpu(F):
    """Rank-deficient branch of Solver::calculateCovariance (src/nlls/solver.cpp:143-149) on the device path: corridor
    with its ends out of sight, x unobservable -> variance 3.0 along x, (J^T J)-eigen pairs elsewhere."""
    from _worlds import open_corridor, open_corridor_scan
    obst = open_corridor()
    o = O.Loc()
    dm = o.dm()
    for x, y in obst:
        c = O.w2m([x, y, 0.0])
        dm.add(int(c[0]), int(c[1]))
    dm.update()
    h = F.Loc2D()
    h.set_obstacles_world(obst)
    assert h.engine_origin().endswith("liblama_hip.so")
    truth = np.array([1.3, 1.7, 0.12])
    scan = open_corridor_scan(*truth, beams=1080)
    start = truth + np.array([0.0, 0.06, -0.02])
    o.set_pose(O.se2(*start)); h.set_pose(*start)
    assert o.update(scan, O.se2(*start), 0.0, force=True) == h.update(scan, start, 0.0, force=True)
    assert o.rank_deficient()
    assert np.abs(o.pose() - h.pose()).max() < 1e-8
    assert o.iterations() == h.iterations()
    co, ch = o.covar().reshape(3, 3), h.covar().reshape(3, 3)
    assert np.allclose(co, ch, rtol=1e-6, atol=1e-10)
    assert abs(ch[0, 0] - 3.0) < 1e-9 and abs(ch[0, 1]) < 1e-9 and abs(ch[0, 2]) < 1e-9
    h.close()


def test_sharded_collective_path_on_rccl(F):
    """The multi-rank code path of ShardedPF (device all-gather of the log-likelihoods over RCCL, resample planning, barrier /
    max-over-ranks) forced on a world of ONE rank: must reproduce the plain PFSlam2D run bit for bit.  (More ranks need more
    GPUs; the sharding logic itself is covered with gloo, world_size 2, in test_distributed_cpu.py.)"""
    import os
    import torch
    import torch.distributed as dist
    from iris_lama_amd.distributed import ShardedPF
    from test_distributed_cpu import _free_port
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(_free_port()))
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        P, steps = 12, 8
        pts, odom, truth = F.corridor_log(steps, 1080)
        kw = dict(particles=P, seed=11, gpu_device=0, meas_sigma_gain=0.01)       # small gain: resampling happens
        a = ShardedPF(F.pf_options(**kw), force_collectives=True)
        b = F.PFSlam2D(F.pf_options(**kw))
        assert a.backend == "nccl" and a.device.type == "cuda"
        a.set_prior(*odom[0]); b.set_prior(*odom[0])
        for k in range(steps + 1):
            assert a.update(pts[k], odom[k], float(k)) == b.update(pts[k], odom[k], float(k))
            assert np.array_equal(a.pf.poses(), b.poses()), k
            assert a.pf.best() == b.best()
        assert a.max_over_ranks(1.25) == 1.25
        a.barrier()
        assert b.num_resamples() > 0 and a.pf.num_resamples() == b.num_resamples()
        a.close(); b.close()
    finally:
        if created:
            dist.destroy_process_group()


def test_canonical_brushfire_mode(F):
    """cfg.brushfire_mode = 1 (level-synchronous, canonical tie rule): bit-exact against the oracle's update_canonical(),
    and -- against the FAITHFUL oracle -- identical in everything but the obstacle offsets of tie cells."""
    P, steps = 6, 10
    pts, odom, truth = F.corridor_log(steps, 1080)
    rng = np.random.default_rng(9)
    pose0 = O.se2(*odom[0])
    faithful = O.PF(O.default_options(particles=P, seed=7))
    faithful.set_prior(pose0)
    faithful.update(pts[0], pose0)
    O.set_canonical_default(True)
    try:
        canon = O.PF(O.default_options(particles=P, seed=7))
        canon.set_prior(pose0)
        canon.update(pts[0], pose0)
    finally:
        O.set_canonical_default(False)
    ctx = F.HipContext(F.default_cfg(particles=P, brushfire_mode=1))
    ctx.init(pts[0], pose0)
    tie_cells = dist_cells = total_cells = 0
    for k in range(0, steps + 1):
        if k > 0:
            poses = _perturbed(rng, O.se2(*truth[k]), P, 0.01, 0.003)
            for pf in (faithful, canon):
                pf.set_poses(poses)
                pf.stage_set_scan(pts[k])
                pf.stage_update_maps()
            ctx.set_poses(poses)
            ctx.update_maps(pts[k])
            if k % 3 == 0:
                idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
                faithful.stage_resample_with(idx); canon.stage_resample_with(idx); ctx.resample(idx)
        for i in range(P):
            g_dm = ctx.download_map(i, F.MAP_DISTANCE)
            assert_maps_equal(g_dm, canon.dm(i).dump(), DM_FIELDS, f"scan {k} canonical dm p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), canon.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            from _cmp import diff_maps
            d = diff_maps(g_dm, faithful.dm(i).dump(), DM_FIELDS)
            assert d["mask_words"] == 0 and d["patches_only_dev"] == 0 and d["patches_only_orc"] == 0 and d["queued"] == 0, (k, i, d)
            tie_cells += d["obstacle"]
            dist_cells += max(d["sqdist"], d["valid"])
            total_cells += 1024 * len(g_dm)
    print(f"canonical vs faithful over {steps + 1} scans x {P} particles ({total_cells} cells compared): obstacle offset differs in "
          f"{tie_cells} cells (ties), sqdist/valid differs in {dist_cells} cells")
    # the distance VALUE may differ only in rare cells where the tie order leaks into a distance (measured: ~1e-6 of the cells)
    assert dist_cells <= 1e-5 * total_cells
    ctx.close()


@pytest.mark.parametrize("trunc_ray,trunc_range", [(0.0, 0.0), (3.0, 0.0), (0.0, 6.0), (2.5, 8.0)])
@pytest.mark.parametrize("seq_ray", [1, 2])
def test_options_truncation_and_sensor_frame(F, trunc_ray, trunc_range, seq_ray):
    """Non-default options of the path: truncated_ray / truncated_range (src/pf_slam2d.cpp:467-491), a sensor mounted
    off-centre with a yaw+roll orientation and points with z != 0 (moving_tf, 3-axis Bresenham count)."""
    steps, P = 5, 3
    pts, odom, truth = F.corridor_log(steps, 540)
    rng = np.random.default_rng(3)
    pts = pts.copy()
    pts[:, :, 2] = rng.normal(0, 0.02, size=pts.shape[:2])           # slightly non-planar returns
    origin = np.array([0.2, -0.1, 0.3])
    yaw, roll = 0.15, 0.05
    qz = np.array([np.cos(yaw / 2), 0, 0, np.sin(yaw / 2)])
    qx = np.array([np.cos(roll / 2), np.sin(roll / 2), 0, 0])
    quat = np.array([qz[0] * qx[0] - qz[3] * qx[3] * 0, qz[0] * qx[1], qz[3] * qx[1], qz[3] * qx[0]])   # q = qz * qx
    quat = quat / np.linalg.norm(quat)
    opts = O.default_options(particles=P, seed=5, truncated_ray=trunc_ray, truncated_range=trunc_range)
    pf = O.PF(opts)
    pose0 = O.se2(*odom[0])
    pf.set_prior(pose0)
    assert pf.update(pts[0], pose0, 0.0, origin=origin, quat=quat)
    ctx = F.HipContext(F.default_cfg(particles=P, truncated_ray=trunc_ray, truncated_range=trunc_range, sequential_raycast=seq_ray))
    ctx.init(pts[0], pose0, origin=origin, quat=quat)
    for k in range(0, steps + 1):
        if k > 0:
            start = _perturbed(rng, O.se2(*truth[k]), P, 0.02, 0.005)
            pf.set_poses(start)
            pf.set_weights(w=np.zeros(P), ws=np.zeros(P))
            pf.stage_set_scan(pts[k], origin=origin, quat=quat)
            pf.stage_scan_match()
            ctx.set_poses(start)
            g_poses, g_ll, g_it = ctx.scan_match(pts[k], origin=origin, quat=quat)
            o_poses = pf.poses()
            same = g_it == np.array([pf.counters(i)["iterations"] for i in range(P)])
            assert np.abs(g_poses - o_poses)[same].max() < 1e-8
            assert np.allclose(g_ll[same], pf.weights()[0][same], rtol=1e-9)
            ctx.set_poses(o_poses)
            pf.stage_update_maps()
            ctx.update_maps(pts[k], origin=origin, quat=quat)
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    ctx.close()


def test_tiny_and_degenerate_scans(F):
    """Edge cases: one particle, scans of 1..3 beams, a beam that ends in the sensor's own cell (from == to), duplicates."""
    pose0 = O.se2(0.0, 0.0, 0.0)
    scans = [np.array([[1.0, 0.0, 0.0]]),
             np.array([[0.01, 0.01, 0.0], [2.0, 0.5, 0.0]]),                 # first beam: hit cell == start cell
             np.array([[1.0, 0.0, 0.0], [1.0, 0.0, 0.0], [1.001, 0.001, 0.0]])]  # duplicate beams
    pf = O.PF(O.default_options(particles=1, seed=1))
    pf.set_prior(pose0)
    pf.update(scans[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=1))
    ctx.init(scans[0], pose0)
    for k, sc in enumerate(scans[1:] + scans):
        pose = O.se2(0.02 * k, -0.01 * k, 0.03 * k)
        pf.set_poses(pose[None])
        pf.stage_set_scan(sc)
        pf.stage_update_maps()
        ctx.set_poses(pose[None])
        ctx.update_maps(sc)
        assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), pf.occ(0).dump(), OCC_FIELDS, f"tiny {k} occ")
        assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), pf.dm(0).dump(), DM_FIELDS, f"tiny {k} dm")
        pf.stage_scan_match()
        g_poses, g_ll, g_it = ctx.scan_match(sc)
        assert np.abs(g_poses - pf.poses()).max() < 1e-8
        ctx.set_poses(pf.poses())
    ctx.close()


def test_loc2d_global_localization_and_sampling_covariance_gpu(F):
    """Loc2D::globalLocalization (3000 candidates evaluated in one batch by lama_hip_eval_batch) and
    addSamplingCovariance (lama_hip_map_sample_likelihood) on the device against the oracle: identical candidates (host
    RNG), squared residual norms within 1e-11 relative, the same winner, localisation recovers the true pose."""
    from _worlds import corridor_free_cells, corridor_obstacles
    obst = corridor_obstacles()
    free = corridor_free_cells(O.w2m)
    steps = 5
    pts, odom, truth = F.corridor_log(steps, 1080)
    kw = dict(gloc_particles=3000, gloc_iters=5, gloc_thresh=0.15, cov_blend=0.35)
    o = O.Loc(**kw)
    dm = o.dm()
    ocells = np.array([[int(c[0]), int(c[1])] for c in (O.w2m([x, y, 0.0]) for x, y in obst)], dtype=np.uint32)
    for cx, cy in ocells:
        dm.add(int(cx), int(cy))
    dm.update()
    o.occ_set_cells(free, -1)
    o.occ_set_cells(ocells, 1)
    h = F.Loc2D(**kw)
    h.occ_set_cells(free, -1)
    h.set_obstacles_world(obst)
    assert h.engine_origin().endswith("liblama_hip.so")
    O.random_set_seed(77)
    F.random_set_seed(77)
    start = np.array([20.0, 1.0, 2.0])
    o.set_pose(O.se2(*start))
    h.set_pose(*start)
    o.trigger_global_localization()
    h.trigger_global_localization()
    ran = 0
    for k in range(steps + 1):
        assert o.update(pts[k], O.se2(*odom[k]), float(k), force=True) == h.update(pts[k], odom[k], float(k), force=True)
        op, oe = o.gloc_candidates()
        hp, he = h.gloc_candidates()
        assert np.array_equal(op, hp), k
        assert np.allclose(oe, he, rtol=1e-11, atol=0), (k, np.abs(oe - he).max())
        assert int(np.argmin(oe)) == int(np.argmin(he))
        ran += len(he) > 0
        ol, hl = o.sampling_likelihoods(), h.sampling_likelihoods()
        assert len(hl) == 161 and np.allclose(ol, hl, rtol=1e-12, atol=1e-300), (k, np.abs(ol - hl).max())
        assert np.abs(o.pose() - h.pose()).max() < 1e-7, k
        assert o.iterations() == h.iterations()
        assert o.global_localization_active() == h.global_localization_active()
        assert abs(o.rmse() - h.rmse()) < 1e-9
        assert np.allclose(o.covar(), h.covar(), rtol=1e-6, atol=1e-12)
    assert ran >= 1 and not h.global_localization_active()
    g = h.pose()
    assert np.hypot(g[2] - truth[steps][0], g[3] - truth[steps][1]) < 0.05
    h.close()


def test_eval_batch_matches_oracle_and_match_batch(F):
    """lama_hip_eval_batch: squared residual norm and log-likelihood per pose on one particle's map."""
    pts, odom, truth = F.corridor_log(2, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=1, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=1, profile=1))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(11)
    B = 3000
    poses = np.stack([O.se2(rng.uniform(0.5, 27.5), rng.uniform(0.5, 3.5), rng.uniform(-np.pi, np.pi)) for _ in range(B)])
    ctx.reset_counters()
    sq, ll = ctx.eval_batch(0, pts[1], poses)
    ms = ctx.counters()["ms_eval_batch"]
    ll2 = ctx.match_batch(0, pts[1], poses)
    assert np.array_equal(ll, ll2)
    dmo = pf.dm(0)
    for b in range(0, B, 97):
        r = O.eval_(dmo, pts[1], poses[b], jac=False)
        assert abs(sq[b] - float(np.dot(r, r))) <= 1e-11 * max(1.0, sq[b])
    print(f"eval_batch: {B} poses x 1080 beams in {ms:.3f} ms")


def test_lidar_odometry_gpu_vs_oracle(F):
    """lama::LidarOdometry2D on the device -- ProbabilisticOccupancyMap log-odds cells (float, bit-exact), the
    last-metre ray rule, distance map with 1 m range, transient-map patch deletion -- against the oracle, free running."""
    steps = 26
    pts, odom, truth = F.corridor_log(steps, 1080)
    o = O.LidarOdometry()
    h = F.LidarOdometry2D()
    assert h.engine_origin().endswith("liblama_hip.so")
    deleted = 0
    for k in range(steps + 1):
        p = pts[k][np.hypot(pts[k][:, 0], pts[k][:, 1]) < 4.0]
        assert o.update(p, float(k)) == h.update(p, float(k))
        assert np.abs(o.odom() - h.odom()).max() < 1e-7, (k, o.odom(), h.odom())
        assert o.iterations() == h.iterations(), k
        deleted += h.deleted_patches()
        assert h.deleted_patches() == o.deleted_last(), k
        if k % 5 == 0 or k == steps:
            ctx = h.hip_context()
            assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), o.dm().dump(), DM_FIELDS, f"lo dm {k}")
            got, ref = ctx.download_map(0, F.MAP_OCCUPANCY), o.occ().dump()
            assert sorted(got) == sorted(ref), k
            for pid in ref:
                assert np.array_equal(np.ascontiguousarray(got[pid][0]).view(np.float32).reshape(-1), ref[pid][0]["prob"]), (k, pid)
                assert np.array_equal(got[pid][1], ref[pid][1]), (k, pid)
    assert deleted > 0
    h.close()


def test_full_size_3000_particles_properties(F):
    """BASELINE config 2 size (3000 particles on one GPU), size-independent properties: particles driven with identical
    poses must end with identical maps (each equal to the oracle's single particle), the patch counters add up, a scan-match
    from identical start poses gives identical results for all 3000, and resampling is a pure gather of whole particles."""
    P = 3000
    pts, odom, truth = F.corridor_log(3, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=1, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
    ctx.init(pts[0], pose0)
    for k in (1, 2):
        start = np.tile(O.se2(*truth[k]), (P, 1))
        pf.set_poses(start[:1])
        pf.set_weights(w=np.zeros(1), ws=np.zeros(1))
        pf.stage_set_scan(pts[k])
        pf.stage_scan_match()
        ctx.set_poses(start)
        g_poses, g_ll, g_it = ctx.scan_match(pts[k])
        assert (g_poses == g_poses[0]).all() and (g_ll == g_ll[0]).all() and (g_it == g_it[0]).all()     # same input -> same bits
        assert np.abs(g_poses[0] - pf.poses()[0]).max() < 1e-8
        ctx.set_poses(np.tile(pf.poses()[0], (P, 1)))
        pf.stage_update_maps()
        ctx.update_maps(pts[k])
    c = ctx.counters()
    ref_dm, ref_occ = pf.dm(0).dump(), pf.occ(0).dump()
    assert c["dm_patches"] == P * len(ref_dm) and c["occ_patches"] == P * len(ref_occ)
    for i in (0, 1, 511, 1499, 2998, 2999):
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), ref_dm, DM_FIELDS, f"dm p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), ref_occ, OCC_FIELDS, f"occ p{i}")
    # make particle 7 different, then resample everything from it and from particle 8 alternately
    poses = np.tile(pf.poses()[0], (P, 1))
    poses[7] = O.se2_mul(poses[7], O.se2(0.1, -0.05, 0.02))
    ctx.set_poses(poses)
    ctx.update_maps(pts[3])
    d7, d8 = ctx.download_map(7, F.MAP_DISTANCE), ctx.download_map(8, F.MAP_DISTANCE)
    idx = np.where(np.arange(P) % 2 == 0, 7, 8).astype(np.int32)
    ctx.resample(idx)
    for i in (0, 1, 1000, 2999):
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), d7 if i % 2 == 0 else d8, DM_FIELDS, f"resampled dm p{i}")
    assert np.array_equal(ctx.get_poses()[::2], np.tile(poses[7], (P // 2, 1)))
    ctx.close()


def test_full_size_3000_distinct_particles_vs_oracle(F):
    """BASELINE config 2 size with 3000 DIFFERENT particles: scan match of all 3000 against the oracle (poses <= 1e-8, log-lik
    rel 1e-9, identical Gauss-Newton iteration counts), then -- at the oracle's poses -- the map update: the map checksums of ALL
    3000 particles equal the oracle's (checksum of checksums), a sample of maps compared cell by cell, patch counters add up."""
    import os
    P = 3000
    pts, odom, truth = F.corridor_log(2, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=P, seed=5, threads=min(64, os.cpu_count() or 1)))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(12)
    for k in (1, 2):
        noise = rng.normal(0, [0.04, 0.04, 0.015], size=(P, 3))
        start = np.stack([O.se2_mul(O.se2(*truth[k]), O.se2(*noise[i])) for i in range(P)])
        pf.set_poses(start)
        pf.set_weights(w=np.zeros(P), ws=np.zeros(P))
        pf.stage_set_scan(pts[k])
        pf.stage_scan_match()
        ctx.set_poses(start)
        ctx.reset_counters()
        g_poses, g_ll, g_it = ctx.scan_match(pts[k])
        o_poses, o_ll = pf.poses(), pf.weights()[0]
        assert np.abs(g_poses - o_poses).max() < 1e-8
        assert np.allclose(g_ll, o_ll, rtol=1e-9, atol=0)
        o_it = np.array([pf.counters(i)["iterations"] for i in range(P)])
        assert ctx.counters()["gn_iterations"] == int(o_it.sum())
        assert len(np.unique(np.round(g_poses, 6), axis=0)) > P // 2          # the particles really differ
        ctx.set_poses(o_poses)                                               # teacher forcing: integer maps stay comparable
        pf.stage_update_maps()
        ctx.update_maps(pts[k])
    c = ctx.counters()
    assert c["dm_patches"] == sum(len(pf.dm(i).patch_ids()) for i in range(P))
    assert c["occ_patches"] == sum(len(pf.occ(i).patch_ids()) for i in range(P))
    # a checksum of every particle's maps (patch set, every cell, every mask bit), computed on the device and by the checker
    g_dm, g_occ = ctx.map_checksums(F.MAP_DISTANCE), ctx.map_checksums(F.MAP_OCCUPANCY)
    assert np.array_equal(g_dm, pf.map_checksums(0)) and np.array_equal(g_occ, pf.map_checksums(1))
    assert len(np.unique(g_dm)) > P // 2 and len(np.unique(g_occ)) > P // 2      # different particles, different maps
    for i in (0, 1, 777, 1500, 2222, 2999):
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"dm p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"occ p{i}")
    ctx.close()


def test_limits_fail_loudly_with_status_codes(F):
    """A cell outside the device window and an invalid configuration are errors with a status code and a message -- never silent
    truncation (include/lama_hip.h status codes); full patch arenas grow, too many order-sensitive visits go beam by beam."""
    pts, odom, truth = F.corridor_log(1, 1080)
    pose0 = O.se2(*odom[0])
    # a window of 8 patches = 12.8 m does not hold the 28 m corridor: since round 4 the window GROWS (the reference's maps have no
    # extent) -- same maps as from the default window
    ctx = F.HipContext(F.default_cfg(particles=2, window_patches=8))
    ref = F.HipContext(F.default_cfg(particles=2))
    ctx.init(pts[0], pose0); ref.init(pts[0], pose0)
    cw = ctx.counters()
    assert cw["window_growths"] >= 1 and cw["window_patches"] > 8 and ref.counters()["window_growths"] == 0
    for kind, fields in ((F.MAP_DISTANCE, DM_FIELDS), (F.MAP_OCCUPANCY, OCC_FIELDS)):
        assert_maps_equal(ctx.download_map(1, kind), ref.download_map(1, kind), fields, "grown window")
    ctx.close(); ref.close()
    # 4 patches per particle used to be an error; since round 3 an update that needs more patches than are free says so before it
    # modifies anything; round 5: the particle's region is then grown by what the update asked for (one particle at a time, not
    # doubled for everybody) and the update runs again (the reference's maps just allocate)
    for small in (dict(dm_patch_capacity=4), dict(occ_patch_capacity=4)):
        ctx = F.HipContext(F.default_cfg(particles=2, **small))
        ctx.init(pts[0], pose0)
        c = ctx.counters()
        assert c["arena_growths"] >= 1 and c["hbm_bytes_used"] <= c["hbm_bytes_allocated"], c
        ctx.close()
    # the parallel ray-cast keeps order-sensitive visits in a list of active_capacity entries (1080 hits alone exceed 64): such
    # a scan is cast beam by beam instead (round 2; it used to be an error)
    ctx = F.HipContext(F.default_cfg(particles=2, active_capacity=64, sequential_raycast=2))
    ctx.init(pts[0], pose0)
    ctx.close()
    for bad in (dict(patch_size=16), dict(window_patches=252), dict(window_patches=1024), dict(resolution=0.0), dict(l2_max=13.0)):      # (13 m = 260 cells: beyond the 255 of the wide library)
        with pytest.raises(F.LamaError, match="lama_hip_ctx_create failed"):
            F.HipContext(F.default_cfg(particles=2, **bad))
    # calls before init / with bad arguments
    ctx = F.HipContext(F.default_cfg(particles=2))
    with pytest.raises(F.LamaError, match="before"):
        ctx.scan_match(pts[0])
    ctx.init(pts[0], pose0)
    with pytest.raises(F.LamaError):
        ctx.download_map(5, F.MAP_DISTANCE)          # particle out of range
    # a map update that was only queued reports its error at the next synchronising call
    far = np.tile(O.se2(3000.0, 2.0, 0.0), (2, 1))   # 3 km away: the mapped area would be wider than the largest window (1016 patches = 1.6 km)
    ctx.set_poses(far)
    ctx.update_maps_begin(pts[0])                    # returns before the kernels have run
    with pytest.raises(F.LamaError, match=r"deferred from lama_hip_pf_update_maps_begin.*window"):
        ctx.sync()
    ctx.close()
    # ... and begin + sync is the same as the synchronous call
    a, b = F.HipContext(F.default_cfg(particles=2)), F.HipContext(F.default_cfg(particles=2))
    a.init(pts[0], pose0); b.init(pts[0], pose0)
    a.update_maps(pts[1])
    b.update_maps_begin(pts[1])
    assert_maps_equal(b.download_map(1, F.MAP_DISTANCE), a.download_map(1, F.MAP_DISTANCE), DM_FIELDS, "begin/sync")   # download collects the status
    a.close(); b.close()


def test_slam2d_transient_map_gpu_vs_oracle(F):
    """Slam2D with transient_map (src/slam2d.cpp:322-379) on the device: patch deletion keeps both maps bit-exact."""
    steps = 30
    pts, odom, truth = F.corridor_log(steps, 1080)
    kw = dict(transient_map=True, truncated_range=3.5)
    o = O.Slam(**kw)
    h = F.Slam2D(**kw)
    assert h.engine_origin().endswith("liblama_hip.so")
    o.set_pose(O.se2(*odom[0]))
    h.set_pose(*odom[0])
    deleted = 0
    for k in range(steps + 1):
        p = pts[k][np.hypot(pts[k][:, 0], pts[k][:, 1]) < 4.0]
        assert o.update(p, O.se2(*odom[k]), float(k)) == h.update(p, odom[k], float(k))
        assert np.abs(o.pose() - h.pose()).max() < 1e-7, k
        assert o.deleted_last() == h.deleted_patches(), k
        deleted += h.deleted_patches()
    assert deleted > 0
    ctx = h.hip_context()
    assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), o.occ().dump(), OCC_FIELDS, "transient occ")
    assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), o.dm().dump(), DM_FIELDS, "transient dm")
    h.close()


def test_lm_strategy_gpu_vs_oracle(F):
    """Levenberg-Marquardt strategy (Slam2D / Loc2D Options::strategy = "lm") on the device against the oracle: poses within
    1e-7, identical iteration counts (applied + rejected steps), free running."""
    from _worlds import corridor_obstacles
    steps = 10
    pts, odom, truth = F.corridor_log(steps, 1080)
    o = O.Slam()
    o.set_lm(True)
    h = F.Slam2D(lm=1)
    o.set_pose(O.se2(*odom[0]))
    h.set_pose(*odom[0])
    flips = 0
    for k in range(steps + 1):
        assert o.update(pts[k], O.se2(*odom[k]), float(k)) == h.update(pts[k], odom[k], float(k))
        assert np.abs(o.pose() - h.pose()).max() < 1e-6, (k, o.pose(), h.pose())
        flips += int(o.iterations() != h.iterations())
    assert flips <= 1                          # LM's gain ratio test may flip on a last-ulp difference; never observed > 0
    obst = corridor_obstacles()
    ol = O.Loc()
    ol.set_lm(True)
    dm = ol.dm()
    for x, y in obst:
        c = O.w2m([x, y, 0.0])
        dm.add(int(c[0]), int(c[1]))
    dm.update()
    hl = F.Loc2D(strategy="lm")
    hl.set_obstacles_world(obst)
    start = truth[0] + np.array([0.05, -0.04, 0.01])
    ol.set_pose(O.se2(*start))
    hl.set_pose(*start)
    for k in range(steps + 1):
        assert ol.update(pts[k], O.se2(*odom[k]), float(k), force=(k == 0)) == hl.update(pts[k], odom[k], float(k), force=(k == 0))
        assert np.abs(ol.pose() - hl.pose()).max() < 1e-6, k
        assert abs(ol.rmse() - hl.rmse()) < 1e-8
    print("lm iteration-count flips:", flips)
    hl.close()
    h.close()


from _stress import random_room_scan as _random_room_scan, random_rooms_case      # (shared with the lane-simulator tests)


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("LAMA_STRESS_SEEDS", "12")))))
def test_randomized_rooms_maps_bit_exact(F, seed):
    """Randomised stress of the exactness claim: random star-shaped rooms (radius 2.5 - 14 m), random beam counts, ray
    truncation options and particle counts, 4 scans from perturbed poses with resampling in between -- occupancy and distance
    maps must stay bit-identical to the oracle (libstdc++ tie order included) in both ray-cast forms and both brushfire forms."""
    random_rooms_case(F, seed)


@pytest.mark.parametrize("order", ["reversed", "shuffled", "two_sweeps", "ragged"])
def test_point_order_is_free_for_the_ray_cast(F, order):
    """k_ray_patches skips whole chunks of 64 consecutive beams by their cone and box; the cone exists only when the beams of a
    chunk fan out in angular order.  The reference accepts ANY point order (PointCloudXYZ is just a list): clockwise scans, shuffled
    points, two sweeps in one cloud and beam counts that are no multiple of 64 must all give the oracle's maps bit for bit."""
    rng = np.random.default_rng(77)
    kind = {"R": 7.0, "coef": [(m, rng.uniform(0.02, 0.12), rng.uniform(0, 2 * np.pi)) for m in (2, 3, 5, 7)]}
    n_beams = {"ragged": 333}.get(order, 720)
    P = 3

    def reorder(sc):
        if order == "reversed":
            return sc[::-1].copy()
        if order == "shuffled":
            return sc[rng.permutation(len(sc))]
        if order == "two_sweeps":
            return np.concatenate([sc[0::2], sc[1::2]])
        return sc

    base = np.array([0.4, -0.3, 0.7])
    pf = O.PF(O.default_options(particles=P, seed=5))
    scan0 = reorder(_random_room_scan(rng, base, n_beams, kind))
    pose0 = O.se2(*base)
    pf.set_prior(pose0)
    assert pf.update(scan0, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, dm_patch_capacity=1024, occ_patch_capacity=1024))
    ctx.init(scan0, pose0)
    for k in range(3):
        truth = base + np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3)])
        scan = reorder(_random_room_scan(rng, truth, n_beams, kind))
        poses = np.stack([O.se2(*(truth + rng.normal(0, [0.02, 0.02, 0.01]))) for _ in range(P)])
        pf.set_poses(poses)
        pf.stage_set_scan(scan)
        pf.stage_update_maps()
        ctx.set_poses(poses)
        ctx.update_maps(scan)
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"{order} scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"{order} scan {k} dm p{i}")
        base = truth
    assert ctx.counters()["parallel_raycast_scans"] == 4
    ctx.close()


@pytest.mark.parametrize("n_beams,parallel", [(4096, True), (3000, True), (5000, False)])
def test_scans_with_thousands_of_points(F, n_beams, parallel):
    """scans with more than 2,048 points used to drop silently to the one-wave ray-cast.  The visit key now carries the
    beam index in 13 bits and k_ray_patches keeps 64 chunk records per wave: up to 4,096 points stay on the parallel, patch-centric
    form (asserted through the counters); beyond that the beam-sequential form takes over.  Bit-exact either way."""
    rng = np.random.default_rng(5)
    kind = {"R": 6.0, "coef": [(m, rng.uniform(0.02, 0.1), rng.uniform(0, 2 * np.pi)) for m in (2, 3, 5)]}
    P = 2
    base = np.array([0.2, 0.1, -0.4])
    pf = O.PF(O.default_options(particles=P, seed=5))
    scan0 = _random_room_scan(rng, base, n_beams, kind)
    pose0 = O.se2(*base)
    pf.set_prior(pose0)
    assert pf.update(scan0, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, dm_patch_capacity=1024, occ_patch_capacity=1024))
    ctx.init(scan0, pose0)
    for k in range(2):
        truth = base + np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2)])
        scan = _random_room_scan(rng, truth, n_beams, kind)
        poses = np.stack([O.se2(*(truth + rng.normal(0, [0.02, 0.02, 0.01]))) for _ in range(P)])
        pf.set_poses(poses); pf.stage_set_scan(scan); pf.stage_update_maps()
        ctx.set_poses(poses); ctx.update_maps(scan)
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"{n_beams} beams scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"{n_beams} beams scan {k} dm p{i}")
        base = truth
    c = ctx.counters()
    assert (c["parallel_raycast_scans"], c["sequential_raycast_scans"]) == ((3, 0) if parallel else (0, 3)), c
    ctx.close()


def test_dynamic_scene_moving_box(F):
    """every GPU world so far was static (removals came from pose noise only).  Here a 1 m box crosses a 10 m room,
    0.2 m per scan for 20 scans, in front of a standing scanner: every scan frees the cells the box left (removeObstacle -> long
    raise waves that clear and re-lower whole disks) and occupies new ones.  Occupancy and distance maps bit-exact after every scan."""
    from _worlds import segment_world_scan
    P = 2
    room = [(-5.0, -5.0, 5.0, -5.0), (5.0, -5.0, 5.0, 5.0), (5.0, 5.0, -5.0, 5.0), (-5.0, 5.0, -5.0, -5.0)]

    def world(k):
        bx = -3.0 + 0.2 * k
        box = [(bx, 1.0, bx + 1.0, 1.0), (bx + 1.0, 1.0, bx + 1.0, 2.0), (bx + 1.0, 2.0, bx, 2.0), (bx, 2.0, bx, 1.0)]
        return np.array(room + box)
    rng = np.random.default_rng(3)
    pose = O.se2(0.0, -1.0, np.pi / 2)
    scan0 = segment_world_scan(world(0), 0.0, -1.0, np.pi / 2, beams=1080, max_range=30.0, noise=rng.normal(0.0, 0.005, 1080))
    pf = O.PF(O.default_options(particles=P, seed=5))
    pf.set_prior(pose)
    assert pf.update(scan0, pose)
    ctx = F.HipContext(F.default_cfg(particles=P))
    ctx.init(scan0, pose)
    raise_pops = 0
    for k in range(1, 21):
        scan = segment_world_scan(world(k), 0.0, -1.0, np.pi / 2, beams=1080, max_range=30.0, noise=rng.normal(0.0, 0.005, 1080))
        poses = _perturbed(rng, pose, P, sxy=0.004, sth=0.001)
        pf.set_poses(poses); pf.stage_set_scan(scan); pf.stage_update_maps()
        ctx.set_poses(poses); ctx.reset_counters(); ctx.update_maps(scan)
        raise_pops += ctx.counters()["bf_cells"]
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    assert raise_pops > 20 * P * 500                 # the brushfire really had work (a static scene settles to a few hundred cells per scan)
    ctx.close()


@pytest.mark.parametrize("seed", list(range(8)))
def test_randomized_rooms_scan_match_parity(F, seed):
    """Scan matching in random rooms: from start poses up to 15 cm / 5 deg off, device and oracle must reach the same pose
    (<= 1e-8) with the same number of Gauss-Newton iterations and the same log-likelihood (rel 1e-9)."""
    rng = np.random.default_rng(5000 + seed)
    kind = {"R": rng.uniform(3.0, 12.0), "coef": [(m, rng.uniform(0.02, 0.12), rng.uniform(0, 2 * np.pi)) for m in (2, 3, 5, 7)]}
    n_beams = int(rng.choice([360, 720, 1080]))
    P = 16
    base = np.array([rng.uniform(-0.2, 0.2) * kind["R"], rng.uniform(-0.2, 0.2) * kind["R"], rng.uniform(-np.pi, np.pi)])
    pf = O.PF(O.default_options(particles=P, seed=seed + 1))
    scan0 = _random_room_scan(rng, base, n_beams, kind)
    pose0 = O.se2(*base)
    pf.set_prior(pose0)
    assert pf.update(scan0, pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, dm_patch_capacity=1024, occ_patch_capacity=1024))
    ctx.init(scan0, pose0)
    truth = base + np.array([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), rng.uniform(-0.2, 0.2)])
    scan = _random_room_scan(rng, truth, n_beams, kind)
    start = np.stack([O.se2(*(truth + rng.uniform(-1, 1, 3) * [0.15, 0.15, 0.09])) for _ in range(P)])
    pf.set_poses(start)
    pf.set_weights(w=np.zeros(P), ws=np.zeros(P))
    pf.stage_set_scan(scan)
    pf.stage_scan_match()
    o_poses, o_ll = pf.poses(), pf.weights()[0]
    o_it = np.array([pf.counters(i)["iterations"] for i in range(P)])
    ctx.set_poses(start)
    g_poses, g_ll, g_it = ctx.scan_match(scan)
    same = g_it == o_it
    assert (~same).sum() <= 1, (g_it, o_it)              # a stop test may flip on a last-ulp difference; poses then differ by one step
    assert np.abs(g_poses[same] - o_poses[same]).max() < 1e-8
    assert np.allclose(g_ll[same], o_ll[same], rtol=1e-9, atol=0)
    ctx.close()
