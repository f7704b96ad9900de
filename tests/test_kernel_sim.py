"""The kernel SOURCES of iris_lama_amd/csrc executed lane by lane on the CPU (tests/sim: every thread a fiber, wave64 rendezvous for
ballot / shuffle / readlane / DPP, workgroup barriers) and compared with the oracle.  This is test infrastructure for the build
container, which has no GPU: it checks the kernels' LOGIC (the same C++ the device compiler gets, compiled for the host against
tests/sim/hip/hip_runtime.h), not their timing, memory ordering or generated code -- the `-m gpu` tests on the MI355X remain the
parity tests proper.  Small cases only (a particle-scan takes seconds here)."""
import os
import subprocess

import numpy as np
import pytest

import _oracle as O
from _cmp import DM_FIELDS, OCC_FIELDS, assert_maps_equal

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_LIB = os.path.join(HERE, "sim", "_build", "liblama_hip_sim.so")
SIM_LIB_WIDE = os.path.join(HERE, "sim", "_build", "liblama_hip_sim_wide.so")
SIM_LIB_SMALLQ = os.path.join(HERE, "sim", "_build", "liblama_hip_sim_smallq.so")      # first-stage brushfire queues of 320 / 80 entries


def _with_lib(path):
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "sim")], check=True)
    import iris_lama_amd.ffi as F
    saved, saved_lib = F.HIP_LIB, getattr(F, "_hip", None)
    F.HIP_LIB = path
    F._hip = None
    yield F
    F.HIP_LIB, F._hip = saved, saved_lib


@pytest.fixture()
def Fsim():
    yield from _with_lib(SIM_LIB)


@pytest.fixture()
def Fsim_smallq():
    yield from _with_lib(SIM_LIB_SMALLQ)


@pytest.fixture()
def Fsim_wide():
    """the sources compiled with -DLAMA_WIDE_DM (liblama_hip_wide.so on the device): l2_max of 128 .. 255 cells"""
    for F in _with_lib(SIM_LIB):
        saved, saved_lib = F.HIP_LIB_WIDE, F._hip_wide
        F.HIP_LIB_WIDE, F._hip_wide = SIM_LIB_WIDE, None
        yield F
        F.HIP_LIB_WIDE, F._hip_wide = saved, saved_lib


def _run(F, P, steps, **cfg):
    pts, odom, truth = F.corridor_log(steps, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=P, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, device=0, **cfg))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(0)
    for k in range(1, steps + 1):
        start = np.stack([O.se2_mul(O.se2(*truth[k]), O.se2(*rng.normal(0, [0.03, 0.03, 0.01]))) for _ in range(P)])
        pf.set_poses(start)
        pf.set_weights(w=np.zeros(P), ws=np.zeros(P))
        pf.stage_set_scan(pts[k])
        pf.stage_scan_match()
        ctx.set_poses(start)
        g_poses, g_ll, g_it = ctx.scan_match(pts[k])
        o_poses, o_ll = pf.poses(), pf.weights()[0]
        assert np.abs(g_poses - o_poses).max() < 1e-8
        assert np.allclose(g_ll, o_ll, rtol=1e-9)
        if k == 1 and P > 1:                         # a resample in between: particle copies
            idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
            pf.stage_resample_with(idx)
            ctx.resample(idx)
            o_poses = pf.poses()
        ctx.set_poses(o_poses)
        pf.stage_update_maps()
        ctx.update_maps(pts[k])
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    c = ctx.counters()
    ctx.close()
    return c


def test_default_kernels_under_the_lane_simulator(Fsim):
    """Patch-centric parallel ray-cast, fused scan match, wave-pair exact brushfire: bit-exact maps, poses within 1e-8."""
    c = _run(Fsim, 2, 2)
    assert c["parallel_raycast_scans"] == 3 and c["brushfire_waves"] == 2 and c["brushfire_mode"] == 0


def test_sequential_raycast_and_one_wave_brushfire_under_the_lane_simulator(Fsim):
    c = _run(Fsim, 1, 1, sequential_raycast=1, brushfire_waves=1)
    assert c["sequential_raycast_scans"] == 2 and c["brushfire_waves"] == 1


@pytest.mark.parametrize("waves", [2, 1])
def test_hand_over_to_the_resume_stage_in_the_middle_of_an_update(Fsim_smallq, waves):
    """First-stage queues of 320 / 80 entries: particles outgrow them while the brushfire runs and are handed, state intact, to the
    resume stage (wave-pair and one-wave first stage)."""
    _run(Fsim_smallq, 2, 2, brushfire_waves=waves)


def test_long_chains_are_routed_to_the_big_queue_stage(Fsim, monkeypatch):
    """k_bf_route: the particles with the most obstacle events go to the big-queue brushfire stage before the first stage starts
    (which skips them).  Forced here on two particles (threshold = 90 % of the mean, one place): in every update one of the two
    runs its whole brushfire in the big-queue stage -- maps bit-exact."""
    monkeypatch.setenv("LAMA_HIP_BF_ROUTE", "1,0,90,1")
    c = _run(Fsim, 2, 2)
    assert c["brushfire_routed"] >= 2 and c["brushfire_waves"] == 2, c
    # ... and a particle routed in one update runs the next one in the EARLY lane (its modifying ray-cast kernels and its brushfire on
    # a stream of their own, the main lane skips it) as long as no resample has moved the particles in between
    assert c["brushfire_early"] >= 1, c


@pytest.mark.parametrize("seq_ray", [0, 1])
def test_update_that_runs_out_of_patches_is_repeated_after_growth(Fsim, seq_ray):
    """Arenas of 8 patches against a first scan that needs ~55 / ~70: the allocation phase (hit cells, ray patches, the bound on the
    distance-map patches still to come) fails BEFORE any cell is modified, the host doubles the arenas and runs the update again --
    maps bit-exact, several growths, both ray-cast forms."""
    c = _run(Fsim, 1, 1, dm_patch_capacity=8, occ_patch_capacity=8, sequential_raycast=seq_ray)
    assert c["arena_growths"] >= 3, c


def test_in_place_resampling_with_regions_of_different_sizes(Fsim):
    """Round 5: ONE particle set.  A resample is a permutation of the particle table: survivors keep their home and regions, only
    the further copies of a multiply drawn particle are copied into the homes of the particles that died.  Regions of 8 patches to
    start with, so the particles' regions are moved, grown and recycled through the allocator all the time; three resamples
    (a dead particle in front of / behind its source, one source drawn three times, the identity) -- maps bit-exact against the
    oracle after every update, and only the clones were copied."""
    F, P, steps = Fsim, 4, 3
    pts, odom, truth = F.corridor_log(steps, 360)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=P, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, device=0, dm_patch_capacity=8, occ_patch_capacity=8))
    ctx.init(pts[0], pose0)
    c0 = ctx.counters()
    assert c0["resample_clones"] == P - 1 and c0["hbm_bytes_used"] <= c0["hbm_bytes_allocated"], c0
    rng = np.random.default_rng(5)
    plans = {1: [1, 1, 3, 3], 2: [0, 2, 2, 2], 3: [0, 1, 2, 3]}
    clones = P - 1
    for k in range(1, steps + 1):
        # particles spread over the corridor: their maps (and region sizes) differ
        start = np.stack([O.se2_mul(O.se2(*truth[k]), O.se2(*rng.normal(0, [0.4, 0.1, 0.05]))) for _ in range(P)])
        idx = np.array(plans[k], dtype=np.int32)
        pf.set_poses(start)
        pf.stage_resample_with(idx)
        ctx.set_poses(start)
        ctx.resample(idx)
        clones += P - len(set(plans[k]))
        assert np.array_equal(ctx.get_poses(), pf.poses())
        pf.stage_set_scan(pts[k])
        pf.stage_update_maps()
        ctx.update_maps(pts[k])
        for i in range(P):
            assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"scan {k} occ p{i}")
            assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"scan {k} dm p{i}")
    c = ctx.counters()
    assert c["resample_clones"] == clones, (c["resample_clones"], clones)
    assert c["arena_growths"] >= 2 and c["hbm_bytes_used"] <= c["hbm_bytes_allocated"], c
    ctx.close()


def test_uploaded_map_replaces_a_particles_map(Fsim):
    """lama_hip_pf_upload_map (Map::read for a device map): particle 1's maps are replaced by particle 0's downloaded ones -- a
    smaller map over a larger one, so slots are re-zeroed --, the uploaded maps equal the oracle's and stay live."""
    F, P = Fsim, 2
    pts, odom, truth = F.corridor_log(2, 360)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=P, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, device=0, dm_patch_capacity=8, occ_patch_capacity=8))
    ctx.init(pts[0], pose0)
    # particle 1 maps further down the corridor: more patches than particle 0
    poses = np.stack([O.se2(*truth[0]), O.se2(truth[1][0] + 4.0, truth[1][1], 0.0)])
    pf.set_poses(poses); pf.stage_set_scan(pts[1]); pf.stage_update_maps()
    ctx.set_poses(poses); ctx.update_maps(pts[1])
    d0, o0 = ctx.download_map(0, F.MAP_DISTANCE), ctx.download_map(0, F.MAP_OCCUPANCY)
    assert len(ctx.download_map(1, F.MAP_OCCUPANCY)) > len(o0)
    ctx.upload_map(1, F.MAP_DISTANCE, d0); ctx.upload_map(1, F.MAP_OCCUPANCY, o0)
    pf.stage_resample_with(np.array([0, 0], dtype=np.int32))
    for i in range(P):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"dm p{i}")
    nxt = np.stack([O.se2(*truth[2]), O.se2(truth[2][0] + 0.3, truth[2][1] - 0.1, 0.02)])
    pf.set_poses(nxt); pf.stage_set_scan(pts[2]); pf.stage_update_maps()
    ctx.set_poses(nxt); ctx.update_maps(pts[2])
    for i in range(P):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"after the update: occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"after the update: dm p{i}")
    ctx.close()


def test_batched_export_and_import_between_two_contexts(Fsim):
    """The resample of a sharded pool: all outgoing particles of one context leave in ONE export launch, arrive in another context
    in ONE import launch (two slots take the same blob), and the receiving context carries on -- maps bit-exact against the oracle
    that resampled in place."""
    F, P = Fsim, 3
    pts, odom, truth = F.corridor_log(2, 1080)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=P, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    a = F.HipContext(F.default_cfg(particles=P, device=0))
    b = F.HipContext(F.default_cfg(particles=P, device=0))
    a.init(pts[0], pose0)
    b.init(pts[0], pose0)
    rng = np.random.default_rng(1)
    start = np.stack([O.se2_mul(O.se2(*truth[1]), O.se2(*rng.normal(0, [0.05, 0.05, 0.02]))) for _ in range(P)])
    pf.set_poses(start); pf.stage_set_scan(pts[1]); pf.stage_update_maps()
    a.set_poses(start); a.update_maps(pts[1])
    src = np.array([2, 0], dtype=np.uint32)
    sizes = a.export_sizes(src)
    bufs = [np.zeros(int(n), dtype=np.uint8) for n in sizes]
    got = a.export_particles(src, [x.ctypes.data for x in bufs], sizes)
    assert np.array_equal(got, sizes)
    for k, p in enumerate(src):                               # the batch writes what the single-particle call writes
        one = np.zeros(int(sizes[k]), dtype=np.uint8)
        a.export_particle(int(p), one.ctypes.data, int(sizes[k]))
        assert np.array_equal(one, bufs[k])
    b.import_particles([0, 1, 2], [bufs[0].ctypes.data, bufs[0].ctypes.data, bufs[1].ctypes.data], [sizes[0], sizes[0], sizes[1]])
    pf.stage_resample_with(np.array([2, 2, 0], dtype=np.int32))
    assert np.array_equal(b.get_poses(), pf.poses())
    for rnd in range(2):
        for i in range(P):
            assert_maps_equal(b.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"round {rnd} occ p{i}")
            assert_maps_equal(b.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"round {rnd} dm p{i}")
        if rnd == 0:                                          # ... and the imported particles are live: one more update
            nxt = np.stack([O.se2_mul(O.se2(*truth[2]), O.se2(*rng.normal(0, [0.05, 0.05, 0.02]))) for _ in range(P)])
            pf.set_poses(nxt); pf.stage_set_scan(pts[2]); pf.stage_update_maps()
            b.set_poses(nxt); b.update_maps(pts[2])
    a.close(); b.close()


def test_shuffled_points_on_the_simulator(Fsim):
    """A point cloud in random order: the chunks of 64 beams have no cone, the patch kernel tests them beam by beam."""
    F = Fsim
    pts, odom, truth = F.corridor_log(1, 360)
    rng = np.random.default_rng(4)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=1, seed=3))
    pf.set_prior(pose0)
    s0 = pts[0][rng.permutation(len(pts[0]))]
    pf.update(s0, pose0)
    ctx = F.HipContext(F.default_cfg(particles=1, device=0))
    ctx.init(s0, pose0)
    s1 = pts[1][rng.permutation(len(pts[1]))]
    start = O.se2(*truth[1])[None]
    pf.set_poses(start); pf.stage_set_scan(s1); pf.stage_update_maps()
    ctx.set_poses(start); ctx.update_maps(s1)
    assert_maps_equal(ctx.download_map(0, F.MAP_OCCUPANCY), pf.occ(0).dump(), OCC_FIELDS, "occ")
    assert_maps_equal(ctx.download_map(0, F.MAP_DISTANCE), pf.dm(0).dump(), DM_FIELDS, "dm")
    ctx.close()


@pytest.mark.parametrize("seed", list(range(10)))
def test_randomized_rooms_on_the_simulator(Fsim, seed):
    """The randomised exactness case of the GPU suite (tests/_stress.py) at simulator size: random rooms, beam counts, truncation
    options, ray-cast and brushfire forms; perturbed poses re-draw walls a cell off, so the raise wave runs as well as the lower
    wave; a resample in between.  (Possible since round 6: the simulator's fiber switch no longer makes a system call.  Seeds 2 and 4
    -- the one-wave brushfire in a small room -- found a gap of the SIMULATOR, not of the kernels: a wavefront-scope fence has to be a
    rendezvous of the fibers, tests/sim/hip/hip_runtime.h; 40 of 40 such cases are exact on the device.)"""
    from _stress import random_rooms_case
    random_rooms_case(Fsim, seed, small=True)


def test_seventy_particles_outgrow_their_regions_together(Fsim):
    """set_capacities moves the particles that outgrow their regions in groups of 64, ordered by the stream, and a region
    released by one group may be handed to the next.  70 particles with regions of 8 patches, a resample with many clones, then an
    update that makes all of them grow (two groups, regions recycled through the allocator) -- the device-side checksums of both maps
    of every particle equal the oracle's, a sample is compared cell by cell."""
    F, P = Fsim, 70
    pts, odom, truth = F.corridor_log(1, 36)                     # (sparse scans: the simulator pays ~0.3 ms per brushfire pop)
    pose0 = O.se2(*odom[0])
    pf = O.PF(O.default_options(particles=P, seed=3))
    pf.set_prior(pose0)
    pf.update(pts[0], pose0)
    ctx = F.HipContext(F.default_cfg(particles=P, device=0, dm_patch_capacity=8, occ_patch_capacity=8))
    ctx.init(pts[0], pose0)
    rng = np.random.default_rng(2)
    poses = np.stack([O.se2_mul(O.se2(*truth[1]), O.se2(*rng.normal(0, [0.3, 0.05, 0.03]))) for _ in range(P)])
    pf.set_poses(poses); ctx.set_poses(poses)
    idx = np.sort(rng.integers(0, P, size=P)).astype(np.int32)
    pf.stage_resample_with(idx); ctx.resample(idx)
    pf.stage_set_scan(pts[1]); pf.stage_update_maps()
    ctx.update_maps(pts[1])
    assert np.array_equal(ctx.map_checksums(F.MAP_DISTANCE), pf.map_checksums(0)), "distance maps"
    assert np.array_equal(ctx.map_checksums(F.MAP_OCCUPANCY), pf.map_checksums(1)), "occupancy maps"
    for i in (0, 33, 64, 69):
        assert_maps_equal(ctx.download_map(i, F.MAP_OCCUPANCY), pf.occ(i).dump(), OCC_FIELDS, f"occ p{i}")
        assert_maps_equal(ctx.download_map(i, F.MAP_DISTANCE), pf.dm(i).dump(), DM_FIELDS, f"dm p{i}")
    c = ctx.counters()
    assert c["arena_growths"] >= 1 and c["resample_clones"] > P - 1, c
    ctx.close()


@pytest.mark.parametrize("seed,l2_max", [(0, 6.6), (7, 8.0)])
def test_wide_build_randomized_small_rooms(Fsim_wide, seed, l2_max):
    """The wide build of the kernels (4-byte distance plane, 9-bit obstacle offsets in the queue entries) on the lane simulator: a
    distance map that reaches 132 / 160 cells floods the whole small room on every scan (seeds 1 - 5 at 255 cells, 1.5 - 3 minutes each,
    were run by hand when the build was made: exact); occupancy and distance maps of every
    particle after every scan against the oracle, bit for bit (the oracle itself is pinned against the reference's build in this
    range by tests/test_oracle_vs_reference.py::test_dynamic_brushfire_beyond_127_cells)."""
    from _stress import random_rooms_case
    assert Fsim_wide.needs_wide(l2_max, 0.05)
    random_rooms_case(Fsim_wide, seed, small=True, l2_max=l2_max)
