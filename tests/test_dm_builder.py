"""The host facade's FIRST build of Loc2D's distance map (iris_lama_amd/host/dm_builder.cpp: addObstacle x N on an empty map + one
update(), replayed on the host and uploaded to the device) against the checker: every byte of every distance_t record -- obstacle
offsets in tie cells included --, the Container masks, the patch set and update()'s return value.  CPU only; the device side
(upload, later incremental updates on the uploaded map) is tests/test_gpu_parity.py::test_loc2d_*."""
import numpy as np
import pytest

import _oracle as O
import _worlds
from _cmp import DM_FIELDS, assert_maps_equal

import iris_lama_amd.ffi as F

OFF = (2642244 >> 1) * 32          # Map's origin offset in cells (src/sdm/map.cpp:55-58)


def _oracle_build(cells, l2_max):
    dm = O.DM.new(l2_max=l2_max)
    for x, y in cells:
        dm.add(int(x), int(y))
    n = dm.update()
    return n, dm


def _check(cells, l2_max):
    n, dm = _oracle_build(cells, l2_max)
    got = F.dm_build(cells, dm.max_sqdist())
    assert got is not None
    assert got[0] == n, (got[0], n)
    assert_maps_equal(got[1], dm.dump(), DM_FIELDS, f"l2_max {l2_max}")
    return n, len(got[1])


@pytest.mark.parametrize("l2_max", [0.5, 1.0, 2.0, 7.0])
def test_first_build_of_the_corridor_map_equals_the_oracle(l2_max):
    """the corridor world of the Loc2D tests: walls and pillars, in the order Loc2D's caller adds them"""
    pts = _worlds.corridor_obstacles()
    cells = np.stack([np.floor(pts[:, 0] / 0.05 + OFF + 0.5), np.floor(pts[:, 1] / 0.05 + OFF + 0.5)], axis=1).astype(np.uint32)
    n, patches = _check(cells, l2_max)
    assert n > len(cells) and patches > 20


@pytest.mark.parametrize("seed", range(6))
def test_first_build_of_random_obstacle_sets_equals_the_oracle(seed):
    """random clutter: duplicates, single cells, diagonal lines (the tie-heavy case), clusters straddling patch borders, shuffled order"""
    rng = np.random.default_rng(seed)
    cells = []
    for _ in range(int(rng.integers(3, 9))):
        x0, y0 = rng.integers(-90, 90, size=2)
        kind = int(rng.integers(0, 4))
        m = int(rng.integers(1, 60))
        if kind == 0:
            cells += [(x0 + k, y0) for k in range(m)]
        elif kind == 1:
            cells += [(x0 + k, y0 + k) for k in range(m)]
        elif kind == 2:
            cells += [(x0 + int(dx), y0 + int(dy)) for dx, dy in rng.integers(-6, 7, size=(m, 2))]
        else:
            cells += [(x0 + k, y0 - 2 * k) for k in range(m)]
    cells = np.array(cells, dtype=np.int64) + OFF
    cells = np.concatenate([cells, cells[rng.integers(0, len(cells), size=5)]])      # some cells twice
    rng.shuffle(cells)
    _check(cells.astype(np.uint32), [0.5, 1.0, 1.5][seed % 3])


def test_first_build_of_a_floor_plan_with_many_equal_priorities():
    """rooms with walls two cells thick (thousands of queue entries of equal priority at every level): the pop order among them is
    libstdc++'s heap order, which the builder has by using std::priority_queue itself"""
    W, H, step, door = 360, 240, 80, 20
    occ = np.zeros((H, W), dtype=bool)
    for x in range(0, W, step):
        occ[:, x:x + 2] = True
    for y in range(0, H, step):
        occ[y:y + 2, :] = True
    occ[:, W - 2:] = True
    occ[H - 2:, :] = True
    for x in range(step, W - step, step):
        for y in range(0, H - step, step):
            occ[y + step // 2 - door // 2:y + step // 2 + door // 2, x:x + 2] = False
    ys, xs = np.nonzero(occ)
    cells = np.stack([xs + OFF, ys + OFF], axis=1).astype(np.uint32)
    n, patches = _check(cells, 1.0)
    assert n > 5 * len(cells)


def test_the_host_does_not_build_what_the_device_plane_cannot_hold():
    assert F.dm_build(np.array([[OFF, OFF]], dtype=np.uint32), 65026) is None           # beyond 255 cells: the end of the reference's uint16_t sqdist and of the wide device library
    assert F.dm_build(np.array([[OFF, OFF]], dtype=np.uint32), 65025) is not None
    assert F.dm_build(np.zeros((0, 2), dtype=np.uint32), 100) is None
