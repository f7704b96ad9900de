"""Pins the CPU oracle against the source-derived known answers of SURVEY.md Appendix A.9
(the reference ships no tests / golden vectors, so these are the only available pins) and
against brute-force self-consistency checks."""
import math

import numpy as np

import _oracle as O

OFF = 42275904  # (2642244 >> 1) * 32, include/lama/sdm/map.h:68 + src/sdm/map.cpp:55-58


def test_kat1_world_origin_addressing():
    m = O.w2m([0.0, 0.0, 0.0])
    assert m.tolist() == [OFF, OFF, OFF]
    c = np.array([OFF, OFF, OFF], dtype=np.uint32)
    assert O.lib().orc_m2p(0.05, 32, O._p(c)) == 1321122 * 2642244 + 1321122 == 3490727998890
    assert O.lib().orc_m2c(0.05, 32, O._p(c)) == 0


def test_kat2_record_sizes():
    assert O.lib().orc_sizeof_distance_t() == 10
    assert O.lib().orc_sizeof_frequency() == 4
    dm = O.DM.new()
    dm.add(OFF, OFF)
    cells, mask = dm.patch(dm.patch_ids()[0])
    assert cells.nbytes == 10240 and mask.size == 16


def test_kat3_max_distance():
    assert O.DM.new(l2_max=0.5).max_sqdist() == 100
    assert O.DM.new(l2_max=1.0).max_sqdist() == 400
    assert O.DM.new(l2_max=0.5).distance_cell(OFF, OFF) == math.sqrt(100.0) * 0.05   # unknown cell
    assert O.DM.new(l2_max=1.0).distance_cell(OFF, OFF) == math.sqrt(400.0) * 0.05


def _cells_by_coord(dm):
    out = {}
    for pid, (cells, mask) in dm.dump().items():
        ax, ay = (pid // 2642244) << 5, (pid % 2642244) << 5
        for ci in range(1024):
            if (int(mask[ci >> 6]) >> (ci & 63)) & 1:
                out[(ax + (ci & 31), ay + (ci >> 5))] = cells[ci]
    return out


def test_kat4_single_obstacle_disk():
    dm = O.DM.new()
    o = (OFF + 7, OFF + 40)  # not patch aligned on purpose
    dm.add(*o)
    dm.update()
    cells = _cells_by_coord(dm)
    valid = {k: v for k, v in cells.items() if v["valid"]}
    expect = {(dx, dy) for dx in range(-10, 11) for dy in range(-10, 11) if dx * dx + dy * dy < 100}
    assert len(expect) == 305
    assert {(x - o[0], y - o[1]) for (x, y) in valid} == expect
    for (x, y), c in valid.items():
        dx, dy = x - o[0], y - o[1]
        assert c["sqdist"] == dx * dx + dy * dy
        assert c["obstacle"].tolist() == [-dx, -dy, 0]
        assert c["queued"] == 0
    # every other touched cell is invalid
    assert all(not v["valid"] for k, v in cells.items() if k not in valid)


def test_kat5_bilinear_centre_and_ramp_gradient():
    dm = O.DM.new()
    # vertical wall x = OFF+20: distance field is a ramp in x: v = |i - 20| cells
    for y in range(-40, 41):
        dm.add(OFF + 20, OFF + y)
    dm.update()
    # exactly on a cell centre -> that cell's value (world = (cell - OFF)*res)
    for i in (12, 15, 19):
        d = dm.distance([i * 0.05, 0.0, 0.0])
        assert d == dm.distance_cell(OFF + i, OFF)
        assert abs(d - (20 - i) * 0.05) < 1e-15
    # gradient of a linear ramp v = a*i (a = -res per cell moving +x) is (a*scale, 0) = (-1, 0)
    d, g = dm.distance([15.3 * 0.05, 0.2 * 0.05, 0.0], grad=True)
    assert abs(g[0] - (-1.0)) < 1e-9 and abs(g[1]) < 1e-9 and g[2] == 0
    # fp64 ulp at map coordinate 4.2e7 is 7.5e-9 cells -> ~4e-10 m of inherent noise
    assert abs(d - (20 - 15.3) * 0.05) < 1e-9


def test_kat6_compute_ray():
    assert O.compute_ray([0, 0, 0], [5, 0, 0])[:, :2].tolist() == [[1, 0], [2, 0], [3, 0], [4, 0]]
    assert O.compute_ray([0, 0, 0], [3, 3, 0])[:, :2].tolist() == [[1, 1], [2, 2]]
    assert len(O.compute_ray([0, 0, 0], [1, 0, 0])) == 0
    assert len(O.compute_ray([4, 4, 0], [4, 4, 0])) == 0
    # closed form used by the device kernel: steps on axis j after t iterations = floor((2 t d_j + n) / (2 n))
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(100, 400, size=3)
        b = rng.integers(100, 400, size=3)
        a[2] = b[2] = 7
        ray = O.compute_ray(a, b)
        d = np.abs(b.astype(np.int64) - a)
        n = int(d.max())
        assert len(ray) == max(n - 1, 0)
        sgn = np.where(b.astype(np.int64) - a < 0, -1, 1)
        for t in range(1, n):
            s = (2 * t * d + n) // (2 * n)
            assert (a + sgn * s).tolist() == ray[t - 1].tolist()


def test_kat7_frequency_cell_state_machine():
    occ = O.Occ.new()
    c = (OFF + 3, OFF + 5)
    assert occ.set_occupied(*c) is True          # 1/1
    assert occ.set_free(*c) is False             # 1/2
    assert occ.set_free(*c) is False             # 1/3
    assert occ.set_free(*c) is False             # 1/4 == 0.25 exactly: still not free
    assert occ.set_free(*c) is True              # 1/5
    assert occ.probability(*c) == 0.2
    cells, mask = occ.patch(occ.patch_ids()[0])
    ci = 3 | (5 << 5)
    assert cells[ci]["occupied"] == 1 and cells[ci]["visited"] == 5
    # miss on a brand-new cell: 0/1 < 0.25 and it was "not free" (0.25) before -> True
    assert occ.set_free(OFF + 9, OFF + 9) is True
    assert occ.set_free(OFF + 9, OFF + 9) is False


def test_kat8_se2_exp():
    e = O.se2_exp([0.3, -0.2, 0.0])
    assert e.tolist() == [1.0, 0.0, 0.3, -0.2]
    th = 0.7
    e = O.se2_exp([0.0, 0.0, th])
    assert abs(e[0] - math.cos(th)) < 1e-15 and abs(e[1] - math.sin(th)) < 1e-15 and e[2] == 0 and e[3] == 0
    # exp(v) with theta != 0: translation = V(theta) v
    v = np.array([0.4, 0.1, 0.5])
    e = O.se2_exp(v)
    a, b = math.sin(0.5) / 0.5, (1 - math.cos(0.5)) / 0.5
    assert abs(e[2] - (a * 0.4 - b * 0.1)) < 1e-15 and abs(e[3] - (b * 0.4 + a * 0.1)) < 1e-15
    # compose/inverse
    A, B = O.se2(1.0, 2.0, 0.3), O.se2(-0.5, 0.25, -1.1)
    AB = O.se2_mul(A, B)
    assert abs(math.atan2(AB[1], AB[0]) - (0.3 - 1.1)) < 1e-15
    inv = np.zeros(4)
    O.lib().orc_se2_inverse(O._p(A), O._p(inv))
    I = O.se2_mul(inv, A)
    assert np.allclose(I, [1, 0, 0, 0], atol=1e-15)


def test_kat9_equal_weights_no_resample():
    P = 16
    pf = O.PF(O.default_options(particles=P, seed=1))
    pf.set_prior(O.se2(0, 0, 0))
    pts = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    pf.update(pts, O.se2(0, 0, 0))
    pf.set_weights(w=np.full(P, -3.25))
    assert abs(pf.stage_normalize() - P) < 1e-9
    assert pf.stage_resample_indices(0.5).tolist() == list(range(P))


def test_kat10_cauchy():
    assert O.lib().orc_cauchy(0.15, 0.15) == 0.5


def test_ldlt3_matches_numpy():
    rng = np.random.default_rng(3)
    for _ in range(50):
        M = rng.normal(size=(20, 3)) * rng.uniform(0.1, 10, size=3)
        A = M.T @ M
        b = rng.normal(size=3)
        x = np.zeros(3)
        O.lib().orc_ldlt3_solve(O._p(np.ascontiguousarray(A)), O._p(b), O._p(x))
        assert np.allclose(x, np.linalg.solve(A, b), rtol=1e-9, atol=1e-12)


def _brute_edt(obstacles, cells, max_sq):
    ob = np.array(sorted(obstacles), dtype=np.int64)
    out = {}
    for c in cells:
        d = ((ob - np.array(c)) ** 2).sum(axis=1).min() if len(ob) else max_sq
        out[c] = int(d)
    return out


def test_brushfire_vs_bruteforce_edt_with_removals():
    """Dynamic brushfire (add + remove waves) against a brute-force truncated EDT.
    4-connected brushfire is exact on these sparse sets (walls + isolated cells)."""
    rng = np.random.default_rng(11)
    dm = O.DM.new()
    obstacles = set()
    base = OFF + 100
    for y in range(0, 60):
        obstacles.add((base, base + y))
    for x in range(0, 45):
        obstacles.add((base + x, base + 60))
    for _ in range(25):
        obstacles.add((base + int(rng.integers(5, 60)), base + int(rng.integers(0, 55))))
    for o in sorted(obstacles):
        dm.add(*o)
    dm.update()
    # remove a third, add a few new ones, update again
    rem = [o for i, o in enumerate(sorted(obstacles)) if i % 3 == 0]
    for o in rem:
        dm.remove(*o)
        obstacles.discard(o)
    for _ in range(10):
        o = (base + int(rng.integers(5, 60)), base + int(rng.integers(0, 55)))
        dm.add(*o)
        obstacles.add(o)
    dm.update()
    cells = _cells_by_coord(dm)
    truth = _brute_edt(obstacles, list(cells.keys()), 100)
    bad = 0
    for k, c in cells.items():
        t = truth[k]
        if t < 100:
            if not c["valid"] or c["sqdist"] != t:
                bad += 1
        else:
            if c["valid"]:
                bad += 1
        assert c["queued"] == 0
        if c["valid"]:
            ox, oy = k[0] + int(c["obstacle"][0]), k[1] + int(c["obstacle"][1])
            assert (ox, oy) in obstacles
            assert (ox - k[0]) ** 2 + (oy - k[1]) ** 2 == c["sqdist"]
    assert bad == 0


def test_gn_recovers_pose_offset_on_wall_map():
    """GN + Cauchy on an analytic map (two perpendicular walls) pulls a perturbed pose back."""
    dm = O.DM.new()
    for i in range(0, 121):
        dm.add(OFF + 100, OFF + i)   # wall x = 5 m, y in [0,6]
        dm.add(OFF + i - 20, OFF + 120)   # wall y = 6 m
    dm.update()
    # scan from truth pose (2, 3, 0.1): points exactly on the walls (cell centres)
    th = 0.1
    truth = O.se2(2.0, 3.0, th)
    world = [(5.0, 0.05 * i) for i in range(10, 115, 2)] + [(0.05 * (i - 20), 6.0) for i in range(30, 115, 2)]
    c, s = math.cos(th), math.sin(th)
    pts = np.array([[c * (wx - 2.0) + s * (wy - 3.0), -s * (wx - 2.0) + c * (wy - 3.0), 0.0] for wx, wy in world])
    r = O.eval_(dm, pts, truth, jac=False)
    assert np.abs(r).max() < 1e-9
    start = O.se2(2.04, 2.97, th + 0.01)
    pose, iters, evals = O.solve(dm, pts, start)
    assert iters >= 1 and evals >= 2 * iters
    assert abs(pose[2] - 2.0) < 5e-3 and abs(pose[3] - 3.0) < 5e-3
    assert abs(math.atan2(pose[1], pose[0]) - th) < 2e-3
    assert O.loglik(dm, pts, pose) > O.loglik(dm, pts, start)
