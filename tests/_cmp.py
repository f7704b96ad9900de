"""Map comparison helpers: device download (reference record format) vs oracle dump."""
import numpy as np


def diff_maps(dev, orc, fields):
    """dev/orc: {patch_id: (cells[1024] structured, mask[16])}.  Returns a dict of mismatch counts."""
    out = {"patches_only_dev": 0, "patches_only_orc": 0, "mask_words": 0}
    for f in fields:
        out[f] = 0
    for pid in dev.keys() - orc.keys():
        out["patches_only_dev"] += 1
    for pid in orc.keys() - dev.keys():
        out["patches_only_orc"] += 1
    for pid in dev.keys() & orc.keys():
        dc, dm = dev[pid]
        oc, om = orc[pid]
        out["mask_words"] += int((dm != om).sum())
        for f in fields:
            a, b = dc[f], oc[f]
            ne = (a != b)
            if ne.ndim > 1:
                ne = ne.any(axis=1)
            out[f] += int(ne.sum())
    return out


DM_FIELDS = ["sqdist", "valid", "obstacle", "queued"]
OCC_FIELDS = ["occupied", "visited"]


def assert_maps_equal(dev, orc, fields, what=""):
    d = diff_maps(dev, orc, fields)
    assert all(v == 0 for v in d.values()), f"{what}: {d} (dev patches {len(dev)}, oracle patches {len(orc)})"

import numpy as np

def check_slam_views(o, h, rng):
    """Slam2D::getOccupancyMap() / getDistanceMap() snapshots (lama/sdm_maps.h) answer the reference's const map queries like
    the oracle's maps: bounds, visit_all_cells, isFree / isOccupied / isUnknown / getProbability, distance(cell),
    distance(point, gradient)."""
    for which, om in ((0, o.occ()), (1, o.dm())):
        mn, mx, wmn, wmx = h.view_bounds(which)
        omn, omx, owmn, owmx = om.bounds()
        assert np.array_equal(mn, omn) and np.array_equal(mx, omx), which
        assert np.array_equal(wmn, owmn) and np.array_equal(wmx, owmx), which
        hc, oc = h.view_cells(which), om.cells()
        assert len(hc) == len(oc) and len(hc) > 1000
        key = lambda a: np.sort(a[:, 0].astype(np.uint64) << np.uint64(32) | a[:, 1].astype(np.uint64))
        assert np.array_equal(key(hc), key(oc)), which
    cells = o.occ().cells()
    pick = cells[rng.choice(len(cells), size=600, replace=False)]
    far = pick + np.array([5000, 7000], dtype=np.uint32)                 # no patch there
    near = pick + np.array([3, 2], dtype=np.uint32)                      # some with the mask bit off
    q = np.concatenate([pick, far, near])
    fr, oc, un, pr = h.view_occupancy(q)
    seen = set()
    for i, (x, y) in enumerate(q):
        f, occ, unk = o.occ().state(int(x), int(y))
        assert (fr[i], oc[i], un[i]) == (f, occ, unk), (i, x, y)
        assert pr[i] == o.occ().probability(int(x), int(y))
        seen.add((bool(fr[i]), bool(oc[i]), bool(un[i])))
    assert {(True, False, False), (False, True, False), (False, False, True)} <= seen
    dcells = o.dm().cells()
    dq = np.concatenate([dcells[rng.choice(len(dcells), size=600, replace=False)], far[:50]])
    hd = h.view_distance_cells(dq)
    assert np.array_equal(hd, np.array([o.dm().distance_cell(int(x), int(y)) for x, y in dq]))
    assert hd.min() == 0.0 and hd.max() == 0.5
    pts = np.stack([rng.uniform(1.0, 27.0, 300), rng.uniform(0.2, 3.8, 300)], 1)
    hv = h.view_distance_points(pts)
    for i, (x, y) in enumerate(pts):
        d, g = o.dm().distance([x, y, 0.0], grad=True)
        assert hv[i, 0] == d and hv[i, 1] == g[0] and hv[i, 2] == g[1], i


def check_match_surface_and_solver(O, o, h, scan, rng, pose_tol, exact):
    """lama::MatchSurface2D::eval / error and lama::Solve on Slam2D::getDistanceMap() against the oracle's MatchSurface2D / solver
    on the oracle's distance map (built from the same scans at the same poses)."""
    import pytest
    base = o.pose()
    for trial in range(3):
        pose = O.se2_mul(base, O.se2(*rng.normal(0, [0.04, 0.04, 0.015])))
        r, J, rmse = h.match_eval(scan, pose)
        orr, oJ = O.eval_(o.dm(), scan, pose)
        if exact:
            assert np.array_equal(r, orr) and np.array_equal(J, oJ)
        else:
            assert np.abs(r - orr).max() < 1e-9 and np.abs(J - oJ).max() < 1e-6
        assert abs(rmse - O.match_error(o.dm(), scan, pose)) < 1e-12
        for strategy in ("gn", "lm"):
            got, cov, it = h.match_solve(scan, pose, strategy=strategy)
            want, oit, ocov = O.solve_full(o.dm(), scan, pose, lm=(strategy == "lm"))
            assert np.abs(got - want).max() <= pose_tol, (trial, strategy)
            assert it == oit
            assert np.allclose(cov, ocov, rtol=1e-6, atol=1e-12)
        got5, _, it5 = h.match_solve(scan, pose, max_iterations=2)
        assert it5 <= 2
    # a configuration without a device kernel is an error, not a CPU computation
    with pytest.raises(Exception, match="no device kernel|no CPU path"):
        h.match_solve(scan, base, weight="tukey", weight_param=4.6851)
    with pytest.raises(Exception, match="no device kernel|no CPU path"):
        h.match_solve(scan, base, weight="cauchy", weight_param=0.3)
