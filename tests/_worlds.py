"""Synthetic static maps for the Loc2D tests: obstacle points (world metres, one per 0.05 m cell) of the corridor world."""
import numpy as np


def corridor_obstacles(res=0.05):
    pts = []

    def seg(x0, y0, x1, y1):
        n = int(round(max(abs(x1 - x0), abs(y1 - y0)) / res)) + 1
        for k in range(n):
            t = k / max(n - 1, 1)
            pts.append((x0 + t * (x1 - x0), y0 + t * (y1 - y0)))

    def box(xa, ya, xb, yb):
        seg(xa, ya, xb, ya); seg(xb, ya, xb, yb); seg(xb, yb, xa, yb); seg(xa, yb, xa, ya)

    box(0.0, 0.0, 28.0, 4.0)
    for k in range(6):
        cx = 5.0 + 4.0 * k
        cy = 0.8 if k % 2 == 0 else 3.2
        box(cx - 0.2, cy - 0.2, cx + 0.2, cy + 0.2)
    return np.array(pts)


def corridor_free_cells(w2m, res=0.05):
    """Map cells (x, y) of the corridor's interior (pillars excluded), for occupancy_map->setFree."""
    cells = []
    xs = np.arange(0.1, 27.9 + 1e-9, res)
    ys = np.arange(0.1, 3.9 + 1e-9, res)
    pillars = [(5.0 + 4.0 * k, 0.8 if k % 2 == 0 else 3.2) for k in range(6)]
    for x in xs:
        for y in ys:
            if any(abs(x - cx) <= 0.3 and abs(y - cy) <= 0.3 for cx, cy in pillars):
                continue
            c = w2m([x, y, 0.0])
            cells.append((int(c[0]), int(c[1])))
    return np.array(sorted(set(cells)), dtype=np.uint32)


def open_corridor(res=0.05, half_len=40.0, width=4.0):
    """Two parallel walls y = 0 and y = width, x in [-half_len, half_len]: a corridor whose ends are out of sight, so
    the along-corridor direction is unobservable (exactly rank-deficient scan-matching Jacobian)."""
    n = int(round(2 * half_len / res)) + 1
    xs = -half_len + res * np.arange(n)
    return np.concatenate([np.stack([xs, np.zeros(n)], 1), np.stack([xs, np.full(n, width)], 1)])


def open_corridor_scan(x, y, yaw, beams=360, fov=np.deg2rad(270.0), max_range=12.0, width=4.0):
    """Sensor-frame points (n, 3) of the beams of a scanner at (x, y, yaw) that reach one of the two walls of
    open_corridor() within max_range (the others have no return and are dropped)."""
    out = []
    for k in range(beams):
        phi = -fov / 2 + fov * k / (beams - 1)
        d = np.sin(yaw + phi)
        if abs(d) < 1e-9:
            continue
        r = (width - y) / d if d > 0 else (0.0 - y) / d
        if 0.2 < r <= max_range:
            out.append((r * np.cos(phi), r * np.sin(phi), 0.0))
    return np.array(out)


def long_corridor_segments(length=460.0, width=4.0, pillar_every=4.0):
    """Wall segments (S, 4) = x0, y0, x1, y1 of a closed corridor [0, length] x [0, width] with square pillars (0.4 m) every
    `pillar_every` metres alternating between y = 0.8 and y = width - 0.8 (the SURVEY 8(d) corridor, made long)."""
    segs = [(0.0, 0.0, length, 0.0), (length, 0.0, length, width), (length, width, 0.0, width), (0.0, width, 0.0, 0.0)]
    k = 0
    cx = 5.0
    while cx < length - 1.0:
        cy = 0.8 if k % 2 == 0 else width - 0.8
        xa, xb, ya, yb = cx - 0.2, cx + 0.2, cy - 0.2, cy + 0.2
        segs += [(xa, ya, xb, ya), (xb, ya, xb, yb), (xb, yb, xa, yb), (xa, yb, xa, ya)]
        cx += pillar_every
        k += 1
    return np.array(segs)


def segment_world_scan(segs, x, y, yaw, beams=1080, fov=np.deg2rad(270.0), max_range=30.0, noise=None):
    """Sensor-frame points (n, 3) of a 2-D scanner at (x, y, yaw) in a world of wall segments: one point per beam that hits a wall
    within max_range (beams without a return are dropped, as iris_lama_ros drops out-of-range readings)."""
    phi = -fov / 2 + fov * np.arange(beams) / (beams - 1)
    dx, dy = np.cos(yaw + phi)[:, None], np.sin(yaw + phi)[:, None]
    near = (np.minimum(segs[:, 0], segs[:, 2]) < x + max_range) & (np.maximum(segs[:, 0], segs[:, 2]) > x - max_range)
    s = segs[near]
    ex, ey = (s[:, 2] - s[:, 0])[None, :], (s[:, 3] - s[:, 1])[None, :]
    ax, ay = (s[:, 0] - x)[None, :], (s[:, 1] - y)[None, :]
    den = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (ax * ey - ay * ex) / den           # along the beam
        u = (ax * dy - ay * dx) / den           # along the segment
    ok = (np.abs(den) > 1e-12) & (t > 0.05) & (u >= 0.0) & (u <= 1.0)
    t = np.where(ok, t, np.inf)
    r = t.min(axis=1)
    if noise is not None:
        r = r + noise
    keep = np.isfinite(r) & (r <= max_range)
    return np.stack([r[keep] * np.cos(phi[keep]), r[keep] * np.sin(phi[keep]), np.zeros(keep.sum())], axis=1)


def hall_segments(side=104.0, rooms=4, gap=6.0):
    """Wall segments (S, 4) of a square hall of `side` metres divided into rooms x rooms bays by interior walls with a `gap`-metre
    opening in the middle of every bay side (so that a robot can drive through all of them), plus a 1 m pillar in every bay."""
    segs = [(0.0, 0.0, side, 0.0), (side, 0.0, side, side), (side, side, 0.0, side), (0.0, side, 0.0, 0.0)]
    bay = side / rooms
    for i in range(1, rooms):
        c = i * bay
        for j in range(rooms):
            a, b = j * bay, (j + 1) * bay
            m = 0.5 * (a + b)
            segs += [(c, a, c, m - gap / 2), (c, m + gap / 2, c, b)]       # wall x = c, opening around y = m
            segs += [(a, c, m - gap / 2, c), (m + gap / 2, c, b, c)]       # wall y = c, opening around x = m
    for i in range(rooms):
        for j in range(rooms):
            cx, cy = (i + 0.3) * bay, (j + 0.7) * bay
            segs += [(cx - .5, cy - .5, cx + .5, cy - .5), (cx + .5, cy - .5, cx + .5, cy + .5), (cx + .5, cy + .5, cx - .5, cy + .5), (cx - .5, cy + .5, cx - .5, cy - .5)]
    return np.array(segs)


def hall_tour(side=104.0, rooms=4, step=6.5):
    """Poses (x, y, yaw) of a serpentine tour through the centres of all bays of hall_segments, every `step` metres, heading along
    the path."""
    bay = side / rooms
    centres = []
    for j in range(rooms):
        cols = range(rooms) if j % 2 == 0 else range(rooms - 1, -1, -1)
        centres += [((i + 0.5) * bay, (j + 0.5) * bay) for i in cols]
    out = []
    for (x0, y0), (x1, y1) in zip(centres[:-1], centres[1:]):
        d = np.hypot(x1 - x0, y1 - y0)
        n = max(int(np.ceil(d / step)), 1)
        yaw = np.arctan2(y1 - y0, x1 - x0)
        for k in range(n):
            t = k / n
            out.append((x0 + t * (x1 - x0), y0 + t * (y1 - y0), yaw))
    out.append((centres[-1][0], centres[-1][1], out[-1][2]))
    return np.array(out)
