#!/usr/bin/env python
"""Generates tests/golden/corridor_golden.npz from the CPU oracle (the reference ships no golden vectors and cannot
be built here -- SURVEY F2/F4 -- so these fixtures pin the ORACLE itself against regressions and give the GPU tests a
target that does not depend on rebuilding the oracle).

Content (seeded corridor log of SURVEY 8(d), 8 scans x 1080 beams, P = 4, seed 42):
  odom, truth                : the log's odometry / ground truth (x, y, yaw)
  poses[k]                   : oracle particle poses {c,s,tx,ty} after update k (free running)
  weights[k], neff[k]        : oracle weights / Neff after update k
  dm_digest[k], occ_digest[k]: sha256 over particle 0's maps after update k (patch ids + cells + masks, reference
                               record formats, ascending patch id)
  kat_*                      : small known-answer vectors (SE2 exp/compose, ray, bilinear distance + gradient)
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as O                      # noqa: E402
import iris_lama_amd.ffi as F            # noqa: E402  (workload generator only)


def map_digest(dump):
    h = hashlib.sha256()
    for pid in sorted(dump):
        cells, mask = dump[pid]
        h.update(np.uint64(pid).tobytes())
        h.update(cells.tobytes())
        h.update(mask.tobytes())
    return h.hexdigest()


def main():
    steps, P = 8, 4
    pts, odom, truth = F.corridor_log(steps, 1080)
    pf = O.PF(O.default_options(particles=P, seed=42))
    pf.set_prior(O.se2(*odom[0]))
    poses, weights, neff, dmd, occd = [], [], [], [], []
    for k in range(steps + 1):
        assert pf.update(pts[k], O.se2(*odom[k]), float(k))
        poses.append(pf.poses())
        weights.append(pf.weights()[0])
        neff.append(pf.neff())
        dmd.append(map_digest(pf.dm(0).dump()))
        occd.append(map_digest(pf.occ(0).dump()))
    # KATs
    v = np.array([[0.3, -0.2, 0.0], [0.0, 0.0, 0.7], [0.4, 0.1, 0.5], [-0.05, 0.02, -1e-12]])
    kat_exp = np.stack([O.se2_exp(x) for x in v])
    a, b = O.se2(1.0, 2.0, 0.3), O.se2(-0.5, 0.25, -1.1)
    kat_mul = O.se2_mul(a, b)
    ray = O.compute_ray([100, 200, 7], [131, 187, 7])
    dm = pf.dm(0)
    q = np.array([[3.1, 1.7, 0.0], [5.02, 0.93, 0.0], [10.0, 3.9, 0.0], [25.0, 2.0, 0.0]])
    kat_dist = np.array([list((lambda d, g: (d, g[0], g[1]))(*dm.distance(p, grad=True))) for p in q])
    out = os.path.join(HERE, "corridor_golden.npz")
    np.savez_compressed(out, odom=odom, truth=truth, poses=np.stack(poses), weights=np.stack(weights), neff=np.array(neff),
                        dm_digest=np.array(dmd), occ_digest=np.array(occd), kat_exp_in=v, kat_exp=kat_exp, kat_mul=kat_mul,
                        kat_ray=ray, kat_dist_in=q, kat_dist=kat_dist)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
