#!/usr/bin/env python
"""Generates tests/golden/reference_golden.npz by RUNNING THE REFERENCE ITSELF: oracle/_ref/liblama_ref.so is the reference's
own sources compiled from /root/reference (recipe oracle/Makefile.ref; Eigen3 is absent from this image, so they compile
against the stand-in under oracle/ref_shim/ -- see its header for what that pins).  The fixtures travel to boxes where
/root/reference does not exist: the CPU oracle and the HIP path are both checked against them (tests/test_reference_golden.py).

Content (seeded corridor log of SURVEY 8(d), 1080 beams):
  pf_*      PFSlam2D, P = 4, seed 42, 9 scans, default gains: poses / weights / Neff / best index after every update and sha256
            digests of particle 0's distance and occupancy maps (patch ids + cell records + Container masks)
  rs_*      the same with meas_sigma_gain = 0.01 (resampling happens), P = 6, 12 scans; digests of the best particle's maps
  slam_*    Slam2D: pose after every update, map digests
  loc_*     Loc2D on the corridor's static map (config 1): pose / covariance / RMSE after every update
  lo_*      LidarOdometry2D on range-limited scans: odometry after every update, map digests at the end
  pgo_*     minisam's linearzationLowerHessian on a SimplePGO-shaped graph (tests/_posegraph.make_graph(40, 30, seed 11)
            plus a duplicate pair and a backward loop closure) at the dead-reckoned poses: whitened errors, Atb, the dense Hessian
  kat_*     known answers: SE2 exp / compose / inverse-compose, Map::computeRay, bilinear distance + gradient, CauchyWeight,
            lama::random after setSeed(7), MatchSurface2D::eval residual / Jacobian rows, Solve() results (GN and LM, with covariance)
Run from the repo root (needs /root/reference):  make -f oracle/Makefile.ref && python tests/golden/make_reference_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import _oracle as O                      # noqa: E402  (option struct only)
import _reference as R                   # noqa: E402
import iris_lama_amd.ffi as F            # noqa: E402  (workload generator only)
from golden.make_golden import map_digest  # noqa: E402

KAT_EXP_IN = np.array([[0.3, -0.2, 0.0], [0.0, 0.0, 0.7], [0.4, 0.1, 0.5], [-0.05, 0.02, -1e-12], [1.5, -2.5, 3.0]])
KAT_POSE_A, KAT_POSE_B = np.array([1.0, 2.0, 0.3]), np.array([-0.5, 0.25, -1.1])
KAT_DIST_IN = np.array([[3.1, 1.7, 0.0], [5.02, 0.93, 0.0], [10.0, 3.9, 0.0], [25.0, 2.0, 0.0], [2.0, 2.0, 0.0]])
KAT_RAYS = [([100, 200, 0], [131, 187, 0]), ([42275904, 42275904, 0], [42275890, 42275950, 0]), ([10, 10, 0], [10, 40, 0]), ([7, 9, 0], [7, 9, 0])]
PF_STEPS, PF_P, RS_STEPS, RS_P, SLAM_STEPS, LOC_STEPS, LO_STEPS = 8, 4, 11, 6, 10, 10, 26
LOC_START_OFFSET = np.array([0.05, -0.04, 0.01])


def pgo_case():
    """The pose graph of the pgo_* fixtures (built from the seeded generator; the oracle's SE2 operations are only the generator)."""
    from _posegraph import make_graph
    fi, fj, meas, sq, truth, init = make_graph(40, 30, seed=11)
    fi = np.concatenate([fi, [3, 17]]).astype(np.int32); fj = np.concatenate([fj, [4, 5]]).astype(np.int32)
    meas = np.concatenate([meas, meas[4:5], [O.se2_mul(O.se2_inverse(truth[17]), truth[5])]])
    sq = np.concatenate([sq, [[2.0, 2.0, 10.0], [1.0, 3.0, 7.0]]])
    return {"fi": fi, "fj": fj, "meas": meas, "sq": sq, "x": init}


def dense_hessian(N, fi, fj, lin):
    """Blocks of a pgo_linearize result (oracle or device) scattered the way minisam's value_ptr walk accumulates them."""
    H = np.zeros((3 * N, 3 * N))
    for v in range(N):
        H[3 * v:3 * v + 3, 3 * v:3 * v + 3] = lin["Hdiag"][v]
    for k in range(len(fi)):
        i, j = int(fi[k]), int(fj[k])
        if j >= 0:
            H[3 * i:3 * i + 3, 3 * j:3 * j + 3] += lin["Hoff"][k]
            H[3 * j:3 * j + 3, 3 * i:3 * i + 3] += lin["Hoff"][k].T
    return H


def run_pf(pts, odom, steps, P, seed, gain):
    pf = R.PF(O.default_options(particles=P, seed=seed, meas_sigma_gain=gain, threads=1))
    pf.set_prior(odom[0])
    out = dict(poses=[], weights=[], neff=[], best=[], dm=[], occ=[])
    for k in range(steps + 1):
        assert pf.update(pts[k], odom[k], float(k))
        b = pf.best()
        out["poses"].append(pf.poses()); out["weights"].append(pf.weights()[0]); out["neff"].append(pf.neff()); out["best"].append(b)
        which = 0 if gain == 3.0 else b
        out["dm"].append(map_digest(pf.dm(which).dump())); out["occ"].append(map_digest(pf.occ(which).dump()))
    return pf, {k: np.array(v) for k, v in out.items()}


def main():
    L = R.lib()
    pts, odom, truth = F.corridor_log(max(PF_STEPS, RS_STEPS, SLAM_STEPS, LOC_STEPS, LO_STEPS), 1080)
    # random stream first (PFSlam2D's constructor re-seeds the global generator)
    L.ref_random_set_seed(7)
    kat_random = np.array([L.ref_random_uniform() for _ in range(4)] + [L.ref_random_normal(0.5) for _ in range(4)] + [L.ref_random_uniform()])
    pf, g = run_pf(pts, odom, PF_STEPS, PF_P, 42, 3.0)
    _, rs = run_pf(pts, odom, RS_STEPS, RS_P, 11, 0.01)
    # KATs
    kat_exp = np.zeros((len(KAT_EXP_IN), 4))
    for i, v in enumerate(KAT_EXP_IN):
        L.ref_se2_exp(O._p(np.ascontiguousarray(v)), O._p(kat_exp[i]))
    kat_plus, kat_minus = np.zeros(4), np.zeros(4)
    L.ref_pose_plus_xyr(O._p(KAT_POSE_A), O._p(KAT_POSE_B), O._p(kat_plus))
    L.ref_pose_minus_xyr(O._p(KAT_POSE_A), O._p(KAT_POSE_B), O._p(kat_minus))
    dm = pf.dm(0)
    rays = [dm.compute_ray(a, b) for a, b in KAT_RAYS]
    kat_dist = np.array([list((lambda d, gr: (d, gr[0], gr[1]))(*dm.distance(p, grad=True))) for p in KAT_DIST_IN])
    kat_cauchy = np.array([L.ref_cauchy(0.15, x) for x in (0.0, 0.15, -0.4, 2.0)])
    xyr = np.array([odom[PF_STEPS][0] + 0.07, odom[PF_STEPS][1] - 0.05, odom[PF_STEPS][2] + 0.02])
    r, J = R.eval_(dm, pts[PF_STEPS], xyr)
    sel = np.arange(0, 1080, 45)
    gn, gn_cov = R.solve(dm, pts[PF_STEPS], xyr, cov=True)
    lm, lm_cov = R.solve(dm, pts[PF_STEPS], xyr, lm=True, cov=True)
    # Slam2D
    s = L.ref_slam_new(0.5, 0.5, 0.5, 0.0, 0.0, 0.05, 32, 100, 0, 0)
    L.ref_slam_set_pose(s, O._p(np.ascontiguousarray(odom[0])))
    slam_poses, slam_dm, slam_occ = [], [], []
    I, Z = R.IDENT_Q, R.ZERO3
    for k in range(SLAM_STEPS + 1):
        L.ref_slam_update(s, O._p(np.ascontiguousarray(pts[k])), len(pts[k]), O._p(Z), O._p(I), O._p(np.ascontiguousarray(odom[k])), float(k))
        p = np.zeros(4); L.ref_slam_get_pose(s, O._p(p)); slam_poses.append(p)
        slam_dm.append(map_digest(R.DM(L.ref_slam_dm(s)).dump())); slam_occ.append(map_digest(R.Occ(L.ref_slam_occ(s)).dump()))
    L.ref_slam_free(s)
    # Loc2D (config 1): static map from the corridor's obstacle points, scan match + covariance per update
    from _worlds import corridor_obstacles
    cells = np.array([[int(c[0]), int(c[1])] for c in (O.w2m([x, y, 0.0]) for x, y in corridor_obstacles())], dtype=np.uint32)
    a = L.ref_loc_new(0.5, 0.5, 1.0, 0.05, 32, 100, 0)
    L.ref_loc_occ_set(a, O._p(cells), len(cells), 1)
    start = truth[0] + LOC_START_OFFSET
    L.ref_loc_set_pose(a, O._p(np.ascontiguousarray(start)))
    loc_poses, loc_cov, loc_rmse, loc_upd = [], [], [], []
    for k in range(LOC_STEPS + 1):
        p = np.ascontiguousarray(pts[k])
        loc_upd.append(L.ref_loc_update(a, O._p(p), len(p), O._p(Z), O._p(I), O._p(np.ascontiguousarray(odom[k])), float(k), 0))
        pp, cc = np.zeros(4), np.zeros(9)
        L.ref_loc_get_pose(a, O._p(pp)); L.ref_loc_covar(a, O._p(cc))
        loc_poses.append(pp); loc_cov.append(cc.reshape(3, 3)); loc_rmse.append(L.ref_loc_rmse(a))
    L.ref_loc_free(a)
    # LidarOdometry2D on range-limited scans
    lo = L.ref_lo_new(0.05, 100)
    lo_odom, lo_upd = [], []
    for k in range(LO_STEPS + 1):
        p = np.ascontiguousarray(pts[k][np.hypot(pts[k][:, 0], pts[k][:, 1]) < 4.0])
        lo_upd.append(L.ref_lo_update(lo, O._p(p), len(p), O._p(Z), O._p(I), float(k)))
        pp = np.zeros(4); L.ref_lo_get_odom(lo, O._p(pp)); lo_odom.append(pp)
    lo_dm, lo_occ = map_digest(R.DM(L.ref_lo_dm(lo)).dump()), map_digest(R.POcc(L.ref_lo_occ(lo)).dump())
    L.ref_lo_free(lo)
    # pose-graph linearisation (minisam itself)
    pgo = pgo_case()
    pgo_ref = R.pgo_linearize(pgo["x"], pgo["fi"], pgo["fj"], pgo["meas"], pgo["sq"])
    out = os.path.join(HERE, "reference_golden.npz")
    np.savez_compressed(
        out, odom=odom, truth=truth,
        pf_poses=g["poses"], pf_weights=g["weights"], pf_neff=g["neff"], pf_best=g["best"], pf_dm_digest=g["dm"], pf_occ_digest=g["occ"],
        rs_poses=rs["poses"], rs_weights=rs["weights"], rs_neff=rs["neff"], rs_best=rs["best"], rs_dm_digest=rs["dm"], rs_occ_digest=rs["occ"],
        slam_poses=np.stack(slam_poses), slam_dm_digest=np.array(slam_dm), slam_occ_digest=np.array(slam_occ),
        loc_poses=np.stack(loc_poses), loc_cov=np.stack(loc_cov), loc_rmse=np.array(loc_rmse), loc_updated=np.array(loc_upd),
        lo_odom=np.stack(lo_odom), lo_updated=np.array(lo_upd), lo_dm_digest=np.array(lo_dm), lo_occ_digest=np.array(lo_occ),
        kat_random=kat_random, kat_exp=kat_exp, kat_plus=kat_plus, kat_minus=kat_minus,
        kat_ray_sizes=np.array([len(x) for x in rays]), kat_rays=np.concatenate(rays) if len(rays) else np.zeros((0, 3)),
        kat_dist=kat_dist, kat_cauchy=kat_cauchy, kat_eval_xyr=xyr, kat_eval_r=r[sel], kat_eval_J=J[sel],
        kat_gn=gn, kat_gn_cov=gn_cov, kat_lm=lm, kat_lm_cov=lm_cov,
        pgo_err=pgo_ref["err"], pgo_b=pgo_ref["b"], pgo_H=pgo_ref["H"])
    print("wrote", out, os.path.getsize(out), "bytes; resampling steps in rs run:",
          int(np.sum(np.any(np.diff(rs["weights"], axis=0) != 0, axis=1))))


if __name__ == "__main__":
    main()
