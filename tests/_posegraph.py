"""Synthetic SE2 pose graphs shaped like SimplePGO::optimize's (src/simple_pgo.cpp:48-105): a prior on pose 0, odometry
BetweenFactors i -> i+1 (sigmas 0.5, 0.5, 0.1) and loop-closure BetweenFactors between random earlier poses."""
import numpy as np

import _oracle as O


def make_graph(n_poses, n_loops, seed=0, noise=(0.05, 0.05, 0.01)):
    rng = np.random.default_rng(seed)
    # ground truth: a meandering trajectory
    th = np.cumsum(rng.normal(0, 0.15, n_poses))
    xy = np.cumsum(np.stack([0.5 * np.cos(th), 0.5 * np.sin(th)], axis=1), axis=0)
    truth = np.stack([O.se2(xy[i, 0], xy[i, 1], th[i]) for i in range(n_poses)])

    def rel(i, j):      # truth_i^-1 * truth_j perturbed by measurement noise
        d = O.se2_mul(O.se2_inverse(truth[i]), truth[j])
        return O.se2_mul(d, O.se2(rng.normal(0, noise[0]), rng.normal(0, noise[1]), rng.normal(0, noise[2])))

    fi, fj, meas, sq = [0], [-1], [truth[0]], [[1.0, 1.0, 1.0]]
    for i in range(n_poses - 1):
        fi.append(i); fj.append(i + 1); meas.append(rel(i, i + 1)); sq.append([2.0, 2.0, 10.0])
    for _ in range(n_loops):
        a = int(rng.integers(0, n_poses - 2))
        b = int(rng.integers(a + 2, min(n_poses, a + 200)))
        if rng.random() < 0.5:
            a, b = b, a                      # loop closures go both ways (from > to happens in the reference)
        fi.append(a); fj.append(b); meas.append(rel(a, b)); sq.append([2.0, 2.0, 10.0])
    # initial guess: dead reckoning along the odometry factors
    init = [truth[0]]
    for i in range(n_poses - 1):
        init.append(O.se2_mul(init[-1], meas[1 + i]))
    return (np.array(fi, dtype=np.int32), np.array(fj, dtype=np.int32), np.array(meas), np.array(sq), truth, np.array(init))
