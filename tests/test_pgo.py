"""SE2 pose-graph linearisation (minisam linearzationLowerHessian for PriorFactor / BetweenFactor with DiagonalLoss,
vendor/minisam/minisam/nonlinear/linearization.cpp:150-272).  CPU: the oracle restatement pinned by source-derived known
answers; GPU: lama_hip_pgo_linearize against the oracle and a Gauss-Newton loop that converges with the device linearisation."""
import numpy as np
import pytest

import _oracle as O
import iris_lama_amd.ffi as F
from _posegraph import make_graph


def _dense(N, fi, fj, lin):
    """Assemble the dense symmetric Hessian and gradient from the block outputs (what the caller's sparse scatter does)."""
    H = np.zeros((3 * N, 3 * N))
    for v in range(N):
        H[3 * v:3 * v + 3, 3 * v:3 * v + 3] += lin["Hdiag"][v]
    for k in range(len(fi)):
        if fj[k] >= 0:
            i, j = int(fi[k]), int(fj[k])
            H[3 * i:3 * i + 3, 3 * j:3 * j + 3] += lin["Hoff"][k]
            H[3 * j:3 * j + 3, 3 * i:3 * i + 3] += lin["Hoff"][k].T
    return H, lin["b"].reshape(-1)


def test_kat_consistent_graph_has_zero_error_and_prior_block():
    """Measurements taken from the poses themselves: every error is 0, b = 0; a prior alone contributes diag(sqrt_info^2)."""
    fi, fj, meas, sq, truth, init = make_graph(12, 5, seed=3, noise=(0, 0, 0))
    lin = O.pgo_linearize(truth, fi, fj, meas, sq)
    assert np.abs(lin["err"]).max() < 1e-12 and np.abs(lin["b"]).max() < 1e-11 and lin["chi2"] < 1e-22
    one = O.pgo_linearize(truth[:1], fi[:1], fj[:1], meas[:1], np.array([[2.0, 3.0, 4.0]]))
    assert np.allclose(one["Hdiag"][0], np.diag([4.0, 9.0, 16.0]))            # PriorFactor: identity Jacobian, DiagonalLoss rows
    assert np.all(one["Hoff"] == 0)


def test_kat_between_error_is_log_of_relative_pose_and_jacobians_follow_minisam():
    """BetweenFactor::error = Local(z, x_i^-1 x_j) = log(z^-1 x_i^-1 x_j) (slam/BetweenFactor.h:50-56, geometry/Sophus.h:45-50);
    with z = identity and x_i = identity the error is log(x_j): theta and V^-1 t (se2.hpp:519-542).  jacobians():
    J_j = I, J_i = Adj(x_j^-1) * (-Adj(x_i)) (BetweenFactor.h:59-67, Sophus.h:63-74, se2.hpp:125-133)."""
    ident = O.se2(0, 0, 0)
    xj = O.se2(0.3, -0.2, 0.4)
    lin = O.pgo_linearize(np.stack([ident, xj]), [0], [1], [ident], [[1.0, 1.0, 1.0]])
    th = 0.4
    half = 0.5 * th
    h = -(half * np.sin(th)) / (np.cos(th) - 1.0)
    expect = np.array([h * 0.3 + half * -0.2, -half * 0.3 + h * -0.2, th])
    assert np.allclose(lin["err"][0], expect, rtol=1e-13, atol=1e-15)
    # J_i = Adj(x_j^-1) * (-I) for x_i = identity
    inv = O.se2_inverse(xj)
    adj = np.array([[inv[0], -inv[1], inv[3]], [inv[1], inv[0], -inv[2]], [0, 0, 1]])
    Ji = -adj
    assert np.allclose(lin["Hdiag"][0], Ji.T @ Ji, rtol=1e-13, atol=1e-15)
    assert np.allclose(lin["Hdiag"][1], np.eye(3), atol=0)
    assert np.allclose(lin["Hoff"][0], Ji.T @ np.eye(3), rtol=1e-13, atol=1e-15)
    assert np.allclose(lin["b"][0], -(Ji.T @ expect), rtol=1e-12, atol=1e-15)
    # whitening scales error and Jacobian rows (core/LossFunction.cpp:103-113)
    lin2 = O.pgo_linearize(np.stack([ident, xj]), [0], [1], [ident], [[2.0, 2.0, 10.0]])
    W = np.diag([2.0, 2.0, 10.0])
    assert np.allclose(lin2["err"][0], W @ expect, rtol=1e-13)
    assert np.allclose(lin2["Hdiag"][0], (W @ Ji).T @ (W @ Ji), rtol=1e-13)


def test_kat_accumulation_is_the_sum_over_incident_factors_in_factor_order():
    fi, fj, meas, sq, truth, init = make_graph(30, 25, seed=5)
    lin = O.pgo_linearize(init, fi, fj, meas, sq)
    for v in (0, 7, 29):
        acc, g = np.zeros((3, 3)), np.zeros(3)
        for k in range(len(fi)):                       # same order, one factor at a time
            if fi[k] == v or fj[k] == v:
                sub = O.pgo_linearize(init, fi[k:k + 1], fj[k:k + 1], meas[k:k + 1], sq[k:k + 1])
                acc = acc + sub["Hdiag"][v]
                g = g + sub["b"][v]
        assert np.array_equal(acc, lin["Hdiag"][v])
        assert np.allclose(g, lin["b"][v], rtol=1e-13, atol=1e-13)
    H, b = _dense(30, fi, fj, lin)
    assert np.allclose(H, H.T) and np.linalg.eigvalsh(H).min() > 0            # prior makes it positive definite


def _gauss_newton(linearize, N, fi, fj, x0, iters=8):
    x = x0.copy()
    hist = []
    for _ in range(iters):
        lin = linearize(x)
        hist.append(lin["chi2"])
        H, b = _dense(N, fi, fj, lin)
        dx = np.linalg.solve(H, b).reshape(N, 3)
        x = np.stack([O.se2_mul(x[v], O.se2_exp(dx[v])) for v in range(N)])   # Retract: origin * exp(v) (Sophus.h:53-57)
    hist.append(linearize(x)["chi2"])
    return x, hist


def test_gauss_newton_with_oracle_linearisation_converges():
    N = 60
    fi, fj, meas, sq, truth, init = make_graph(N, 40, seed=9)
    x, hist = _gauss_newton(lambda p: O.pgo_linearize(p, fi, fj, meas, sq), N, fi, fj, init)
    assert hist[-1] < 0.05 * hist[0] and abs(hist[-1] - hist[-2]) < 1e-6 * max(hist[-1], 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("N,loops", [(50, 30), (2000, 8000)])
def test_device_linearisation_matches_oracle(N, loops):
    fi, fj, meas, sq, truth, init = make_graph(N, loops, seed=N)
    g = F.PoseGraph(N, fi, fj, meas, sq)
    dev = g.linearize(init)
    orc = O.pgo_linearize(init, fi, fj, meas, sq)
    assert np.allclose(dev["err"], orc["err"], rtol=1e-12, atol=1e-13)
    assert np.allclose(dev["Hoff"], orc["Hoff"], rtol=1e-12, atol=1e-13)
    assert np.allclose(dev["Hdiag"], orc["Hdiag"], rtol=1e-12, atol=1e-12)
    assert np.allclose(dev["b"], orc["b"], rtol=1e-11, atol=1e-11)
    assert abs(dev["chi2"] - orc["chi2"]) <= 1e-11 * orc["chi2"]
    # a second call with other poses reuses the uploaded graph
    dev2 = g.linearize(truth)
    orc2 = O.pgo_linearize(truth, fi, fj, meas, sq)
    assert np.allclose(dev2["b"], orc2["b"], rtol=1e-11, atol=1e-11)
    g.close()


@pytest.mark.gpu
def test_gauss_newton_with_device_linearisation_converges_like_oracle():
    N = 80
    fi, fj, meas, sq, truth, init = make_graph(N, 60, seed=21)
    g = F.PoseGraph(N, fi, fj, meas, sq)
    xd, hd = _gauss_newton(lambda p: g.linearize(p), N, fi, fj, init)
    xo, ho = _gauss_newton(lambda p: O.pgo_linearize(p, fi, fj, meas, sq), N, fi, fj, init)
    assert hd[-1] < 0.05 * hd[0]
    assert np.allclose(hd, ho, rtol=1e-8)
    assert np.abs(xd - xo).max() < 1e-8
    g.close()


@pytest.mark.gpu
def test_device_linearisation_at_config5_size():
    """BASELINE config 5: 10k poses / 50k edges."""
    N = 10000
    fi, fj, meas, sq, truth, init = make_graph(N, 40000, seed=1)
    assert len(fi) == 50000
    g = F.PoseGraph(N, fi, fj, meas, sq)
    g.linearize(init)
    dev = g.linearize(init)
    orc = O.pgo_linearize(init, fi, fj, meas, sq)
    assert np.allclose(dev["Hdiag"], orc["Hdiag"], rtol=1e-12, atol=1e-11)
    assert np.allclose(dev["b"], orc["b"], rtol=1e-10, atol=1e-10)
    print(f"pgo linearize 10k poses / 50k factors: {dev['kernel_ms']:.3f} ms on the device")
    g.close()
