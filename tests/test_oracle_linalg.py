"""Known-answer tests pinning the oracle's Eigen/Sophus restatement (oracle/efo_linalg.h): the reference's
third-party arithmetic is un-vendored and unpinned, so the textbook identities are the pin."""
import ctypes as C

import numpy as np

import efo


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_ldlt6_solves_spd_and_semidefinite():
    rng = np.random.RandomState(0)
    for _ in range(20):
        J = rng.randn(40, 6)
        A = np.ascontiguousarray(J.T @ J)
        b = rng.randn(6)
        x = np.zeros(6)
        efo.lib().efo_ldlt6(_p(A), _p(b), _p(x))
        assert np.allclose(A @ x, b, rtol=1e-9, atol=1e-9)
    # badly scaled (ICP-like: translation vs rotation blocks)
    S = np.diag([1e4, 1e4, 1e4, 1, 1, 1.0])
    A = np.ascontiguousarray(S @ A @ S)
    x = np.zeros(6)
    efo.lib().efo_ldlt6(_p(A), _p(b), _p(x))
    assert np.allclose(A @ x, b, rtol=1e-7, atol=1e-7)


def test_ldlt3f():
    rng = np.random.RandomState(1)
    J = rng.randn(30, 3).astype(np.float32)
    A = np.ascontiguousarray(J.T @ J)
    b = rng.randn(3).astype(np.float32)
    x = np.zeros(3, np.float32)
    efo.lib().efo_ldlt3f(_p(A), _p(b), _p(x))
    assert np.allclose(A @ x, b, rtol=1e-4, atol=1e-4)


def test_rodrigues_matches_expm():
    from scipy.linalg import expm
    rng = np.random.RandomState(2)
    for s in (1e-12, 1e-6, 0.01, 1.0, 3.0):
        v = rng.randn(3)
        v = v / np.linalg.norm(v) * s
        R = np.zeros(9)
        efo.lib().efo_rodrigues(_p(v), _p(R))
        K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        assert np.allclose(R.reshape(3, 3), expm(K), atol=1e-12)


def test_polar3_is_nearest_rotation():
    rng = np.random.RandomState(3)
    for _ in range(20):
        Q, _ = np.linalg.qr(rng.randn(3, 3))
        if np.linalg.det(Q) < 0:
            Q[:, 0] *= -1
        A = np.ascontiguousarray(Q + 1e-3 * rng.randn(3, 3))
        R = np.zeros(9)
        efo.lib().efo_polar3(_p(A), _p(R))
        R = R.reshape(3, 3)
        U, _, Vt = np.linalg.svd(A)
        assert np.allclose(R, U @ Vt, atol=1e-12)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-13)


def test_se3_inverse_and_log():
    from scipy.linalg import logm
    rng = np.random.RandomState(4)
    for s in (1e-9, 1e-3, 0.5):
        w = rng.randn(3) * s
        R = np.zeros(9)
        efo.lib().efo_rodrigues(_p(w), _p(R))
        T = np.eye(4)
        T[:3, :3] = R.reshape(3, 3)
        T[:3, 3] = rng.randn(3) * s
        Ti = np.zeros(16)
        efo.lib().efo_se3_inverse(_p(np.ascontiguousarray(T.reshape(16))), _p(Ti))
        assert np.allclose(Ti.reshape(4, 4) @ T, np.eye(4), atol=1e-12)
        out6 = np.zeros(6)
        n = efo.lib().efo_se3_log_norm(_p(np.ascontiguousarray(T.reshape(16))), _p(out6))
        L = np.real(logm(T))
        ref = np.array([L[0, 3], L[1, 3], L[2, 3], L[2, 1], L[0, 2], L[1, 0]])
        assert np.allclose(out6, ref, atol=1e-9 + 1e-6 * s)
        assert abs(n - np.linalg.norm(ref)) < 1e-9 + 1e-6 * s


def test_expf_spec_accuracy():
    xs = np.linspace(-87, 0, 20001).astype(np.float32)
    got = np.array([efo.lib().efo_expf_spec(C.c_float(float(x))) for x in xs[::40]], np.float64)
    ref = np.exp(xs[::40].astype(np.float64))
    assert np.max(np.abs(got - ref) / ref) < 2.5e-7   # ~2 ulp
    assert efo.lib().efo_expf_spec(C.c_float(-100.0)) == 0.0
    assert efo.lib().efo_expf_spec(C.c_float(0.0)) == 1.0


def test_covariance_is_the_partial_pivot_lu_inverse():
    """RGBDOdometry::getCovariance = lastA.lu().inverse() (RGBDOdometry.cpp:573-575): known-answer against numpy on SPD
    normal matrices of the size and conditioning the tracker produces, and on a matrix that needs row pivoting."""
    rng = np.random.RandomState(3)
    for _ in range(20):
        J = rng.randn(200, 6) * np.array([1, 1, 1, 0.3, 0.3, 0.3])
        A = J.T @ J
        C = efo.covariance(A)
        assert np.allclose(C @ A, np.eye(6), atol=1e-9)
        assert np.allclose(C, np.linalg.inv(A), rtol=1e-8, atol=1e-12)
    P = np.eye(6)[[3, 0, 5, 1, 4, 2]] + 1e-3 * rng.randn(6, 6)
    assert np.allclose(efo.covariance(P) @ P, np.eye(6), atol=1e-10)
