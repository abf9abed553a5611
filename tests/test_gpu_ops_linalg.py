"""GPU parity, operator tier: the tracking driver's small linear algebra (Eigen / Sophus restatement) as the DEVICE
evaluates it (elasticfusion_amd/csrc/ef_linalg_dev.hpp, through ef_op_linalg) against the oracle's efo_linalg.h.
Everything built from IEEE +,-,*,/,sqrt must agree bit for bit; sin/cos/atan2 come from two different libms
(glibc on the host, ocml on the device), so results that pass through them get a 2-ulp bar."""
import ctypes as C

import numpy as np
import pytest

import efo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from elasticfusion_amd import api
    return api.ops


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def ulps(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.spacing(np.maximum(np.abs(a), np.abs(b))), 5e-324)


def test_fp64_scalar_ops_are_ieee(ops):
    rng = np.random.RandomState(3)
    for _ in range(200):
        a, b = float(rng.uniform(1e-6, 50.0)), float(rng.uniform(-30.0, 30.0)) or 1.0
        out = ops.linalg("scalar", [a, b], 5)
        assert out[0] == np.sqrt(a), (a, out[0], np.sqrt(a))
        assert out[1] == a / b
        assert ulps(out[2:5], [np.sin(a), np.cos(a), np.arctan2(a, b)]).max() <= 1


def test_ldlt6_bit_exact(ops):
    rng = np.random.RandomState(0)
    for t in range(20):
        J = rng.randn(40, 6) * np.array([100, 100, 100, 1, 1, 1.0])
        A = np.ascontiguousarray(J.T @ J)
        if t % 5 == 4:
            A[2, :] = 0; A[:, 2] = 0   # a singular direction (Eigen leaves the component at 0)
        b = rng.randn(6)
        x_r = np.zeros(6)
        efo.lib().efo_ldlt6(_p(A), _p(b), _p(x_r))
        for which in ("ldlt6", "ldlt6_wave"):   # scalar restatement and the wave-parallel version the tracker runs
            x = ops.linalg(which, np.concatenate([A.reshape(-1), b]), 6)
            assert np.array_equal(x.view(np.uint64), x_r.view(np.uint64)), (which, t, x, x_r)


def test_ldlt6_degenerate_systems(ops):
    """what a frame without correspondences hands the solver (a covered lens: A = 0, b = 0), a rank-one system, and a system with a zero
    block: Eigen::LDLT's zero-pivot branch leaves the factorisation where it stands and the solve sets those components to 0"""
    rng = np.random.RandomState(5)
    cases = []
    cases.append((np.zeros((6, 6)), np.zeros(6)))
    cases.append((np.zeros((6, 6)), rng.randn(6)))
    v = rng.randn(6)
    cases.append((np.outer(v, v), rng.randn(6)))
    J = rng.randn(30, 3)
    A = np.zeros((6, 6))
    A[3:, 3:] = J.T @ J                                            # rotation constrained, translation not at all
    cases.append((A, rng.randn(6)))
    for t, (A, b) in enumerate(cases):
        A = np.ascontiguousarray(A, np.float64)
        x_r = np.zeros(6)
        efo.lib().efo_ldlt6(_p(A), _p(b), _p(x_r))
        for which in ("ldlt6", "ldlt6_wave"):
            x = ops.linalg(which, np.concatenate([A.reshape(-1), b]), 6)
            assert np.array_equal(x.view(np.uint64), x_r.view(np.uint64)), (which, t, x, x_r)
    assert not np.any(x_r[:3]) and np.all(np.isfinite(x_r))


def test_ldlt3f_bit_exact(ops):
    rng = np.random.RandomState(1)
    for _ in range(20):
        J = rng.randn(30, 3).astype(np.float32)
        A = np.ascontiguousarray(J.T @ J)
        b = rng.randn(3).astype(np.float32)
        x_r = np.zeros(3, np.float32)
        efo.lib().efo_ldlt3f(_p(A), _p(b), _p(x_r))
        x = ops.linalg("ldlt3f", np.concatenate([A.reshape(-1), b]).astype(np.float64), 3).astype(np.float32)
        assert np.array_equal(x.view(np.uint32), x_r.view(np.uint32))


def _small_rot(rng, s=0.05):
    v = rng.randn(3) * s
    R = np.zeros(9)
    efo.lib().efo_rodrigues(_p(np.ascontiguousarray(v)), _p(R))
    return R.reshape(3, 3)


def test_polar3_bit_exact(ops):
    rng = np.random.RandomState(2)
    for _ in range(20):
        A = np.ascontiguousarray((_small_rot(rng).astype(np.float32) + rng.randn(3, 3).astype(np.float32) * 1e-6).astype(np.float64))
        R_r = np.zeros(9)
        efo.lib().efo_polar3(_p(A), _p(R_r))
        R = ops.linalg("polar3", A, 9)
        assert np.array_equal(R.view(np.uint64), R_r.view(np.uint64)), np.abs(R - R_r).max()


def test_rodrigues_and_se3(ops):
    rng = np.random.RandomState(4)
    for _ in range(20):
        v = np.ascontiguousarray(rng.randn(3) * 0.02)
        R_r = np.zeros(9)
        efo.lib().efo_rodrigues(_p(v), _p(R_r))
        R = ops.linalg("rodrigues", v, 9)
        assert ulps(R, R_r).max() <= 2 or np.abs(R - R_r).max() < 1e-18      # sin / cos from two libms
        T = np.eye(4)
        T[:3, :3] = R_r.reshape(3, 3)
        T[:3, 3] = rng.randn(3) * 0.01
        T = np.ascontiguousarray(T)
        Ti_r = np.zeros(16)
        efo.lib().efo_se3_inverse(_p(T), _p(Ti_r))
        Ti = ops.linalg("se3_inverse", T, 16)
        assert np.array_equal(Ti.view(np.uint64), Ti_r.view(np.uint64))       # +,-,*,/,sqrt only
        efo.lib().efo_se3_log_norm.restype = C.c_double
        ln_r = efo.lib().efo_se3_log_norm(_p(T), None)
        ln = ops.linalg("se3_log_norm", T, 1)[0]
        assert abs(ln - ln_r) <= 1e-12 * max(ln_r, 1e-6)                      # atan2 / sin / cos inside
