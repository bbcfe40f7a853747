"""Pins the oracle's SE(3) restatement (Sophus / Eigen semantics) against float64 matrix exponentials."""
import numpy as np
from scipy.linalg import expm

from oracle import oracle as O


def hat6(a):
    v, w = a[:3], a[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = v
    return M


def test_exp_matches_matrix_exponential():
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = np.concatenate([rng.normal(0, 1.0, 3), rng.normal(0, 0.3, 3)]).astype(np.float32)
        M = O.se3_exp(a)
        np.testing.assert_allclose(M, expm(hat6(a.astype(np.float64))), atol=3e-6)
        assert np.array_equal(M[3], [0, 0, 0, 1])


def test_exp_small_angle_branch():
    a = np.array([0.5, -0.25, 1.0, 3e-6, -2e-6, 1e-6], np.float32)   # theta < 1e-5: Taylor branch, V = R
    M = O.se3_exp(a)
    # V = R drops the 0.5 * omega x upsilon term (ref: se3.hpp:775-777): error ~ theta * |upsilon|
    np.testing.assert_allclose(M, expm(hat6(a.astype(np.float64))), atol=5e-6)
    Z = O.se3_exp(np.zeros(6, np.float32))
    assert np.array_equal(Z, np.eye(4, dtype=np.float32))


def test_exp_reference_quirk_cancellation():
    # theta just above the Taylor threshold: (1 - cosf(theta)) / theta^2 cancels to 0 in fp32 (Sophus float
    # behaviour, ref: se3.hpp:780-782) -> translation loses the 0.5*omega x upsilon term. The restatement keeps it.
    a = np.array([1.0, 0.0, 0.0, 0.0, 0.0, 1e-4], np.float32)
    M = O.se3_exp(a)
    assert M[0, 3] == 1.0 and abs(M[1, 3]) < 1e-6   # exact expm would give ~5e-5


def test_rotation_roundtrip_is_near_identity():
    rng = np.random.default_rng(1)
    for _ in range(100):
        a = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 1.5, 3)]).astype(np.float32)
        M = O.se3_exp(a)
        R = O.se3_roundtrip(M)
        np.testing.assert_allclose(R, M, atol=5e-7)


def test_quaternion_branch_trace_negative():
    # rotation by ~pi about each axis exercises the three non-trace branches of Eigen's R->q
    for axis in range(3):
        a = np.zeros(6, np.float32)
        a[3 + axis] = 3.1
        M = O.se3_exp(a)
        assert np.trace(M[:3, :3]) < 0
        np.testing.assert_allclose(O.se3_roundtrip(M), M, atol=1e-6)


def test_left_update_composition():
    rng = np.random.default_rng(2)
    cur = O.se3_exp(np.array([0.1, -0.2, 0.3, 0.02, -0.01, 0.03], np.float32))
    d = rng.normal(0, 0.05, 6).astype(np.float32)
    out = O.se3_left_update(d, cur)
    np.testing.assert_allclose(out, expm(hat6(d.astype(np.float64))) @ cur.astype(np.float64), atol=2e-6)


def test_sincos_agree_with_libm():
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-0.8, 0.8, 20000), rng.uniform(-50, 50, 20000)]).astype(np.float32)
    lib = O.lib()
    s = np.array([lib.orc_sinf(float(v)) for v in x], np.float32)
    c = np.array([lib.orc_cosf(float(v)) for v in x], np.float32)
    rs = np.sin(x.astype(np.float64)).astype(np.float32)   # correctly rounded reference
    rc = np.cos(x.astype(np.float64)).astype(np.float32)
    assert np.mean(s == rs) > 0.9999 and np.mean(c == rc) > 0.9999
    assert np.max(np.abs(s - rs)) <= np.spacing(np.float32(1.0))


def test_solve_damped_matches_numpy():
    rng = np.random.default_rng(4)
    for _ in range(50):
        J = rng.normal(0, 1, (200, 6)) * np.array([1, 1, 1, 30, 30, 30])
        r = rng.normal(0, 5, 200)
        A = J.T @ J
        acc = np.zeros(29)
        k = 0
        for a in range(6):
            for b in range(a, 6):
                acc[k] = A[a, b]
                k += 1
        acc[21:27] = J.T @ r
        lam = np.float32(0.01)
        ref = np.linalg.solve(A + float(lam) * np.diag(np.diag(A)), -(J.T @ r))
        np.testing.assert_allclose(O.solve_damped(acc, lam), ref, rtol=1e-5, atol=1e-9)
    # singular system: zero column -> zero step component, others solved
    acc = np.zeros(29)
    acc[0] = 4.0
    acc[21] = -2.0
    d = O.solve_damped(acc, 0.0)
    assert d[0] == 0.5 and np.all(d[1:] == 0)
