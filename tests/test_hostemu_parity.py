"""odometry_amd/csrc/odo_math.h (the arithmetic the HIP kernels run) compiled for the host and stepped
serially must agree BIT FOR BIT with the oracle — the CPU-side half of the parity argument (no GPU needed)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from odometry_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fp = C.POINTER(C.c_float)


def load_emu():
    """Builds (if stale) and loads tests/hostemu.cpp: odometry_amd/csrc/odo_math.h compiled for the host."""
    so = os.path.join(ROOT, "tests", "_build_hostemu.so")
    src = os.path.join(ROOT, "tests", "hostemu.cpp")
    hdr = os.path.join(ROOT, "odometry_amd", "csrc", "odo_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-shared",
                               "-fPIC", "-o", so, src])
    lib = C.CDLL(so)
    lib.emu_ssd8.restype = C.c_float
    lib.emu_cx_level.restype = C.c_float
    lib.emu_cx_level.argtypes = [C.c_float, C.c_int]
    return lib


@pytest.fixture(scope="module")
def emu():
    return load_emu()


def P(a):
    return a.ctypes.data_as(fp)


def test_se3_functions_bit_exact(emu):
    rng = np.random.default_rng(0)
    for i in range(300):
        scale = [1e-6, 0.01, 0.3, 2.5][i % 4]
        a = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, scale, 3)]).astype(np.float32)
        M = np.zeros(16, np.float32)
        emu.emu_se3_exp(P(a), P(M))
        assert np.array_equal(M.reshape(4, 4).T, O.se3_exp(a))
        R = np.zeros(16, np.float32)
        emu.emu_se3_roundtrip(P(M), P(R))
        assert np.array_equal(R.reshape(4, 4).T, O.se3_roundtrip(M.reshape(4, 4).T))
        d = rng.normal(0, 0.05, 6).astype(np.float32)
        U = np.zeros(16, np.float32)
        emu.emu_se3_left_update(P(d), P(M), P(U))
        assert np.array_equal(U.reshape(4, 4).T, O.se3_left_update(d, M.reshape(4, 4).T))


def test_sincos_bit_exact(emu):
    lib = O.lib()
    xs = np.concatenate([np.linspace(-7, 7, 3001), np.random.default_rng(1).uniform(-1e3, 1e3, 2000)]).astype(np.float32)
    for x in xs:
        s, c = C.c_float(0), C.c_float(0)
        emu.emu_sincos(C.c_float(float(x)), C.byref(s), C.byref(c))
        assert s.value == lib.orc_sinf(float(x)) and c.value == lib.orc_cosf(float(x))


def test_solve_damped_bit_exact(emu):
    rng = np.random.default_rng(2)
    for _ in range(100):
        J = rng.normal(0, 1, (50, 6)) * np.array([1, 1, 1, 300, 300, 300])
        A = J.T @ J
        acc = np.zeros(29)
        k = 0
        for a in range(6):
            for b in range(a, 6):
                acc[k] = A[a, b]
                k += 1
        acc[21:27] = rng.normal(0, 10, 6)
        out = np.zeros(6, np.float32)
        emu.emu_solve_damped(acc.ctypes.data_as(C.POINTER(C.c_double)), C.c_float(0.01), P(out))
        assert np.array_equal(out, O.solve_damped(acc, 0.01))


def test_cx_level_bit_exact(emu):
    for c in (607.1928, 185.2157, 959.5, 80.0):
        for l in range(6):
            assert emu.emu_cx_level(c, l) == O.lib().orc_cx_level(c, l)


@pytest.fixture(scope="module")
def pyrs(kitti_seq):
    L0, L1, Z0 = kitti_seq["left"][0], kitti_seq["left"][1], kitti_seq["depth"][0]
    inv = synth.semi_dense_inverse_depth(Z0, L0)
    return O.image_pyramid(L0, flat=True), O.depth_pyramid(inv, flat=True), O.image_pyramid(L1, flat=True)


@pytest.mark.parametrize("robust", [0, 1])
def test_accumulators_bit_exact(emu, pyrs, robust):
    ip0, dp0, ip1 = pyrs
    T = O.se3_exp(np.array([0.02, -0.01, -0.3, 0.002, 0.01, -0.003], np.float32))
    Tc = np.ascontiguousarray(T.T)
    for lvl in range(4):
        r, c = O.level_dims(376, 1241, lvl)
        off = int(O.pyramid_size(376, 1241, lvl))
        I1, I2, D1 = (a[off:off + r * c].reshape(r, c) for a in (ip0, ip1, dp0))
        ref = O.lm_accumulate(I1, I2, D1, lvl, T, robust=robust)
        acc = np.zeros(29)
        emu.emu_lm_accumulate(P(I1), P(I2), P(D1), r, c, lvl, P(Tc), robust, C.c_float(28.0), C.c_float(718.856),
                              C.c_float(607.1928), C.c_float(185.2157), acc.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(acc, ref["acc"])


def test_full_solve_bit_exact(emu, pyrs):
    ip0, dp0, ip1 = pyrs
    ref = O.lm_solve(ip0, dp0, ip1, 376, 1241, O.lm_params())
    out = np.zeros(16, np.float32)
    ne = C.c_int(0)
    mi = (C.c_int * 4)(10, 20, 30, 30)
    init = np.eye(4, dtype=np.float32)
    st = emu.emu_lm_solve(P(ip0), P(dp0), P(ip1), 376, 1241, 4, mi, C.c_float(0.01), C.c_float(0.995), 1,
                          C.c_float(28.0), C.c_float(718.856), C.c_float(607.1928), C.c_float(185.2157), P(init), P(out),
                          C.byref(ne))
    assert st == ref["status"] == 0 and ne.value == ref["n_evals"]
    assert np.array_equal(out.reshape(4, 4).T, ref["pose"])


def test_ssd_tree_bit_exact(emu):
    rng = np.random.default_rng(3)
    for _ in range(1000):
        Lp = (rng.random(8) * 255).astype(np.float32)
        Rp = (rng.random(8) * 255).astype(np.float32)
        s = ((Lp - Rp) * (Lp - Rp)).astype(np.float32)
        assert emu.emu_ssd8(P(Lp), P(Rp)) == O.lib().orc_ssd8_tree(P(s))


def test_bilinear_sampling_mode_bit_exact_and_continuous(emu, pyrs):
    """The non-parity bilinear-sampling option (odo_lm_set_sampling): the device arithmetic (odo_math.h) against its oracle
    mode bit for bit on every level and through a whole Solve; and the property that motivates it — the objective is
    continuous in the pose, where floor sampling jumps."""
    ip0, dp0, ip1 = pyrs
    T = O.se3_exp(np.array([0.02, -0.01, -0.3, 0.002, 0.01, -0.003], np.float32))
    Tc = np.ascontiguousarray(T.T)
    O.set_sampling(True)
    emu.emu_set_sampling(1)
    try:
        floor_ref = None
        for lvl in range(4):
            r, c = O.level_dims(376, 1241, lvl)
            off = int(O.pyramid_size(376, 1241, lvl))
            I1, I2, D1 = (a[off:off + r * c].reshape(r, c) for a in (ip0, ip1, dp0))
            ref = O.lm_accumulate(I1, I2, D1, lvl, T, robust=1)
            acc = np.zeros(29)
            emu.emu_lm_accumulate(P(I1), P(I2), P(D1), r, c, lvl, P(Tc), 1, C.c_float(28.0), C.c_float(718.856),
                                  C.c_float(607.1928), C.c_float(185.2157), acc.ctypes.data_as(C.POINTER(C.c_double)))
            assert ref["status"] == 0 and np.array_equal(acc, ref["acc"])
            if lvl == 0:
                floor_ref = acc.copy()
        ref = O.lm_solve(ip0, dp0, ip1, 376, 1241, O.lm_params())
        out = np.zeros(16, np.float32)
        ne = C.c_int(0)
        mi = (C.c_int * 4)(10, 20, 30, 30)
        init = np.eye(4, dtype=np.float32)
        st = emu.emu_lm_solve(P(ip0), P(dp0), P(ip1), 376, 1241, 4, mi, C.c_float(0.01), C.c_float(0.995), 1,
                              C.c_float(28.0), C.c_float(718.856), C.c_float(607.1928), C.c_float(185.2157), P(init), P(out),
                              C.byref(ne))
        assert st == ref["status"] == 0 and ne.value == ref["n_evals"]
        assert np.array_equal(out.reshape(4, 4).T, ref["pose"])
        bil_pose = ref["pose"]
        # continuity: mean squared residual along a tiny translation sweep changes smoothly with bilinear sampling
        r0, c0 = 376, 1241
        I1, I2, D1 = (a[:r0 * c0].reshape(r0, c0) for a in (ip0, ip1, dp0))

        def cost(tx):
            Tt = T.copy()
            Tt[0, 3] += tx
            a = O.lm_accumulate(I1, I2, D1, 0, Tt, robust=0)["acc"]
            return a[27] / a[28]
        xs = np.linspace(0, 2e-3, 41)
        cb = np.array([cost(x) for x in xs])
    finally:
        O.set_sampling(False)
        emu.emu_set_sampling(0)
    cf = np.array([O.lm_accumulate(I1, I2, D1, 0, np.block([[T[:3, :3], (T[:3, 3] + [x, 0, 0]).reshape(3, 1)], [0, 0, 0, 1]]).astype(np.float32),
                                   robust=0)["acc"][27] / 1.0 for x in xs])
    assert np.abs(np.diff(cb)).max() < 0.05 * abs(cb.mean())          # smooth
    assert not np.array_equal(floor_ref, O.lm_accumulate(I1, I2, D1, 0, T, robust=1)["acc"])   # and a different objective
    # the two sampling modes agree on the motion to well under a pixel's worth
    floor_pose = O.lm_solve(ip0, dp0, ip1, 376, 1241, O.lm_params())["pose"]
    assert np.abs(bil_pose - floor_pose).max() < 0.05
    _ = cf
