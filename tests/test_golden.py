"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle).
CPU half: the oracle still reproduces them exactly. GPU half: the HIP path meets them through the C ABI."""
import os

import numpy as np
import pytest

from conftest import se3_log_norm

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(G, name))


# ------------------------------------------------------------------ CPU: oracle vs fixtures
def test_oracle_reproduces_pyramids():
    from oracle import oracle as O
    g = load("pyramid_64x48.npz")
    img = g["img"].astype(np.float32)
    for l, p in enumerate(O.image_pyramid(img, 3, True)):
        assert np.array_equal(p, g[f"smooth_l{l}"])
    for l, p in enumerate(O.image_pyramid(img, 3, False)):
        assert np.array_equal(p, g[f"plain_l{l}"])
    for l, p in enumerate(O.depth_pyramid(g["dep"], 3)):
        assert np.array_equal(p, g[f"depth_l{l}"])


def test_oracle_reproduces_lm():
    from oracle import oracle as O
    g = load("lm_120x160.npz")
    K = dict(f0=float(g["K"][0]), cx0=float(g["K"][1]), cy0=float(g["K"][2]))
    L0, L1, inv = g["L0"].astype(np.float32), g["L1"].astype(np.float32), g["inv"]
    p0, pd, p1 = O.image_pyramid(L0, 3, True), O.depth_pyramid(inv, 3), O.image_pyramid(L1, 3, True)
    for robust in range(3):
        for l in range(3):
            r = O.lm_accumulate(p0[l], p1[l], pd[l], l, g["T"], robust=robust, huber_delta=28.0, K=K)
            assert np.array_equal(r["acc"], g["accs"][robust, l])
        s = O.lm_solve(O.image_pyramid(L0, 3, flat=True), O.depth_pyramid(inv, 3, flat=True),
                       O.image_pyramid(L1, 3, flat=True), 120, 160, O.lm_params(max_iters=(10, 20, 30), robust=robust, K=K))
        assert np.array_equal(s["pose"], g[f"pose_r{robust}"])
        tr = np.array([[t["level"], t["iter"], t["n_res"], t["accepted"], t["stop"]] for t in s["trace"]], np.int32)
        assert np.array_equal(tr, g[f"trace_r{robust}"])
        assert np.array_equal(np.array([t["err"] for t in s["trace"]], np.float32), g[f"err_r{robust}"])


def test_oracle_reproduces_se3_and_ssd():
    from oracle import oracle as O
    g = load("se3_ssd.npz")
    for v, e in zip(g["a"], g["exps"]):
        assert np.array_equal(O.se3_exp(v), e)
    for i, v in enumerate(g["a"]):
        assert np.array_equal(O.se3_left_update((0.05 * v).astype(np.float32), g["exps"][(i + 1) % 40]), g["upd"][i])
    for r, t in zip(np.ascontiguousarray(g["s8"]), g["tree"]):
        assert O.lib().orc_ssd8_tree(r.ctypes.data_as(O._fp)) == t


def test_oracle_reproduces_depth():
    from oracle import oracle as O
    g = load("depth_120x160.npz")
    lm = load("lm_120x160.npz")
    dp = O.depth_params(grad_th=4.0, baseline=0.5, f0=float(lm["K"][0]), any_size=1)
    L, R = g["L"].astype(np.float32), g["R"].astype(np.float32)
    d1 = O.compute_depth(L, R, dp, stage=1)
    d2 = O.compute_depth(L, R, dp, stage=2)
    assert np.array_equal(d1["val"], g["val1"]) and np.array_equal(d1["disp"].astype(np.int16), g["disp1"])
    assert np.array_equal(d1["dep"], g["dep1"])
    assert np.array_equal(d2["val"], g["val2"]) and np.array_equal(d2["dep"], g["dep2"])
    assert [d1["n_selected"], d1["n_matched"], d2["n_valid"], d2["iters"], d2["status"]] == g["stats"].tolist()


# ------------------------------------------------------------------ GPU: HIP path vs fixtures
@pytest.mark.gpu
def test_gpu_pyramids_match_golden():
    from odometry_amd import api
    g = load("pyramid_64x48.npz")
    img = g["img"].astype(np.float32)
    ps, pn, pd = api.ImagePyramid(3, img, True), api.ImagePyramid(3, img, False), api.DepthPyramid(3, g["dep"], False)
    for l in range(3):
        assert np.array_equal(ps.GetPyramidImage(l), g[f"smooth_l{l}"])
        assert np.array_equal(pn.GetPyramidImage(l), g[f"plain_l{l}"])
        assert np.array_equal(pd.GetPyramidDepth(l), g[f"depth_l{l}"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [1, 2])
def test_gpu_lm_matches_golden(mode):
    from odometry_amd import api
    g = load("lm_120x160.npz")
    K = tuple(float(v) for v in g["K"])
    L0, L1, inv = g["L0"].astype(np.float32), g["L1"].astype(np.float32), g["inv"]
    p0, pd, p1 = api.ImagePyramid(3, L0, True), api.DepthPyramid(3, inv, False), api.ImagePyramid(3, L1, True)
    for robust in range(3):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30], np.eye(4), None, robust, 28.0, intrinsics=K)
        lm.set_mode(mode)
        for l in range(3):
            st, acc = lm.accumulate(p0, pd, p1, l, g["T"])
            assert st == 0 and acc[28] == g["accs"][robust, l][28]
            np.testing.assert_allclose(acc, g["accs"][robust, l], rtol=1e-11, atol=1e-9)
        T = lm.Solve(p0, pd, p1)
        assert lm.last_status == 0
        assert se3_log_norm(g[f"pose_r{robust}"], T) < 1e-5
        tr = lm.trace()
        got = np.array([[t["level"], t["iter"], t["n_res"], t["accepted"], t["stop"]] for t in tr], np.int32)
        assert np.array_equal(got, g[f"trace_r{robust}"])
        np.testing.assert_allclose([t["err"] for t in tr], g[f"err_r{robust}"], rtol=1e-6)
        np.testing.assert_allclose(np.array([t["delta"] for t in tr]), g[f"delta_r{robust}"], rtol=1e-5, atol=1e-9)


@pytest.mark.gpu
def test_gpu_depth_matches_golden():
    from odometry_amd import api
    g = load("depth_120x160.npz")
    lm = load("lm_120x160.npz")
    K = tuple(float(v) for v in lm["K"])
    L, R = g["L"].astype(np.float32), g["R"].astype(np.float32)
    de = api.DepthEstimator(4.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, 0.5, 80000,
                            intrinsics=K, any_size=True)
    val, disp, dep = np.zeros(L.shape, np.uint8), np.zeros(L.shape, np.float32), np.zeros(L.shape, np.float32)
    assert de.DisparityDepthEstimate(L, R, val, disp, dep) == 0
    assert np.array_equal(val, g["val1"]) and np.array_equal(disp.astype(np.int16), g["disp1"])
    assert np.array_equal(dep, g["dep1"])
    st = de.ComputeDepth(L, R, val, disp, dep)
    assert st == int(g["stats"][4])
    assert np.array_equal(val, g["val2"])
    np.testing.assert_allclose(dep, g["dep2"], rtol=0, atol=1e-7)
    rep = de.report()
    assert [rep["n_selected"], rep["n_matched"], rep["n_valid"], rep["iters"]] == g["stats"][:4].tolist()
