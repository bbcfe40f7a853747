"""BASELINE.json configs[2] at full size, against the oracle: 1920x1080, every pixel a residual (dense levels), intrinsics
f = 1100, c = image centre. A whole Solve (ref: src/lm_optimizer.cpp:73-160 at ~2 M residuals per level-0 evaluation), one
evaluation per level, ComputeDepth with the 376x1241 guard lifted (ref: src/depth_estimate.cpp:46-49), and the
shared-reciprocal division path of the dense kernel against the plain IEEE divisions."""
import os

import numpy as np
import pytest

from conftest import se3_log_norm

pytestmark = pytest.mark.gpu

K = (1100.0, 959.5, 539.5)
KD = dict(f0=K[0], cx0=K[1], cy0=K[2])
ROWS, COLS = 1080, 1920


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def api():
    from odometry_amd import api
    api.default_context()
    return api


@pytest.fixture(scope="module")
def scene():
    """Three 1080p frames of a forward drive with dense ground-truth inverse depth, and a stereo partner of frame 0."""
    from odometry_amd import synth
    sc = synth.Scene(1)
    poses = synth.trajectory(3, 1)
    out = dict(left=[], inv=[], poses=poses)
    for T in poses:
        L, Z = sc.render(T, ROWS, COLS, *K)
        out["left"].append(L)
        out["inv"].append(np.where(Z < 99.0, 1.0 / np.maximum(Z, 1e-3), 0.0).astype(np.float32))
    out["right0"], _ = sc.render(poses[0], ROWS, COLS, *K, 0.5)
    return out


def _pyrs(api, scene, k0, k1):
    return (api.ImagePyramid(4, scene["left"][k0], False), api.DepthPyramid(4, scene["inv"][k0], False),
            api.ImagePyramid(4, scene["left"][k1], False))


@pytest.mark.parametrize("robust,handover", [(1, True), (0, True), (1, False)])
def test_dense_1080p_solve_matches_oracle(api, O, scene, robust, handover, monkeypatch):
    """Full dense Solve on a 1080p pair, the test_optimizer.cpp call pattern (unsmoothed pyramids, identity start,
    ref: test_optimizer.cpp:53-54,90): identical evaluation trace (level, iteration, residual count, accept / stop) and pose
    within 1e-5 on the SE(3) log-map norm. handover: the two coarse levels (27 k and 108 k pixels, all with depth) run on their
    point lists through the fused pipeline, which hands the Solve over to the dense pipeline at level 1; without it
    (ODO_FUSE_DENSE_MAX=0) every level runs the dense scan."""
    if not handover:
        monkeypatch.setenv("ODO_FUSE_DENSE_MAX", "0")
    p0, d0, p1 = _pyrs(api, scene, 0, 1)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0, intrinsics=K)
    T = lm.Solve(p0, d0, p1)
    assert lm.points()[1][:4] == ([0, 0, 1, 1] if handover else [0, 0, 0, 0])   # which levels ran on their point list
    ref = O.lm_solve(O.image_pyramid(scene["left"][0], 4, False, flat=True), O.depth_pyramid(scene["inv"][0], 4, flat=True),
                     O.image_pyramid(scene["left"][1], 4, False, flat=True), ROWS, COLS, O.lm_params(robust=robust, K=KD))
    assert lm.last_status == 0 and ref["status"] == 0
    tr = lm.trace()
    assert len(tr) == ref["n_evals"]
    for a, b in zip(tr, ref["trace"]):
        assert (a["level"], a["iter"], a["n_res"], a["accepted"], a["stop"]) == \
               (b["level"], b["iter"], b["n_res"], b["accepted"], b["stop"])
        assert abs(a["err"] - b["err"]) <= 1e-6 * abs(b["err"])
    assert tr[-1]["level"] == 0 and max(t["n_res"] for t in tr) > 1.5e6
    d = se3_log_norm(ref["pose"], T)
    assert d < 1e-5, f"pose delta {d} vs oracle"
    gt = np.linalg.inv(scene["poses"][1]) @ scene["poses"][0]
    assert abs(T[2, 3] - gt[2, 3]) < 0.05


@pytest.mark.parametrize("variant", ["multi", "single", "fault"])
def test_dense_1080p_tdistribution_solve_matches_oracle(api, O, scene, variant, monkeypatch):
    """t-distribution weights on dense levels (ref: src/lm_optimizer.cpp:257-261,338-358 at 2 M residuals): the scale iteration runs on
    up to 128 workgroups that meet once per pass (lm_tdist_scale_multi_kernel), on one workgroup (ODO_TDIST_SINGLE=1: the fall-back),
    and through that fall-back after a workgroup of the multi launch never published (ODO_TDIST_MULTI_FAULT=1): the oracle's trace
    and pose every time."""
    if variant == "single":
        monkeypatch.setenv("ODO_TDIST_SINGLE", "1")
    if variant == "fault":
        monkeypatch.setenv("ODO_TDIST_MULTI_FAULT", "1")
        monkeypatch.setenv("ODO_LM_FINE_WAIT_US", "300")
    p0, d0, p1 = _pyrs(api, scene, 0, 1)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 2, 28.0, intrinsics=K)
    T = lm.Solve(p0, d0, p1)
    ref = O.lm_solve(O.image_pyramid(scene["left"][0], 4, False, flat=True), O.depth_pyramid(scene["inv"][0], 4, flat=True),
                     O.image_pyramid(scene["left"][1], 4, False, flat=True), ROWS, COLS, O.lm_params(robust=2, K=KD))
    assert lm.last_status == 0 and ref["status"] == 0
    tr = lm.trace()
    assert len(tr) == ref["n_evals"]
    for a, b in zip(tr, ref["trace"]):
        assert (a["level"], a["iter"], a["n_res"], a["accepted"], a["stop"]) == \
               (b["level"], b["iter"], b["n_res"], b["accepted"], b["stop"])
        assert abs(a["err"] - b["err"]) <= 1e-6 * abs(b["err"])
    assert tr[-1]["level"] == 0 and max(t["n_res"] for t in tr) > 1.5e6
    d = se3_log_norm(ref["pose"], T)
    assert d < 1e-5, f"pose delta {d} vs oracle"
    # odo_lm_tdist_stats: the multi-workgroup launch took the large levels; its fall-back ran exactly when a workgroup never published
    # (the two kernels add in different orders: after a fall-back sigma may differ in the last bits, which is why they are counted)
    multi, fallbacks = lm.tdist_stats()
    if variant == "single":
        assert multi == 0 and fallbacks == 0
    elif variant == "multi":
        assert multi > 0 and fallbacks == 0
    else:
        assert multi > 0 and 0 < fallbacks <= multi     # (launches queued for a level the loop had already left return at once, both kernels)


def test_dense_1080p_stream_matches_oracle(api, O, scene):
    """Two consecutive frames tracked the way test_optimizer.cpp does (ref: :86-105): Solve(frame k-1 -> k), Reset to the
    identity; both poses against the oracle."""
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K)
    prm = O.lm_params(robust=1, K=KD)
    for k in (1, 2):
        p0, d0, p1 = _pyrs(api, scene, k - 1, k)
        T = lm.Solve(p0, d0, p1)
        ref = O.lm_solve(O.image_pyramid(scene["left"][k - 1], 4, False, flat=True),
                         O.depth_pyramid(scene["inv"][k - 1], 4, flat=True),
                         O.image_pyramid(scene["left"][k], 4, False, flat=True), ROWS, COLS, prm)
        assert se3_log_norm(ref["pose"], T) < 1e-5
        assert lm.launch_stats()[0] == ref["n_evals"]
        assert lm.Reset(np.eye(4), 0.01) == 0
        for o in (p0, d0, p1):
            o.close()


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_dense_1080p_accumulators_match_oracle(api, O, scene, level):
    """One evaluation of every level at the rendered motion: N exact, the 29 sums to fp64 rounding."""
    p0, d0, p1 = _pyrs(api, scene, 0, 1)
    T = (np.linalg.inv(scene["poses"][1]) @ scene["poses"][0]).astype(np.float32)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K)
    st, acc = lm.accumulate(p0, d0, p1, level, T)
    i0 = O.split_levels(O.image_pyramid(scene["left"][0], 4, False, flat=True), ROWS, COLS, 4)[level]
    i1 = O.split_levels(O.image_pyramid(scene["left"][1], 4, False, flat=True), ROWS, COLS, 4)[level]
    dd = O.split_levels(O.depth_pyramid(scene["inv"][0], 4, flat=True), ROWS, COLS, 4)[level]
    ref = O.lm_accumulate(i0, i1, dd, level, T, robust=1, huber_delta=28.0, K=KD)
    assert st == 0 and ref["status"] == 0
    assert acc[28] == ref["acc"][28] > 0
    np.testing.assert_allclose(acc, ref["acc"], rtol=1e-11, atol=1e-6)


def test_shared_reciprocal_divisions_equal_plain_divisions(api, scene):
    """The dense kernel's shared-reciprocal division sequences (dense.hip.h) against the plain IEEE divisions it stands
    for: the same kernel with ODO_DENSE_PLAIN_DIV=1 must give the same 29 sums BIT FOR BIT on every level (same
    association order, so any difference would be a per-pixel difference)."""
    p0, d0, p1 = _pyrs(api, scene, 0, 1)
    T = (np.linalg.inv(scene["poses"][1]) @ scene["poses"][0]).astype(np.float32)
    res = {}
    for plain in (0, 1):
        if plain:
            os.environ["ODO_DENSE_PLAIN_DIV"] = "1"
        try:
            lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K)
        finally:
            os.environ.pop("ODO_DENSE_PLAIN_DIV", None)
        res[plain] = [lm.accumulate(p0, d0, p1, level, T)[1] for level in range(4)]
        lm.close()
    for level in range(4):
        assert np.array_equal(res[0][level], res[1][level]), f"level {level}: shared-reciprocal path differs"
        assert res[0][level][28] > 0


def test_compute_depth_1080p_any_size_matches_oracle(api, O, scene):
    """ComputeDepth at 1920x1080 with the hard 376x1241 check lifted (any_size, ref: src/depth_estimate.cpp:46-49):
    selection mask and integer disparities bit-exact, inverse depths equal after the depth LM."""
    L, R = scene["left"][0], scene["right0"]
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, 0.5, 80000,
                            intrinsics=K, any_size=True)
    val = np.zeros(L.shape, np.uint8)
    disp, dep = np.zeros(L.shape, np.float32), np.zeros(L.shape, np.float32)
    st = de.ComputeDepth(L, R, val, disp, dep)
    ref = O.compute_depth(L, R, O.depth_params(baseline=0.5, f0=K[0], any_size=1))
    assert st == ref["status"] == 0
    assert np.array_equal(val, ref["val"]) and int(val.sum()) == ref["n_valid"] > 500
    assert np.array_equal(disp, ref["disp"])          # integer argmin index: bit-exact
    np.testing.assert_allclose(dep, ref["dep"], rtol=0, atol=1e-7)
    # the guard itself still stands without the knob
    de2 = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, 0.5, 80000, intrinsics=K)
    assert de2.ComputeDepth(L, R, val, disp, dep) == -1


@pytest.mark.parametrize("robust", [1, 0])
def test_kitti_size_dense_solve_hands_over_at_level_0(api, O, kitti_seq, robust):
    """376x1241 with dense ground-truth depth: levels 3..1 (7 k / 28 k / 113 k pixels, all with depth) run on their point lists
    through the fused pipeline, level 0 (454 k) on the dense scan — the hand-over happens between levels 1 and 0. Trace and
    pose against the oracle; the tracker's early start (odo_lm_solve_begin) works on such a Solve too."""
    L, Z = kitti_seq["left"], kitti_seq["depth"]
    inv = np.where(Z[0] < 99.0, 1.0 / np.maximum(Z[0], 1e-3), 0.0).astype(np.float32)
    p0, d0, p1 = api.ImagePyramid(4, L[0], True), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, L[1], True)
    ref = O.lm_solve(O.image_pyramid(L[0], 4, True, flat=True), O.depth_pyramid(inv, 4, flat=True),
                     O.image_pyramid(L[1], 4, True, flat=True), 376, 1241, O.lm_params(robust=robust))
    for begin_early in (False, True):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0)
        if begin_early:
            assert lm.SolveBegin(p0, d0, p1) == 0
        T = lm.Solve(p0, d0, p1)
        assert lm.points()[1][:4] == [0, 1, 1, 1]
        assert lm.last_status == 0 and ref["status"] == 0
        tr = lm.trace()
        assert len(tr) == ref["n_evals"]
        for a, b in zip(tr, ref["trace"]):
            assert (a["level"], a["iter"], a["n_res"], a["accepted"], a["stop"]) == \
                   (b["level"], b["iter"], b["n_res"], b["accepted"], b["stop"])
        assert se3_log_norm(ref["pose"], T) < 1e-5
        lm.close()


# ---- several dense streams in the same launches (odo_lm_solve_batch over dense pyramids) -------------------------------------

@pytest.mark.parametrize("handover", [True, False])
def test_batched_dense_1080p_solves_equal_separate_solves(api, scene, handover, monkeypatch):
    """Three 1080p dense Solves (frame 0 -> 1, 1 -> 2, and 0 -> 2: different images, different iteration counts) in ONE batched
    call — fused point-list levels in the batched step launches, the dense levels below the hand-over in the batched
    evaluation + update launches — against the same three Solves one after the other: poses, evaluation counts and the per-
    level iteration counts bit for bit. handover = False (ODO_FUSE_DENSE_MAX=0): every level dense, the whole Solve unfused."""
    if not handover:
        monkeypatch.setenv("ODO_FUSE_DENSE_MAX", "0")
    pairs = [(0, 1), (1, 2), (0, 2)]
    pyrs = [_pyrs(api, scene, a, b) for a, b in pairs]
    want = []
    for p0, d0, p1 in pyrs:
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K)
        T = lm.Solve(p0, d0, p1)
        want.append((T, lm.launch_stats()[0], lm.report()[0]))
        lm.close()
    lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K) for _ in pairs]
    poses, status = api.solve_batch(lms, [p[0] for p in pyrs], [p[1] for p in pyrs], [p[2] for p in pyrs])
    assert status == [0, 0, 0]
    assert len({w[1] for w in want}) > 1          # the streams really differ in length
    for i, lm in enumerate(lms):
        assert np.array_equal(poses[i], want[i][0]), f"stream {i}"
        assert lm.launch_stats()[0] == want[i][1] and lm.report()[0] == want[i][2]
    # again on the same optimisers after a Reset (tables, progress words and states are reused)
    for lm in lms:
        lm.Reset(np.eye(4), 0.01)
    poses2, _ = api.solve_batch(lms, [p[0] for p in pyrs], [p[1] for p in pyrs], [p[2] for p in pyrs])
    assert np.array_equal(poses2, poses)


def test_batched_dense_failing_stream_fails_alone(api, scene):
    """A stream whose keyframe has no depth at all fails (pseudo-identity, ref: src/lm_optimizer.cpp:48-52) without disturbing the
    dense streams batched with it."""
    p0, d0, p1 = _pyrs(api, scene, 0, 1)
    dz = api.DepthPyramid(4, np.zeros_like(scene["inv"][0]), False)
    ref = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K)
    want = ref.Solve(p0, d0, p1)
    lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K) for _ in range(2)]
    poses, status = api.solve_batch(lms, [p0, p0], [d0, dz], [p1, p1])
    assert status[0] == 0 and np.array_equal(poses[0], want)
    assert status[1] == -1 and poses[1][3, 3] == 0 and poses[1][0, 0] == 1


@pytest.mark.parametrize("level", [0, 1])
def test_batched_dense_evaluation_counts_every_stream(api, scene, level):
    """odo_lm_time_eval_batch (bench.py's batched roofline leg): S streams in one launch produce S times the residuals."""
    p0, d0, p1 = _pyrs(api, scene, 0, 1)
    T = (np.linalg.inv(scene["poses"][1]) @ scene["poses"][0]).astype(np.float32)
    lms = [api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K) for _ in range(3)]
    one = lms[0].time_eval(p0, d0, p1, level, T, reps=3)
    t = api.time_eval_batch(lms, [p0] * 3, [d0] * 3, [p1] * 3, level, T, reps=3)
    assert t["n_points"] == 3 * one["n_points"] and abs(t["bytes"] - 3 * one["bytes"]) < 1.0
    assert t["mean_us"] > 0
