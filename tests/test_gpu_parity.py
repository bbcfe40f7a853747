"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same inputs."""
import numpy as np
import pytest

from conftest import se3_log_norm

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def api():
    from odometry_amd import api
    api.default_context()  # raises if the HIP library or the device is missing: no silent fallback
    return api


# ---------------------------------------------------------------- pyramids ----------------------
@pytest.mark.parametrize("shape", [(376, 1241), (48, 64), (95, 131), (1080, 1920)])
@pytest.mark.parametrize("smooth", [True, False])
def test_image_pyramid_bit_exact(api, O, shape, smooth):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, shape).astype(np.float32)
    ref = O.image_pyramid(img, 4, smooth)
    pyr = api.ImagePyramid(4, img, smooth)
    assert pyr.GetNumberLevels() == 4
    for l in range(4):
        got = pyr.GetPyramidImage(l)
        assert got.shape == ref[l].shape
        assert np.array_equal(got, ref[l]), f"level {l} differs: max {np.abs(got - ref[l]).max()}"


def test_image_pyramid_non_integer_input_bit_exact(api, O):
    rng = np.random.default_rng(2)
    img = (rng.random((200, 333)) * 255).astype(np.float32)  # arbitrary fp32: rounding order matters at every level
    ref = O.image_pyramid(img, 4, True)
    pyr = api.ImagePyramid(4, img, True)
    for l in range(4):
        assert np.array_equal(pyr.GetPyramidImage(l), ref[l])


def test_depth_pyramid_bit_exact(api, O):
    rng = np.random.default_rng(3)
    dep = rng.random((376, 1241)).astype(np.float32)
    dep[rng.random(dep.shape) < 0.9] = 0
    ref = O.depth_pyramid(dep, 4)
    pyr = api.DepthPyramid(4, dep, False)
    for l in range(4):
        assert np.array_equal(pyr.GetPyramidDepth(l), ref[l])


@pytest.mark.parametrize("levels,shape", [(4, (376, 1241)), (4, (61, 83)), (6, (200, 264))])
def test_depth_pyramid_median_smoothing_bit_exact(api, O, levels, shape):
    """DepthPyramid(.., smooth = true): cv::medianBlur 3x3 into level 0, the levels below decimated from it (ref:
    src/image_processing_global.cpp:76-80,85-106) — the one-launch pyramid (<= 4 levels) and the level-by-level path (> 4)."""
    rng = np.random.default_rng(11)
    dep = rng.random(shape).astype(np.float32)
    dep[rng.random(dep.shape) < 0.6] = 0
    ref = O.depth_pyramid(dep, levels, smooth=True)
    pyr = api.DepthPyramid(levels, dep, True)
    for l in range(levels):
        assert np.array_equal(pyr.GetPyramidDepth(l), ref[l]), f"level {l}"
    assert not np.array_equal(ref[0], dep)     # the filter did something


def test_pyramid_bad_level(api):
    pyr = api.ImagePyramid(4, np.zeros((64, 64), np.float32), True)
    with pytest.raises(IndexError):
        pyr.GetPyramidImage(4)


# ---------------------------------------------------------------- LM accumulate -----------------
def _kitti_pyrs(api, O, seq, inv=None):
    from odometry_amd import synth
    L0, L1, Z0 = seq["left"][0], seq["left"][1], seq["depth"][0]
    if inv is None:
        inv = synth.semi_dense_inverse_depth(Z0, L0)
    return (L0, L1, inv, api.ImagePyramid(4, L0, True), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, L1, True),
            O.image_pyramid(L0, 4, True), O.depth_pyramid(inv, 4), O.image_pyramid(L1, 4, True))


@pytest.mark.parametrize("mode", [0, 1, 2])   # 0 auto, 1 dense scan, 2 keyframe point list
@pytest.mark.parametrize("robust", [0, 1, 2])
def test_lm_accumulate_matches_oracle(api, O, kitti_seq, robust, mode):
    L0, L1, inv, p0, d0, p1, r0, rd, r1 = _kitti_pyrs(api, O, kitti_seq)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0)
    lm.set_mode(mode)
    T = np.eye(4, dtype=np.float32)
    T[0, 3], T[2, 3] = 0.02, -0.3
    T[:3, :3] = O.se3_exp(np.array([0, 0, 0, 0.002, 0.01, -0.003], np.float32))[:3, :3]
    for level in range(4):
        st, acc = lm.accumulate(p0, d0, p1, level, T)
        ref = O.lm_accumulate(r0[level], r1[level], rd[level], level, T, robust=robust, huber_delta=28.0)
        assert st == 0 and ref["status"] == 0
        assert acc[28] == ref["acc"][28], "number of residuals must match exactly"
        np.testing.assert_allclose(acc, ref["acc"], rtol=1e-11, atol=1e-9)


def test_lm_accumulate_dense_matches_oracle(api, O, kitti_seq):
    Z0 = kitti_seq["depth"][0]
    inv = np.where(Z0 < 100, 1.0 / np.maximum(Z0, 1e-3), 0).astype(np.float32)  # every pixel valid (config 3 shape)
    L0, L1, inv, p0, d0, p1, r0, rd, r1 = _kitti_pyrs(api, O, kitti_seq, inv)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = -0.4
    ref = O.lm_accumulate(r0[0], r1[0], rd[0], 0, T, robust=1, huber_delta=28.0)
    for mode in (0, 2):
        lm.set_mode(mode)
        st, acc = lm.accumulate(p0, d0, p1, 0, T)
        assert acc[28] == ref["acc"][28] and acc[28] > 400000
        np.testing.assert_allclose(acc, ref["acc"], rtol=1e-11, atol=1e-9)
        assert lm.points()[1][0] == (1 if mode == 2 else 0)   # auto keeps the dense scan when every pixel has depth


def test_lm_accumulate_is_deterministic(api, O, kitti_seq):
    L0, L1, inv, p0, d0, p1, *_ = _kitti_pyrs(api, O, kitti_seq)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = -0.2
    for mode in (1, 2):
        lm.set_mode(mode)
        a = [lm.accumulate(p0, d0, p1, 0, T)[1] for _ in range(3)]
        assert np.array_equal(a[0], a[1]) and np.array_equal(a[0], a[2])


def test_keyframe_point_list_counts(api, O, kitti_seq):
    """The compacted list holds exactly the pixels the reference's scan would visit (|d| >= 0.01 inside the border)."""
    L0, L1, inv, p0, d0, p1, r0, rd, r1 = _kitti_pyrs(api, O, kitti_seq)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    lm.set_mode(0)   # auto (the default unless the ODO_LM_MODE diagnostic knob says otherwise)
    lm.accumulate(p0, d0, p1, 0, np.eye(4, dtype=np.float32))
    npts, use = lm.points()
    for l in range(4):
        d = rd[l][4:-4, 4:-4]
        assert npts[l] == int((np.abs(d) >= np.float32(0.01)).sum())
        assert use[l] == 1


# ---------------------------------------------------------------- LM solve ----------------------
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("robust", [1, 0, 2])
def test_lm_solve_pose_parity(api, O, kitti_seq, robust, mode):
    L0, L1, inv, p0, d0, p1, *_ = _kitti_pyrs(api, O, kitti_seq)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0)
    lm.set_mode(mode)
    T = lm.Solve(p0, d0, p1)
    ref = O.lm_solve(O.image_pyramid(L0, flat=True), O.depth_pyramid(inv, flat=True), O.image_pyramid(L1, flat=True),
                     376, 1241, O.lm_params(robust=robust))
    assert lm.last_status == 0 and ref["status"] == 0
    d = se3_log_norm(ref["pose"], T)
    assert d < 1e-5, f"pose delta {d} vs oracle"   # north_star tolerance: 1e-5 on the SE(3) log-map norm
    tr = lm.trace()
    assert len(tr) == ref["n_evals"]
    for a, b in zip(tr, ref["trace"]):
        assert (a["level"], a["iter"], a["n_res"], a["accepted"], a["stop"]) == \
               (b["level"], b["iter"], b["n_res"], b["accepted"], b["stop"])
        assert abs(a["err"] - b["err"]) <= 1e-6 * abs(b["err"])
    # recovered motion is close to the rendered one (sanity of the whole chain, not parity)
    gt = np.linalg.inv(kitti_seq["poses"][1]) @ kitti_seq["poses"][0]
    assert abs(T[2, 3] - gt[2, 3]) < 0.05


@pytest.mark.parametrize("robust", [1, 0])
def test_lean_kernels_give_the_recording_kernels_pose_bit_for_bit(api, O, kitti_seq, robust):
    """odo_lm_set_record(lm, 0) — what the trackers' own optimisers and the drop-in C++ class run — selects the lean builds of
    lm_coarse_kernel / lm_fine_kernel (no trace rows, no cost statistics, no t-distribution or bilinear path compiled in); a
    recording optimiser runs lm_coarse_full_kernel / lm_fine_trace_kernel. Same pose, same evaluation counts, bit for bit; the lean
    one has no trace to hand out."""
    L0, L1, inv, p0, d0, p1, *_ = _kitti_pyrs(api, O, kitti_seq)
    rec = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0)
    lean = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0)
    lean.set_record(False)
    Ta, Tb = rec.Solve(p0, d0, p1), lean.Solve(p0, d0, p1)
    assert rec.last_status == 0 and lean.last_status == 0
    assert np.array_equal(np.asarray(Ta).view(np.uint32), np.asarray(Tb).view(np.uint32))
    assert rec.launch_stats()[0] == lean.launch_stats()[0]        # evaluations
    assert len(rec.trace()) == rec.launch_stats()[0]
    with pytest.raises(Exception):
        lean.trace()
    lean.set_record(True)                                          # ... and back: the next Solve records again
    Tc = lean.Solve(p0, d0, p1)
    assert np.array_equal(np.asarray(Ta).view(np.uint32), np.asarray(Tc).view(np.uint32)) and len(lean.trace()) == rec.launch_stats()[0]


def test_lm_solve_sequence_with_reset(api, O, kitti_seq):
    """Reset(pose, lambda) then track the next frame against the same keyframe (runner usage :215,:268)."""
    from odometry_amd import synth
    L = kitti_seq["left"]
    inv = synth.semi_dense_inverse_depth(kitti_seq["depth"][0], L[0])
    p0, d0 = api.ImagePyramid(4, L[0], True), api.DepthPyramid(4, inv, False)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    lp = O.lm_params()
    ref_init = np.eye(4, dtype=np.float32)
    r0, rd = O.image_pyramid(L[0], flat=True), O.depth_pyramid(inv, flat=True)
    for k in (1, 2):
        T = lm.Solve(p0, d0, api.ImagePyramid(4, L[k], True))
        ref = O.lm_solve(r0, rd, O.image_pyramid(L[k], flat=True), 376, 1241, lp, init=ref_init)
        assert se3_log_norm(ref["pose"], T) < 1e-5
        assert lm.Reset(T, 0.01) == 0
        ref_init = ref["pose"]


def test_idle_callback_runs_while_solve_waits_and_changes_nothing(api, kitti_seq):
    """odo_lm_set_idle_callback: host work for the time a Solve leaves the calling thread idle (the cv::Mat build of the drop-in
    classes builds ComputeDepth's output images there). The callback runs on the caller's thread, many times per Solve, the pose is the
    pose without it, and NULL removes it. odo_ctx_mark_reached: never blocks; 1 once the stream has passed the mark."""
    import ctypes as C
    from odometry_amd import synth
    Ls = kitti_seq["left"]
    inv = synth.semi_dense_inverse_depth(kitti_seq["depth"][0], Ls[0])
    p0, d0, p1 = api.ImagePyramid(4, Ls[0], True), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, Ls[1], True)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    want = lm.Solve(p0, d0, p1)
    calls = []
    import threading
    me = threading.get_ident()
    cb = C.CFUNCTYPE(None, C.c_void_p)(lambda arg: calls.append(threading.get_ident()))
    lib = lm.ctx.lib
    assert lib.odo_lm_set_idle_callback(lm.h, C.cast(cb, C.c_void_p), None) == 0
    assert lm.Reset(np.eye(4), 0.01) == 0
    got = lm.Solve(p0, d0, p1)
    assert np.array_equal(got, want)
    assert len(calls) > 3 and set(calls) == {me}, len(calls)
    assert lib.odo_lm_set_idle_callback(lm.h, None, None) == 0
    n = len(calls)
    assert lm.Reset(np.eye(4), 0.01) == 0
    assert np.array_equal(lm.Solve(p0, d0, p1), want) and len(calls) == n
    # marks
    mark = lib.odo_ctx_mark(lm.ctx.h)
    assert mark > 0 and lib.odo_ctx_mark_reached(lm.ctx.h, mark) in (0, 1)
    assert lib.odo_ctx_wait_mark(lm.ctx.h, mark) == 0 and lib.odo_ctx_mark_reached(lm.ctx.h, mark) == 1


def test_lm_solve_fails_without_depth(api):
    img = np.random.default_rng(0).integers(0, 255, (376, 1241)).astype(np.float32)
    p0 = api.ImagePyramid(4, img, True)
    d0 = api.DepthPyramid(4, np.zeros_like(img), False)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    T = lm.Solve(p0, d0, p0)
    assert lm.last_status == -1
    expect = np.eye(4, dtype=np.float32)
    expect[3, 3] = 0.0   # pseudo-identity (ref: src/lm_optimizer.cpp:48-52)
    assert np.array_equal(T, expect)


def test_lm_solve_rejects_mismatched_pyramids(api):
    a = api.ImagePyramid(4, np.zeros((376, 1241), np.float32), True)
    b = api.ImagePyramid(4, np.zeros((370, 1241), np.float32), True)
    d = api.DepthPyramid(4, np.zeros((376, 1241), np.float32), False)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    lm.Solve(a, d, b)
    assert lm.last_status == -1


def test_lm_solve_identical_frames_stays_identity(api, kitti_seq):
    """Size-independent property: tracking a frame against itself from identity must not move."""
    from odometry_amd import synth
    L0 = kitti_seq["left"][0]
    inv = synth.semi_dense_inverse_depth(kitti_seq["depth"][0], L0)
    p0, d0 = api.ImagePyramid(4, L0, True), api.DepthPyramid(4, inv, False)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    T = lm.Solve(p0, d0, p0)
    assert se3_log_norm(np.eye(4), T) < 1e-4   # floor(u) of a re-projected integer pixel may land one pixel left: not exactly 0


def test_lm_small_image_custom_intrinsics(api, O, small_seq):
    from odometry_amd import synth
    K = small_seq["K"]
    L0, L1, Z0 = small_seq["left"][0], small_seq["left"][1], small_seq["depth"][0]
    inv = synth.semi_dense_inverse_depth(Z0, L0, grad_th=6.0)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30], np.eye(4), None, 1, 28.0,
                                         intrinsics=(K["f0"], K["cx0"], K["cy0"]))
    T = lm.Solve(api.ImagePyramid(3, L0, True), api.DepthPyramid(3, inv, False), api.ImagePyramid(3, L1, True))
    ref = O.lm_solve(O.image_pyramid(L0, 3, flat=True), O.depth_pyramid(inv, 3, flat=True),
                     O.image_pyramid(L1, 3, flat=True), 120, 160, O.lm_params(max_iters=(10, 20, 30), K=K))
    assert lm.last_status == ref["status"] == 0
    assert se3_log_norm(ref["pose"], T) < 1e-5


# ---------------------------------------------------------------- depth estimator ---------------
def _depth_est(api, **kw):
    from odometry_amd.synth import KITTI_BASELINE
    return api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                              float(np.float32(386.1448) / np.float32(718.856)), 80000, **kw)


def _bufs(shape):
    return np.zeros(shape, np.uint8), np.zeros(shape, np.float32), np.zeros(shape, np.float32)


@pytest.mark.parametrize("max_disparity", [0, 128])
def test_disparity_bit_exact(api, O, kitti_seq, max_disparity):
    L, R = kitti_seq["left"][0], kitti_seq["right"][0]
    de = _depth_est(api, max_disparity=max_disparity)
    val, disp, dep = _bufs(L.shape)
    assert de.DisparityDepthEstimate(L, R, val, disp, dep) == 0
    ref = O.compute_depth(L, R, O.depth_params(max_disparity=max_disparity), stage=1)
    assert np.array_equal(val, ref["val"]), "selection mask differs"
    assert np.array_equal(disp, ref["disp"]), "integer disparity (argmin index) differs"
    assert np.array_equal(dep, ref["dep"])
    rep = de.report()
    assert rep["n_selected"] == ref["n_selected"] and rep["n_matched"] == ref["n_matched"]


def test_disparity_known_answer(api):
    """Integer-disparity KAT at full size: interior points of each band recover the exact disparity."""
    from odometry_amd import synth
    L, R, gt = synth.integer_disparity_pair(seed=1)
    de = _depth_est(api)
    val, disp, dep = _bufs(L.shape)
    assert de.DisparityDepthEstimate(L, R, val, disp, dep) == 0
    band = 24
    yy = np.arange(L.shape[0])[:, None] % band
    interior = (yy >= 3) & (yy < band - 3) & (val == 1) & (disp > 0)
    xs = np.arange(L.shape[1])[None, :]
    interior &= (xs - gt) >= 6
    assert interior.sum() > 2000
    assert np.array_equal(disp[interior], gt[interior].astype(np.float32))


def test_compute_depth_matches_oracle(api, O, kitti_seq):
    for k in (0, 1):
        L, R = kitti_seq["left"][k], kitti_seq["right"][k]
        de = _depth_est(api)
        val, disp, dep = _bufs(L.shape)
        st = de.ComputeDepth(L, R, val, disp, dep)
        ref = O.compute_depth(L, R, O.depth_params(), stage=2)
        assert st == ref["status"] == 0
        rep = de.report()
        assert rep["iters"] == ref["iters"]
        assert np.array_equal(val, ref["val"])
        assert np.array_equal(disp, ref["disp"])
        np.testing.assert_allclose(dep, ref["dep"], rtol=0, atol=1e-7)
        assert abs(rep["cost"] - ref["cost"]) <= 1e-5 * abs(ref["cost"])
        assert rep["n_valid"] == ref["n_valid"] == int(val.sum())


def test_depth_lm_persistent_launch_and_its_fallback_give_the_same_depths(O, kitti_seq, monkeypatch):
    """DepthOptimization (ref: src/depth_estimate.cpp:141-191) runs as ONE persistent launch whose 32 workgroups wait for each
    other (depth_lm_persistent_kernel); a launch per iteration (depth_lm_step_kernel) is its fall-back. Same mask, disparities,
    inverse depths, iteration count and cost bit for bit: with the persistent launch, with it switched off (ODO_DEPTH_NO_PERSIST),
    and when a workgroup never publishes (ODO_DEPTH_PERSIST_FAULT: the launch gives up within its wait bound, the job is run again
    on the step launches, and after three such calls the estimator stays on them). Different iteration budgets, incl. 0 and 1."""
    from odometry_amd import api
    L, R = kitti_seq["left"][1], kitti_seq["right"][1]
    base = float(np.float32(386.1448) / np.float32(718.856))

    def run(max_iters, n=1):
        de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, max_iters, 4, None, None, base, 80000)
        outs = []
        for _ in range(n):
            val, disp, dep = _bufs(L.shape)
            st = de.ComputeDepth(L, R, val, disp, dep)
            rep = de.report()
            outs.append((st, val, disp, dep, rep["iters"], rep["cost"], rep["n_valid"]))
        ps = de.persistent_stats()
        de.close()
        return outs, ps

    def same(a, b):
        return a[0] == b[0] and all(np.array_equal(x, y) for x, y in zip(a[1:4], b[1:4])) and a[4:] == b[4:]

    for max_iters in (50, 7, 1, 0):
        ref = O.compute_depth(L, R, O.depth_params(max_iters=max_iters), stage=2)
        on, ps = run(max_iters)
        assert ps == (1, 0)
        assert on[0][0] == ref["status"] and on[0][4] == ref["iters"] and np.array_equal(on[0][1], ref["val"])
        np.testing.assert_allclose(on[0][3], ref["dep"], rtol=0, atol=1e-7)
        monkeypatch.setenv("ODO_DEPTH_NO_PERSIST", "1")
        off, ps = run(max_iters)
        monkeypatch.delenv("ODO_DEPTH_NO_PERSIST")
        assert ps == (0, 0) and same(on[0], off[0]), f"max_iters {max_iters}"
    monkeypatch.setenv("ODO_DEPTH_PERSIST_FAULT", "1")
    bad, ps = run(50, n=4)
    monkeypatch.delenv("ODO_DEPTH_PERSIST_FAULT")
    assert ps == (0, 3)                        # three calls were run again, the fourth went to the step launches directly
    on, _ = run(50)
    for b in bad:
        assert same(b, on[0])


@pytest.mark.parametrize("precision", [0.995, 1.0])
@pytest.mark.parametrize("max_disparity", [0, 128])
def test_compute_depth_with_test_disparity_cpp_parameters(O, max_disparity, precision, monkeypatch):
    """BASELINE.json configs[4] by its own program's parameters: test_disparity.cpp:68-75 constructs DepthEstimator(35, 1000, 10, 3, 17,
    0.01, 28, 0.995, 100, 4, nullptr, nullptr, baseline, 5000) — another operating point than the runner's (8, 900, 15, 0.1, 30, ..., 50):
    a 35-grey-level selection threshold, a 3-17 m depth window and a 100-iteration budget for the inverse-depth LM, all inside the
    persistent launch. Whole ComputeDepth on a KITTI-shaped pair (the reference's 376x1241 guard; KITTI baseline) against the oracle,
    reference range and +-128 px: mask and disparity bit-exact, inverse depth to 1e-7, iteration count and cost equal; the persistent
    launch and the step launches agree bit for bit. (max_residuals = 5000 is a buffer size the reference overruns when more points are
    valid, ref: src/depth_estimate.cpp:105-110 — undefined there, ignored here and in the oracle.) On the drives' textures the
    precision stop (0.995) ends the loop after 13-21 iterations; precision = 1.0 (a stop that never fires) makes the loop use its whole
    100-iteration budget inside the one launch."""
    from odometry_amd import api, synth
    seq = synth.make_sequence(2, seed=0, drive="dense")
    L, R = seq["left"][1], seq["right"][1]
    base = float(np.float32(386.1448) / np.float32(718.856))
    prm = O.depth_params(grad_th=35.0, ssd_th=1000.0, photo_th=10.0, min_depth=3.0, max_depth=17.0, lam=0.01, huber_delta=28.0,
                         precision=precision, max_iters=100, boundary=4, max_residuals=5000, max_disparity=max_disparity)
    ref = O.compute_depth(L, R, prm, stage=2)
    ref_scan = O.compute_depth(L, R, prm, stage=1)
    assert ref["n_selected"] > 1000 and 0 < ref["n_valid"] <= ref["n_matched"]
    assert ref["iters"] > 15 if precision < 1.0 else ref["iters"] >= 60, ref["iters"]
    # (at this operating point the drive keeps ~500 depths: 498 with the precision stop — "number of valid after optimization is too
    #  small", status -1, ref: src/depth_estimate.cpp:192-194 — and 528 without; the status is part of the parity)
    assert ref["status"] == (0 if ref["n_valid"] >= 500 else -1)

    def run():
        de = api.DepthEstimator(35.0, 1000.0, 10.0, 3.0, 17.0, 0.01, 28.0, precision, 100, 4, None, None, base, 5000, max_disparity=max_disparity)
        val, disp, dep = _bufs(L.shape)
        st = de.ComputeDepth(L, R, val, disp, dep)
        rep, ps = de.report(), de.persistent_stats()
        v1, d1, p1 = _bufs(L.shape)
        assert de.DisparityDepthEstimate(L, R, v1, d1, p1) == 0                      # DisparityDepthEstimate alone (:244-401)
        de.close()
        return st, val, disp, dep, rep, ps, (v1, d1, p1)

    st, val, disp, dep, rep, ps, scan = run()
    assert st == ref["status"] and ps == (1, 0)                               # one persistent launch, never gave up
    assert np.array_equal(scan[0], ref_scan["val"]) and np.array_equal(scan[1], ref_scan["disp"]) and np.array_equal(scan[2], ref_scan["dep"])
    assert np.array_equal(val, ref["val"]) and np.array_equal(disp, ref["disp"])
    np.testing.assert_allclose(dep, ref["dep"], rtol=0, atol=1e-7)
    assert rep["iters"] == ref["iters"] and rep["n_selected"] == ref["n_selected"] and rep["n_matched"] == ref["n_matched"]
    assert rep["n_valid"] == ref["n_valid"] and abs(rep["cost"] - ref["cost"]) <= 1e-6 * abs(ref["cost"])
    if max_disparity:
        assert float(disp.max()) <= max_disparity
    monkeypatch.setenv("ODO_DEPTH_NO_PERSIST", "1")
    st2, val2, disp2, dep2, rep2, ps2, _ = run()
    assert ps2 == (0, 0) and st2 == st and rep2 == rep
    assert np.array_equal(val2, val) and np.array_equal(disp2, disp) and np.array_equal(dep2, dep)


def test_compute_depth_started_ahead_on_another_stream_gives_the_same_outputs(kitti_seq, monkeypatch):
    """odo_depth_compute_begin_dev / _end_dev (what the drop-in classes do beside the Solve): the whole ComputeDepth enqueued on a second
    context's stream and collected later is bit-identical to odo_depth_compute_dev; a job collected with other stamps, or never
    collected (the next plain call), is dropped and recomputed; with the persistent depth-LM launch off nothing is started; a
    persistent launch that gives up inside a job started ahead is run again by the collecting call."""
    from odometry_amd import api
    ctx, side = api.default_context(), api.Context(0)
    base = float(np.float32(386.1448) / np.float32(718.856))
    rows, cols = kitti_seq["left"][0].shape
    n = rows * cols

    def outs():
        return ctx.alloc(n), ctx.alloc(4 * n), ctx.alloc(4 * n)

    def get(o):
        return (ctx.download(o[0], (rows, cols), np.uint8), ctx.download(o[1], (rows, cols), np.float32),
                ctx.download(o[2], (rows, cols), np.float32))

    def make():
        return api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, base, 80000)

    de = make()
    frames = [(ctx.upload(kitti_seq["left"][k]), ctx.upload(kitti_seq["right"][k])) for k in (1, 2)]
    ref = []
    for l, r in frames:
        o = outs()
        assert de.compute_dev(l, r, rows, cols, *o) == 0
        ref.append((get(o), de.report()))
    ctx.synchronize()

    def check(o, k):
        got = get(o)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref[k][0])) and de.report() == ref[k][1]

    # started ahead, collected with the same arguments
    for k, (l, r) in enumerate(frames):
        o = outs()
        assert de.compute_begin_dev(side, l, r, rows, cols, *o, 11 + k, 21 + k) == 0 and de.early_pending()
        assert de.compute_end_dev(l, r, rows, cols, *o, 11 + k, 21 + k) == 0 and not de.early_pending()
        check(o, k)
    # started for frame 0, collected for frame 1 (other images): dropped, frame 1 computed now
    o0, o1 = outs(), outs()
    assert de.compute_begin_dev(side, *frames[0], rows, cols, *o0, 31, 32) == 0
    assert de.compute_end_dev(*frames[1], rows, cols, *o1, 33, 34) == 0 and not de.early_pending()
    check(o1, 1)
    # the same images under another stamp (the Mat was refilled in between): dropped as well
    o2 = outs()
    assert de.compute_begin_dev(side, *frames[0], rows, cols, *o0, 41, 42) == 0
    assert de.compute_end_dev(*frames[0], rows, cols, *o2, 41, 43) == 0
    check(o2, 0)
    # never collected: the next plain call drops it
    assert de.compute_begin_dev(side, *frames[1], rows, cols, *o0, 51, 52) == 0
    assert de.compute_dev(*frames[0], rows, cols, *o2) == 0 and not de.early_pending()
    check(o2, 0)
    ps = de.persistent_stats()
    de.close()
    assert ps == (1, 0)
    # persistent depth-LM launch off: nothing is started (the step launches are paced by the host)
    monkeypatch.setenv("ODO_DEPTH_NO_PERSIST", "1")
    de = make()
    o = outs()
    assert de.compute_begin_dev(side, *frames[0], rows, cols, *o, 61, 62) == 1 and not de.early_pending()
    assert de.compute_end_dev(*frames[0], rows, cols, *o, 61, 62) == 0
    check(o, 0)
    de.close()
    monkeypatch.delenv("ODO_DEPTH_NO_PERSIST")
    # a launch that gives up inside the job started ahead: the collecting call runs the job again on the step launches
    monkeypatch.setenv("ODO_DEPTH_PERSIST_FAULT", "1")
    de = make()
    o = outs()
    assert de.compute_begin_dev(side, *frames[1], rows, cols, *o, 71, 72) == 0
    assert de.compute_end_dev(*frames[1], rows, cols, *o, 71, 72) == 0
    check(o, 1)
    assert de.persistent_stats()[1] == 1
    de.close()
    monkeypatch.delenv("ODO_DEPTH_PERSIST_FAULT")
    side.close()


def test_compute_depth_size_guard(api):
    de = _depth_est(api)
    val, disp, dep = _bufs((480, 640))
    img = np.zeros((480, 640), np.float32)
    assert de.ComputeDepth(img, img, val, disp, dep) == -1   # ref: src/depth_estimate.cpp:46-49


def test_compute_depth_flat_image_fails(api, O):
    img = np.full((376, 1241), 100.0, np.float32)
    de = _depth_est(api)
    val, disp, dep = _bufs(img.shape)
    st = de.ComputeDepth(img, img, val, disp, dep)
    ref = O.compute_depth(img, img, O.depth_params(), stage=2)
    assert st == ref["status"] == -1   # fewer than 500 valid points (ref: :192-197)
    assert np.array_equal(val, ref["val"])


def test_compute_depth_any_size(api, O, small_seq):
    L, R = small_seq["left"][0], small_seq["right"][0]
    K = small_seq["K"]
    de = api.DepthEstimator(4.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, 0.5, 80000,
                            intrinsics=(K["f0"], K["cx0"], K["cy0"]), any_size=True)
    val, disp, dep = _bufs(L.shape)
    st = de.ComputeDepth(L, R, val, disp, dep)
    ref = O.compute_depth(L, R, O.depth_params(grad_th=4.0, baseline=0.5, f0=K["f0"], any_size=1), stage=2)
    assert st == ref["status"]
    assert np.array_equal(val, ref["val"]) and np.array_equal(disp, ref["disp"])
    np.testing.assert_allclose(dep, ref["dep"], rtol=0, atol=1e-7)


# ---------------------------------------------------------------- tracker (runner loop) ---------
def test_tracker_matches_oracle_runner(api, kitti_seq):
    """Full frame loop (ref: run_odometry_kitti_offline.cpp:95-145,198-271): GPU tracker vs oracle runner."""
    from oracle import runner as orunner
    L, R = kitti_seq["left"], kitti_seq["right"]
    for overlap in (2, 1, 0):
        trk = api.Tracker(0, overlap_depth=overlap)
        dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(L, R)]
        trk.init(*dev[0])
        ref = orunner.OracleRunner()
        ref.init(L[0], R[0])
        for k in (1, 2, 1, 0, 1):
            g = trk.track(*dev[k])
            c = ref.track(L[k], R[k])
            assert g["solve_status"] == c["solve_status"] == 0
            assert se3_log_norm(c["pose_to_keyframe"], g["pose_to_keyframe"]) < 1e-5
            assert se3_log_norm(c["abs_pose"], g["abs_pose"]) < 1e-5
            assert g["new_keyframe"] == c["new_keyframe"]
            assert abs(g["motion"] - c["motion"]) < 1e-5
            val, disp, dep = trk.outputs(376, 1241)
            assert np.array_equal(val, c["val"]) and np.array_equal(disp, c["disp"])
            np.testing.assert_allclose(dep, c["dep"], rtol=0, atol=1e-7)
            assert trk.stats()["n_valid_depth"] == c["n_valid"]
        trk.close()


def test_tracker_keyframe_switch(api, kitti_seq):
    """A low motion threshold forces a keyframe switch on every frame; the next Solve must use the new keyframe."""
    from oracle import runner as orunner
    L, R = kitti_seq["left"], kitti_seq["right"]
    trk = api.Tracker(0, keyframe_motion_th=0.05)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(L, R)]
    trk.init(*dev[0])
    ref = orunner.OracleRunner(motion_th=0.05)
    ref.init(L[0], R[0])
    switched = 0
    for k in (1, 2, 1):
        g = trk.track(*dev[k])
        c = ref.track(L[k], R[k])
        assert g["new_keyframe"] == c["new_keyframe"]
        switched += g["new_keyframe"]
        assert se3_log_norm(c["pose_to_keyframe"], g["pose_to_keyframe"]) < 1e-5
        assert se3_log_norm(c["abs_pose"], g["abs_pose"]) < 1e-5
    assert switched >= 2 and trk.stats()["n_keyframes"] == ref.n_keyframes
    trk.close()


def test_wave_solver_matches_oracle_bit_for_bit(api, O):
    """The lane-parallel 6x6 solve of the update kernel vs the oracle's sequential one (incl. singular systems)."""
    import ctypes as C
    from odometry_amd import _lib
    ctx = api.default_context()
    rng = np.random.default_rng(7)
    for trial in range(60):
        J = rng.normal(0, 1, (40, 6)) * np.array([1, 1, 1, 300, 300, 300])
        if trial % 10 == 3:
            J[:, 2] = 0          # zero column -> zero pivot
        if trial % 10 == 7:
            J[:, 4] = J[:, 1]    # rank deficient (exactly dependent columns)
        A = J.T @ J
        acc = np.zeros(29)
        k = 0
        for a in range(6):
            for b in range(a, 6):
                acc[k] = A[a, b]
                k += 1
        acc[21:27] = rng.normal(0, 10, 6)
        lam = [0.01, 0.0, 6.25][trial % 3]
        out = np.zeros(6, np.float32)
        st = ctx.lib.odo_debug_solve(ctx.h, acc.ctypes.data_as(C.POINTER(C.c_double)), lam,
                                     out.ctypes.data_as(C.POINTER(C.c_float)))
        assert st == 0
        ref = O.solve_damped(acc, lam)
        assert np.array_equal(out, ref, equal_nan=True), (trial, out, ref)


# ---------------------------------------------------------------- further edge cases ------------
def test_pyramid_five_levels_and_single_level(api, O):
    """More than four levels takes the level-by-level kernels, one level is just the (blurred) input."""
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (160, 224)).astype(np.float32)
    for levels in (1, 2, 5):
        ref = O.image_pyramid(img, levels, True)
        pyr = api.ImagePyramid(levels, img, True)
        for l in range(levels):
            assert np.array_equal(pyr.GetPyramidImage(l), ref[l]), (levels, l)
    dep = (rng.random((160, 224)) * (rng.random((160, 224)) < 0.2)).astype(np.float32)
    refd = O.depth_pyramid(dep, 5)
    pd = api.DepthPyramid(5, dep, False)
    for l in range(5):
        assert np.array_equal(pd.GetPyramidDepth(l), refd[l])


def test_lm_single_level_and_tiny_interior(api, O, small_seq):
    from odometry_amd import synth
    K = small_seq["K"]
    L0, L1, Z0 = small_seq["left"][0], small_seq["left"][1], small_seq["depth"][0]
    inv = synth.semi_dense_inverse_depth(Z0, L0, grad_th=6.0)
    # one level only
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [15], np.eye(4), None, 1, 28.0, intrinsics=(K["f0"], K["cx0"], K["cy0"]))
    T = lm.Solve(api.ImagePyramid(1, L0, True), api.DepthPyramid(1, inv, False), api.ImagePyramid(1, L1, True))
    ref = O.lm_solve(O.image_pyramid(L0, 1, flat=True), O.depth_pyramid(inv, 1, flat=True), O.image_pyramid(L1, 1, flat=True),
                     120, 160, O.lm_params(max_iters=(15,), K=K))
    assert lm.last_status == ref["status"] == 0 and se3_log_norm(ref["pose"], T) < 1e-5
    # a 12x14 image has a 4x6 interior after the 4-px border; 8x8 has none -> the Solve fails like the reference
    for shape, expect_fail in (((12, 14), False), ((8, 8), True)):
        img = np.random.default_rng(5).integers(0, 255, shape).astype(np.float32)
        d = np.full(shape, 0.2, np.float32)
        lm1 = api.LevenbergMarquardtOptimizer(0.01, 0.995, [5], np.eye(4), None, 0, 28.0, intrinsics=(20.0, 7.0, 6.0))
        T1 = lm1.Solve(api.ImagePyramid(1, img, False), api.DepthPyramid(1, d, False), api.ImagePyramid(1, img, False))
        r1 = O.lm_solve(img.ravel(), d.ravel(), img.ravel(), shape[0], shape[1],
                        O.lm_params(max_iters=(5,), robust=0, K=dict(f0=20.0, cx0=7.0, cy0=6.0)))
        assert lm1.last_status == r1["status"] == (-1 if expect_fail else 0)
        assert np.array_equal(T1, r1["pose"]) or se3_log_norm(r1["pose"], T1) < 1e-5


def test_lm_inverse_depth_threshold_and_negative(api, O):
    """|d| < 0.01 is skipped, |d| == 0.01 kept, negative inverse depth warps behind the camera (ref: :193, h:45)."""
    rng = np.random.default_rng(9)
    img = rng.integers(0, 255, (64, 80)).astype(np.float32)
    d = np.zeros((64, 80), np.float32)
    d[10:50:3, 10:70:3] = 0.25
    d[12, 12] = 0.0099
    d[13, 13] = 0.01
    d[14, 14] = -0.25
    K = (60.0, 40.0, 32.0)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [5], np.eye(4), None, 1, 28.0, intrinsics=K)
    p, dp = api.ImagePyramid(1, img, False), api.DepthPyramid(1, d, False)
    for mode in (1, 2):
        lm.set_mode(mode)
        st, acc = lm.accumulate(p, dp, p, 0, np.eye(4, dtype=np.float32))
        ref = O.lm_accumulate(img, img, d, 0, np.eye(4, dtype=np.float32), robust=1, K=dict(f0=K[0], cx0=K[1], cy0=K[2]))
        assert acc[28] == ref["acc"][28]
        np.testing.assert_allclose(acc, ref["acc"], rtol=1e-11, atol=1e-9)


def test_depth_other_boundary(api, O, kitti_seq):
    L, R = kitti_seq["left"][2], kitti_seq["right"][2]
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 6, None, None,
                            float(np.float32(386.1448) / np.float32(718.856)), 80000)
    val, disp, dep = _bufs(L.shape)
    st = de.ComputeDepth(L, R, val, disp, dep)
    ref = O.compute_depth(L, R, O.depth_params(boundary=6))
    assert st == ref["status"]
    assert np.array_equal(val, ref["val"]) and np.array_equal(disp, ref["disp"])
    np.testing.assert_allclose(dep, ref["dep"], rtol=0, atol=1e-7)


def test_tracker_reports_depth_failure(api):
    """Flat images: ComputeDepth finds < 500 points -> the runner breaks (ref: run_odometry_kitti_offline.cpp:230-232)."""
    from odometry_amd import _lib
    flat = np.full((376, 1241), 90.0, np.float32)
    trk = api.Tracker(0)
    a, b = trk.upload_frame(flat), trk.upload_frame(flat)
    with pytest.raises(_lib.OdoError):
        trk.init(a, b)
    trk.close()


def test_tracker_solve_failure_is_not_fatal(api, kitti_seq):
    """No depth on the keyframe -> Solve returns the pseudo-identity, the loop carries on (ref: :215, lm_optimizer.cpp:60-65)."""
    L, R = kitti_seq["left"], kitti_seq["right"]
    trk = api.Tracker(0, min_depth=1000.0, max_depth=2000.0)   # every depth is filtered out -> init fails? no: < 500 -> -1
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(L, R)]
    from odometry_amd import _lib
    with pytest.raises(_lib.OdoError):
        trk.init(*dev[0])
    trk.close()


def test_tracker_next_frame_hint_changes_nothing(api, kitti_seq):
    """odo_tracker_hint_next only moves the next frame's pyramid build earlier: poses must be identical."""
    L, R = kitti_seq["left"], kitti_seq["right"]
    out = []
    for use_hint in (False, True):
        trk = api.Tracker(0)
        dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(L, R)]
        trk.init(*dev[0])
        seq = [1, 2, 1, 0, 2]
        poses = []
        for j, k in enumerate(seq):
            if use_hint and j + 1 < len(seq):
                trk.hint_next(dev[seq[j + 1]][0])
            poses.append(trk.track(*dev[k])["abs_pose"])
        out.append(np.stack(poses))
        trk.close()
    assert np.array_equal(out[0], out[1])


def test_bilinear_sampling_option_matches_its_oracle(api, O, kitti_seq):
    """odo_lm_set_sampling(ODO_SAMPLE_BILINEAR): a non-parity knob (the reference floors, ref: src/lm_optimizer.cpp:208-217)
    with its own oracle mode. Accumulators of every level (point list, dense scan) and a whole Solve (fused list pipeline and
    dense pipeline) against the oracle; the default stays floor sampling."""
    L0, L1, inv, p0, d0, p1, *_ = _kitti_pyrs(api, O, kitti_seq)
    T = np.eye(4, dtype=np.float32)
    T[2, 3] = -0.35
    T[0, 3] = 0.02
    ref_floor = O.lm_solve(O.image_pyramid(L0, flat=True), O.depth_pyramid(inv, flat=True), O.image_pyramid(L1, flat=True),
                           376, 1241, O.lm_params())
    O.set_sampling(True)
    try:
        i0, i1, dd = O.image_pyramid(L0), O.image_pyramid(L1), O.depth_pyramid(inv)
        for mode in (2, 1):   # keyframe point list, dense scan
            lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
            lm.set_sampling(True)
            lm.set_mode(mode)
            for level in range(4):
                st, acc = lm.accumulate(p0, d0, p1, level, T)
                ref = O.lm_accumulate(i0[level], i1[level], dd[level], level, T, robust=1)
                assert st == 0 and acc[28] == ref["acc"][28] > 0
                np.testing.assert_allclose(acc, ref["acc"], rtol=1e-11, atol=1e-7)
            Tg = lm.Solve(p0, d0, p1)
            ref = O.lm_solve(O.image_pyramid(L0, flat=True), O.depth_pyramid(inv, flat=True), O.image_pyramid(L1, flat=True),
                             376, 1241, O.lm_params())
            assert lm.last_status == 0 and ref["status"] == 0
            assert se3_log_norm(ref["pose"], Tg) < 1e-5
            assert lm.launch_stats()[0] == ref["n_evals"]
            lm.close()
    finally:
        O.set_sampling(False)
    # a fresh optimiser still floors
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
    assert se3_log_norm(ref_floor["pose"], lm.Solve(p0, d0, p1)) < 1e-5


@pytest.mark.gpu
def test_persistent_launch_and_its_fallback_give_the_same_pose(monkeypatch):
    """The fine levels run in ONE persistent launch whose workgroups wait for each other (lm_fine_kernel). With a workgroup that
    never publishes (ODO_LM_FINE_FAULT) the launch gives up within its wall-clock wait bound (4 ms) instead of hanging, the host
    redoes the Solve on the step launches, and after three such Solves the optimiser stays on them — for 4 096 Solves, then it tries
    once more (odo_lm_persistent_backoff): same pose bit for bit, as with the persistent launch switched off altogether."""
    from odometry_amd import api, synth
    seq = synth.make_sequence(2, seed=5, with_depth=True)
    L0, L1 = seq["left"][0], seq["left"][1]
    inv = synth.semi_dense_inverse_depth(seq["depth"][0], L0)
    p0, d0, p1 = api.ImagePyramid(4, L0, True), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, L1, True)
    K = (718.856, 607.1928, 185.2157)

    def solve(n=1):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0, intrinsics=K)
        out = [lm.Solve(p0, d0, p1).copy() for _ in range(n)]
        st = lm.persistent_stats()
        solve.backoff = lm.persistent_backoff()
        assert lm.last_status == 0
        lm.close()
        return out, st

    ref, st = solve()
    assert st[0] > 0 and st[1] == 0            # persistent launch on, never fell back
    assert solve.backoff == (0, 4096, 0)
    monkeypatch.setenv("ODO_LM_NO_FINE", "1")
    off, st = solve()
    assert st == (0, 0) and np.array_equal(off[0], ref[0])
    monkeypatch.delenv("ODO_LM_NO_FINE")
    monkeypatch.setenv("ODO_LM_FINE_FAULT", "1")
    import time
    t0 = time.perf_counter()
    bad, st = solve(4)
    assert st == (0, 3)                        # three Solves were redone, the fourth went to the step launches directly
    assert solve.backoff == (3, 4096, 4094)    # switched off; two Solves (the third redo + the fourth) of the 4 096 done
    assert time.perf_counter() - t0 < 2.0      # a give-up is a bounded wait of milliseconds (it was ~0.2 s each)
    for T in bad:
        assert np.array_equal(T, ref[0])


@pytest.mark.gpu
def test_tdistribution_weights_same_pose_on_every_pipeline(O, monkeypatch):
    """configs[0]'s own parameter set (ref: test_optimizer.cpp:53-54,59-67: unsmoothed pyramids, identity start, robust_estimator = 2)
    through (a) the coarse + persistent launches with the scale iteration inside them, (b) the coarse launch + the unfused pipeline
    below it (ODO_LM_NO_FINE), (c) the unfused pipeline alone (ODO_LM_UNFUSED), (d) a persistent launch that gives up
    (ODO_LM_FINE_FAULT: redone on (b)): the same pose and the same evaluation trace bit for bit — one summation order for the scale
    passes everywhere — and the oracle's trace."""
    from odometry_amd import api, synth
    seq = synth.make_sequence(2, seed=5, with_depth=True)
    L0, L1 = seq["left"][0], seq["left"][1]
    inv = synth.semi_dense_inverse_depth(seq["depth"][0], L0, stride_keep=0.1, seed=3)   # ~12 k points on level 0: every level fits
    p0, d0, p1 = api.ImagePyramid(4, L0, False), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, L1, False)
    ref = O.lm_solve(O.image_pyramid(L0, 4, False, flat=True), O.depth_pyramid(inv, 4, flat=True), O.image_pyramid(L1, 4, False, flat=True),
                     376, 1241, O.lm_params(robust=2))

    def solve(n=1):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 2, 28.0)
        out = []
        for _ in range(n):
            T = lm.Solve(p0, d0, p1).copy()
            assert lm.last_status == 0
            out.append((T, [(a["level"], a["iter"], a["n_res"], a["accepted"], a["stop"], a["err"], a["lambda_after"]) + tuple(a["delta"])
                            for a in lm.trace()], lm.launch_stats()[1]))
            lm.Reset(np.eye(4), 0.01)                      # ref: test_optimizer.cpp:104
        st = lm.persistent_stats()
        lm.close()
        return out, st

    fused, st = solve(2)
    assert st[0] > 0 and st[1] == 0
    assert fused[0][2] <= 3, f"{fused[0][2]} launches: the t-distribution Solve did not stay inside the coarse + persistent launches"
    assert se3_log_norm(ref["pose"], fused[0][0]) < 1e-5 and len(fused[0][1]) == ref["n_evals"]
    for a, b in zip(fused[0][1], ref["trace"]):
        assert a[:5] == (b["level"], b["iter"], b["n_res"], b["accepted"], b["stop"])
        assert abs(a[5] - b["err"]) <= 1e-6 * abs(b["err"])
    assert np.array_equal(fused[0][0], fused[1][0]) and fused[0][1] == fused[1][1]
    for env in ("ODO_LM_NO_FINE", "ODO_LM_UNFUSED", "ODO_LM_FINE_FAULT"):
        monkeypatch.setenv(env, "1")
        other, st = solve(2 if env == "ODO_LM_FINE_FAULT" else 1)
        monkeypatch.delenv(env)
        if env == "ODO_LM_FINE_FAULT":
            assert st[1] == 2                              # both Solves gave up and were redone
        else:
            assert st == (0, 0) or env == "ODO_LM_UNFUSED"
        for T, tr, _ in other:
            assert np.array_equal(T, fused[0][0]), env
            assert tr == fused[0][1], env
    for o in (p0, d0, p1):
        o.close()


def _trace_key(tr):
    return [(t["level"], t["iter"], t["n_res"], t["accepted"], t["stop"], float(t["err"]), float(t["lambda_after"]),
             tuple(float(v) for v in t["delta"])) for t in tr]


@pytest.mark.gpu
def test_two_pass_levels_inside_the_persistent_launch_equal_step_launches(api, kitti_seq, monkeypatch):
    """A level of 65-128 virtual blocks (a keyframe near the reference's 40 960-point selection cap, ref: src/depth_estimate.cpp:
    300-339) takes two passes per evaluation inside the persistent launch: the whole Solve is two launches, and pose and trace are
    those of the same level on step launches behind the persistent launch (ODO_LM_FINE_PASSES=1), of step launches only, and of a
    persistent launch whose two-pass level is forced to give up (ODO_LM_FINE_FAULT: redone on the step launches)."""
    from odometry_amd import synth
    L0, L1 = kitti_seq["left"][0], kitti_seq["left"][1]
    inv = synth.semi_dense_inverse_depth(kitti_seq["depth"][0], L0, stride_keep=0.2, seed=5)
    p0, d0, p1 = api.ImagePyramid(4, L0, True), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, L1, True)
    runs = {}
    for name, env in (("two passes", {}), ("step launches behind", {"ODO_LM_FINE_PASSES": "1"}), ("step launches only", {"ODO_LM_NO_FINE": "1"}),
                      ("two passes, forced give-up", {"ODO_LM_FINE_FAULT": "1", "ODO_LM_FINE_WAIT_US": "300"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
        T = lm.Solve(p0, d0, p1)
        runs[name] = (T, _trace_key(lm.trace()), lm.launch_stats()[1], lm.points()[0][:4], lm.persistent_stats())
        lm.close()
        for k in env:
            monkeypatch.delenv(k)
    n0 = runs["two passes"][3][0]
    assert 64 * 256 < n0 <= 128 * 256, n0                       # level 0: 65-128 virtual blocks
    assert runs["two passes"][2] == 2 and runs["step launches behind"][2] > 2 and runs["two passes"][4][1] == 0
    for name in ("step launches behind", "step launches only", "two passes, forced give-up"):
        assert np.array_equal(runs[name][0], runs["two passes"][0]) and runs[name][1] == runs["two passes"][1], name
    # virtual block 0 of a two-pass level never published (the test hook reaches that branch of lm_fine_body too): the launch gave up
    # within its bound and the Solve was redone on the step launches
    assert runs["two passes, forced give-up"][4][1] == 1, runs["two passes, forced give-up"][4]
    for o in (p0, d0, p1):
        o.close()


@pytest.mark.gpu
@pytest.mark.parametrize("robust", [1, 2])
@pytest.mark.parametrize("keep,iters", [(1.0, [10, 20, 30, 30]), (0.45, [10, 20, 30, 30]), (0.12, [10, 20, 30, 30, 30]),
                                        (0.45, [10, 20]), (0.03, [10, 20, 30, 30]), (1.0, [0, 20, 0, 30])])
def test_lm_solve_across_keyframe_densities_and_pyramid_depths(api, O, kitti_seq, keep, iters, robust):
    """Which kernel runs which level depends on the keyframe's point counts (coarse launch: levels of one round of its workgroup;
    persistent launch: the run of levels of <= 64 virtual blocks under it; step launches: what is left) — dense keyframes, sparse
    ones, two to five pyramid levels, zero budgets: the evaluation trace and the pose match the oracle in every split, and the
    persistent launch never falls back. robust = 2 (t-distribution weights, ref: src/lm_optimizer.cpp:257-261,338-358;
    test_optimizer.cpp:65): the scale iteration runs inside the coarse and the persistent launch, and what neither takes goes to
    the unfused pipeline below the hand-over level."""
    from odometry_amd import synth
    n = len(iters)
    L0, L1 = kitti_seq["left"][0], kitti_seq["left"][1]
    inv = synth.semi_dense_inverse_depth(kitti_seq["depth"][0], L0, stride_keep=keep, seed=3)
    p0, d0, p1 = api.ImagePyramid(n, L0, True), api.DepthPyramid(n, inv, False), api.ImagePyramid(n, L1, True)
    lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, iters, np.eye(4), None, robust, 28.0)
    T = lm.Solve(p0, d0, p1)
    ref = O.lm_solve(O.image_pyramid(L0, n, flat=True), O.depth_pyramid(inv, n, flat=True), O.image_pyramid(L1, n, flat=True),
                     376, 1241, O.lm_params(max_iters=tuple(iters), robust=robust))
    assert lm.last_status == ref["status"] == 0
    assert se3_log_norm(ref["pose"], T) < 1e-5
    tr = lm.trace()
    assert len(tr) == ref["n_evals"]
    for a, b in zip(tr, ref["trace"]):
        assert (a["level"], a["iter"], a["n_res"], a["accepted"], a["stop"]) == \
               (b["level"], b["iter"], b["n_res"], b["accepted"], b["stop"])
        assert abs(a["err"] - b["err"]) <= 1e-6 * abs(b["err"])
    npts, _ = lm.points()
    st = lm.persistent_stats()
    assert st[0] > 0 and st[1] == 0, (npts, st)
    lm.close()
