"""The pin: the oracle (and, with -m gpu, the HIP path) against fixtures produced by the REFERENCE'S OWN LINES.

tests/golden/ssd_ref.npz and cx_level_ref.npz are written by oracle/make_ref_fixtures.py, which compiles
src/depth_estimate.cpp:435-453 (ComputeSsdPattern8Sse), :380-395 (template taps, candidate loop, strict-< first minimum,
threshold, disparity, inverse depth) and include/image_processing_global.h:22-28 (GetCxLevel) and src/camera.cpp:61-65 (the camera pyramid's intrinsic rule) straight out of /root/reference
with the reference's build flags. These are the only fixtures in this repository that do not come from the restatement itself;
they pin D3's arithmetic and the principal-point rule. Everything else stays "parity unpinned" (DESIGN.md section 2)."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ssd():
    return dict(np.load(os.path.join(GOLD, "ssd_ref.npz")))


def test_cx_level_matches_reference_code():
    g = np.load(os.path.join(GOLD, "cx_level_ref.npz"))
    for i, c in enumerate(g["c"]):
        for j, l in enumerate(g["levels"]):
            assert np.float32(O.lib().orc_cx_level(float(c), int(l))) == g["out"][i, j]
    # the two constants of the hot path (ref: include/image_processing_global.h:35-36), levels 0-3 as SURVEY section 8 lists them
    assert np.allclose(g["out"][0, :4], [607.1928, 304.3464, 152.9232, 77.2116], rtol=0, atol=1e-4)


def test_camera_intrinsic_pyramid_matches_reference_code():
    """CameraPyramid::ConfigureCamera's per-level rule (ref: src/camera.cpp:61-65, the five update statements compiled as they
    stand): the oracle's orc_camera_intrinsics bit for bit, 32 cameras x 6 levels, in double."""
    g = np.load(os.path.join(GOLD, "cx_level_ref.npz"))
    for cin, want in zip(g["cam_in"], g["cam_out"]):
        P = np.zeros((3, 4))
        P[0, 0], P[1, 1], P[0, 1], P[0, 2], P[1, 2], P[2, 2] = cin[0], cin[1], cin[2], cin[3], cin[4], 1.0
        got = O.camera_intrinsics(P, 6)
        assert np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_camera_intrinsics_match_reference_code():
    """odo_camera_configure / odo_camera_intrinsics through the C ABI against the same fixture (KITTI-00 and the reference's own
    calibration-file camera)."""
    from odometry_amd import api
    g = np.load(os.path.join(GOLD, "cx_level_ref.npz"))
    for cin, want in zip(g["cam_in"][:4], g["cam_out"][:4]):
        cam = api.CameraPyramid(4, cin[0], cin[1], cin[2], cin[3], cin[4], 0.0, 0.0, 0.0, 0.0, 6.4, 4.8, 128, 96)
        P = np.zeros((3, 4))
        P[0, 0], P[1, 1], P[0, 1], P[0, 2], P[1, 2], P[2, 2] = cin[0], cin[1], cin[2], cin[3], cin[4], 1.0
        cam.ConfigureCamera(np.eye(3), P, (128, 96))
        for l in range(4):
            got = [cam.fx_double(l), cam.fy_double(l), cam.f_theta_double(l), cam.cx_double(l), cam.cy_double(l)]
            assert got == list(want[l]), (l, got, list(want[l]))
        cam.close()


def _schedule(fn, errs, lam, prec, mi):
    errs = np.ascontiguousarray(errs, np.float32)
    rec = np.zeros((len(errs), 5), np.int32)
    fin = C.c_int(0)
    n = fn(errs.ctypes.data_as(C.POINTER(C.c_float)), len(errs), C.c_float(lam), C.c_float(prec), int(mi),
           rec.ctypes.data_as(C.POINTER(C.c_int)), C.byref(fin))
    return rec[:n], n, fin.value


def test_lm_driver_schedule_matches_reference_code():
    """The LM loop of one pyramid level (ref: src/lm_optimizer.cpp:110-115,117,131-143,154-155 compiled as they stand, estimates as
    tags) replayed on 240 error sequences — accept / reject, lambda x5 and /5 with its floor, the precision and lambda breaks, the
    iteration budget, which estimate is current afterwards: the oracle's loop AND the device's state machine (odo_math.h
    lm_consume, host-compiled) reproduce every recorded word."""
    from test_hostemu_parity import load_emu
    g = np.load(os.path.join(GOLD, "lm_schedule_ref.npz"))
    emu = load_emu()
    n_break = n_budget = n_reject = 0
    for errs, meta, recs, outs in zip(g["errs"], g["meta"], g["recs"], g["outs"]):
        m, lam, prec, mi = int(meta[0]), float(meta[1]), float(meta[2]), int(meta[3])
        want = recs[:outs[0]]
        for fn in (O.lib().orc_lm_schedule, emu.emu_lm_schedule):
            rec, n, fin = _schedule(fn, errs[:m], lam, prec, mi)
            assert n == outs[0] and fin == outs[1]
            assert np.array_equal(rec, want)
        n_break += int(want[-1, 4])
        n_budget += int(outs[0] == mi and not want[-1, 4])
        n_reject += int(sum(1 for k in range(1, len(want)) if want[k, 2] == want[k - 1, 3] and want[k, 0] != want[k - 1, 0]))
    assert n_break > 50 and n_budget > 20 and n_reject > 200     # the fixture walks every branch


def _depth_schedule(fn, errs, lam, prec, mi):
    errs = np.ascontiguousarray(errs, np.float32)
    rec = np.zeros((max(len(errs), 1), 5), np.int32)
    fin, it = C.c_int(0), C.c_int(0)
    n = fn(errs.ctypes.data_as(C.POINTER(C.c_float)), len(errs), C.c_float(lam), C.c_float(prec), int(mi),
           rec.ctypes.data_as(C.POINTER(C.c_int)), C.byref(fin), C.byref(it))
    return rec[:n], n, fin.value, it.value


def test_depth_lm_driver_schedule_matches_reference_code():
    """The inverse-depth LM loop (ref: src/depth_estimate.cpp:92-96,141,150-161,167-168 compiled as they stand, depth vectors as
    tags) replayed on 240 error sequences — lambda x10 and /10 with its 1e-7 floor, the 1e5 and precision breaks, the iteration
    budget (0 included), which vector is current and which is `pre` afterwards, iter_count: the oracle's loop AND the rule every
    block of depth_lm_step_kernel runs (odo_math.h depth_lm_begin / depth_lm_decide / depth_lm_advance, host-compiled) reproduce
    every recorded word."""
    from test_hostemu_parity import load_emu
    g = np.load(os.path.join(GOLD, "depth_lm_schedule_ref.npz"))
    emu = load_emu()
    n_break = n_budget = n_zero = 0
    for errs, meta, recs, outs in zip(g["errs"], g["meta"], g["recs"], g["outs"]):
        m, lam, prec, mi = int(meta[0]), float(meta[1]), float(meta[2]), int(meta[3])
        want = recs[:outs[0]]
        for fn in (O.lib().orc_depth_lm_schedule, emu.emu_depth_lm_schedule):
            rec, n, fin, it = _depth_schedule(fn, errs[:m], lam, prec, mi)
            assert n == outs[0] and fin == outs[1] and it == outs[2]
            assert np.array_equal(rec, want)
        n_zero += int(outs[0] == 0)
        n_break += int(want[-1, 4]) if len(want) else 0
        n_budget += int(len(want) > 0 and outs[0] == mi and not want[-1, 4])
    assert n_break > 50 and n_budget > 20 and n_zero > 10


def test_ssd_tree_kats_match_reference_code(ssd):
    # left8 is in _mm256_set_ps ARGUMENT order (ref: src/depth_estimate.cpp:380-381); the oracle holds lanes low -> high
    for l8, r5, x, want in zip(ssd["kat_left8"], ssd["kat_rows5"], ssd["kat_x"], ssd["kat_ssd"]):
        got = O.ssd8_at(l8[::-1].copy(), r5, x, 2)
        assert np.float32(got) == want


@pytest.mark.parametrize("key", ["a", "b"])
def test_oracle_scan_matches_reference_code(ssd, key):
    r = O.disparity_scan(ssd[f"{key}_left_blur"], ssd[f"{key}_right_blur"], ssd[f"{key}_val"], int(ssd["boundary"]),
                         float(ssd["ssd_th"]), float(ssd["fx"]), float(ssd["baseline"]))
    assert r["n_selected"] == int(ssd[f"{key}_val"].sum()) > 400
    for name in ("best_ssd", "disp", "dep"):
        assert np.array_equal(r[name], ssd[f"{key}_{name}"]), name
    # the winning column wherever there was a candidate at all (x > boundary); at x == boundary the loop of :382 is empty and the
    # reference's match_coord is whatever the previous point left there (it is never reset, :277) — unobservable: :388 skips
    scanned = ssd[f"{key}_best_ssd"] < np.float32(1e10)
    assert np.array_equal(r["best_col"][scanned], ssd[f"{key}_best_col"][scanned])
    assert (ssd[f"{key}_disp"] > 0).sum() > 100   # the fixture exercises the matched branch (:390-394), not only `continue`


@pytest.mark.parametrize("key", ["a", "b"])
def test_fixture_inputs_are_what_the_oracle_front_end_produces(ssd, key):
    """The blurred pair and the mask the reference lines were fed are the oracle's own blur / selection of the raw pair, so a GPU
    run on the raw pair (below) is comparable with the fixture's outputs."""
    assert np.array_equal(O.blur3x3(ssd[f"{key}_left"]), ssd[f"{key}_left_blur"])
    assert np.array_equal(O.blur3x3(ssd[f"{key}_right"]), ssd[f"{key}_right_blur"])
    d = O.compute_depth(ssd[f"{key}_left"], ssd[f"{key}_right"], O.depth_params(any_size=1), stage=1)
    assert np.array_equal(d["val"], ssd[f"{key}_val"])
    assert np.array_equal(d["disp"], ssd[f"{key}_disp"]) and np.array_equal(d["dep"], ssd[f"{key}_dep"])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference checkout exists in the build container only")
def test_oracle_scan_matches_live_reference_build_on_fresh_inputs(tmp_path_factory):
    """In the build container the reference lines are compiled again — hash-checked, into a temporary directory OUTSIDE the
    repository, so nothing of the reference ever lies in the tree that travels — and run on inputs the fixture does not hold:
    20 random pairs, random masks, full-mantissa values."""
    from oracle import make_ref_fixtures as M
    L = M.load(M.build(str(tmp_path_factory.mktemp("odo_ref"))))
    rng = np.random.default_rng(99)
    for trial in range(20):
        rows, cols = int(rng.integers(12, 40)), int(rng.integers(24, 160))
        lb = (rng.random((rows, cols), np.float32) * 255).astype(np.float32)
        rb = (lb + rng.normal(0, 4.0, lb.shape).astype(np.float32)).astype(np.float32) if trial % 2 else np.roll(lb, -3, 1).copy()
        val = np.zeros((rows, cols), np.uint8)
        val[4:rows - 4, 4:cols - 4] = rng.random((rows - 8, cols - 8)) < 0.2
        base = float(np.float32(386.1448) / np.float32(718.856))
        disp, dep, best, col = M.ref_scan(L, lb, rb, val, 4, 900.0, base)
        r = O.disparity_scan(lb, rb, val, 4, 900.0, 718.856, base)
        assert np.array_equal(r["best_ssd"], best) and np.array_equal(r["best_col"][best < 1e10], col[best < 1e10])
        assert np.array_equal(r["disp"], disp) and np.array_equal(r["dep"], dep)
    for c in (607.1928, 185.2157, 0.0, 1e6):
        for l in range(8):
            assert O.lib().orc_cx_level(C.c_float(c), l) == L.ref_cx_level(C.c_float(c), l)
    rng2 = np.random.default_rng(123)
    for trial in range(50):   # fresh error sequences through the reference's LM loop lines and the oracle's
        errs = (rng2.uniform(100, 500) * np.cumprod(rng2.uniform(0.9, 1.08, int(rng2.integers(3, 40))))).astype(np.float32)
        want, fin = M.ref_lm_schedule(L, errs, 0.01, 0.995, 30)
        rec, n, f2 = _schedule(O.lib().orc_lm_schedule, errs, 0.01, 0.995, 30)
        assert n == len(want) and f2 == fin and np.array_equal(rec, want)
    for trial in range(50):   # and through the inverse-depth LM's
        errs = (rng2.uniform(20, 300) * np.cumprod(rng2.uniform(0.9, 1.08, int(rng2.integers(3, 60))))).astype(np.float32)
        want, fin, it = M.ref_depth_lm_schedule(L, errs, 0.01, 0.995, 50)
        rec, n, f2, i2 = _depth_schedule(O.lib().orc_depth_lm_schedule, errs, 0.01, 0.995, 50)
        assert n == len(want) and f2 == fin and i2 == it and np.array_equal(rec, want)
    out = np.zeros((5, 5))
    L.ref_camera_pyramid(1100.0, 1090.5, 0.25, 959.5, 539.5, 5, out.ctypes.data_as(C.POINTER(C.c_double)))
    P = np.zeros((3, 4))
    P[0, 0], P[1, 1], P[0, 1], P[0, 2], P[1, 2], P[2, 2] = 1100.0, 1090.5, 0.25, 959.5, 539.5, 1.0
    assert np.array_equal(O.camera_intrinsics(P, 5), out)


def test_reference_lines_are_never_built_inside_the_repository():
    """make_ref_fixtures.build() refuses a target directory under the repository root, and a line range whose text is not the
    recorded one is refused before anything is compiled (every path goes through extract())."""
    from oracle import make_ref_fixtures as M
    with pytest.raises(M.ReferencePinError):
        M.build(os.path.join(ROOT, "oracle", "_ref"))
    assert not os.path.exists(os.path.join(ROOT, "oracle", "_ref"))
    if os.path.isdir(M.REF):
        old = M.HASHES["scan"]
        M.HASHES["scan"] = "0" * 64
        try:
            with pytest.raises(M.ReferencePinError):
                M.extract("scan")
        finally:
            M.HASHES["scan"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("key", ["a", "b"])
def test_hip_disparity_matches_reference_code(ssd, key):
    """blur3x3_kernel -> depth_select_kernel -> depth_disparity_kernel through the C ABI (odo_depth_disparity) on the raw pair:
    mask, integer disparity and inverse depth bit-exact against what the reference's own scan lines produced."""
    from odometry_amd import api
    left, right = ssd[f"{key}_left"], ssd[f"{key}_right"]
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, float(ssd["baseline"]), 80000,
                            any_size=True)
    val, disp, dep = np.zeros(left.shape, np.uint8), np.zeros(left.shape, np.float32), np.zeros(left.shape, np.float32)
    assert de.DisparityDepthEstimate(left, right, val, disp, dep) == 0
    de.close()
    assert np.array_equal(val, ssd[f"{key}_val"])
    assert np.array_equal(disp, ssd[f"{key}_disp"])
    assert np.array_equal(dep, ssd[f"{key}_dep"])


def test_shared_device_arithmetic_matches_reference_code(ssd):
    """odometry_amd/csrc/odo_math.h — the header the HIP kernels compile — built for the host: its cx_level and ssd8_tree against
    the reference's GetCxLevel and ComputeSsdPattern8Sse outputs."""
    from test_hostemu_parity import load_emu
    lib = load_emu()
    g = np.load(os.path.join(GOLD, "cx_level_ref.npz"))
    for i, c in enumerate(g["c"]):
        for j, l in enumerate(g["levels"]):
            assert np.float32(lib.emu_cx_level(float(c), int(l))) == g["out"][i, j]
    fp = C.POINTER(C.c_float)
    for l8, r5, x, want in zip(ssd["kat_left8"], ssd["kat_rows5"], ssd["kat_x"], ssd["kat_ssd"]):
        Ll = np.ascontiguousarray(l8[::-1], np.float32)                      # lanes low -> high
        Rl = np.array([r5[4][x], r5[3][x - 1], r5[2][x + 2], r5[2][x], r5[2][x - 2], r5[1][x + 1], r5[1][x - 1], r5[0][x]], np.float32)
        assert np.float32(lib.emu_ssd8(Ll.ctypes.data_as(fp), Rl.ctypes.data_as(fp))) == want
