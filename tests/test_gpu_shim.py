"""The C++ drop-in surface (include/odometry_shim.hpp): the example runner, which is the reference runner's frame loop
written against the shim, is compiled with g++, run on the GPU and compared with the oracle runner."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_runner_over_shim(tmp_path, kitti_seq):
    from oracle import runner as orunner
    exe = str(tmp_path / "run_odometry_synth")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "run_odometry_synth.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib")])
    frames = str(tmp_path / "frames.bin")
    L, R = kitti_seq["left"], kitti_seq["right"]
    with open(frames, "wb") as f:
        np.array([len(L), L[0].shape[0], L[0].shape[1]], np.int32).tofile(f)
        for l, r in zip(L, R):
            l.astype(np.float32).tofile(f)
            r.astype(np.float32).tofile(f)
    out = subprocess.run([exe, frames], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    rows = re.findall(r"frame (\d+) kf (\d+) motion ([\d.]+)\s+t = \[\s*([-\d.]+)\s+([-\d.]+)\s+([-\d.]+)\]", out.stdout)
    assert len(rows) == len(L) - 1
    ref = orunner.OracleRunner()
    ref.init(L[0], R[0])
    for k, row in enumerate(rows, start=1):
        c = ref.track(L[k], R[k])
        t = np.array([float(v) for v in row[3:6]])
        np.testing.assert_allclose(t, c["abs_pose"][:3, 3], atol=2e-5)
        assert abs(float(row[2]) - c["motion"]) < 2e-4
    assert "LM Optimizer failed! Invalid camera pointer!" in out.stdout   # the reference's null-camera warning


_BUILT = {}   # (source, flavour) -> executable: a program is compiled once per test session (the header is ~1 500 lines: 5-10 s each time)


def _build_once(src, flavour, flags):
    import atexit
    import shutil
    import tempfile
    if (src, flavour) not in _BUILT:
        d = tempfile.mkdtemp(prefix="odo_shim_build_", dir="/tmp")
        atexit.register(shutil.rmtree, d, True)
        exe = os.path.join(d, os.path.splitext(os.path.basename(src))[0] + "_" + flavour)
        subprocess.check_call(["g++", "-O2", "-std=c++17"] + flags + ["-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, src), "-o", exe,
                               "-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip",
                               "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib")])
        _BUILT[(src, flavour)] = exe
    return _BUILT[(src, flavour)]


def _build(tmp_path, src, name):
    return _build_once(src, "std", [])


def test_shim_mat_device_mirror(tmp_path):
    """The stand-in Mat's device mirror (uploads deduplicated, outputs downloaded lazily) never serves stale pixels: writes
    through the Mat or a header copy invalidate it, device-side results are fetched on first host access."""
    exe = _build(tmp_path, "tests/shim_mat_harness.cpp", "shim_mat_harness")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("OK"), out.stdout[-2000:] + out.stderr[-2000:]


def test_shim_poses_bit_identical_to_tracker(tmp_path):
    """The drop-in classes (host Mats, PCIe inside) and odo_tracker (device-resident frames) run the same kernels in the same
    order: pose_to_keyframe of every frame is bit-identical, and the timed mode reproduces it pass after pass."""
    from odometry_amd import api, synth
    seq = synth.make_sequence(24, seed=2)
    L, R = seq["left"], seq["right"]
    exe = _build(tmp_path, "examples/run_odometry_synth.cpp", "run_odometry_synth")
    frames = str(tmp_path / "frames.bin")
    with open(frames, "wb") as f:
        np.array([len(L), L[0].shape[0], L[0].shape[1]], np.int32).tofile(f)
        for l, r in zip(L, R):
            l.astype(np.float32).tofile(f)
            r.astype(np.float32).tofile(f)
    rel = str(tmp_path / "rel.bin")
    out = subprocess.run([exe, frames, "--time", "2", "--rel-bin", rel], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    m = re.search(r"SHIM_FPS ([\d.]+) FRAMES (\d+) PASSES 2", out.stderr)
    assert m and float(m.group(1)) > 0 and int(m.group(2)) == 2 * (len(L) - 1)
    got = np.fromfile(rel, np.float32).reshape(-1, 4, 4).transpose(0, 2, 1)   # column-major on file
    trk = api.Tracker(0)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(L, R)]
    trk.init(*dev[0])
    for k in range(1, len(L)):
        T = trk.track(*dev[k])["pose_to_keyframe"]
        assert np.array_equal(T, got[k - 1]), f"frame {k}"
    trk.close()
    # what the classes start ahead of the call that needs it (the partner's upload, ComputeDepth beside the Solve, the shared
    # :205 / :251 pyramid) changes nothing: the same poses with all of it switched off
    rel_off = str(tmp_path / "rel_off.bin")
    out = subprocess.run([exe, frames, "--time", "1", "--rel-bin", rel_off], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ODOMETRY_SHIM_NO_LOOKAHEAD="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert np.array_equal(np.fromfile(rel_off, np.float32), np.fromfile(rel, np.float32))


def _write_frames(path, L, R):
    with open(path, "wb") as f:
        np.array([len(L), L[0].shape[0], L[0].shape[1]], np.int32).tofile(f)
        for l, r in zip(L, R):
            l.astype(np.float32).tofile(f)
            r.astype(np.float32).tofile(f)


def _harness_runner(exe, frames, n):
    def run(mode, off=False, **extra_env):
        env = dict(os.environ, **extra_env)
        if off:
            env["ODOMETRY_SHIM_NO_LOOKAHEAD"] = "1"
        out = subprocess.run([exe, frames, mode], capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        lines = out.stdout.strip().splitlines()
        assert len(lines) == n and all(ln.split()[1] == "0" for ln in lines), out.stdout[-2000:]
        st = re.search(r"SHIM_STATS (.*)", out.stderr)
        stats = dict(zip(st.group(1).split()[0::2], map(int, st.group(1).split()[1::2])))
        return lines, stats
    return run


def test_lookahead_survives_refilled_swapped_and_poked_mats(tmp_path, monkeypatch):
    """What the drop-in classes start ahead (the partner's upload, the whole ComputeDepth, the next frame's upload and pyramid) is keyed
    by content stamps: two Mats refilled per frame like the reference's load_data, the two Mats swapping roles every frame, a right
    image modified between Solve and ComputeDepth and a LEFT image modified in place between Solve (:215) and ComputeDepth (:229) all
    give the poses and depth outputs of a run with the look-ahead switched off, bit for bit — and the refilled runs give those of the
    run with every frame in its own Mat."""
    from odometry_amd import synth
    seq = synth.make_sequence(10, seed=3)
    L, R = seq["left"], seq["right"]
    frames = str(tmp_path / "frames.bin")
    _write_frames(frames, L, R)
    run = _harness_runner(_build(tmp_path, "tests/shim_lookahead_harness.cpp", "shim_lookahead_harness"), frames, len(L))
    base, _ = run("vector", True)
    for mode in ("vector", "refill", "swap"):
        lines, stats = run(mode)
        assert lines == base, mode
        if mode != "swap":
            assert stats["early_adopted"] >= len(L) - 2, (mode, stats)     # ComputeDepth ran beside the Solve
    assert run("refill", True)[0] == base and run("swap", True)[0] == base
    poked = run("poke", True)[0]
    assert poked != base and run("poke")[0] == poked
    poked_left = run("poke_left", True)[0]
    assert poked_left != base and poked_left != poked and run("poke_left")[0] == poked_left
    # a persistent depth launch that gives up inside the job started ahead: ComputeDepth runs the job again
    monkeypatch.setenv("ODO_DEPTH_PERSIST_FAULT", "1")
    assert run("vector")[0] == base and run("refill")[0] == base


def test_cv_mat_build_has_the_lookahead_and_detects_in_place_writes(tmp_path, monkeypatch):
    """The build INTEGRATION.md prescribes (-DODOMETRY_SHIM_WITH_OPENCV, here against tests/stubs): a cv::Mat reports no writes, so its
    device mirror is keyed by the pixels' address and checked by a full-image fingerprint at every use.
      * the same lines as the stand-in build in every mode of tests/shim_lookahead_harness.cpp, look-ahead on and off;
      * refill (the reference's load_data shape): ONE upload per image per frame — the left image of :205 is still the mirror at :229
        and :251 —, and ComputeDepth of every frame but the first two is found finished beside the Solve;
      * poke / poke_left: a right image modified between Solve and ComputeDepth, and a LEFT image modified IN PLACE between :205 and
        :229 (an 8x8 patch the 48-word sample cannot see), are detected by the fingerprint: the job started ahead is dropped, the new
        pixels are uploaded, results equal the look-ahead-off run;
      * ODOMETRY_SHIM_VERIFY_MIRRORS=1 (every "unchanged" verdict checked by downloading the mirror and comparing bytes): no failure;
      * ODOMETRY_SHIM_LAZY_OUTPUTS=1 (left_disp / left_dep stay on the device until odometry::Download): the same lines;
      * outputs are written IN PLACE by default (the reference does, ref: src/depth_estimate.cpp:176-191,388-397 — a raw
        pointer taken from an output Mat before ComputeDepth stays that Mat's data pointer: mode raw_pointers); with the opt-in
        ODOMETRY_SHIM_SWAP_OUTPUTS=1 the three output images are built while Solve waits and handed over by header assignment when
        the caller's output Mats are theirs alone (refill: every frame but the first two) — and still written in place when they are
        not (shared_outputs: a second header on left_disp, left_dep in user memory): the same lines either way."""
    from odometry_amd import synth
    seq = synth.make_sequence(10, seed=3)
    L, R = seq["left"], seq["right"]
    n = len(L)
    frames = str(tmp_path / "frames.bin")
    _write_frames(frames, L, R)
    std = _harness_runner(_build(tmp_path, "tests/shim_lookahead_harness.cpp", "harness_std"), frames, n)
    cv = _harness_runner(_build_cv(tmp_path, "tests/shim_lookahead_harness.cpp", "harness_cv"), frames, n)
    base = std("vector", True)[0]
    for mode in ("vector", "refill", "swap"):
        for off in (False, True):
            assert cv(mode, off)[0] == base, (mode, off)
    lines, st = cv("refill")
    assert st["uploads"] == 2 * n, st                                  # one per image per frame
    assert st["early_adopted"] >= n - 2 and st["early_dropped"] == 0, st
    assert st["delivered"] == 3 * n and st["verify_failures"] == 0, st
    assert st["changed"] >= 2 * (n - 1) - 2, st                        # every refill was noticed
    assert st["outputs_prepared"] == 0, st                             # default: every output written into the caller's buffer
    lines, st = cv("raw_pointers")                                     # (exit code 4 = an output Mat came back with another buffer)
    assert lines == base and st["outputs_prepared"] == 0 and st["early_adopted"] >= n - 2, st
    assert std("raw_pointers")[0] == base
    swap = dict(ODOMETRY_SHIM_SWAP_OUTPUTS="1")
    lines, st = cv("refill", **swap)
    assert lines == base and st["outputs_prepared"] >= n - 2, st       # opt-in: the output images were built while Solve waited
    lines, st = cv("shared_outputs", **swap)
    assert lines == base and st["outputs_prepared"] == 0 and st["early_adopted"] >= n - 2, st   # ... and never swapped in under a second header
    assert std("shared_outputs")[0] == base and cv("shared_outputs")[0] == base
    lines, st = cv("refill", ODOMETRY_SHIM_SWAP_OUTPUTS="1", ODOMETRY_SHIM_KEEP_OUTPUT_BUFFERS="1")   # round 5's opt-out still wins
    assert lines == base and st["outputs_prepared"] == 0, st
    exe_cv = _build_cv(tmp_path, "tests/shim_lookahead_harness.cpp", "harness_cv")
    p = subprocess.run([exe_cv, frames, "raw_pointers"], capture_output=True, text=True, timeout=300, env=dict(os.environ, **swap))
    assert p.returncode == 4, p.returncode                             # the hand-over IS observable through a raw pointer: hence opt-in
    for mode in ("poke", "poke_left"):
        want = std(mode, True)[0]
        got, st = cv(mode)
        assert got == want and cv(mode, True)[0] == want, mode
        if mode == "poke_left":   # (refill shape; in "poke" every frame sits in a cv::Mat never seen before: nothing is started ahead)
            assert st["early_dropped"] >= 3, (mode, st)                # frames 1, 4, 7: the job started ahead was for other pixels
    # every "unchanged" verdict re-checked against the mirror's bytes
    lines, st = cv("refill", ODOMETRY_SHIM_VERIFY_MIRRORS="1")
    assert lines == base and st["verify_failures"] == 0 and st["unchanged"] > n, st
    lines, st = cv("poke_left", ODOMETRY_SHIM_VERIFY_MIRRORS="1")
    assert lines == std("poke_left", True)[0] and st["verify_failures"] == 0, st
    # outputs left on the device until asked for
    lines, st = cv("refill", ODOMETRY_SHIM_LAZY_OUTPUTS="1")
    assert lines == base and st["delivered"] == 3 * n, st              # (val by ComputeDepth, the other two by the harness's Download)
    assert cv("vector", ODOMETRY_SHIM_LAZY_OUTPUTS="1")[0] == base
    # a persistent depth launch that gives up inside the job started ahead: ComputeDepth runs the job again, outputs re-delivered
    monkeypatch.setenv("ODO_DEPTH_PERSIST_FAULT", "1")
    assert cv("refill")[0] == base and cv("vector")[0] == base


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_walk_over_caller_behaviour_same_results_in_every_build(tmp_path, seed):
    """tests/shim_fuzz_harness.cpp: 80 random operations a caller can perform on the classes' images — refill in place, small in-place
    writes to images and to ComputeDepth's inverse-depth output, pyramids of Mats and of header copies, ComputeDepth with left and right
    exchanged, into reused or fresh outputs, outputs handed back in, Solve + ComputeDepth + the :251-252 pyramids back to back. The
    stand-in build with the look-ahead off (writes seen through ptr<T>(), nothing started ahead) defines the lines; the stand-in build
    with the look-ahead on and the cv::Mat build (every decision rests on fingerprints; also with ODOMETRY_SHIM_VERIFY_MIRRORS and with
    lazy outputs) must print the same."""
    from odometry_amd import synth
    seq = synth.make_sequence(6, seed=4)
    frames = str(tmp_path / "frames.bin")
    _write_frames(frames, seq["left"], seq["right"])
    exes = {"std": _build(tmp_path, "tests/shim_fuzz_harness.cpp", "fuzz_std"), "cv": _build_cv(tmp_path, "tests/shim_fuzz_harness.cpp", "fuzz_cv")}

    def run(which, **env):
        out = subprocess.run([exes[which], frames, str(seed), "80"], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
        st = re.search(r"SHIM_STATS (.*)", out.stderr)
        return out.stdout.strip().splitlines(), dict(zip(st.group(1).split()[0::2], map(int, st.group(1).split()[1::2])))

    want, _ = run("std", ODOMETRY_SHIM_NO_LOOKAHEAD="1")
    assert len(want) > 60 and sum("depth" in ln for ln in want) > 10 and sum("solve" in ln for ln in want) > 5
    assert run("std")[0] == want
    got, st = run("cv")
    assert got == want, next((a, b) for a, b in zip(got, want) if a != b)
    assert st["unchanged"] > 10 and st["changed"] > 5                      # mirrors were reused AND in-place writes were noticed
    got, st = run("cv", ODOMETRY_SHIM_VERIFY_MIRRORS="1")
    assert got == want and st["verify_failures"] == 0
    assert run("cv", ODOMETRY_SHIM_LAZY_OUTPUTS="1")[0] == want
    assert run("cv", ODOMETRY_SHIM_NO_LOOKAHEAD="1")[0] == want
    assert run("cv", ODOMETRY_SHIM_SWAP_OUTPUTS="1")[0] == want             # opt-in: outputs handed over by header assignment where unobservable


def test_runner_with_load_data_inside_the_loop(tmp_path):
    """examples/run_odometry_synth.cpp --load-per-frame — the two Mats of a frame refilled inside the loop by convertTo from 8-bit
    images, the reference runner's own frame source (run_odometry_kitti_offline.cpp:200,334-359) — in both builds: the poses of the
    run with every frame preloaded, bit for bit, pass after pass."""
    from odometry_amd import synth
    seq = synth.make_sequence(14, seed=0)     # past the first keyframe switch (frame 9)
    frames = str(tmp_path / "frames.bin")
    _write_frames(frames, seq["left"], seq["right"])
    want = None
    for exe in (_build(tmp_path, "examples/run_odometry_synth.cpp", "synth_std"),
                _build_cv(tmp_path, "examples/run_odometry_synth.cpp", "synth_cv")):
        for extra in ([], ["--load-per-frame"]):
            rel = exe + ".rel"
            out = subprocess.run([exe, frames, "--time", "2", "--rel-bin", rel] + extra, capture_output=True, text=True, timeout=300)
            assert out.returncode == 0 and "SHIM_MISMATCH" not in out.stderr, out.stdout[-1500:] + out.stderr[-1500:]
            got = np.fromfile(rel, np.float32)
            want = got if want is None else want
            assert got.shape == want.shape and np.array_equal(got, want), (exe, extra)


_CV_FLAGS = ["-DODOMETRY_SHIM_WITH_OPENCV", "-DODOMETRY_SHIM_WITH_EIGEN", "-I" + os.path.join(ROOT, "tests", "stubs")]


def _build_cv(tmp_path, src, name):
    """The cv::Mat / Eigen branch of the shim — the one a maintainer of the reference builds — against tests/stubs (this image has
    neither library: the stubs are written from the documented APIs, test scaffolding only)."""
    return _build_once(src, "cv", ["-Wall", "-Werror"] + _CV_FLAGS)


def test_opencv_eigen_branch_tracks_bit_identically_to_the_stand_in_build(tmp_path):
    """examples/run_odometry_synth.cpp built twice — with the stand-in Mat / Affine4f and with -DODOMETRY_SHIM_WITH_OPENCV
    -DODOMETRY_SHIM_WITH_EIGEN (cv::Mat inputs staged and uploaded at every use, outputs downloaded at once, Eigen poses): the same
    pose_to_keyframe for every frame, bit for bit."""
    from odometry_amd import synth
    seq = synth.make_sequence(12, seed=2)
    L, R = seq["left"], seq["right"]
    frames = str(tmp_path / "frames.bin")
    with open(frames, "wb") as f:
        np.array([len(L), L[0].shape[0], L[0].shape[1]], np.int32).tofile(f)
        for l, r in zip(L, R):
            l.astype(np.float32).tofile(f)
            r.astype(np.float32).tofile(f)
    rels = []
    for exe in (_build(tmp_path, "examples/run_odometry_synth.cpp", "synth_standin"),
                _build_cv(tmp_path, "examples/run_odometry_synth.cpp", "synth_opencv")):
        rel = exe + ".rel"
        out = subprocess.run([exe, frames, "--rel-bin", rel], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        rels.append(np.fromfile(rel, np.float32).reshape(-1, 16))
    assert rels[0].shape == (len(L) - 1, 16) and np.array_equal(rels[0], rels[1])


def test_opencv_branch_honours_step_and_rejects_views_like_the_reference(tmp_path):
    """cv::Mat views (rows `step` bytes apart) through the drop-in classes: pyramids of a view equal those of its continuous clone;
    ComputeDepth refuses non-continuous inputs and outputs with the reference's message (ref: src/depth_estimate.cpp:259-263)."""
    exe = _build_cv(tmp_path, "tests/shim_opencv_harness.cpp", "shim_opencv_harness")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1].startswith("OK"), out.stdout[-2000:] + out.stderr[-2000:]


def test_cpp_kitti_runner_with_png_ingest(tmp_path, kitti_seq):
    """examples/run_odometry_kitti.cpp: KITTI directory layout, PNG decoding, tracking, error evaluation, KITTI pose file."""
    from oracle import runner as orunner
    from test_io import write_png
    exe = str(tmp_path / "run_odometry_kitti")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "run_odometry_kitti.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib")])
    root = tmp_path / "dataset"
    L, R, P = kitti_seq["left"], kitti_seq["right"], kitti_seq["poses"]
    for cam, imgs in ((0, L), (1, R)):
        d = root / "sequences" / "00" / f"image_{cam}"
        d.mkdir(parents=True)
        for i, im in enumerate(imgs):
            write_png(str(d / f"{i:06d}.png"), im.astype(np.uint8), "mix", 6, 65536)
    (root / "poses").mkdir()
    with open(root / "poses" / "00.txt", "w") as f:
        for T in P:
            f.write(" ".join("%e" % v for v in T[:3, :].reshape(-1)) + "\n")
    out_txt = str(tmp_path / "pred.txt")
    out = subprocess.run([exe, str(root), "00", str(len(L)), out_txt], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    pred = np.loadtxt(out_txt).reshape(-1, 3, 4)
    assert pred.shape[0] == len(L)
    ref = orunner.OracleRunner()
    ref.init(L[0], R[0], abs_pose0=np.eye(4, dtype=np.float32))
    np.testing.assert_allclose(pred[0], np.eye(4)[:3], atol=1e-6)
    for k in range(1, len(L)):
        c = ref.track(L[k], R[k])
        np.testing.assert_allclose(pred[k], c["abs_pose"][:3], atol=2e-6)     # 6 decimals in the file
    assert "avg error over" in out.stdout and "save completed." in out.stdout


def test_cpp_camera_pyramid_over_shim(tmp_path):
    """examples/camera_rectify.cpp: calibration file -> CameraPyramid x 2 -> ConfigureCamera -> UndistortRectify, against the
    oracle's restatement (bit-exact) on the calibration the reference ships with its camera test."""
    from oracle import oracle as O
    exe = str(tmp_path / "camera_rectify")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "camera_rectify.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib")])
    yaml = os.path.join(ROOT, "tests", "golden", "camchain.yaml")
    raw = [np.array([427.32814323885566, 429.48081105226316, 0.0, 367.1148716890002, 242.03387791215218]),
           np.array([425.28226969376584, 427.5013362691404, 0.0, 342.6156277602674, 233.38645927695092])]
    dist = [np.array([-0.35292630520315216, 0.09970701156068408, -0.0003265055193558261, -0.003400767380536901]),
            np.array([-0.34242635946786465, 0.09353275937137827, 0.000332922660566574, -0.001440982693394223])]

    def rot(rx, ry, rz):
        cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
        return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
                @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))
    Rs = [rot(0.0016, -0.0011, 0.0026), rot(-0.0016, 0.0011, -0.0026)]
    Ps = [np.array([[380.0, 0, 352.0, 0], [0, 380.0, 238.5, 0], [0, 0, 1, 0]]),
          np.array([[380.0, 0, 352.0, -22.95], [0, 380.0, 238.5, 0], [0, 0, 1, 0]])]
    with open(tmp_path / "RP.txt", "w") as f:
        for R, P in zip(Rs, Ps):
            f.write(" ".join(repr(float(v)) for v in R.reshape(-1)) + "\n")
            f.write(" ".join(repr(float(v)) for v in P.reshape(-1)) + "\n")
    rng = np.random.default_rng(11)
    frames = [rng.integers(0, 256, (480, 640)).astype(np.float32) for _ in range(2)]
    with open(tmp_path / "frames.bin", "wb") as f:
        for fr in frames:
            fr.tofile(f)
    out = subprocess.run([exe, yaml, str(tmp_path / "RP.txt"), str(tmp_path / "frames.bin"), str(tmp_path / "out.bin")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    got = np.fromfile(tmp_path / "out.bin", np.float32).reshape(2, 480, 640)
    for cam in range(2):
        mx, my = O.camera_init_maps(raw[cam], dist[cam], Rs[cam], Ps[cam], 480, 640)
        assert np.array_equal(got[cam], O.camera_remap(frames[cam], mx, my, 0.0))
        intr = O.camera_intrinsics(Ps[cam], 4)
        for l in range(4):
            m = re.search(rf"cam{cam} level {l} fx ([\d.]+) fy ([\d.]+) cx ([\d.]+) cy ([\d.]+) f_m ([\d.]+)", out.stdout)
            vals = [float(v) for v in m.groups()]
            np.testing.assert_allclose(vals[:4], [intr[l, 0], intr[l, 1], intr[l, 3], intr[l, 4]], rtol=0, atol=1e-8)
            np.testing.assert_allclose(vals[4], intr[l, 0] / (640 / 5.76), rtol=1e-8)   # f_meters, ref: include/camera.h:85
    assert "camera raw image is not 480x640!" in out.stdout
    assert "stereo configuration done!" in out.stdout


def test_c_example_over_the_tracker_abi(tmp_path):
    """examples/track_resident.c — plain C over odo_tracker_*, frames resident on the device, the next pair announced before
    every frame: the same pose_to_keyframe as the C++ runner over the drop-in classes, bit for bit, with and without the
    announcements, pass after pass."""
    from odometry_amd import synth
    seq = synth.make_sequence(14, seed=0)     # long enough for the first keyframe switch (frame 9)
    L, R = seq["left"], seq["right"]
    frames = str(tmp_path / "frames.bin")
    with open(frames, "wb") as f:
        np.array([len(L), L[0].shape[0], L[0].shape[1]], np.int32).tofile(f)
        for l, r in zip(L, R):
            l.astype(np.float32).tofile(f)
            r.astype(np.float32).tofile(f)
    shim = _build(tmp_path, "examples/run_odometry_synth.cpp", "run_odometry_synth")
    rel_shim = str(tmp_path / "rel_shim.bin")
    out = subprocess.run([shim, frames, "--rel-bin", rel_shim], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    want = np.fromfile(rel_shim, np.float32)
    exe = str(tmp_path / "track_resident")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "track_resident.c"), "-o", exe,
                           "-L" + os.path.join(ROOT, "odometry_amd", "lib"), "-lodometry_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "odometry_amd", "lib"), "-lm"])
    for extra in ([], ["--no-announce"]):
        rel = str(tmp_path / "rel_c.bin")
        out = subprocess.run([exe, frames, "--passes", "2", "--rel-bin", rel] + extra, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert re.search(r"TRACK_FPS ([\d.]+) FRAMES 26 PASSES 2", out.stderr), out.stderr[-500:]
        assert "Total keyframes: 1" in out.stdout or "Total keyframes:" in out.stdout
        got = np.fromfile(rel, np.float32)
        assert got.shape == want.shape and np.array_equal(got, want), extra
