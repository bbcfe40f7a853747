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
