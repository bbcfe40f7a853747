import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def kitti_seq():
    """Three KITTI-shaped synthetic stereo frames with ground-truth depth (seed 0)."""
    from odometry_amd import synth
    return synth.make_sequence(3, seed=0, with_depth=True)


@pytest.fixture(scope="session")
def small_seq():
    """Three 120x160 synthetic stereo frames (f=150, principal point at the centre)."""
    from odometry_amd import synth
    K = dict(f0=150.0, cx0=80.0, cy0=60.0)
    scene = synth.Scene(3, texels_per_m=12.0, tile_texels=(5, 11, 23))
    poses = synth.trajectory(3, 3, fwd_range=(0.15, 0.25))
    out = dict(left=[], right=[], depth=[], poses=poses, K=K, baseline=0.5)
    for T in poses:
        L, Z = scene.render(T, 120, 160, K["f0"], K["cx0"], K["cy0"], 0.0)
        R, _ = scene.render(T, 120, 160, K["f0"], K["cx0"], K["cy0"], 0.5)
        out["left"].append(L)
        out["right"].append(R)
        out["depth"].append(Z)
    return out


def se3_log_norm(Ta, Tb):
    """|| log(Ta^-1 Tb) ||_2 of the 6-vector twist, evaluated in float64 (parity metric)."""
    from scipy.linalg import logm
    D = np.linalg.inv(np.asarray(Ta, np.float64)) @ np.asarray(Tb, np.float64)
    Lg = np.real(logm(D))
    w = np.array([Lg[2, 1], Lg[0, 2], Lg[1, 0]])
    v = Lg[:3, 3]
    return float(np.sqrt(np.sum(w * w) + np.sum(v * v)))
