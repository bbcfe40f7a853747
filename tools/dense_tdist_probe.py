"""t-distribution weights on DENSE levels (1920x1080, every pixel a residual): Solve time with robust = 2 against robust = 1, and
the pose against the oracle.   python tools/dense_tdist_probe.py [--oracle]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import api, synth  # noqa: E402

K = (1100.0, 959.5, 539.5)
ROWS, COLS = 1080, 1920


def main():
    sc = synth.Scene(1)
    poses = synth.trajectory(3, 1)
    left, inv = [], []
    for T in poses[:2]:
        L, Z = sc.render(T, ROWS, COLS, *K)
        left.append(L)
        inv.append(np.where(Z < 99.0, 1.0 / np.maximum(Z, 1e-3), 0.0).astype(np.float32))
    p0, d0, p1 = api.ImagePyramid(4, left[0], False), api.DepthPyramid(4, inv[0], False), api.ImagePyramid(4, left[1], False)
    for robust in (1, 2):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0, intrinsics=K)
        T = lm.Solve(p0, d0, p1)
        ts = []
        for _ in range(5):
            lm.Reset(np.eye(4), 0.01)
            t0 = time.perf_counter()
            T = lm.Solve(p0, d0, p1)
            ts.append(time.perf_counter() - t0)
        ev = len(lm.trace())
        print(f"robust {robust}: Solve {1e3 * min(ts):.3f} ms, {ev} evaluations, {1e6 * min(ts) / ev:.1f} us per evaluation, status {lm.last_status}, "
              f"list levels {lm.points()[1][:4]}")
        if "--oracle" in sys.argv:
            from oracle import oracle as O
            KD = dict(f0=K[0], cx0=K[1], cy0=K[2])
            t0 = time.perf_counter()
            ref = O.lm_solve(O.image_pyramid(left[0], 4, False, flat=True), O.depth_pyramid(inv[0], 4, flat=True),
                             O.image_pyramid(left[1], 4, False, flat=True), ROWS, COLS, O.lm_params(robust=robust, K=KD))
            print(f"   oracle {time.perf_counter() - t0:.2f} s, {ref['n_evals']} evaluations, max |pose delta| {np.abs(ref['pose'] - T).max():.3g}")


if __name__ == "__main__":
    main()
