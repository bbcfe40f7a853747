"""Solve time of one KITTI-sized pair against the keyframe's point count (synth.semi_dense_inverse_depth's stride_keep), for the
setting of ODO_LM_FINE_PASSES in the environment (how many passes per evaluation a level may take inside the persistent launch
before it goes to step launches).   ODO_LM_FINE_PASSES=2 python tools/passes_probe.py 0.6 0.8 1.0"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import api, synth  # noqa: E402


def main():
    seq = synth.make_sequence(2, seed=0, drive="dense", with_depth=True)
    L0, L1, Z = seq["left"][0], seq["left"][1], seq["depth"][0]
    for keep in [float(a) for a in sys.argv[1:]] or [1.0]:
        inv = synth.semi_dense_inverse_depth(Z, L0, stride_keep=keep, seed=3)
        p0, d0, p1 = api.ImagePyramid(4, L0, True), api.DepthPyramid(4, inv, False), api.ImagePyramid(4, L1, True)
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, 1, 28.0)
        lm.Solve(p0, d0, p1)
        ts = []
        for _ in range(30):
            lm.Reset(np.eye(4), 0.01)
            t0 = time.perf_counter()
            lm.Solve(p0, d0, p1)
            ts.append(time.perf_counter() - t0)
        ev, launches, _ = lm.launch_stats()
        print(f"passes {os.environ.get('ODO_LM_FINE_PASSES', 'default')}: keep {keep}: points {lm.points()[0][:4]}, Solve median {1e3 * np.median(ts):.4f} ms, "
              f"{ev} evaluations, {launches} launches, persistent stats {lm.persistent_stats()}")
        lm.close()
        for o in (p0, d0, p1):
            o.close()


if __name__ == "__main__":
    main()
