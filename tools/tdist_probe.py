#!/usr/bin/env python3
"""configs[0] timing probe: one 1241x376 pair through the test_optimizer.cpp parameter set (t-distribution weights) and through the
runner's Huber weights; median Solve time over N Solves. With ODO_COARSE_STAMPS=1 the library prints per-phase cycle counts of the
coarse and the persistent kernel to stderr every 100 Solves (the scale passes are inside the 'eval' phase)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import api, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seq = synth.make_sequence(2, seed=0, drive="natural")
    L0, R0, L1 = seq["left"][0], seq["right"][0], seq["left"][1]
    ctx = api.Context(0)
    de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None,
                            float(np.float32(386.1448) / np.float32(718.856)), 80000, ctx=ctx)
    val, disp, dep = np.zeros(L0.shape, np.uint8), np.zeros(L0.shape, np.float32), np.zeros(L0.shape, np.float32)
    de.ComputeDepth(L0, R0, val, disp, dep)
    de.close()
    p0, d0, p1 = api.ImagePyramid(4, L0, False, ctx=ctx), api.DepthPyramid(4, dep, False, ctx=ctx), api.ImagePyramid(4, L1, False, ctx=ctx)
    for name, robust in (("t_distribution", 2), ("huber", 1)):
        lm = api.LevenbergMarquardtOptimizer(0.01, 0.995, [10, 20, 30, 30], np.eye(4), None, robust, 28.0, ctx=ctx)
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            lm.Solve(p0, d0, p1)
            ts.append(time.perf_counter() - t0)
            lm.Reset(np.eye(4), 0.01)
        ev, launches, _ = lm.launch_stats()
        print(f"{name}: median {np.median(ts[5:]) * 1e3:.4f} ms, p10 {np.percentile(ts[5:], 10) * 1e3:.4f}, evaluations {ev}, launches {launches}, "
              f"points {lm.points()[0]}, {np.median(ts[5:]) * 1e6 / max(ev, 1):.2f} us / evaluation", flush=True)
        lm.close()


if __name__ == "__main__":
    main()
