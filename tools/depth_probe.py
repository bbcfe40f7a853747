#!/usr/bin/env python3
"""ComputeDepth timing probe on device-resident frames (odo_depth_compute_dev), nothing else running: DepthOptimization as one
persistent launch against a launch per iteration (ODO_DEPTH_NO_PERSIST). ODO_DEPTH_STAMPS=1: per-phase cycles of the persistent
kernel on stderr."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import api, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seq = synth.make_sequence(int(sys.argv[2]) if len(sys.argv) > 2 else 3, seed=0, drive="natural")
    ctx = api.Context(0)
    frames = [(ctx.upload(l), ctx.upload(r)) for l, r in zip(seq["left"], seq["right"])]
    rows, cols = seq["left"][0].shape
    val, disp, dep = ctx.alloc(rows * cols), ctx.alloc(4 * rows * cols), ctx.alloc(4 * rows * cols)
    base = float(np.float32(386.1448) / np.float32(718.856))
    for name, env in (("persistent launch", None), ("launch per iteration", "ODO_DEPTH_NO_PERSIST")):
        if env:
            os.environ[env] = "1"
        de = api.DepthEstimator(8.0, 900.0, 15.0, 0.1, 30.0, 0.01, 28.0, 0.995, 50, 4, None, None, base, 80000, ctx=ctx)
        if env:
            del os.environ[env]
        ts, its = [], []
        for i in range(n):
            l, r = frames[i % len(frames)]
            t0 = time.perf_counter()
            assert de.compute_dev(l, r, rows, cols, val, disp, dep) == 0
            ts.append(time.perf_counter() - t0)
            its.append(de.report()["iters"])
        st = de.time_stages(frames[0][0], frames[0][1], rows, cols)
        front = st["blur_us"] + st["select_us"] + st["scan_us"]
        print(f"{name}: ComputeDepth median {np.median(ts[5:]) * 1e6:.1f} us (front end blur + selection + scan {front:.1f} us), "
              f"depth-LM iterations mean {np.mean(its):.1f}, persistent {de.persistent_stats()}", flush=True)
        de.close()


if __name__ == "__main__":
    main()
