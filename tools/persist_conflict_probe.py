#!/usr/bin/env python3
"""Do the two persistent launches of a tracker (pose LM on stream A, depth LM on stream B) ever starve each other? Tracks N frames of
the natural drive with the next pair announced (bench.py's loop) and prints frames/s and how many Solves / depth jobs gave up and
were redone on the step launches. Environment knobs are read by the library: ODO_LM_NO_FINE, ODO_DEPTH_NO_PERSIST, ODO_LM_PRIORITY=0."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from odometry_amd import api  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    uniq = 40
    seq = bench.render_sequence(uniq, 0, min(8, os.cpu_count() or 1), drive="natural")
    order = bench.frame_order(uniq, n)
    trk = api.Tracker(0)
    dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
    pk, pa = np.zeros(16, np.float32), np.zeros(16, np.float32)
    trk.init(*dev[0])
    t0 = time.perf_counter()
    slow = 0
    for k in range(n):
        if bench.begins_pass(order, k):
            trk.init(*dev[0])
        if k + 1 < n:
            trk.hint_next(*dev[order[k + 1]])
        t1 = time.perf_counter()
        trk.track_into(dev[order[k]][0], dev[order[k]][1], pk, pa)
        slow += (time.perf_counter() - t1) > 2e-3
    trk._sync()
    dt = time.perf_counter() - t0
    print(f"{n} frames: {n / dt:.0f} frames/s, calls over 2 ms: {slow}, pose LM (workgroups, Solves redone) {trk.persistent_stats()}, "
          f"depth LM (on, jobs redone) {trk.depth_persistent_stats()}, chained Solves (adopted, wasted) {trk.chain_stats()}, env "
          f"{ {k: v for k, v in os.environ.items() if k.startswith('ODO_')} }", flush=True)
    trk.close()


if __name__ == "__main__":
    main()
