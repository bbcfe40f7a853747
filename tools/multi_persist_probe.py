"""Probe: S separate trackers (one host thread each) in ONE process on one GPU, with and without the persistent LM launch. Their
persistent launches can land on the same XCD and then cannot all be resident: how often do they give up, and what does it cost?
(The supported way to run several sequences on one GPU is odo_tracker_batch, which gives every sequence its own XCD.)"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from odometry_amd import api  # noqa: E402


def run(n_seq, fine, steps=300):
    if fine:
        os.environ.pop("ODO_LM_NO_FINE", None)
    else:
        os.environ["ODO_LM_NO_FINE"] = "1"
    seq = bench.render_sequence(24, 0, 1)
    order = bench.frame_order(24, 600)
    trks = [api.Tracker(0) for _ in range(n_seq)]
    devs = []
    for t in trks:
        d = [(t.upload_frame(l), t.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
        t.init(*d[0])
        devs.append(d)
    bar = threading.Barrier(n_seq + 1)

    def work(k):
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        for i in order[:10]:
            trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        bar.wait()
        for j, i in enumerate(order[10:10 + steps], start=10):
            if bench.begins_pass(order, j):
                trks[k].init(*devs[k][0])
            trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        bar.wait()

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_seq)]
    [t.start() for t in th]
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    [t.join() for t in th]
    stats = [t.persistent_stats() for t in trks]
    [t.close() for t in trks]
    return round(n_seq * steps / dt), stats


if __name__ == "__main__":
    for n in (1, 2, 4):
        for fine in (1, 0):
            fps, st = run(n, fine)
            print(f"{n} trackers, persistent launch {'on ' if fine else 'off'}: {fps} frames/s, (workgroups, fall-backs) per tracker {st}", flush=True)
