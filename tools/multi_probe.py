"""Probe: throughput of S trackers in one process under different LM-stream priority settings (diagnostic)."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import api, synth
import bench

seq = synth.make_sequence(16, seed=0)
order = bench.frame_order(16, 400)

def run(n_seq, prio, steps=200):
    os.environ["ODO_LM_PRIORITY"] = prio
    trks = [api.Tracker(0) for _ in range(n_seq)]
    devs = []
    for t in trks:
        d = [(t.upload_frame(l), t.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
        t.init(*d[0]); devs.append(d)
    bar = threading.Barrier(n_seq + 1)
    def work(k):
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        for i in order[:10]: trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        bar.wait()
        for j, i in enumerate(order[10:10 + steps], start=10):
            if bench.begins_pass(order, j): trks[k].init(*devs[k][0])
            trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        bar.wait()
    th = [threading.Thread(target=work, args=(k,)) for k in range(n_seq)]
    [t.start() for t in th]
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    [t.join() for t in th]; [t.close() for t in trks]
    return n_seq * steps / dt

for rep in range(2):
    for prio in ("0", "1"):
        print("prio", prio, [round(run(n, prio)) for n in (1, 2, 4, 8)], flush=True)
