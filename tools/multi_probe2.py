import os, sys, time, threading, numpy as np
sys.path.insert(0, os.getcwd())
from odometry_amd import api, synth
import bench
seq = bench.render_sequence(64, 0, 16)
order = bench.frame_order(64, 400)
def run(n_seq, overlap, prio, steps=200):
    os.environ["ODO_LM_PRIORITY"] = prio
    trks = [api.Tracker(0, overlap_depth=overlap) for _ in range(n_seq)]
    devs = []
    for t in trks:
        d = [(t.upload_frame(l), t.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
        t.init(*d[0]); devs.append(d)
    barrier = threading.Barrier(n_seq + 1)
    def work(k):
        a, b = np.zeros(16, np.float32), np.zeros(16, np.float32)
        for i in order[:10]: trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        barrier.wait()
        for j, i in enumerate(order[10:10 + steps], start=10):
            if bench.begins_pass(order, j): trks[k].init(*devs[k][0])
            trks[k].track_into(devs[k][i][0], devs[k][i][1], a, b)
        barrier.wait()
    th = [threading.Thread(target=work, args=(k,)) for k in range(n_seq)]
    for t in th: t.start()
    barrier.wait(); t0 = time.perf_counter(); barrier.wait(); dt = time.perf_counter() - t0
    for t in th: t.join()
    for t in trks: t.close()
    print(f"S={n_seq} overlap={overlap} prio={prio}: {n_seq*steps/dt:.0f} frames/s", flush=True)
for S, ov, pr in [(1,2,"1"),(1,0,"1"),(2,2,"1"),(2,0,"1"),(2,0,"0"),(3,0,"0"),(4,0,"0"),(4,0,"1"),(4,1,"0"),(6,0,"0"),(8,0,"0")]:
    run(S, ov, pr)
