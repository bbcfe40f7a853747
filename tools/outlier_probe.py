import sys, time, gc, numpy as np
sys.path.insert(0, '.')
import bench
from odometry_amd import api
seq = bench.render_sequence(200, 0, 8)
trk = api.Tracker(0, overlap_depth=2)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
T = np.zeros(16, np.float32); A = np.zeros(16, np.float32)
for rep in range(8):
    trk.init(*dev[0])
    ts = []; kf = []
    gc.collect(); gc.disable()
    for i in range(1, 200):
        if i + 1 < 200: trk.hint_next(*dev[i + 1])
        t0 = time.perf_counter()
        f = trk.track_into(dev[i][0], dev[i][1], T, A)
        ts.append(time.perf_counter() - t0); kf.append(f)
    gc.enable()
    ts = np.array(ts) * 1e6; kf = np.array(kf)
    slow = [(i + 1, int(ts[i]), int(kf[i]), int(kf[i-1]) if i else 0) for i in range(len(ts)) if ts[i] > 420]
    print("pass", rep, "mean %.1f median %.1f" % (ts.mean(), np.median(ts)), "slow (frame, us, promote, after promote):", slow[:12], trk.persistent_stats(), trk.depth_persistent_stats())
