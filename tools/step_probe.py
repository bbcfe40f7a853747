import sys, time, gc, numpy as np
sys.path.insert(0, '.')
import bench
from odometry_amd import api
seq = bench.render_sequence(200, 0, 8)
trk = api.Tracker(0, overlap_depth=2)
dev = [(trk.upload_frame(l), trk.upload_frame(r)) for l, r in zip(seq["left"], seq["right"])]
import torch
res = {}
for rep in range(3):
    trk.init(*dev[0])
    T = np.zeros(16, np.float32); A = np.zeros(16, np.float32)
    ts = []; kf = []
    gc.collect(); gc.disable()
    for i in range(1, 200):
        if i + 1 < 200: trk.hint_next(*dev[i + 1])
        t0 = time.perf_counter()
        f = trk.track_into(dev[i][0], dev[i][1], T, A)
        ts.append(time.perf_counter() - t0); kf.append(f)
    gc.enable()
    ts = np.array(ts) * 1e6; kf = np.array(kf)
    ev = None
    print("pass", rep, "mean %.1f median %.1f" % (ts.mean(), np.median(ts)), "kf frames", int(kf.sum()), "mean on promote %.1f" % ts[kf != 0].mean(), "after promote %.1f" % ts[np.roll(kf != 0, 1)].mean(), "other %.1f" % ts[(kf == 0) & ~np.roll(kf != 0, 1)].mean())
    st = trk.stats() if hasattr(trk, 'stats') else None
print([ (i+1, round(t)) for i, t in enumerate(ts[:40])])
print(kf[:40])
